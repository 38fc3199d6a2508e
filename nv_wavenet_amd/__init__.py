"""nv_wavenet_amd -- MI355X-native autoregressive WaveNet inference (drop-in for NVIDIA/nv-wavenet).

The product is the HIP library ``libwavenet_infer.so`` (C ABI in ``include/``); this package is the
Python host side mirroring the reference's ``pytorch/`` wrapper:

    nv_wavenet_amd.nv_wavenet      NVWaveNet, Impl, column_major   (pytorch/nv_wavenet.py);
                                   NVWaveNetEngine, get_cond_input (engine kept alive)
    nv_wavenet_amd.nv_wavenet_ext  infer, num_*_channels           (pytorch/wavenet_infer_wrapper.cpp)
    nv_wavenet_amd.engine          WavenetEngine: the nvWavenetInfer class surface over ctypes

There is no CPU fallback: importing ``_lib`` without the built library raises.
"""
from . import _lib  # noqa: F401  (fails loudly when the HIP library is missing)
from .engine import WavenetEngine, Impl, Org, supported_configs  # noqa: F401
