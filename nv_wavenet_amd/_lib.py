"""ctypes loader for libwavenet_infer.so (the C ABI declared in include/*.h)."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("NVW_LIB") or os.path.join(_HERE, "libwavenet_infer.so")  # NVW_LIB: timing experiments

if not os.path.exists(LIB_PATH):
    raise ImportError(
        "nv_wavenet_amd: %s is missing. Build it with `python -c 'import __graft_entry__ as g; g.build()'` "
        "or `make -C nv_wavenet_amd/csrc`. There is no CPU fallback for the engine." % LIB_PATH)

# One HIP runtime per process: PyTorch-ROCm bundles its own libamdhip64.so (soname
# libamdhip64.so.7, the same as /opt/rocm's).  Importing torch FIRST puts that copy in the link
# map, and the dynamic loader then satisfies this library's NEEDED libamdhip64.so.7 with it; the
# other order loads two runtimes and the second one finds "no ROCm-capable device".
import torch  # noqa: E402,F401

lib = C.CDLL(LIB_PATH)

_fp = C.c_void_p  # float* / int* arguments are passed as raw addresses (host or device)
CONSUME_FN = C.CFUNCTYPE(None, C.c_void_p, C.c_int, C.c_int, C.c_void_p)

# every symbol of include/nv_wavenet_c.h and include/wavenet_infer.h
SIGNATURES = {
    "nvw_abi_version": (C.c_int, []),
    "nvw_supported": (C.c_int, [C.c_int] * 4),
    "nvw_list_supported": (C.c_int, [C.POINTER(C.c_int), C.c_int]),
    "nvw_create": (C.c_void_p, [C.c_int] * 10),
    "nvw_create_ex": (C.c_void_p, [C.c_int] * 11),
    "nvw_destroy": (None, [C.c_void_p]),
    "nvw_set_embeddings": (None, [C.c_void_p, _fp, _fp]),
    "nvw_set_layer_weights": (None, [C.c_void_p, C.c_int] + [_fp] * 7),
    "nvw_set_out_weights": (None, [C.c_void_p] + [_fp] * 4),
    "nvw_set_inputs": (None, [C.c_void_p, _fp, _fp]),
    "nvw_set_conditioning": (None, [C.c_void_p, _fp]),
    "nvw_set_inputs_n": (None, [C.c_void_p, _fp, _fp, C.c_int]),
    "nvw_set_conditioning_n": (None, [C.c_void_p, _fp, C.c_int]),
    "nvw_pack_conditioning": (None, [C.c_void_p, _fp, C.c_int, C.c_int, C.c_void_p]),
    "nvw_set_conditioning_direct": (None, [C.c_void_p, _fp, C.c_int]),
    "nvw_set_conditioning_direct_t": (C.c_int, [C.c_void_p, _fp, C.c_int, C.c_int]),
    "nvw_set_conditioning_packed": (None, [C.c_void_p, _fp, C.c_int]),
    "nvw_set_conditioning_packed_n": (C.c_int, [C.c_void_p, _fp, C.c_int, C.c_size_t]),
    "nvw_cond_tiles": (C.c_int, [C.c_void_p]),
    "nvw_produce_conditioning_f16": (C.c_int, [_fp, _fp, _fp, _fp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "nvw_max_cond_channels": (C.c_int, []),
    "nvw_set_conditioning_weights": (C.c_int, [C.c_void_p, _fp, _fp, C.c_int]),
    "nvw_feature_fragments": (C.c_int, [C.c_void_p]),
    "nvw_feature_elems": (C.c_size_t, [C.c_void_p, C.c_int]),
    "nvw_set_conditioning_features": (C.c_int, [C.c_void_p, _fp, C.c_int, C.c_size_t]),
    "nvw_pack_features": (C.c_int, [C.c_void_p, _fp, C.c_int, C.c_longlong, C.c_longlong, C.c_longlong, C.c_int, C.c_int, C.c_void_p]),
    "nvw_set_features": (C.c_int, [C.c_void_p, _fp, C.c_int, C.c_longlong, C.c_longlong, C.c_longlong, C.c_int]),
    "nvw_set_upsampling": (C.c_int, [C.c_void_p, _fp, _fp, C.c_int, C.c_int]),
    "nvw_set_mel": (C.c_int, [C.c_void_p, _fp, C.c_int, C.c_longlong, C.c_longlong, C.c_longlong, C.c_int]),
    "nvw_upsample_features": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "nvw_get_features": (C.c_int, [C.c_void_p, _fp, C.c_int, C.c_int]),
    "nvw_generate_stream": (C.c_int, [C.c_void_p, C.c_int, CONSUME_FN, C.c_void_p, C.c_int, C.c_int, _fp, C.c_void_p]),
    "nvw_set_selectors": (None, [C.c_void_p, _fp, C.c_int]),
    "nvw_chain_status": (C.c_uint, [C.c_void_p]),
    "nvw_chain_fallbacks": (C.c_uint, [C.c_void_p]),
    "nvw_chain_last_timeout": (C.c_uint, [C.c_void_p]),
    "nvw_set_chain_timeout_ms": (None, [C.c_void_p, C.c_double]),
    "nvw_set_clock_probe": (None, [C.c_void_p, C.c_int]),
    "nvw_set_ring_in_lds": (None, [C.c_void_p, C.c_int]),
    "nvw_last_launch_clock_ghz": (C.c_double, [C.c_void_p]),
    "nvw_run_range": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "nvw_reset_history": (None, [C.c_void_p, C.c_void_p]),
    "nvw_set_selector_seed": (None, [C.c_void_p, C.c_ulonglong]),
    "nvw_set_audio_out": (None, [C.c_void_p, C.c_void_p]),
    "nvw_set_audio_out_n": (None, [C.c_void_p, C.c_void_p, C.c_size_t]),
    "nvw_kernel_info": (None, [C.c_void_p, C.c_int, C.c_int, C.c_char_p, C.c_int]),
    "nvw_run": (C.c_int, [C.c_void_p, C.c_int, C.c_int, _fp, C.c_int, C.c_int, C.c_void_p]),
    "nvw_run_partial": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, _fp, C.c_int, C.c_int, C.c_void_p]),
    "nvw_run_chunks": (C.c_int, [C.c_void_p, C.c_int, CONSUME_FN, C.c_void_p, C.c_int, C.c_int, _fp, C.c_int,
                                 C.c_int, C.c_void_p]),
    "nvw_get_xt_out": (None, [C.c_void_p, C.c_int, _fp]),
    "nvw_get_skip_out": (None, [C.c_void_p, C.c_int, _fp]),
    "nvw_get_zs": (None, [C.c_void_p, _fp]),
    "nvw_get_za": (None, [C.c_void_p, _fp]),
    "nvw_get_p": (None, [C.c_void_p, _fp]),
    "nvw_get_y_out": (None, [C.c_void_p, _fp, C.c_int, C.c_int, C.c_void_p]),
    "nvw_device_synchronize": (None, []),
    "nvw_time_runs": (C.c_float, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "wavenet_infer": (None, [C.c_int, C.c_int, _fp, _fp, C.c_int, C.c_int] + [C.POINTER(C.c_void_p)] * 7 +
                      [_fp, _fp, C.c_int, _fp, C.c_int, _fp]),
    "get_R": (C.c_int, []),
    "get_S": (C.c_int, []),
    "get_A": (C.c_int, []),
}

ABI_VERSION = 6            # NVW_ABI_VERSION of include/nv_wavenet_c.h this package was written against
# (checked before the other symbols are bound, so that an older library fails with the rebuild hint, not an AttributeError)
_abi = getattr(lib, "nvw_abi_version", None)
_have = None
if _abi is not None:
    _abi.restype, _abi.argtypes = C.c_int, []
    _have = _abi()
if _have != ABI_VERSION:
    raise ImportError("nv_wavenet_amd: %s implements revision %s of include/nv_wavenet_c.h, this package expects %d: rebuild it "
                      "(make -C nv_wavenet_amd/csrc)" % (LIB_PATH, "<none: too old>" if _have is None else _have, ABI_VERSION))

for _name, (_res, _args) in SIGNATURES.items():
    _fn = getattr(lib, _name)  # AttributeError here = the library does not export a declared symbol
    _fn.restype = _res
    _fn.argtypes = _args


def addr(x):
    """Raw address of a numpy array (host) or torch tensor (host or device); None -> NULL."""
    if x is None:
        return None
    if hasattr(x, "data_ptr"):  # torch.Tensor
        assert x.is_contiguous(), "tensor must be contiguous"
        return x.data_ptr()
    if hasattr(x, "ctypes"):  # numpy.ndarray
        assert x.flags["C_CONTIGUOUS"], "array must be C-contiguous"
        return x.ctypes.data
    if isinstance(x, int):
        return x
    raise TypeError("expected numpy array, torch tensor or int address, got %r" % type(x))
