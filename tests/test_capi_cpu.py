"""CPU tests (no GPU): the C-ABI library loads and exports every symbol include/*.h declares;
host-side logic of the Python mirror of the reference wrapper. No compute calls."""
import ctypes
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    syms = set()
    for h in ("wavenet_infer.h", "nv_wavenet_c.h"):
        txt = open(os.path.join(ROOT, "include", h)).read()
        txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
        for m in re.finditer(r"^\s*(?:[A-Za-z_][\w\s\*]*?)\b(\w+)\s*\(", txt, flags=re.M):
            name = m.group(1)
            if name not in ("defined", "nvw_consume_fn") and not name.startswith("__"):
                syms.add(name)
    syms.discard("void")
    return syms


def test_library_exports_every_declared_symbol():
    from nv_wavenet_amd import _lib
    syms = _declared_symbols()
    assert {"wavenet_infer", "get_R", "get_S", "get_A", "nvw_create", "nvw_run_chunks"} <= syms
    lib = ctypes.CDLL(_lib.LIB_PATH)
    missing = [s for s in sorted(syms) if not hasattr(lib, s)]
    assert not missing, "declared in include/*.h but not exported: %s" % missing
    # and the Python binding table covers exactly the declared set
    assert set(_lib.SIGNATURES) == syms


def test_compiled_channel_counts_and_instantiations():
    from nv_wavenet_amd import nv_wavenet_ext, supported_configs
    assert (nv_wavenet_ext.num_res_channels(), nv_wavenet_ext.num_skip_channels(),
            nv_wavenet_ext.num_out_channels()) == (64, 256, 256)  # pytorch/wavenet_infer.cu:35-37
    cfg = set(supported_configs())
    # the reference's tested channel combinations (README.md:5-10), fp32 and fp16
    for rsa in ((32, 128, 256), (64, 128, 256), (64, 256, 256), (128, 256, 256)):
        assert rsa + (32,) in cfg and rsa + (16,) in cfg


def test_unsupported_config_is_rejected_loudly():
    from nv_wavenet_amd import WavenetEngine
    with pytest.raises(ValueError):
        WavenetEngine(48, 128, 256, 4, 8, 1, 8)
    with pytest.raises(ValueError):
        WavenetEngine(64, 256, 256, 4, 8, 1, 8, impl=7)


def test_column_major_matches_reference_semantics():
    import torch
    from nv_wavenet_amd.nv_wavenet import column_major
    w = torch.arange(6, dtype=torch.float32).reshape(2, 3)       # M=2 rows, K=3
    cm = column_major(w)                                          # -> [K][M] contiguous
    assert cm.shape == (3, 2) and cm.is_contiguous()
    assert cm.flatten().tolist() == [0, 3, 1, 4, 2, 5]            # W[m + k*M]
    assert column_major(w.reshape(2, 3, 1)).equal(cm)
    b = torch.arange(4.0)
    assert column_major(b) is b
    c = torch.arange(2 * 3 * 4 * 5, dtype=torch.float32).reshape(2, 3, 4, 5)   # (2R,B,L,N)
    cc = column_major(c)
    assert cc.shape == (5, 4, 3, 2) and cc[1, 2, 0, 1] == c[1, 0, 2, 1]


def test_nvwavenet_constructor_checks_shapes():
    import torch
    from nv_wavenet_amd.nv_wavenet import NVWaveNet
    R, S, A, L = 64, 256, 256, 3
    mk = lambda *s: torch.zeros(*s)
    kw = dict(embedding_prev=mk(A, R), embedding_curr=mk(A, R), conv_out_weight=mk(A, S, 1),
              conv_end_weight=mk(A, A, 1), dilate_weights=[mk(2 * R, R, 2) for _ in range(L)],
              dilate_biases=[mk(2 * R) for _ in range(L)], max_dilation=4,
              res_weights=[mk(R, R, 1) for _ in range(L - 1)], res_biases=[mk(R) for _ in range(L - 1)],
              skip_weights=[mk(S, R, 1) for _ in range(L)], skip_biases=[mk(S) for _ in range(L)],
              use_embed_tanh=False)
    m = NVWaveNet(**kw)
    assert m.num_layers == L and len(m.layers) == 7 * L
    assert m.layers[0].shape == (R, 2 * R)        # Wprev of layer 0, column-major
    bad = dict(kw, embedding_prev=mk(A, R + 1))
    with pytest.raises(AssertionError):
        NVWaveNet(**bad)
    bad = dict(kw, dilate_weights=[mk(2 * R, R, 3) for _ in range(L)])
    with pytest.raises(AssertionError):
        NVWaveNet(**bad)


def test_get_cond_input_matches_torch_modules():
    """nv_wavenet.get_cond_input restates WaveNet.get_cond_input (pytorch/wavenet.py:190-202) over
    the module's tensors; compare with the torch modules the reference builds (ConvTranspose1d
    upsample, 1x1 Conv1d cond_layers) in both output layouts."""
    import torch
    from nv_wavenet_amd.nv_wavenet import get_cond_input, column_major
    torch.manual_seed(0)
    n_cond, R, L, B, frames, win, stride = 8, 4, 3, 2, 5, 12, 4
    up = torch.nn.ConvTranspose1d(n_cond, n_cond, win, stride)
    cl = torch.nn.Conv1d(n_cond, 2 * R * L, 1)
    feats = torch.randn(B, n_cond, frames)
    with torch.no_grad():
        x = up(feats)
        x = x[:, :, :-(win - stride)]
        x = cl(x)
        x = x.view(x.size(0), L, -1, x.size(2)).permute(2, 0, 1, 3)          # 2R x B x L x N
        got = get_cond_input(feats, up.weight, up.bias, stride, cl.weight, cl.bias, L)
        got2 = get_cond_input(feats, up.weight, up.bias, stride, cl.weight, cl.bias, L, layout="NLBC")
    assert got.shape == (2 * R, B, L, frames * stride) and torch.equal(got, x)
    assert got2.is_contiguous() and torch.equal(got2, column_major(x.contiguous()))


@pytest.mark.parametrize("n_cond,R,L,B,frames,win,stride", [(8, 4, 3, 2, 5, 12, 4), (80, 64, 20, 32, 1, 1024, 256), (16, 8, 2, 4, 7, 32, 8),
                                                            (8, 4, 3, 2, 2, 16, 4), (8, 4, 2, 3, 6, 10, 4)])
def test_get_cond_input_as_matrix_products_equals_the_convolutions(n_cond, R, L, B, frames, win, stride):
    """On the GPU get_cond_input runs its two convolutions as matrix products (the library convolutions compile their kernels at
    first use on a machine without a cache: bench.py's default run did not finish for that).  Same sums in another order: every
    layout must agree with the torch convolutions to fp32 rounding, the fragment-order output (fp16) to one unit in the last
    place, also when it is written into an existing packed buffer; a kernel that is no multiple of the stride falls back."""
    import torch
    from nv_wavenet_amd.nv_wavenet import get_cond_input
    torch.manual_seed(1)
    up_w, up_b = torch.randn(n_cond, n_cond, win) * 0.1, torch.randn(n_cond)
    cw, cb = torch.randn(2 * R * L, n_cond, 1) * 0.2, torch.randn(2 * R * L)
    f = torch.randn(B, n_cond, frames)
    for lay in ("NLBC", "CBLN"):
        a = get_cond_input(f, up_w, up_b, stride, cw, cb, L, layout=lay, via_gemm=False)
        b = get_cond_input(f, up_w, up_b, stride, cw, cb, L, layout=lay, via_gemm=True)
        assert a.shape == b.shape == ((frames * stride, L, B, 2 * R) if lay == "NLBC" else (2 * R, B, L, frames * stride))
        assert torch.allclose(a, b, rtol=0, atol=2e-5 * float(a.abs().max())), float((a - b).abs().max())
    if R == 64:
        tiles = B // 16
        a = get_cond_input(f, up_w, up_b, stride, cw, cb, L, layout="packed", precision=16, tiles=tiles, via_gemm=False)
        b = get_cond_input(f, up_w, up_b, stride, cw, cb, L, layout="packed", precision=16, tiles=tiles, via_gemm=True)
        assert a.dtype == b.dtype == torch.float16 and a.shape == b.shape
        ulp = float(a.float().abs().max()) * 2.0 ** -10
        assert float((a.float() - b.float()).abs().max()) <= ulp
        n = frames * stride
        o1 = torch.zeros(n + 1, L, tiles, 2 * R // 32, 4, 16, 8, dtype=torch.float16)
        o2 = torch.zeros_like(o1)
        get_cond_input(f, up_w, up_b, stride, cw, cb, L, layout="packed", precision=16, tiles=tiles, out=o1[:n], via_gemm=False)
        get_cond_input(f, up_w, up_b, stride, cw, cb, L, layout="packed", precision=16, tiles=tiles, out=o2[:n], via_gemm=True)
        assert float((o1.float() - o2.float()).abs().max()) <= ulp and torch.equal(o1[:n], a[:n]) and not o1[n].any()


def test_reference_style_host_program_compiles_against_nv_wavenet_hpp():
    """a host translation unit written the way the reference's nv_wavenet_test.cu /
    pytorch/wavenet_infer.cu use nvWavenetInfer -- every member, the reference's defaults, a lambda
    run_chunks consumer, fp32 and fp16 -- compiles against nv_wavenet.hpp (hipcc cross-compiles without a GPU)."""
    import subprocess
    import tempfile
    src = os.path.join(ROOT, "tests", "cpp", "api_surface.hip")
    with tempfile.TemporaryDirectory() as tmp:
        r = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O1", "-std=c++17", "-Wno-unused-result",
                            "-I", os.path.join(ROOT, "nv_wavenet_amd", "csrc"), "-c", src, "-o", os.path.join(tmp, "a.o")],
                           capture_output=True, text=True, timeout=1200)
    assert r.returncode == 0, r.stderr[-3000:]


def test_reference_binding_is_built_when_the_reference_tree_is_here():
    """oracle/build_ref_binding.py compiles the reference's own pybind wrapper (pytorch/wavenet_infer_wrapper.cpp with
    the two documented edits) against libwavenet_infer.so; the extension imports and reports the compiled channel
    counts without a GPU.  (The GPU test runs inference through it.)  No bytecode of the reference's Python is left
    beside it: that file does not travel."""
    import subprocess
    import sys
    if not os.path.exists("/root/reference/pytorch/wavenet_infer_wrapper.cpp"):
        pytest.skip("no reference tree on this machine")
    subprocess.check_call([sys.executable, os.path.join(ROOT, "oracle", "build_ref_binding.py")])
    code = ("import sys, torch; sys.path.insert(0, %r); import nv_wavenet_ext as m; "
            "print(m.num_res_channels(), m.num_skip_channels(), m.num_out_channels())") % os.path.join(ROOT, "oracle", "_ref")
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and out.stdout.split() == ["64", "256", "256"], out.stderr[-2000:]
    assert not [f for f in os.listdir(os.path.join(ROOT, "oracle", "_ref")) if f.endswith((".pyc", ".py"))]


def test_reference_python_wrapper_prepares_the_same_tensors():
    """The reference's OWN pytorch/nv_wavenet.py, imported where it lies (this container only: it does not travel), against this
    package's mirror of it (nv_wavenet_amd/nv_wavenet.py): for the same exported weights both classes hand the extension the same
    tensors -- embeddings, output matrices, the 7-per-layer list in the same order with the same column-major images and the
    zero-padded last residual layer --, report the same channel counts and layer count, and turn a conditioning tensor into the same
    [N][L][B][2R] image (column_major).  What follows `nv_wavenet_ext.infer(...)` is the C ABI both share."""
    import sys
    import importlib.util
    import torch
    ref_py = "/root/reference/pytorch/nv_wavenet.py"
    refdir = os.path.join(ROOT, "oracle", "_ref")
    if not (os.path.exists(ref_py) and os.path.exists(os.path.join(refdir, "nv_wavenet_ext.so"))):
        pytest.skip("no reference tree (or no compiled reference extension) on this machine")
    sys.path.insert(0, refdir)                      # the reference's file imports `nv_wavenet_ext` at its top
    try:
        spec = importlib.util.spec_from_file_location("nv_wavenet_reference_py", ref_py)
        ref = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(ref)
    finally:
        sys.path.remove(refdir)
        sys.modules.pop("nv_wavenet_ext", None)
    from nv_wavenet_amd import nv_wavenet as ours
    R, S, A, L = 64, 256, 256, 5
    g = torch.Generator().manual_seed(11)
    rnd = lambda *shape: torch.randn(*shape, generator=g)
    kw = dict(embedding_prev=rnd(A, R), embedding_curr=rnd(A, R), conv_out_weight=rnd(A, S, 1), conv_end_weight=rnd(A, A, 1),
              dilate_weights=[rnd(2 * R, R, 2) for _ in range(L)], dilate_biases=[rnd(2 * R) for _ in range(L)], max_dilation=4,
              res_weights=[rnd(R, R, 1) for _ in range(L - 1)], res_biases=[rnd(R) for _ in range(L - 1)],
              skip_weights=[rnd(S, R, 1) for _ in range(L)], skip_biases=[rnd(S) for _ in range(L)], use_embed_tanh=False)
    a, b = ref.NVWaveNet(**kw), ours.NVWaveNet(**kw)
    assert (a.R, a.S, a.A, a.num_layers, a.max_dilation, a.use_embed_tanh) == (b.R, b.S, b.A, b.num_layers, b.max_dilation, b.use_embed_tanh)
    for name in ("embedding_prev", "embedding_curr", "conv_out", "conv_end"):
        x, y = getattr(a, name), getattr(b, name)
        assert x.shape == y.shape and x.is_contiguous() and y.is_contiguous() and torch.equal(x, y), name
    assert len(a.layers) == len(b.layers) == 7 * L
    for i, (x, y) in enumerate(zip(a.layers, b.layers)):
        assert x.shape == y.shape and torch.equal(x.contiguous(), y.contiguous()) and y.is_contiguous(), "layers[%d]" % i
    cond = rnd(2 * R, 3, L, 7)
    assert torch.equal(ref.column_major(cond), ours.column_major(cond))
    assert (ref.Impl.AUTO, ref.Impl.SINGLE_BLOCK, ref.Impl.DUAL_BLOCK, ref.Impl.PERSISTENT) == \
           (int(ours.Impl.AUTO), int(ours.Impl.SINGLE_BLOCK), int(ours.Impl.DUAL_BLOCK), int(ours.Impl.PERSISTENT))
