"""Device primitives in isolation (GPU), in the spirit of the reference's math_test.cu:262-410: the MFMA GEMM path
(weight packing -> per-wave fragment streams -> prefetch ring -> MFMA, K permutation, gated tile pairing) bit-exact
on small-integer matrices in fp32 AND fp16 for every (M, K) the engine uses, the softmax / inverse-CDF pick block
against matrix_softmax + the oracle's selection rule on crafted logits, and the multi-CU hand-off primitives under
uneven load.  The kernels come from tests/cpp/wn_primitives.hip (libwn_primitives.so: test-only entries that call
the product's own device functions; nothing of it is in the product ABI)."""
import ctypes as C
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
_fp = C.POINTER(C.c_float)


@pytest.fixture(scope="module")
def prim():
    import torch  # noqa: F401  (one HIP runtime per process: torch's, see nv_wavenet_amd/_lib.py)
    path = os.path.join(HERE, "cpp", "libwn_primitives.so")
    assert os.path.exists(path), "build it with python -c 'import __graft_entry__ as g; g.build()'"
    lib = C.CDLL(path)
    lib.wnp_gemm.argtypes = [C.c_int] * 5 + [_fp, _fp, _fp]
    lib.wnp_softmax_pick.argtypes = [C.c_int, C.c_int, _fp, _fp, C.POINTER(C.c_int), _fp]
    lib.wnp_handoff.argtypes = [C.c_int, C.c_int, C.c_int, C.POINTER(C.c_longlong), C.POINTER(C.c_longlong),
                                C.POINTER(C.c_int), C.POINTER(C.c_uint)]
    lib.wnp_stream_walk.argtypes = [C.c_int] * 6 + [_fp, _fp, _fp]
    lib.wnp_gate.argtypes = [C.c_int, C.c_int, _fp, _fp, _fp]
    return lib


def _f(a):
    return a.ctypes.data_as(_fp)


GEMMS = [(128, 64, 4, 1), (64, 64, 4, 0), (256, 64, 4, 0), (128, 64, 4, 0), (256, 256, 4, 0), (256, 128, 4, 0),
         (256, 128, 4, 1), (128, 128, 4, 0), (64, 32, 2, 1), (32, 32, 2, 0), (128, 32, 2, 0), (512, 256, 4, 0)]


@pytest.mark.parametrize("precision", [32, 16])
@pytest.mark.parametrize("M,K,nw,gated", GEMMS)
def test_mfma_gemm_is_exact_on_small_integers(prim, M, K, nw, gated, precision):
    """math_test.cu:283-293 uses closed-form integer matrices so that every product and partial sum is exact in
    half as well: W[m][k] in {-2..2}, X[k][j] in {-1, 0, 1}; |sum| <= 2K <= 512 is exact in fp16 and fp32, so the
    packed, permuted, ring-fed MFMA result must equal W @ X bit for bit.  The gated (tanh | sigmoid) matrices of the
    fp16 engine carry the gate's pre-scale (2 log2 e, -log2 e) folded in at packing time: there the comparison is
    against the scaled product at fp16 weight rounding."""
    rng = np.random.default_rng(M * 1000 + K + precision)
    W = rng.integers(-2, 3, size=(M, K)).astype(np.float32)
    W[0, :] = 1 + (np.arange(K) % 2)               # a row whose sum depends on every k being visited once
    X = rng.integers(-1, 2, size=(K, 16)).astype(np.float32)
    X[:, 0] = 1
    Wcm = np.ascontiguousarray(W.T)                # col-major M x K = [K][M] row-major
    out = np.full((M, 16), np.nan, dtype=np.float32)
    assert prim.wnp_gemm(precision, M, K, nw, gated, _f(Wcm), _f(X), _f(out)) == 0
    ref = W.astype(np.float64) @ X.astype(np.float64)
    if gated and precision == 16:
        scale = np.where(np.arange(M) < M // 2, 2.88539008177792681472, -1.44269504088896340736)
        Ws = (W * scale[:, None]).astype(np.float16).astype(np.float64)
        ref = Ws @ X.astype(np.float64)
        assert np.allclose(out, ref, rtol=1e-6, atol=1e-4), np.abs(out - ref).max()
    else:
        assert np.array_equal(out, ref.astype(np.float32)), "first bad (m, j): %s" % (np.argwhere(out != ref)[:3].tolist(),)


def _expected_pick(logits, sel):
    """matrix_softmax (matrix.cpp:166-183) + the oracle's rule (nv_wavenet_reference.cpp:106-121) in float64:
    first index with sel < cumulative p; None when the draw sits within 1e-6 of an edge (either neighbour is right)."""
    x = logits.astype(np.float64)
    p = np.exp(x - x.max())
    p /= p.sum()
    c = np.cumsum(p)
    idx = int(np.searchsorted(c, sel, side="right"))
    near = np.abs(c - sel).min() < 1e-6
    return idx, near, p


@pytest.mark.parametrize("A,nw", [(256, 4), (256, 2), (512, 4), (1024, 4)])
def test_softmax_pick_against_the_oracle_rule(prim, A, nw):
    """Crafted logits (math_test.cu:356-360 uses tanh((r+c)/M)) and selectors incl. the edge cases: sel = 0 (first
    bin with mass), sel = 1 - 2^-24 (last bins), two-point masses, ties, a sharp peak, and a draw beyond the total
    (the scan falls off the end: 128 like softmax.cuh:154-155)."""
    rng = np.random.default_rng(A + nw)
    logits = np.zeros((16, A), dtype=np.float32)
    sel = np.zeros(16, dtype=np.float32)
    for u in range(16):
        logits[u] = np.tanh((np.arange(A) + u) / A)                        # the reference's softmax test input
        sel[u] = rng.random()
    sel[0] = 0.0
    sel[1] = 1.0 - 2.0 ** -24
    logits[2] = 0.0                                                        # uniform: ties everywhere
    sel[2] = 0.5 + 1.0 / (4 * A)
    logits[3] = -30.0
    logits[3, 7] = 0.0                                                     # a point mass
    logits[4] = -30.0
    logits[4, A - 1] = 0.0                                                 # ... in the last bin
    logits[5] = -30.0
    logits[5, 3], logits[5, A // 2 + 1] = 0.0, 0.0                         # two equal masses, draw in the second
    sel[5] = 0.75
    logits[6] = rng.normal(0, 3, A)                                        # wide dynamic range
    logits[7] = 0.0
    sel[7] = 1.5                                                           # beyond the total: falls off the end
    logits[8] = 40.0 * np.tanh((np.arange(A) - 17.0) / 3.0)               # large magnitudes (max subtraction matters)
    picks = np.full(16, -1, dtype=np.int32)
    probs = np.zeros((16, A), dtype=np.float32)
    assert prim.wnp_softmax_pick(A, nw, _f(logits), _f(sel), picks.ctypes.data_as(C.POINTER(C.c_int)), _f(probs)) == 0
    for u in range(16):
        idx, near, p = _expected_pick(logits[u], float(sel[u]))
        assert np.allclose(probs[u], p, rtol=2e-5, atol=1e-9), (u, np.abs(probs[u] - p).max())
        if u == 7:
            assert picks[u] == 128
            continue
        if u == 1:
            # the largest selector below 1: sel * total may round to total itself, and then no cumulative sum exceeds
            # it -- the scan falls off the end like the reference's GPU code (softmax.cuh:154-155 -> 128); the oracle's
            # own cumulative p is just as fragile there (its assert(y >= 0), nv_wavenet_reference.cpp:119)
            assert int(picks[u]) in (A - 1, 128), (int(picks[u]), idx)
            continue
        if near:
            assert abs(int(picks[u]) - idx) <= 1, (u, picks[u], idx)
        else:
            assert int(picks[u]) == idx, (u, int(picks[u]), idx, float(sel[u]))
    assert picks[0] == 0 and picks[3] == 7 and picks[4] == A - 1 and picks[5] == A // 2 + 1


@pytest.mark.parametrize("force_agent", [0, 1])
def test_handoff_granules_under_uneven_load(prim, force_agent):
    """MI355X_MICROARCH.md: "test every hand-off under UNEVEN load, consumer L1-warm, checking every word".  120 producer /
    consumer workgroup pairs exchange 200 messages each through single-slot mailboxes with unevenly delayed
    producers; every 32-bit word of every message is checked.  force_agent = 1: agent-scope (write-through) stores
    everywhere; 0: workgroup-scope stores where the placement exchange found both ends on one XCD."""
    pairs, rounds = 120, 200
    good = (C.c_longlong * pairs)()
    bad = (C.c_longlong * pairs)()
    same = (C.c_int * pairs)()
    status = C.c_uint(0)
    assert prim.wnp_handoff(pairs, rounds, force_agent, good, bad, same, C.byref(status)) == 0
    assert status.value == 0, "a hand-off timed out: 0x%x" % status.value
    words = rounds * 2 * 4 * 256                                           # NT * NW tiles x 4 registers x 64 lanes
    assert all(b == 0 for b in bad), "corrupted words per pair: %s" % [int(b) for b in bad if b][:5]
    assert all(g == words for g in good), (int(good[0]), words)
    print("pairs on one XCD: %d of %d" % (sum(same), pairs))


@pytest.mark.parametrize("precision,R,S,A,L", [(16, 64, 256, 256, 5), (32, 64, 256, 256, 3), (16, 64, 128, 256, 4), (16, 128, 256, 256, 3),
                                               (32, 32, 128, 256, 4), (16, 64, 256, 256, 2), (16, 32, 256, 256, 3)])
def test_weight_stream_walk_is_exact_on_small_integers(prim, precision, R, S, A, L):
    """What wavenet_wg actually runs since round 2: pack_layer_kernel lays every layer's four matrices into the per-wave
    streams at Cfg::streamPos (consumption order, the skip GEMM of a layer behind the next layer's current tap, the dilated
    tap of layer l+1 at the end of layer l, prev(0) behind the last layer), the head follows padded to whole ring turns;
    the kernel walks that stream through the buffer-resource prefetch ring (gemm_b / take_group / refill_group /
    skip_frags) for TWO samples, so the wrap from the head back to layer 0 is exercised.  Small-integer matrices: every
    W X must come out exact (fp16 gated matrices: at the packed pre-scale, like the GEMM test above), both passes."""
    rng = np.random.default_rng(precision * 7 + R + S + A + L)
    K = max(R, S, A)
    X = rng.integers(-1, 2, size=(K, 16)).astype(np.float32)
    X[:, 0] = 1
    mats, Wflat = [], []
    for l in range(L):
        per = []
        for (M, KK) in ((2 * R, R), (2 * R, R), (R, R), (S, R)):
            W = rng.integers(-2, 3, size=(M, KK)).astype(np.float32)
            W[0, :] = 1 + (np.arange(KK) + l) % 2
            per.append(W)
            Wflat.append(np.ascontiguousarray(W.T).ravel())      # col-major
        mats.append(per)
    Wzs = rng.integers(-2, 3, size=(A, S)).astype(np.float32)
    Wza = rng.integers(-2, 3, size=(A, A)).astype(np.float32)
    Wflat += [np.ascontiguousarray(Wzs.T).ravel(), np.ascontiguousarray(Wza.T).ravel()]
    Wflat = np.concatenate(Wflat).astype(np.float32)
    per_pass = L * (5 * R + S) * 16 + 2 * A * 16
    passes = 2
    out = np.full(per_pass * passes, np.nan, dtype=np.float32)
    assert prim.wnp_stream_walk(precision, R, S, A, L, passes, _f(Wflat), _f(X), _f(out)) == 0
    scale = np.where(np.arange(2 * R) < R, 2.88539008177792681472, -1.44269504088896340736)

    def want(W, gated, KK):
        Wd = W.astype(np.float64)
        if gated and precision == 16:
            Wd = (W * scale[:, None]).astype(np.float16).astype(np.float64)
        return Wd @ X[:KK].astype(np.float64)
    for ps in range(passes):
        o = out[ps * per_pass:(ps + 1) * per_pass]
        at = 0
        for l in range(L):
            for name, W, gated in (("prev", mats[l][0], True), ("cur", mats[l][1], True), ("res", mats[l][2], False), ("skip", mats[l][3], False)):
                M = W.shape[0]
                got = o[at:at + M * 16].reshape(M, 16)
                at += M * 16
                ref = want(W, gated, R)
                if gated and precision == 16:
                    assert np.allclose(got, ref, rtol=1e-6, atol=1e-4), (ps, l, name, np.abs(got - ref).max())
                else:
                    assert np.array_equal(got, ref.astype(np.float32)), (ps, l, name, np.argwhere(got != ref)[:3].tolist())
        for name, W, KK in (("zs", Wzs, S), ("za", Wza, A)):
            got = o[at:at + A * 16].reshape(A, 16)
            at += A * 16
            assert np.array_equal(got, want(W, False, KK).astype(np.float32)), (ps, name)


@pytest.mark.parametrize("way", [0, 1])
def test_fp16_gate_on_prescaled_inputs(prim, way):
    """The fp16 engine's gate (nv_wavenet_util.cuh:78-86 is what it replaces: tanhf * sigmoid in fp32): pre-activations arrive
    pre-scaled by 2 log2 e / -log2 e and the gate is exp2 / rcp only,  tanh a = 1 - 2 / (2^a' + 1),  sigmoid b = 1 / (1 + 2^b').
    v_exp_f32 and v_rcp_f32 are good to 1 ulp, so both factors carry an ABSOLUTE error of a few 1e-7 (the tanh form
    cancels for small |a|: its relative error there is large, its absolute error is not) -- three orders of magnitude
    below the fp16 rounding h gets next (u = 4.9e-4).  Stated bound: |h - tanh(a) sigmoid(b)| <= 1e-6 on [-8, 8]^2 plus the
    saturated corners, for the one-shot form (way 0) and the five-stage form wavenet_wg interleaves with MFMAs (way 1),
    which must agree with each other bit for bit."""
    rng = np.random.default_rng(5)
    n = 1 << 16
    a = rng.uniform(-8, 8, n).astype(np.float32)
    b = rng.uniform(-8, 8, n).astype(np.float32)
    a[:8] = [0, 1e-4, -1e-4, 30, -30, 0.5, -0.5, 88]
    b[:8] = [0, 30, -30, 0, 0, 88, -88, 0.25]
    h = np.full(n, np.nan, dtype=np.float32)
    assert prim.wnp_gate(n, way, _f(a), _f(b), _f(h)) == 0
    ref = np.tanh(a.astype(np.float64)) / (1 + np.exp(-b.astype(np.float64)))
    err = np.abs(h - ref)
    assert np.all(np.isfinite(h)) and err.max() <= 1e-6, (err.max(), a[err.argmax()], b[err.argmax()])
    h0 = np.full(n, np.nan, dtype=np.float32)
    assert prim.wnp_gate(n, 0, _f(a), _f(b), _f(h0)) == 0
    assert np.array_equal(h, h0)
    print("fp16-engine gate, way %d: max |error| %.3g over %d points" % (way, err.max(), n))
