"""Shared helpers for the parity tests. The oracle is the CHECKER here, never the thing tested."""
import os

import numpy as np

from oracle import oracle as O
import cases

# test "mode" names -> nvwOrganisation (wg = exactly one tile per workgroup, the latency kernel as first built)
MODE_ORG = {None: 0, "auto": 0, "wg": 2, "wg2": 3, "wg3": 4, "chain": 5, "chain1": 6, "wg4": 10}
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"))


def gen_inputs(case, half=False):
    t = O.gen_test_inputs(case.seed, case.prior, case.shape, "oracle")
    g = load_golden(case.name)
    assert t.crc() == int(g["crc_inputs"][0]), "glibc rand() stream drifted: inputs differ from the fixture"
    if half:
        t.round_to_half()
    return t


def make_oracle(case, t):
    s = case.shape
    o = O.Oracle(s.L, s.B, s.N, s.R, s.S, s.A, s.maxD)
    o.set_model(t)
    o.set_inputs(t.Lh, t.sel)
    return o


def make_engine(case, t, precision=32, impl=None, device_ptrs=False, mode=None):
    """Build the HIP engine through the C ABI and upload the model + inputs.
    mode: None = the engine's own choice from the Implementation value and the batch size; "wg" / "wg2" / "wg3" /
    "chain" / "chain1" force a kernel organisation (nvw_create_ex)."""
    from nv_wavenet_amd import WavenetEngine
    return _make_engine(WavenetEngine, case, t, precision, impl, device_ptrs, mode)


def _make_engine(WavenetEngine, case, t, precision, impl, device_ptrs, mode=None):
    s = case.shape
    e = WavenetEngine(s.R, s.S, s.A, s.L, s.maxD, s.B, s.N, impl=case.impl if impl is None else impl,
                      tanhEmbed=True, precision=precision, organisation=MODE_ORG[mode])
    conv = (lambda a: a)
    if device_ptrs:
        import torch
        conv = lambda a: torch.from_numpy(a).cuda()
    e.setEmbeddings(conv(t.embP), conv(t.embC))
    for l in range(s.L):
        e.setLayerWeights(l, conv(t.Wprev[l]), conv(t.Wcur[l]), conv(t.Bh[l]), conv(t.Wres[l]), conv(t.Bres[l]),
                          conv(t.Wskip[l]), conv(t.Bskip[l]))
    e.setOutWeights(conv(t.Wzs), conv(t.Bzs), conv(t.Wza), conv(t.Bza))
    e.setInputs(conv(t.Lh), conv(t.sel))
    return e


def engine_getters(e, L):
    return dict(Xout=np.stack([e.getXtOut(l) for l in range(L)]),
                skipOut=np.stack([e.getSkipOut(l) for l in range(L)]),
                Zs=e.getZs(), Za=e.getZa(), P=e.getP())


def matrix_compare(name, ref, got, tol, relu=False, atol_eps=4):
    """The reference's matrix_compare (matrix.cpp:133-151) with two repairs.  It passes when
    |got/ref| - 1 <= tol: (1) that is one-sided and lets a too-small magnitude through, here
    |got - ref| <= tol*|ref| both ways; (2) a purely relative bar is meaningless for elements that
    are themselves cancellation residue (|ref| ~ 1e-8 in tensors of scale 1e-1: fp32 cannot carry
    them to 1e-2 relative on ANY summation order, and the reference's one-sided form hides that),
    so an absolute term at the fp32 rounding level of the tensor is allowed: 4*eps32*max|ref|.
    relu=True: where either value is <= 0 both must be < tol (matrix.cpp:142)."""
    ref = np.asarray(ref, dtype=np.float64)
    got = np.asarray(got, dtype=np.float64)
    assert ref.shape == got.shape, (name, ref.shape, got.shape)
    assert np.all(np.isfinite(got)), name + ": non-finite values"
    atol = atol_eps * 1.1920929e-07 * np.abs(ref).max()
    ok = np.abs(got - ref) <= tol * np.abs(ref) + atol
    if relu:
        r = (ref <= 0) | (got <= 0)
        ok[r] = (ref[r] < tol) & (got[r] < tol)
    if not ok.all():
        idx = np.argwhere(~ok)[0]
        raise AssertionError("%s mismatch at %s: ref %.10e vs got %.10e (tol %g, atol %.3g, %d bad of %d)" %
                             (name, tuple(idx), ref[tuple(idx)], got[tuple(idx)], tol, atol, (~ok).sum(), ok.size))


def compare_activations(ref, got, tols=None, names=("Xout", "skipOut", "Zs", "Za", "P"), atol_eps=4):
    """atol_eps: the absolute term in units of eps32 * max|tensor|.  4 for the reference's own recipe (every tensor is a
    sum of same-sized terms); on the O(1) recipe a K = 256 dot product of order-one terms that cancels to 1e-3 carries a
    summation-order error of ~sqrt(K) eps * rms|term| ~ 16 eps * 0.3 (4 sigma over 1e4 elements: ~20 eps * max), which no
    fp32 implementation with another summation order can avoid: 32 there.  (A broken network moves these tensors by
    1e5 such units: tests/test_parity_bars_cpu.py.)"""
    tols = tols or cases.TOL
    for k in names:
        matrix_compare(k, ref[k], got[k], tols[k], cases.RELU_AWARE[k], atol_eps)


def explain_mismatches(y_ref, y_got, lo, hi, sel_bn, eps):
    """Exact-sample parity is chaotic: one differing pick changes every later sample of that
    utterance. Returns (n_utterances_diverged, unexplained list). A first divergence at (b,t) is
    'explained' when the draw sits within eps of a CDF edge of the oracle's pick
    (SURVEY.md 8c); later samples of a diverged utterance are not compared."""
    B, N = y_ref.shape
    diverged, unexplained = 0, []
    for b in range(B):
        d = np.nonzero(y_ref[b] != y_got[b])[0]
        if d.size == 0:
            continue
        t = int(d[0])
        diverged += 1
        s = float(sel_bn[b, t])
        near = min(abs(s - float(lo[b, t])), abs(s - float(hi[b, t])))
        if not (near <= eps and abs(int(y_ref[b, t]) - int(y_got[b, t])) <= 1):
            unexplained.append((b, t, int(y_ref[b, t]), int(y_got[b, t]), near))
    return diverged, unexplained


# ---------------------------------------------------------------------------------------------------------------------
# The O(1) input recipe and the fp16 parity bars
# ---------------------------------------------------------------------------------------------------------------------
# Under the reference's own recipe (nv_wavenet_test.cu:36-111: every tensor uniform within +-0.25/rows) the logits are
# Bza +- 1.6e-4 and the output distribution is uniform to 0.4 %: the picks test softmax + scan + Bza and nothing else, and
# a network with every layer matrix zeroed produces the same samples (tests/test_parity_bars_cpu.py
# reproduces that).  o1_recipe() rescales the SAME glibc-rand() draws -- so the fixtures' CRC pins still hold -- to the
# magnitudes of a trained network: activations of order one in every layer, logits of order one, all 256 bins in play.
# Target standard deviations (fan-in scaled, He-style): embeddings 0.45 each, conditioning 0.5, gate matrices 0.55/sqrt(R)
# and gate bias 0.3 (pre-activations of order one), residual 0.5/sqrt(R), skip 1.5/sqrt(R L) (the sum over L layers is of
# order one), Wzs 1.6/sqrt(S), Wza 1.5/sqrt(A).
def o1_recipe(t):
    R, S, L, A = t.R, t.S, t.L, t.A
    s12 = np.sqrt(12.0)

    def to(a, width, std):                 # a is uniform of full width `width`: std = width / sqrt(12)
        a *= np.float32(std / (width / s12))
    to(t.embP, 0.5 / R, 0.45), to(t.embC, 0.5 / R, 0.45)
    to(t.Lh, 0.5 / R, 0.5)
    to(t.Wprev, 0.5 / (2 * R), 0.55 / np.sqrt(R)), to(t.Wcur, 0.5 / (2 * R), 0.55 / np.sqrt(R)), to(t.Bh, 0.5 / (2 * R), 0.3)
    to(t.Wres, 0.5 / R, 0.5 / np.sqrt(R)), to(t.Bres, 0.5 / R, 0.05)
    to(t.Wskip, 0.5 / S, 1.5 / np.sqrt(R * L)), to(t.Bskip, 0.5 / S, 0.3 / np.sqrt(L))
    to(t.Wzs, 0.5 / R, 1.6 / np.sqrt(S)), to(t.Bzs, 0.5 / R, 0.2)
    to(t.Wza, 0.5 / R, 1.5 / np.sqrt(A)), to(t.Bza, 0.5 / R, 0.2)
    return t


def gen_o1(case, half):
    """Inputs of `case` in the O(1) recipe; half: every parameter rounded through fp16 (what an fp16 engine stores)."""
    t = O.gen_test_inputs(case.seed, case.prior, case.shape, "oracle")
    o1_recipe(t)
    if half:
        t.round_to_half()
    return t


def teacher_forced_oracle(case, t, y, tanh_embed=True):
    """The fp32 oracle FED the samples y [B][N] (an engine's free run): its own pick at every step given that history,
    the CDF edges of that pick, and its activations at the last sample (same history as the engine's dump)."""
    s = case.shape
    o = O.Oracle(s.L, s.B, s.N, s.R, s.S, s.A, s.maxD)
    o.set_model(t)
    o.set_tanh_embed(tanh_embed)
    o.set_inputs(t.Lh, t.sel)
    y_own, lo, hi = o.run(s.N, forced=np.ascontiguousarray(y, dtype=np.int32), edges=True)
    ref = o.getters()
    o.close()
    ref.update(y=y_own, lo=lo, hi=hi)
    return ref


# fp16 parity bars.  The reference has no fp16 test; the fp16 engine is held to the fp32 oracle fed the same
# fp16-rounded parameters.  What it may differ by is operand rounding: every GEMM operand (weights, x, h, relu(skip),
# relu(zs), conditioning) carries a relative error of at most u = 2^-11 (fp16 unit roundoff), accumulation is fp32, the
# gate and the softmax are fp32.  A dot product of K such terms is off by at most ~2u * sum|terms| and, with errors of
# random sign, by ~u * sqrt(K) * rms|term|: a fraction of u of the tensor's magnitude per GEMM; the residual stream adds
# one such error per layer (random walk over L <= 30 layers: ~sqrt(L) * u / 2).  tests/fp16_model.py applies exactly these
# roundings on the CPU and lands at 0.6-0.9 u * max|tensor| for Xout, skipOut, Zs and Za over 20-30 layers
# (test_parity_bars_cpu.py prints the numbers).  The bar is FP16_K = 8 of those units -- an order of magnitude of
# head-room over what rounding explains, while the mildest mutation of the network we could think of (one bias vector
# zeroed, one layer's dilated tap dropped) moves the same tensors by 100-1000 units.
FP16_U = 2.0 ** -11
FP16_K = 8.0
FP16_MIN_AGREEMENT = 0.98


def fp16_bars(ref, got, sel_bn, what=""):
    """Holds an fp16 run `got` (y [B][N] + last-sample activations) to `ref` = teacher_forced_oracle(.., got["y"]).
    Raises AssertionError naming the first bar that fails; returns the measured statistics.
      * activations: |got - ref| <= FP16_K * u * max|ref| per tensor (per layer for Xout / skipOut), probabilities
        |dP| <= 2 dZa P (d log p = dz - sum p dz);
      * picks: a logit error delta moves every CDF edge by at most 2 delta C (1 - C) <= delta / 2, so a pick that differs
        from the oracle's must be an adjacent bin (two bins when the skipped one is narrower than that) with the draw
        within delta/2 of the oracle's edge, delta = the logit bar; and at least FP16_MIN_AGREEMENT of all picks agree
        (the expected miss rate is the L1 distance of the two CDFs, sum_k |dC_k| ~ A * 1e-4 for the measured errors)."""
    stats = {}
    for k in ("Xout", "skipOut"):
        for l in range(ref[k].shape[0]):
            scale = np.abs(ref[k][l]).max()
            err = np.abs(got[k][l] - ref[k][l]).max()
            stats["%s_units" % k] = max(stats.get("%s_units" % k, 0.0), float(err / (FP16_U * scale)))
            assert err <= FP16_K * FP16_U * scale, "%s %s[%d]: error %.3g = %.1f u*max (bar %.0f)" % (what, k, l, err, err / (FP16_U * scale), FP16_K)
    for k in ("Zs", "Za"):
        scale = np.abs(ref[k]).max()
        err = np.abs(got[k] - ref[k]).max()
        stats["%s_units" % k] = float(err / (FP16_U * scale))
        assert np.all(np.isfinite(got[k])), what + " " + k + ": non-finite"
        assert err <= FP16_K * FP16_U * scale, "%s %s: error %.3g = %.1f u*max (bar %.0f)" % (what, k, err, err / (FP16_U * scale), FP16_K)
    dza = FP16_K * FP16_U * np.abs(ref["Za"]).max()
    perr = np.abs(got["P"] - ref["P"])
    assert np.all(perr <= 2 * dza * ref["P"] + 1e-7), "%s P: probability error %.3g" % (what, perr.max())
    y, y_own = got["y"], ref["y"]
    miss = np.argwhere(y != y_own)
    agree = 1.0 - len(miss) / y.size
    worst = 0.0
    for b, n in miss:
        s = float(sel_bn[b, n])
        near = min(abs(s - float(ref["lo"][b, n])), abs(s - float(ref["hi"][b, n])))
        worst = max(worst, near)
        assert abs(int(y[b, n]) - int(y_own[b, n])) <= 2 and near <= dza / 2, \
            "%s pick (%d,%d): engine %d, oracle %d, draw %.3g from the oracle's CDF edge (bar %.3g)" % (what, b, n, y[b, n], y_own[b, n], near, dza / 2)
    assert agree >= FP16_MIN_AGREEMENT, "%s teacher-forced agreement %.4f < %.2f" % (what, agree, FP16_MIN_AGREEMENT)
    stats.update(agreement=agree, misses=len(miss), worst_edge_distance=worst, logit_bar=float(dza))
    return stats
