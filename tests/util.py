"""Shared helpers for the parity tests. The oracle is the CHECKER here, never the thing tested."""
import os

import numpy as np

from oracle import oracle as O
import cases

# test "mode" names -> nvwOrganisation (wg = exactly one tile per workgroup, the latency kernel as first built)
MODE_ORG = {None: 0, "auto": 0, "wg": 2, "wg2": 3, "stream": 4, "chain": 5, "chain1": 6, "pipe": 7, "wg3": 8}
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
    mode: None = the engine's own choice from the Implementation value and the batch size; "wg" / "wg2" /
    "stream" / "chain" / "chain1" force a kernel organisation (nvw_create_ex)."""
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


def matrix_compare(name, ref, got, tol, relu=False):
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
    atol = 4 * 1.1920929e-07 * np.abs(ref).max()
    ok = np.abs(got - ref) <= tol * np.abs(ref) + atol
    if relu:
        r = (ref <= 0) | (got <= 0)
        ok[r] = (ref[r] < tol) & (got[r] < tol)
    if not ok.all():
        idx = np.argwhere(~ok)[0]
        raise AssertionError("%s mismatch at %s: ref %.10e vs got %.10e (tol %g, atol %.3g, %d bad of %d)" %
                             (name, tuple(idx), ref[tuple(idx)], got[tuple(idx)], tol, atol, (~ok).sum(), ok.size))


def compare_activations(ref, got, tols=None, names=("Xout", "skipOut", "Zs", "Za", "P")):
    tols = tols or cases.TOL
    for k in names:
        matrix_compare(k, ref[k], got[k], tols[k], cases.RELU_AWARE[k])


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
