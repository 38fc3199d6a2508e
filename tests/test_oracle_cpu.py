"""CPU tests (no GPU): the oracle reproduces every committed fixture bit-for-bit.

The fixtures in tests/golden/ were produced by the reference's own nv_wavenet_reference.cpp +
matrix.cpp (tests/golden/make_golden.py); where the prebuilt oracle/_ref library is present the
oracle is additionally pinned against it live.
"""
import numpy as np
import pytest

import cases
import util
from oracle import oracle as O


@pytest.mark.parametrize("case", cases.ALL_CASES, ids=lambda c: c.name)
def test_oracle_matches_golden(case):
    g = util.load_golden(case.name)
    t = util.gen_inputs(case)
    o = util.make_oracle(case, t)
    s = case.shape
    for it in range(case.iters):
        y = o.run(s.N)
        assert np.array_equal(y, g["yOut"][it]), "yOut differs from the fixture"
        got = o.getters()
        crc = [O.crc32(got[k]) for k in ("Xout", "skipOut", "Zs", "Za", "P")]
        assert crc == [int(c) for c in g["crc_act"][it]], "activation CRCs differ from the fixture"
    assert np.array_equal(got["Za"], g["Za"]) and np.array_equal(got["P"], g["P"])
    o.close()


@pytest.mark.skipif(not O.have_ref(), reason="oracle/_ref (reference build) not present")
@pytest.mark.parametrize("case", [c for c in cases.REF_CASES if c.shape.R <= 64][:6], ids=lambda c: c.name)
def test_oracle_matches_reference_build_live(case):
    s = case.shape
    t1 = O.gen_test_inputs(case.seed, case.prior, s, "oracle")
    t2 = O.gen_test_inputs(case.seed, case.prior, s, "ref")
    for a, b in zip(t1.arrays(), t2.arrays()):
        assert np.array_equal(a, b)
    o, r = util.make_oracle(case, t1), O.RefOracle(s.L, s.B, s.N, s.R, s.S, s.A, s.maxD)
    r.set_model(t1)
    r.set_inputs(t1.Lh, t1.sel)
    for _ in range(case.iters):
        assert np.array_equal(o.run(s.N), r.run(s.N))
        go, gr = o.getters(), r.getters()
        for k in go:
            assert np.array_equal(go[k], gr[k]), k
    o.close(), r.close()


@pytest.mark.skipif(not O.have_ref(), reason="oracle/_ref (reference build) not present")
@pytest.mark.parametrize("shape", [cases.Shape(64, 256, 256, 20, 16, 24, 8), cases.Shape(128, 256, 256, 30, 8, 12, 512),
                                   cases.Shape(32, 128, 256, 8, 5, 40, 4)], ids=["C3", "C4", "R32_ragged"])
def test_oracle_matches_reference_build_live_on_the_o1_recipe(shape):
    """The same pin on inputs at the magnitudes of a trained network (tests/util.py: o1_recipe), where the samples depend on
    every part of the network and the gate, the ReLUs and the softmax leave their linear range: the restatement and the
    reference's own nv_wavenet_reference.cpp + matrix.cpp still agree bit for bit (samples and every activation)."""
    case = cases.Case("o1_pin", 30, [], shape, 3, 1, shape.N)
    s = case.shape
    t = util.gen_o1(case, half=False)
    o, r = util.make_oracle(case, t), O.RefOracle(s.L, s.B, s.N, s.R, s.S, s.A, s.maxD)
    r.set_model(t)
    r.set_inputs(t.Lh, t.sel)
    y = o.run(s.N)
    assert np.array_equal(y, r.run(s.N)) and len(np.unique(y)) > min(100, y.size // 3)
    go, gr = o.getters(), r.getters()
    for k in go:
        assert np.array_equal(go[k], gr[k]), k
    assert float(np.abs(go["Za"]).max()) > 0.3, "logits of order one"
    o.close(), r.close()


def test_oracle_chunked_equals_single_run_and_teacher_forcing():
    """run(a)+run(b) from one setInputs is NOT run(a+b) in the reference (sample index restarts, so
    the dilated history is logically cleared); teacher forcing with the oracle's own output
    reproduces it exactly and reports CDF edges that bracket the draw."""
    case = cases.BY_NAME["C1_R32S128A256_L8_B1"]
    t = util.gen_inputs(case)
    s = case.shape
    o = util.make_oracle(case, t)
    y, lo, hi = o.run(s.N, edges=True)
    sel = t.sel.T  # [B][N]
    assert np.all(lo <= sel) and np.all(sel < hi)
    o2 = util.make_oracle(case, t)
    y2 = o2.run(s.N, forced=y)
    assert np.array_equal(y, y2)
    o.close(), o2.close()


def test_oracle_partial_batch():
    """batch_size < maxBatch keeps the maxBatch stride of Lh / selectors (nv_wavenet.cuh:144)."""
    case = cases.BY_NAME["R32S128A256_impl1"]
    t = util.gen_inputs(case)
    s = case.shape
    o = util.make_oracle(case, t)
    full = o.run(s.N)
    o2 = util.make_oracle(case, t)
    part = o2.run(s.N, batch_size=5)
    assert np.array_equal(full[:5], part[:5])
    o.close(), o2.close()


def test_philox_known_answers():
    """Random123's kat_vectors for philox4x32-10 pin the generator behind the in-kernel selectors."""
    kat = [
        ([0, 0, 0, 0], [0, 0], [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]),
        ([0xffffffff] * 4, [0xffffffff] * 2, [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]),
        ([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], [0xa4093822, 0x299f31d0],
         [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]),
    ]
    for ctr, key, want in kat:
        assert O.philox4x32_10(ctr, key) == want
    sel = O.philox_selectors(0x299f31d0a4093822, 5, 7)
    assert sel.shape == (5, 7) and sel.min() >= 0.0 and sel.max() < 1.0
    w0 = O.philox4x32_10([3, 2, 0, 0], [0xa4093822, 0x299f31d0])[0]
    assert sel[3, 2] == np.float32((w0 >> 8) / 16777216.0)
    big = O.philox_selectors(1, 256, 64)
    assert abs(float(big.mean()) - 0.5) < 0.01 and len(np.unique(big)) > 16000


def test_mulaw_pcm_table_matches_reference_python():
    """tests/golden/mulaw_pcm.npz was produced by the reference's utils.mu_law_decode_numpy +
    inference.py's int16 cast (tests/golden/make_mulaw_golden.py)."""
    g = util.load_golden("mulaw_pcm")
    for A in (256, 512, 1024):
        assert np.array_equal(O.mulaw_pcm_table(A), g["pcm_%d" % A])
