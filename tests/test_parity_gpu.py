"""GPU parity tests (run with -m gpu on the MI355X box): the HIP engine, reached through the
C ABI, against the CPU oracle on the same seeded inputs and against the committed fixtures.

fp32: the reference's own bar (nv_wavenet_test.cu:273-304): per-layer activations within
1e-2 (Xout, skipOut), head within 1e-4 (Zs, Za), probabilities 1e-3, and EXACT sample indices
over the whole horizon, for all four Implementation values, with host and device pointers.
"""
import numpy as np
import pytest

import cases
import util

pytestmark = pytest.mark.gpu


def _bspb(B):
    return 4 if B % 4 == 0 else 2 if B % 2 == 0 else 1  # nv_wavenet_test.cu:247


# the kernel organisations: wavenet_wg with one / two / three tiles of 16 utterances per workgroup (wn_kernels.hpp; the engine
# picks by batch size: beyond one / two tiles per CU) and the multi-CU chain with resident weights (wn_chain.hpp; "chain": as
# many layers per CU as stay resident, "chain1": one layer per CU)
# (wn::wavenet_bcast of rounds 3-4 -- every wave its own tile, weights broadcast through an LDS ring -- was removed in round 5)
MODES = ["wg", "wg2", "chain"]
ALL_MODES = MODES + ["chain1"]
FP16_MODES = ["wg", "wg2", "wg3", "chain", "chain1"]
KERNEL_OF = {"wg": "wavenet_wg<", "wg2": "wavenet_wg<", "wg3": "wavenet_wg<", "chain": "wavenet_chain<", "chain1": "wavenet_chain<"}


def _check_mode(e, mode, shape, precision=32):
    """The engine reports the device code it launches: the forced organisation must be the one that ran
    (the chain does not exist for shapes whose single layer exceeds a CU: R = 256)."""
    info = e.kernelInfo()
    if mode in ("chain", "chain1") and shape.R >= 256:
        assert "wavenet_wg<" in info, info
        return
    assert KERNEL_OF[mode] in info, (mode, info)
    if mode == "wg2" and shape.R < 128:   # (two tiles of R >= 128 do not fit the LDS of one workgroup: one tile runs)
        assert "BT=2" in info, info
    if mode == "wg3" and shape.R == 64 and shape.A == 256 and precision == 16:   # (three tiles: fp16, R <= 64, where the LDS holds them)
        assert "BT=3" in info, info
    if mode == "chain1":
        assert "layers/stage=1 " in info, info


@pytest.mark.parametrize("mode", ALL_MODES)
@pytest.mark.parametrize("case", cases.REF_CASES, ids=lambda c: c.name)
def test_reference_harness_fp32(case, mode):
    """Re-creation of runTest<float,float,R,S,A> (nv_wavenet_test.cu:44-329): 2 iterations from one
    setInputs, run_chunks(7, ...) so a 7+1 split and an init_sample != 0 relaunch are exercised."""
    s = case.shape
    g = util.load_golden(case.name)
    t = util.gen_inputs(case)
    o = util.make_oracle(case, t)
    # pointer permutations like nv_wavenet_test.cu:359-365: odd cases upload from device memory
    e = util.make_engine(case, t, precision=32, device_ptrs=(case.impl % 2 == 0), mode=mode)
    _check_mode(e, mode, s)
    for it in range(case.iters):
        y_ref = o.run(s.N)
        y = np.full((s.B, s.N), -1, dtype=np.int32)
        chunks = []
        assert e.run_chunks(case.chunk, lambda yo, i, n: chunks.append((i, n)), s.N, s.B, y, _bspb(s.B))
        e.synchronize()
        assert chunks == [(i, min(case.chunk, s.N - i)) for i in range(0, s.N, case.chunk)]
        util.compare_activations(o.getters(), util.engine_getters(e, s.L))
        assert np.array_equal(y, y_ref), "sample indices differ from the oracle (iteration %d)" % it
        assert np.array_equal(y, g["yOut"][it]), "sample indices differ from the reference fixture"
    assert e.chainStatus() == 0
    e.close(), o.close()


@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("case", cases.EXTRA_CASES, ids=lambda c: c.name)
def test_baseline_config_shapes_fp32(case, mode):
    """BASELINE.json config shapes over long horizons (ring wrap-around, d up to 512, ragged batch).
    Exact indices expected; a divergence is accepted only when the draw is within 1e-5 of a CDF
    edge of the oracle's pick (then that utterance's later samples legitimately differ)."""
    s = case.shape
    g = util.load_golden(case.name)
    t = util.gen_inputs(case)
    o = util.make_oracle(case, t)
    e = util.make_engine(case, t, precision=32, mode=mode)
    _check_mode(e, mode, s)
    y_ref, lo, hi = o.run(s.N, edges=True)
    assert np.array_equal(y_ref, g["yOut"][0])
    y = np.full((s.B, s.N), -1, dtype=np.int32)
    assert e.run_chunks(case.chunk, None, s.N, s.B, y, _bspb(s.B))
    e.synchronize()
    diverged, unexplained = util.explain_mismatches(y_ref, y, lo, hi, t.sel.T, 1e-5)
    assert not unexplained, "unexplained sample mismatches (b,t,ref,got,edge distance): %s" % unexplained[:5]
    if diverged == 0:
        util.compare_activations(o.getters(), util.engine_getters(e, s.L))
    assert diverged <= max(1, s.B // 8), "%d of %d utterances diverged" % (diverged, s.B)
    e.close(), o.close()


def test_run_equals_run_chunks_and_partial_batch():
    """run() == run_chunks() == run_partial() pieces; batch_size < maxBatch generates a prefix."""
    case = cases.BY_NAME["R64S256A256_impl3"]
    s = case.shape
    t = util.gen_inputs(case)
    ys = []
    for mode in ("run", "chunks", "partial"):
        e = util.make_engine(case, t, precision=32)
        y = np.full((s.B, s.N), -1, dtype=np.int32)
        if mode == "run":
            assert e.run(s.N, s.B, y, 4, True)
        elif mode == "chunks":
            assert e.run_chunks(3, None, s.N, s.B, y, 4)
        else:
            # reference idiom: the chunk length is a member set by run_chunks; through the C ABI
            # a partial run without it generates up to num_samples, so emulate with chunks of 1
            assert e.run_chunks(1, None, s.N, s.B, y, 4)
        e.synchronize()
        ys.append(y)
        e.close()
    assert np.array_equal(ys[0], ys[1]) and np.array_equal(ys[0], ys[2])
    e = util.make_engine(case, t, precision=32)
    y = np.full((s.B, s.N), -1, dtype=np.int32)
    assert e.run(s.N, 5, y, 1, False)
    e.synchronize()
    assert np.array_equal(y[:5], ys[0][:5])
    assert np.all(y[5:] <= 0)  # untouched rows (zero-initialised device buffer)
    e.close()


# =====================================================================================================================
# The O(1) recipe (tests/util.py): activations and logits of order one, so that the samples depend on every part of the
# network.  tests/test_parity_bars_cpu.py proves on the CPU that the bars used below FAIL for broken networks (dilated
# taps dropped, conditioning dropped, one layer's sigmoid rows negated, ...) and PASS for a model of the engine's rounding.
# =====================================================================================================================
_S = cases.Shape
O1_CASES = {
    # BASELINE.json configs at their full depth and dilation range (the oracle finishes each in well under a minute):
    "C3": cases.Case("C3_o1", 30, [], _S(64, 256, 256, 20, 16, 256, 32), 3, 1, 100),
    "C2": cases.Case("C2_o1", 10, [], _S(64, 128, 256, 20, 4, 1100, 512), 1, 1, 300),     # every ring wraps, d = 512 live twice
    "C4": cases.Case("C4_o1", 50, [], _S(128, 256, 256, 30, 8, 600, 512), 4, 1, 256),     # 30 layers, R = 128
    # stand-ins for the other instantiations and the awkward shapes
    "R32": cases.Case("R32_o1", 3, [], _S(32, 128, 256, 20, 16, 64, 8), 1, 1, 30),
    "A512": cases.Case("A512_o1", 70, [], _S(64, 128, 512, 20, 16, 32, 8), 3, 1, 15),
    "A1024": cases.Case("A1024_o1", 71, [], _S(128, 256, 1024, 12, 16, 24, 8), 3, 1, 11),
    "R256": cases.Case("R256_o1", 90, [], _S(256, 256, 256, 6, 5, 24, 8), 3, 1, 10),
    "oddL_ragged": cases.Case("oddL_o1", 91, [], _S(64, 128, 256, 7, 19, 40, 4), 1, 1, 13),
    "S8R": cases.Case("S8R_o1", 93, [], _S(32, 256, 256, 6, 5, 24, 8), 3, 1, 10),
}
_ref_cache = {}


def _teacher_forced_ref(case, t, y, tanh_embed=True):
    """util.teacher_forced_oracle, memoised on the engine's samples: the organisations produce bit-identical samples, so
    the (slow) oracle run is shared between them."""
    key = (case.name, tanh_embed, t.crc(), y.tobytes())       # (t.crc(): fp32 and fp16-rounded inputs are different models)
    if key not in _ref_cache:
        _ref_cache[key] = util.teacher_forced_oracle(case, t, y, tanh_embed)
    return _ref_cache[key]


def _engine_o1(case, t, precision, mode, B=None, tanh_embed=True, Lh=None, sel=None):
    from nv_wavenet_amd import WavenetEngine
    s = case.shape
    B = s.B if B is None else B
    e = WavenetEngine(s.R, s.S, s.A, s.L, s.maxD, B, s.N, impl=case.impl, tanhEmbed=tanh_embed, precision=precision,
                      organisation=util.MODE_ORG[mode])
    e.setEmbeddings(t.embP, t.embC)
    for l in range(s.L):
        e.setLayerWeights(l, t.Wprev[l], t.Wcur[l], t.Bh[l], t.Wres[l], t.Bres[l], t.Wskip[l], t.Bskip[l])
    e.setOutWeights(t.Wzs, t.Bzs, t.Wza, t.Bza)
    e.setInputs(t.Lh if Lh is None else Lh, t.sel if sel is None else sel)
    return e


def _run_dumped(e, case, B=None, chunk=None):
    """Free run with the activation dump on (run_chunks: several launches, the last one dumps): samples + getters."""
    s = case.shape
    B = s.B if B is None else B
    y = np.full((B, s.N), -1, dtype=np.int32)
    assert e.run_chunks(chunk or case.chunk, None, s.N, B, y, 1, True)
    e.synchronize()
    assert e.chainStatus() == 0
    got = util.engine_getters(e, s.L)
    got["y"] = y
    return got


def _fp16_checked_run(name, mode, record_property=None):
    """fp16 engine in organisation `mode` on the O(1) inputs of O1_CASES[name], dump on, held to the fp32 oracle (fed the
    same fp16-rounded parameters and the engine's samples) by util.fp16_bars.  Returns (inputs, samples)."""
    case = O1_CASES[name]
    t = util.gen_o1(case, half=True)
    e = _engine_o1(case, t, 16, mode)
    _check_mode(e, mode, case.shape, 16)
    got = _run_dumped(e, case)
    ref = _teacher_forced_ref(case, t, got["y"])
    st = util.fp16_bars(ref, got, t.sel.T, "%s/%s" % (name, mode))
    print("fp16 %s %s: %s" % (name, mode, {k: round(v, 4) for k, v in st.items()}))
    if record_property:
        for k, v in st.items():
            record_property("fp16_" + k, v)
    # the production launch (dumpActivations = false) runs a kernel variant without any dump code:
    # it must generate the same samples from the same inputs, in one launch
    e.setInputs(t.Lh, t.sel)
    y2 = np.full((case.shape.B, case.shape.N), -1, dtype=np.int32)
    assert e.run(case.shape.N, case.shape.B, y2, 1, False)
    e.synchronize()
    assert e.chainStatus() == 0 and e.chainFallbacks() == 0
    assert np.array_equal(y2, got["y"]), "dump and no-dump kernel variants disagree"
    e.close()
    return t, got["y"]


@pytest.mark.parametrize("mode", FP16_MODES)
@pytest.mark.parametrize("name", sorted(O1_CASES))
def test_fp16_engine_against_the_oracle_o1(name, mode, record_property):
    """THE fp16 parity test (unpinned by the reference, which has no half test; what fp16 must approximate:
    nv_wavenet_util.cuh:78-86, matrix_math.cuh:119-157; what is compared: nv_wavenet_test.cu:259-304).  Every organisation,
    at the BASELINE configs proper (C2 with maxDilation 512 over 1100 samples, C3, C4 with 30 layers) and the other
    instantiations, on inputs whose samples depend on the whole network: per-layer residual stream and skip sums, Zs,
    logits and probabilities of the last sample within FP16_K * 2^-11 * max|tensor| of the fp32 oracle, >= 98 % of all
    picks identical to the oracle's own pick given the same history, every other one a neighbouring bin with the draw
    within (logit bar)/2 of the oracle's CDF edge."""
    _fp16_checked_run(name, mode, record_property)


@pytest.mark.parametrize("name", ["C2", "C3", "C4"])
def test_fp16_organisations_are_bit_identical_o1(name):
    """Same arithmetic in the same order in every organisation: the free-running fp16 samples are IDENTICAL."""
    ys = {m: _fp16_checked_run(name, m)[1] for m in (["wg", "chain"] if name == "C4" else ["wg", "wg3", "chain"])}
    first = ys.pop("wg")
    for m, y in ys.items():
        assert np.array_equal(first, y), "wavenet_wg and %s disagree in fp16" % m


@pytest.mark.parametrize("mode", ALL_MODES)
@pytest.mark.parametrize("name", ["C3", "C2", "C4", "R32", "A1024", "oddL_ragged"])
def test_fp32_engine_o1_exact_samples_and_reference_bars(name, mode):
    """fp32 on the O(1) recipe: now the samples depend on the network (on the reference's recipe they test softmax + scan + Bza
    only), and they must be the oracle's EXACTLY; a divergence is accepted only when the draw is within 1e-5 of a CDF
    edge of the oracle's pick.  Activations: the reference harness's own bars (nv_wavenet_test.cu:273-298)."""
    case = O1_CASES[name]
    s = case.shape
    if name == "C2":
        case = case._replace(shape=s._replace(N=600))      # (d = 512 live, rings wrapped; the fp16 test runs the full 1100)
        s = case.shape
    t = util.gen_o1(case, half=False)
    e = _engine_o1(case, t, 32, mode)
    _check_mode(e, mode, s)
    got = _run_dumped(e, case)
    ref = _teacher_forced_ref(case, t, got["y"])
    diverged, unexplained = util.explain_mismatches(ref["y"], got["y"], ref["lo"], ref["hi"], t.sel.T, 1e-5)
    assert not unexplained, "unexplained sample mismatches (b,t,ref,got,edge distance): %s" % unexplained[:5]
    assert diverged <= max(1, s.B // 8), "%d utterances diverged" % diverged
    assert (ref["y"] == got["y"]).mean() >= 0.9995        # teacher-forced: a miss does not propagate
    util.compare_activations(ref, got, atol_eps=32)
    e.close()


@pytest.mark.parametrize("mode", ["wg", "wg3", "chain"])
def test_a_new_utterance_starts_from_silence(mode):
    """An engine that has generated one utterance (rings full of ITS activations) generates the next one, from other conditioning
    and a shorter batch, exactly as a fresh engine does: the taps x[t-d] of t < d are zero (reference nv_wavenet_persistent.cuh:287).
    wavenet_wg reads them from ring slots that the engine clears for a new utterance (resetHistory, i.e. every setInputs-like
    call; a launch at sample 0 without one -- which keeps the sample history, as the reference's does -- clears them itself);
    the chain substitutes zeros itself."""
    case = O1_CASES["C3"]
    s = case.shape
    a = util.gen_o1(case, half=True)
    b = util.gen_o1(case._replace(seed=case.seed + 1), half=True)
    e = _engine_o1(case, a, 16, mode)
    y = np.full((s.B, s.N), -1, dtype=np.int32)
    assert e.run(s.N, s.B, y, 1, False)
    e.setInputs(b.Lh, b.sel)
    nb = max(1, s.B - 5)
    y2 = np.full((s.B, s.N), -1, dtype=np.int32)              # (the output block is [maxBatch][N]; a shorter batch fills a prefix)
    assert e.run(s.N, nb, y2, 1, False)
    e.synchronize()
    e.close()
    f = _engine_o1(case, a, 16, mode, Lh=b.Lh, sel=b.sel)      # same weights, the second utterance's inputs
    y3 = np.full((s.B, s.N), -1, dtype=np.int32)
    assert f.run(s.N, nb, y3, 1, False)
    f.synchronize()
    f.close()
    assert not np.array_equal(y[:nb], y2[:nb])
    assert np.array_equal(y2[:nb], y3[:nb])


@pytest.mark.parametrize("mode", ["wg3", "wg2", "auto"])
def test_a_larger_batch_after_a_smaller_one_starts_from_silence(mode):
    """The batch GROWS between utterances: a workgroup of two or three tiles stores into the rings of ALL its tiles, the
    padding tiles beyond the batch included, so a small batch dirties ring slots of tiles it does not generate.  A following utterance
    with a larger batch must find them zero (taps x[t-d] of t < d: reference nv_wavenet_persistent.cuh:287): same samples as a fresh
    engine.  (The engine records the tiles a launch touches, rounded up to its tiles per workgroup.)"""
    case = O1_CASES["C3"]
    s = case.shape
    a = util.gen_o1(case, half=True)
    b = util.gen_o1(case._replace(seed=case.seed + 1), half=True)
    B = 48
    idx = np.arange(B) % s.B
    rep = lambda t: (np.ascontiguousarray(t.Lh[:, :, idx, :]), np.ascontiguousarray(t.sel[:, idx]))
    LhA, selA = rep(a)
    LhB, selB = rep(b)
    e = _engine_o1(case, a, 16, mode, B=B, Lh=LhA, sel=selA)
    y = np.full((B, s.N), -1, dtype=np.int32)
    assert e.run(s.N, 16, y, 1, False)                         # one tile of the batch; a three-tile workgroup writes the rings of three
    e.setInputs(LhB, selB)
    y2 = np.full((B, s.N), -1, dtype=np.int32)
    assert e.run(s.N, B, y2, 1, False)
    e.synchronize()
    e.close()
    f = _engine_o1(case, a, 16, mode, B=B, Lh=LhB, sel=selB)
    y3 = np.full((B, s.N), -1, dtype=np.int32)
    assert f.run(s.N, B, y3, 1, False)
    f.synchronize()
    f.close()
    assert np.array_equal(y2, y3)
    assert np.array_equal(y2[:16], y2[16:32]) and np.array_equal(y2[:16], y2[32:])      # (the three tiles repeat the same utterances)


# (C3's test shape has maxDilation 32: four layers of dilation 1, whose slots do not fit beside three tiles' images: C3_full below;
#  C4, R = 128 with 30 layers, fills the LDS with its tables: no ring slot fits, the launch is the plain one)
RING_O1_CASES = dict(O1_CASES)
RING_O1_CASES["C3_full"] = cases.Case("C3_full_o1", 30, [], _S(64, 256, 256, 20, 16, 600, 512), 3, 1, 128)      # C3 at BASELINE's dilation range
RING_CASES = [("C3", "wg", 16), ("C3", "wg2", 16), ("C3_full", "wg3", 16), ("C3_full", "wg4", 16), ("C2", "wg", 16), ("C2", "wg2", 16), ("C3", "wg", 32),
              ("R32", "wg", 32), ("oddL_ragged", "wg2", 16)]


@pytest.mark.parametrize("name,mode,precision", RING_CASES)
def test_ring_slots_in_lds_generate_the_same_samples(name, mode, precision):
    """The dilation ring staged in LDS (round 6; north_star "ring buffer staged in LDS with coalesced HBM spill"; the reference stages
    x[t-d] through shared memory out of a global ring, nv_wavenet.cuh:96-127,334-335): a wavenet_wg launch keeps the slots of as many
    short-dilation layers as fit in LDS (the default; kernelInfo says up to which dilation; setRingInLds(-1) switches it off), loads
    them from the HBM ring at its start and spills them back at its end.  Chunked runs (the state crosses launches through the HBM ring), a run that continues
    an utterance generated with the ring in HBM, and the other way round, must all produce the samples of the plain kernel -- which
    the oracle holds (test_fp16_engine_against_the_oracle_o1 / test_fp32_engine_o1_*)."""
    case = RING_O1_CASES[name]
    s = case.shape
    t = util.gen_o1(case, half=(precision == 16))
    e0 = _engine_o1(case, t, precision, mode)
    e0.setRingInLds(-1)
    assert "LR" not in e0.kernelInfo(s.B, False)
    y0 = np.full((s.B, s.N), -1, dtype=np.int32)
    assert e0.run(s.N, s.B, y0, 1, False)
    e0.synchronize()
    e0.close()
    e = _engine_o1(case, t, precision, mode)
    info = e.kernelInfo(s.B, False)                 # (the default)
    assert "LR=1" in info and "ring_in_lds=d<=" in info, info
    held = int(info.split("ring_in_lds=d<=")[1].split()[0])
    assert held >= 1
    for chunk in (None, max(3, s.N // 5)):
        e.setInputs(t.Lh, t.sel)
        y = np.full((s.B, s.N), -1, dtype=np.int32)
        if chunk:
            assert e.run_chunks(chunk, None, s.N, s.B, y, 1)
        else:
            assert e.run(s.N, s.B, y, 1, False)
        e.synchronize()
        assert np.array_equal(y, y0), "ring in LDS (d <= %d), chunk %s: samples differ" % (held, chunk)
    # the first half of the utterance with the ring in LDS, the second with it in HBM, and the other way round
    half = s.N // 2
    for first_mode, second_mode in ((0, -1), (-1, 0)):
        e.setInputs(t.Lh, t.sel)
        e.setRingInLds(first_mode)
        assert e.run_partial_chunk(0, half, s.N, s.B)
        e.setRingInLds(second_mode)
        assert e.run_partial_chunk(half, s.N - half, s.N, s.B)
        y = np.full((s.B, s.N), -1, dtype=np.int32)
        e.getYOut(y, 0, s.N)
        e.synchronize()
        assert np.array_equal(y, y0), "ring %s then %s: samples differ" % (first_mode, second_mode)
    e.close()


def test_whole_ring_in_lds_when_it_fits():
    """A model whose WHOLE ring fits the LDS a workgroup's tables leave free (short maxDilation: R64 S128 A256, 7 layers, maxDilation
    4, fp16) keeps all of it there -- the launch touches the HBM ring at its two ends only -- and generates the samples of the
    launch that keeps the ring in HBM (held to the oracle by test_fp16_engine_against_the_oracle_o1)."""
    case = O1_CASES["oddL_ragged"]
    s = case.shape
    t = util.gen_o1(case, half=True)
    ys = []
    for ring_mode in (0, -1):
        e = _engine_o1(case, t, 16, "wg")
        e.setRingInLds(ring_mode)
        info = e.kernelInfo(s.B, False)
        if ring_mode == 0:
            assert "LR=1" in info and "ring_in_lds=d<=%d" % s.maxD in info, info
        else:
            assert "LR" not in info, info
        y = np.full((s.B, s.N), -1, dtype=np.int32)
        assert e.run_chunks(case.chunk, None, s.N, s.B, y, 1)
        e.synchronize()
        e.close()
        ys.append(y)
    assert np.array_equal(ys[0], ys[1])


def test_chain_fills_the_gpu_by_replication():
    """The multi-CU chain with as many chains as the GPU holds (C3 fp16: 5 workgroups per 16 utterances, all of
    them resident at once, chains spread over every XCD so that some hand-offs cross XCDs): 50 tiles that repeat the
    16 utterances of the oracle-checked case must repeat its samples bit for bit, chunked."""
    import torch
    case = O1_CASES["C3"]
    s = case.shape
    t, y16 = _fp16_checked_run("C3", "wg")
    ncu = torch.cuda.get_device_properties(0).multi_processor_count
    tiles = ncu // 5 - 1
    B = 16 * tiles - 3                                   # ragged last tile
    idx = np.arange(B) % s.B
    e = _engine_o1(case, t, 16, None, B=B, Lh=np.ascontiguousarray(t.Lh[:, :, idx, :]), sel=np.ascontiguousarray(t.sel[:, idx]))
    info = e.kernelInfo(B, False)
    assert "wavenet_chain<" in info and "chains=%d" % tiles in info, info
    y = np.full((B, s.N), -1, dtype=np.int32)
    assert e.run_chunks(100, None, s.N, B, y, 1)
    e.synchronize()
    assert e.chainStatus() == 0 and e.chainFallbacks() == 0
    bad = np.argwhere((y != y16[idx]).any(axis=1))
    assert bad.size == 0, "utterance %d differs" % int(bad[0, 0])
    e.close()


@pytest.mark.parametrize("name,tpc,precision", [("C3", 3, 16), ("C4", 5, 16), ("C4", 8, 16), ("C3", 2, 32)])
def test_chain_with_several_tiles_per_chain(name, tpc, precision):
    """Round 5: batches beyond the chains that are resident at once ride the same chains, `tpc` tiles per chain (every stage works
    through its chain's tiles in turn; role of the reference's persistent blocks looping over the whole batch,
    nv_wavenet_persistent.cuh:110).  Ragged: the last chains hold one tile fewer and the last tile is partial.  Every utterance
    must repeat, bit for bit, what the one-tile kernel generates for the case's utterances alone (which the oracle holds)."""
    import torch
    import re
    case = O1_CASES[name]
    s = case.shape
    if name == "C4":
        case = case._replace(shape=s._replace(N=160))
        s = case.shape
    t = util.gen_o1(case, half=precision == 16)
    e0 = _engine_o1(case, t, precision, "wg")
    y0 = _run_dumped(e0, case)["y"]
    e0.close()
    ncu = torch.cuda.get_device_properties(0).multi_processor_count
    probe = _engine_o1(case, t, precision, "chain")
    stages = int(re.search(r"stages=(\d+)", probe.kernelInfo()).group(1))
    probe.close()
    chains = ncu // stages
    tiles = chains * tpc - chains // 2                   # the upper half of the chains gets tpc - 1 tiles
    B = 16 * tiles - 5                                   # ragged last tile
    idx = np.arange(B) % s.B
    e = _engine_o1(case, t, precision, "chain", B=B, Lh=np.ascontiguousarray(t.Lh[:, :, idx, :]), sel=np.ascontiguousarray(t.sel[:, idx]))
    info = e.kernelInfo(B, False)
    assert "wavenet_chain<" in info and "chains=%d tiles/chain=%d " % (chains, tpc) in info, info
    # (round 6: beyond four tiles per chain the two-layer stages of R = 128 run the instantiation that requests a unit's conditioning up front)
    assert ("HOIST=1" in info) == (name == "C4" and tpc > 4), info
    y = np.full((B, s.N), -1, dtype=np.int32)
    assert e.run_chunks(case.chunk, None, s.N, B, y, 1)
    e.synchronize()
    assert e.chainStatus() == 0 and e.chainFallbacks() == 0
    bad = np.argwhere((y != y0[idx]).any(axis=1))
    assert bad.size == 0, "utterance %d differs" % int(bad[0, 0])
    e.close()


def test_chain_launch_that_cannot_become_resident_is_rerun_on_wavenet_wg():
    """Fault tolerance of the multi-CU organisation (the reference's pollers spin without bound,
    nv_wavenet_persistent.cuh:80-93).  A kernel on another stream holds most CUs, so only part of a 20-chain launch (100
    workgroups) becomes resident: its pollers run into their bound (set to 60 ms here), the launch gives up, and the engine
    -- in stream order, no host involvement -- restores rings and history and re-runs the launch's samples on wavenet_wg.
    The caller gets the same samples as from an undisturbed run (which the oracle checks), run() returns true,
    chainStatus() stays 0 and chainFallbacks() counts the event."""
    import ctypes as C
    import os
    import torch
    case = O1_CASES["C3"]._replace(shape=O1_CASES["C3"].shape._replace(N=96))
    s = case.shape
    t = util.gen_o1(case, half=True)
    tiles = 20
    B = 16 * tiles
    idx = np.arange(B) % s.B
    Lh, sel = np.ascontiguousarray(t.Lh[:, :, idx, :]), np.ascontiguousarray(t.sel[:, idx])
    e = _engine_o1(case, t, 16, "chain", B=B, Lh=Lh, sel=sel)
    assert "wavenet_chain<" in e.kernelInfo(B, False)
    y_quiet = np.full((B, s.N), -1, dtype=np.int32)
    assert e.run_chunks(40, None, s.N, B, y_quiet, 1)
    e.synchronize()
    assert e.chainStatus() == 0 and e.chainFallbacks() == 0
    # the undisturbed run is the oracle-checked one (16 utterances repeated)
    ref = _teacher_forced_ref(case, t, y_quiet[:16])
    assert (ref["y"] == y_quiet[:16]).mean() >= util.FP16_MIN_AGREEMENT
    assert np.array_equal(y_quiet, y_quiet[:16][idx])

    prim = C.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "cpp", "libwn_primitives.so"))
    prim.wnp_hog_start.argtypes = [C.c_int, C.c_double]
    ncu = torch.cuda.get_device_properties(0).multi_processor_count
    e.setChainTimeoutMs(60.0)
    e.setInputs(Lh, sel)
    assert prim.wnp_hog_start(ncu - 48, 1200.0) == 0          # 48 CUs stay free: fewer than the 100 workgroups of the launch
    import time
    time.sleep(0.05)                                         # (the hog is resident before the chain is launched)
    y = np.full((B, s.N), -1, dtype=np.int32)
    side = torch.cuda.Stream()
    assert e.run(s.N, B, y, 1, False, side.cuda_stream)
    side.synchronize()
    assert prim.wnp_hog_wait() == 0
    assert e.chainStatus() == 0, "the give-up must have been repaired"
    assert e.chainFallbacks() >= 1, "the launch was expected to give up under the CU hog"
    assert e.chainLastTimeout() != 0
    assert np.array_equal(y, y_quiet), "samples after the fallback differ from the undisturbed run"
    # ... and the engine is healthy afterwards: an undisturbed chunked run continues to work
    e.setChainTimeoutMs(1500.0)
    e.setInputs(Lh, sel)
    n0 = e.chainFallbacks()
    y3 = np.full((B, s.N), -1, dtype=np.int32)
    assert e.run_chunks(40, None, s.N, B, y3, 1)
    e.synchronize()
    assert e.chainFallbacks() == n0 and np.array_equal(y3, y_quiet)
    e.close()


@pytest.mark.parametrize("B", [16, 40, 100])
def test_fp16_three_tiles_per_workgroup_identical(B):
    """wavenet_wg with three tiles of 16 utterances per workgroup (the launch shape beyond two tiles per CU) against the
    one-tile kernel, which the oracle checks: one tile in a group of three, one partly filled group, several groups; one
    launch and chunked; packed conditioning and conditioning read in place (fp32 and fp16 tensors)."""
    import torch
    case = O1_CASES["C3"]
    s = case.shape
    t, y16 = _fp16_checked_run("C3", "wg")
    idx = np.arange(B) % s.B
    Lh = np.ascontiguousarray(t.Lh[:, :, idx, :])
    sel = np.ascontiguousarray(t.sel[:, idx])
    e = _engine_o1(case, t, 16, "wg3", B=B, Lh=Lh, sel=sel)
    assert "BT=3" in e.kernelInfo(B, False), e.kernelInfo(B, False)
    Lh32 = torch.from_numpy(Lh).cuda()
    Lh16 = Lh32.half()
    for chunk, direct in ((None, None), (100, None), (None, Lh32), (None, Lh16), (100, Lh16)):
        e.setInputs(Lh, sel)
        if direct is not None:
            e.setConditioningDirect(direct)
            assert "RAW=%d" % (2 if direct.dtype == torch.float16 else 1) in e.kernelInfo(B, False)
        y = np.full((B, s.N), -1, dtype=np.int32)
        if chunk:
            assert e.run_chunks(chunk, None, s.N, B, y, 1)
        else:
            assert e.run(s.N, B, y, 1, False)
        e.synchronize()
        bad = np.argwhere((y != y16[idx]).any(axis=1))
        assert bad.size == 0, "utterance %d differs (chunk %s, in place %s)" % (int(bad[0, 0]), chunk, None if direct is None else direct.dtype)
    e.close()


@pytest.mark.parametrize("B", [16, 70, 200])
def test_fp16_four_tiles_per_workgroup_identical(B):
    """Round 6: wavenet_wg with FOUR tiles of 16 utterances per workgroup (dump-free launches with packed conditioning) against the
    one-tile kernel, which the oracle checks: one tile in a group of four, partly filled groups (a ragged last tile), several groups;
    one launch and chunked -- the last chunk of run_chunks can dump and therefore runs three tiles per workgroup on the same rings --;
    conditioning read in place takes the three-tile kernels as well (the engine reports which)."""
    import torch
    case = O1_CASES["C3"]
    s = case.shape
    t, y16 = _fp16_checked_run("C3", "wg")
    idx = np.arange(B) % s.B
    Lh = np.ascontiguousarray(t.Lh[:, :, idx, :])
    sel = np.ascontiguousarray(t.sel[:, idx])
    e = _engine_o1(case, t, 16, "wg4", B=B, Lh=Lh, sel=sel)
    assert "BT=4" in e.kernelInfo(B, False) and "BT=3" in e.kernelInfo(B, True), (e.kernelInfo(B, False), e.kernelInfo(B, True))
    Lh16 = torch.from_numpy(Lh).cuda().half()
    for chunk, direct in ((None, None), (100, None), (37, None), (None, Lh16)):
        e.setInputs(Lh, sel)
        if direct is not None:
            e.setConditioningDirect(direct)
            info = e.kernelInfo(B, False)
            assert "RAW=2" in info and "BT=3" in info, info
        y = np.full((B, s.N), -1, dtype=np.int32)
        if chunk:
            assert e.run_chunks(chunk, None, s.N, B, y, 1)
        else:
            assert e.run(s.N, B, y, 1, False)
        e.synchronize()
        bad = np.argwhere((y != y16[idx]).any(axis=1))
        assert bad.size == 0, "utterance %d differs (chunk %s, in place %s)" % (int(bad[0, 0]), chunk, None if direct is None else direct.dtype)
    e.close()


@pytest.mark.parametrize("precision", [32, 16])
@pytest.mark.parametrize("mode", ["wg", "wg2", "wg3", "chain"])
def test_conditioning_consumed_in_place(mode, precision):
    """SURVEY.md 8f rank 1 / pytorch/README.md:44: device-resident conditioning WITHOUT the copy.  setConditioningDirect hands
    the engine the caller's [N][L][B][2R] device tensor -- fp32, or for the fp16 engine its T_data, fp16 (the reference keeps
    m_Lh in T_data, nv_wavenet.cuh:326) -- and the kernels read it in place (no packed copy exists).  O(1) inputs, ragged
    batch (21 utterances), one launch and chunks (the second chunk starts reading mid-tensor); the dump-capable kernels
    are held to the oracle (fp32: exact samples; fp16: util.fp16_bars), the production kernels must reproduce the packed
    path's samples bit for bit, and the caller's tensor is neither modified nor copied."""
    import torch
    if mode == "wg3" and precision == 32:
        pytest.skip("three tiles per workgroup: fp16 instantiations only")
    case = cases.Case("C3_o1_B21", 31, [], cases.Shape(64, 256, 256, 20, 21, 40, 16), 3, 1, 16)
    s = case.shape
    t = util.gen_o1(case, half=(precision == 16))
    e = _engine_o1(case, t, precision, mode)        # packed: setInputs copies and packs
    _check_mode(e, mode, s, precision)
    y_packed = np.full((s.B, s.N), -1, dtype=np.int32)
    assert e.run(s.N, s.B, y_packed, 1, False)
    e.synchronize()
    Lh32 = torch.from_numpy(t.Lh).cuda()
    tensors = [Lh32] + ([Lh32.half()] if precision == 16 else [])
    for Lh in tensors:
        before = Lh.clone()
        # with the dump on: against the oracle
        e.setInputs(t.Lh, t.sel)                 # (selectors; the packed conditioning it leaves behind must NOT be used)
        e.setConditioningDirect(Lh)
        got = _run_dumped(e, case, chunk=16)
        ref = _teacher_forced_ref(case, t, got["y"])
        if precision == 32:
            assert np.array_equal(got["y"], ref["y"])
            util.compare_activations(ref, got, atol_eps=32)
        else:
            util.fp16_bars(ref, got, t.sel.T, "in place %s/%s" % (Lh.dtype, mode))
        assert np.array_equal(got["y"], y_packed), "in-place conditioning (%s, dump kernels) differs from the packed path" % Lh.dtype
        # production kernels, one launch
        e.setInputs(t.Lh, t.sel)
        e.setConditioningDirect(Lh)
        y = np.full((s.B, s.N), -1, dtype=np.int32)
        assert e.run(s.N, s.B, y, 1, False)
        e.synchronize()
        assert e.chainStatus() == 0
        assert np.array_equal(y, y_packed), "in-place conditioning (%s) differs from the packed path" % Lh.dtype
        assert torch.equal(Lh, before), "the caller's tensor was modified"
        # and the packed copy really is not what is read: scribble over the caller's tensor -> different samples
        Lh.neg_()
        e.setInputs(t.Lh, t.sel)
        e.setConditioningDirect(Lh)
        y2 = np.full((s.B, s.N), -1, dtype=np.int32)
        assert e.run(s.N, s.B, y2, 1, False)
        e.synchronize()
        assert (y2 != y_packed).mean() > 0.5
    # an fp32 engine refuses an fp16 tensor instead of misreading it
    if precision == 32:
        with pytest.raises(TypeError):
            e.setConditioningDirect(Lh32.half())
    e.close()


@pytest.mark.parametrize("mode,precision", [("wg", 32), ("chain", 32), ("wg", 16), ("wg2", 16), ("wg3", 16), ("chain", 16)])
def test_conditioning_produced_in_fragment_order(mode, precision):
    """Round 3: conditioning the caller PRODUCES in the engine's fragment order (setConditioningPacked; a model folds the
    channel permutation and the gate's pre-scale into its conditioning convolution, nv_wavenet.py: get_cond_input(layout=
    "packed")): the generation kernels run their packed path -- the headline kernel as it is -- on the caller's buffer, no
    copy, no second pass, no in-place conversion.  O(1) inputs, ragged batch, chunked.  fp32 (no pre-scale, no rounding): the
    samples must be those of setInputs, bit for bit.  fp16: the producer rounds value * prescale to fp16 itself, so the
    engine is held to the fp32 oracle fed exactly the values it was given (fragment / prescale) by util.fp16_bars, and the
    buffer really is what is read (negating it changes the samples; the caller's tensor is not modified)."""
    import torch
    from nv_wavenet_amd.nv_wavenet import pack_cond_input, cond_fragment_order
    case = cases.Case("C3_o1_B37", 31, [], cases.Shape(64, 256, 256, 20, 37, 40, 16), 3, 1, 16)
    s = case.shape
    t = util.gen_o1(case, half=(precision == 16))
    e = _engine_o1(case, t, precision, mode)
    _check_mode(e, mode, s, precision)
    y_set_inputs = np.full((s.B, s.N), -1, dtype=np.int32)
    assert e.run(s.N, s.B, y_set_inputs, 1, False)
    e.synchronize()
    Lh = torch.from_numpy(t.Lh).cuda()
    frags = pack_cond_input(Lh, precision, e.condTiles())
    assert frags.shape[0] == s.N + 1 and frags.shape[2] == e.condTiles() and frags.numel() == (s.N + 1) * s.L * e.condTiles() * 16 * 2 * s.R
    before = frags.clone()
    e.setConditioningPacked(frags)
    e.setSelectors(t.sel)
    got = _run_dumped(e, case, chunk=16)
    assert torch.equal(frags, before), "the caller's buffer was modified"
    if precision == 32:
        assert np.array_equal(got["y"], y_set_inputs), "fragment-order conditioning differs from setInputs in fp32"
        ref = _teacher_forced_ref(case, t, got["y"])
        assert np.array_equal(got["y"], ref["y"])
        util.compare_activations(ref, got, atol_eps=32)
    else:
        # what the engine was given, as the oracle's conditioning: fragment value / prescale, back in channel order
        perm, scale = cond_fragment_order(s.R, 16)
        idx = torch.tensor(perm, device="cuda")
        sc = torch.tensor(scale, dtype=torch.float32, device="cuda")
        given = ((Lh.index_select(3, idx) * sc).half().float() / sc)
        Lh_eff = torch.empty_like(Lh)
        Lh_eff.index_copy_(3, idx, given)
        import copy
        t_eff = copy.copy(t)
        t_eff.Lh = np.ascontiguousarray(Lh_eff.cpu().numpy())
        ref = util.teacher_forced_oracle(case, t_eff, got["y"])
        st = util.fp16_bars(ref, got, t.sel.T, "fragment order %s" % mode)
        print("fragment-order conditioning fp16 %s: %s; identical to setInputs: %.4f" %
              (mode, {k: round(v, 4) for k, v in st.items()}, (got["y"] == y_set_inputs).mean()))
        assert (np.abs(Lh_eff.cpu().numpy() - t.Lh).max() <= 2.0 ** -10 * np.abs(t.Lh).max())     # the same conditioning to an fp16 ulp
    # production kernels, one launch: the same samples
    e.setConditioningPacked(frags)
    e.setSelectors(t.sel)
    y = np.full((s.B, s.N), -1, dtype=np.int32)
    assert e.run(s.N, s.B, y, 1, False)
    e.synchronize()
    assert e.chainStatus() == 0 and np.array_equal(y, got["y"])
    # the buffer is what is read
    neg = (-frags).contiguous()
    e.setConditioningPacked(neg)
    e.setSelectors(t.sel)
    y2 = np.full((s.B, s.N), -1, dtype=np.int32)
    assert e.run(s.N, s.B, y2, 1, False)
    e.synchronize()
    assert (y2 != y).mean() > 0.5
    if precision == 32:
        with pytest.raises(TypeError):
            e.setConditioningPacked(frags.half())
    e.close()


@pytest.mark.parametrize("tiles_per_cu", [2, 3])
def test_benchmarked_launch_is_the_parity_tested_one(tiles_per_cu):
    """What bench.py times by default IS pinned: C3 at BASELINE depth and dilation range (R64/S256/A256, 20 layers,
    maxDilation 512, fp16) at two or three tiles per CU -- the engine then launches wavenet_wg with that many tiles per
    workgroup, no dump code.  The big batch repeats 16 utterances on O(1) inputs (conditioning tiled on the device),
    N = 640 samples so the d = 512 taps are live and the rings wrap, and must reproduce, bit for bit, the 16-utterance
    run of the one-tile kernel -- which is held to the fp32 oracle by util.fp16_bars in this very test."""
    import torch
    import bench
    from nv_wavenet_amd import WavenetEngine
    case = cases.Case("C3_fp16_benchmarked_launch", 30, [], cases.Shape(64, 256, 256, 20, 16, 640, 512), 3, 1, 128)
    s = case.shape
    t = util.gen_o1(case, half=True)
    e0 = _engine_o1(case, t, 16, "wg")
    got = _run_dumped(e0, case)
    e0.close()
    st = util.fp16_bars(_teacher_forced_ref(case, t, got["y"]), got, t.sel.T, "16-utterance run")
    print("checked 16-utterance run:", {k: round(v, 4) for k, v in st.items()})
    y16 = got["y"]
    ncu = torch.cuda.get_device_properties(0).multi_processor_count
    B = tiles_per_cu * 16 * ncu                         # the batches bench.py settles on
    e = WavenetEngine(s.R, s.S, s.A, s.L, s.maxD, B, s.N, impl=0, tanhEmbed=True, precision=16)
    info = e.kernelInfo(B, False)
    assert info.split(" ")[0] == bench.HEADLINE_KERNELS[tiles_per_cu], info
    e.setEmbeddings(t.embP, t.embC)
    for l in range(s.L):
        e.setLayerWeights(l, t.Wprev[l], t.Wcur[l], t.Bh[l], t.Wres[l], t.Bres[l], t.Wskip[l], t.Bskip[l])
    e.setOutWeights(t.Wzs, t.Bzs, t.Wza, t.Bza)
    idx = torch.arange(B, device="cuda") % s.B
    Lh = torch.from_numpy(t.Lh).cuda()[:, :, idx, :].contiguous()          # [N][L][B][2R]
    sel = torch.from_numpy(t.sel).cuda()[:, idx].contiguous()             # [N][B]
    e.setInputs(Lh, sel)
    del Lh
    y = torch.full((B, s.N), -1, dtype=torch.int32, device="cuda")
    assert e.run(s.N, B, y, 1, False)
    e.synchronize()
    y = y.cpu().numpy()
    ref = y16[idx.cpu().numpy()]
    bad = np.argwhere((y != ref).any(axis=1))
    assert bad.size == 0, "utterance %d of the benchmarked launch differs from the 16-utterance run" % int(bad[0, 0])
    e.close()


@pytest.mark.parametrize("tiles_per_cu", [0, 3, 4, 8])
def test_benchmarked_path_exactly(tiles_per_cu):
    """the launch sequence bench.py times (bench.py: steady_engine + step) reproduced to the letter -- fp16, O(1)
    weights, in-kernel Philox selectors (setSelectorSeed), conditioning packed chunk-wise from one reused 64-sample block
    (packConditioning), run_range(0, 640) and then run_range(640, n): what is compared is nv_wavenet_test.cu:259-304's,
    what the selectors replace is wavenet_infer.cu:92-94's rand() table.  tiles_per_cu = 0: 16 utterances, held to the fp32
    oracle fed philox_selectors(seed) and the same conditioning by util.fp16_bars (the last launch with the dump on; the
    dump-free kernels must generate the same samples); tiles_per_cu = 3 / 4: 3 / 4 x 16 x CUs utterances on the three- / four-tile
    kernel bench.py asserts, every utterance bit-identical to its 16-utterance original."""
    import torch
    import bench
    from nv_wavenet_amd import WavenetEngine
    n_timed, seed = 64, 111
    N = bench.STEADY_FROM + n_timed
    case = cases.Case("C3_fp16_benchmarked_path", 30, [], cases.Shape(64, 256, 256, 20, 16, N, 512), 3, 1, 128)
    s = case.shape
    t = util.gen_o1(case, half=True)
    block = np.ascontiguousarray(t.Lh[:bench.COND_BLOCK])                  # [64][L][16][2R]: the reused block
    t.Lh = np.ascontiguousarray(np.tile(block, (N // bench.COND_BLOCK, 1, 1, 1)))
    t.sel = util.O.philox_selectors(seed, N, s.B)
    assert t.Lh.shape[0] == N

    def engine(B, org):
        e = WavenetEngine(s.R, s.S, s.A, s.L, s.maxD, B, N, impl=0, tanhEmbed=True, precision=16, organisation=org)
        e.setEmbeddings(t.embP, t.embC)
        for l in range(s.L):
            e.setLayerWeights(l, t.Wprev[l], t.Wcur[l], t.Bh[l], t.Wres[l], t.Bres[l], t.Wskip[l], t.Bskip[l])
        e.setOutWeights(t.Wzs, t.Bzs, t.Wza, t.Bza)
        idx = torch.arange(B, device="cuda") % s.B
        blk = torch.from_numpy(block).cuda()[:, :, idx, :].contiguous()
        e.setSelectorSeed(seed)
        e.resetHistory()
        for first in range(0, N, bench.COND_BLOCK):                        # exactly bench.py: steady_engine
            e.packConditioning(blk, first, bench.COND_BLOCK)
        torch.cuda.synchronize()
        return e, idx.cpu().numpy()

    def sequence(B, org, check=None):
        e, idx = engine(B, org)
        if check:
            assert e.kernelInfo(B, False).split(" ")[0] == check, e.kernelInfo(B, False)
        assert e.run_partial_chunk(0, bench.STEADY_FROM, N, B)
        assert e.run_partial_chunk(bench.STEADY_FROM, n_timed, N, B)
        e.synchronize()
        y = torch.full((B, N), -1, dtype=torch.int32, device="cuda")
        e.getYOut(y, 0, N)
        e.synchronize()
        e.close()
        return y.cpu().numpy()

    if tiles_per_cu == 0:
        # 16 utterances, production kernels for the whole sequence ...
        y16 = sequence(s.B, util.MODE_ORG["wg"])
        # ... and once more with the dump on in the last launch: held to the oracle
        e, _ = engine(s.B, util.MODE_ORG["wg"])
        assert e.run_partial_chunk(0, bench.STEADY_FROM, N, s.B)
        yd = np.full((s.B, N), -1, dtype=np.int32)
        assert e.run_partial(bench.STEADY_FROM, N, s.B, yd, 1, True)
        e.synchronize()
        got = util.engine_getters(e, s.L)
        got["y"] = yd
        e.close()
        assert np.array_equal(yd, y16), "dump and dump-free kernels disagree on the benchmarked sequence"
        st = util.fp16_bars(_teacher_forced_ref(case, t, yd), got, t.sel.T, "benchmarked path, 16 utterances")
        print("benchmarked path, 16 utterances:", {k: round(v, 4) for k, v in st.items()})
        # Philox selectors are a function of (sample, utterance NUMBER): utterance 16 k + j shares j's conditioning but draws its
        # own selectors.  48 utterances on the one-tile kernel (whose first 16 are the oracle-held ones above) pin the three-tile
        # kernel itself and the throughput organisation, selectors of utterances >= 16 included
        y48 = sequence(3 * s.B, util.MODE_ORG["wg"])
        assert np.array_equal(y48[:s.B], y16)
        assert not np.array_equal(y48[s.B:2 * s.B], y16), "utterances 16.. drew utterance 0..'s selectors"
        for mode in ("wg3", "wg4"):
            assert np.array_equal(sequence(3 * s.B, util.MODE_ORG[mode]), y48), "%s differs from the one-tile kernel on the benchmarked sequence" % mode
        # ... and the four-tile kernel (round 6) with all four of its tiles in use
        y64 = sequence(4 * s.B, util.MODE_ORG["wg"])
        assert np.array_equal(y64[:3 * s.B], y48)
        assert np.array_equal(sequence(4 * s.B, util.MODE_ORG["wg4"], bench.HEADLINE_KERNELS[4]), y64), "wg4 differs from the one-tile kernel"
        return
    # the headline batch (the engine's own choice: three tiles per workgroup, the kernel bench.py asserts): its first 1024
    # utterances against the one-tile kernel run on 1024 utterances, whose first 16 are the oracle-held ones of the other case
    ncu = torch.cuda.get_device_properties(0).multi_processor_count
    B = tiles_per_cu * 16 * ncu
    # (four tiles per CU: the four-tile kernel of round 6, one workgroup per CU; eight: two rounds of it -- the launch shapes of
    #  bench.py's real-time probe at 16 384 utterances and of its `oversubscribed` entry)
    y = sequence(B, 0, bench.HEADLINE_KERNELS[min(tiles_per_cu, 4)])
    y1k = sequence(1024, util.MODE_ORG["wg"])
    assert np.array_equal(y[:1024], y1k), "the headline batch differs from the one-tile kernel on the benchmarked sequence"
    assert np.array_equal(y1k[:s.B], sequence(s.B, util.MODE_ORG["wg"]))
    assert all(len(np.unique(y[b])) > 8 for b in range(0, B, 997))


def _wrapper_model(R, S, A, L, B, N, seed=7):
    """export_weights()-shaped random tensors (pytorch/wavenet.py:147-188) + a conditioning tensor."""
    import torch
    gen = torch.Generator().manual_seed(seed)
    rnd = lambda *s, sc=0.1: (torch.rand(*s, generator=gen) - 0.5) * sc
    w = dict(embedding_prev=rnd(A, R), embedding_curr=rnd(A, R), conv_out_weight=rnd(A, S, 1),
             conv_end_weight=rnd(A, A, 1), dilate_weights=[rnd(2 * R, R, 2) for _ in range(L)],
             dilate_biases=[rnd(2 * R) for _ in range(L)], max_dilation=4,
             res_weights=[rnd(R, R, 1) for _ in range(L - 1)], res_biases=[rnd(R) for _ in range(L - 1)],
             skip_weights=[rnd(S, R, 1) for _ in range(L)], skip_biases=[rnd(S) for _ in range(L)],
             use_embed_tanh=True)
    cond = rnd(2 * R, B, L, N)
    dev = {k: ([x.cuda() for x in v] if isinstance(v, list) else v.cuda() if torch.is_tensor(v) else v)
           for k, v in w.items()}
    return w, dev, cond


def _wrapper_oracle(w, cond, R, S, A, L, B, N, maxD, sel, half=False):
    """The oracle loaded with the same model the wrapper converts (zero output biases, zero last
    residual layer: wavenet_infer.cu:75-82, nv_wavenet.py:139-141)."""
    import ctypes
    from oracle import oracle as O
    o = O.Oracle(L, B, N, R, S, A, maxD)
    rh = (lambda a: a.astype(np.float16).astype(np.float32)) if half else (lambda a: a)
    f = lambda x: rh(np.ascontiguousarray(x.numpy(), dtype=np.float32))
    cm = lambda x: f(x.squeeze(-1) if x.dim() == 3 else x).T.copy()
    lib = o.lib
    fp = lambda a: a.ctypes.data_as(ctypes.POINTER(ctypes.c_float))
    ep, ec = f(w["embedding_prev"]), f(w["embedding_curr"])
    lib.nvw_oracle_set_embeddings(o.h, fp(ep), fp(ec))
    zR, zRR = np.zeros(R, np.float32), np.zeros((R, R), np.float32)
    for l in range(L):
        dw = w["dilate_weights"][l]
        a = [cm(dw[:, :, 0]), cm(dw[:, :, 1]), np.ascontiguousarray(w["dilate_biases"][l].numpy()),
             cm(w["res_weights"][l]) if l < L - 1 else zRR,
             np.ascontiguousarray(w["res_biases"][l].numpy()) if l < L - 1 else zR,
             cm(w["skip_weights"][l]), np.ascontiguousarray(w["skip_biases"][l].numpy())]
        lib.nvw_oracle_set_layer_weights(o.h, l, *[fp(x) for x in a])
    zA = np.zeros(A, np.float32)
    wzs, wza = cm(w["conv_out_weight"]), cm(w["conv_end_weight"])
    lib.nvw_oracle_set_out_weights(o.h, fp(wzs), fp(zA), fp(wza), fp(zA))
    Lh = rh(np.ascontiguousarray(cond.permute(3, 2, 1, 0).numpy(), dtype=np.float32))
    o.set_inputs(Lh, sel)
    return o


def test_wavenet_infer_c_abi_and_python_wrapper():
    """The reference's PyTorch path: NVWaveNet(**weights).infer(cond, impl) -> nv_wavenet_ext.infer
    -> wavenet_infer() (pytorch/nv_wavenet.py:172-196, wavenet_infer.cu:105-143). Selectors come
    from libc rand() inside the call; seeding srand() makes them reproducible, and the oracle fed
    the same draws must give the same samples."""
    import ctypes
    import torch
    from nv_wavenet_amd.nv_wavenet import NVWaveNet, Impl
    from oracle import oracle as O
    R, S, A, L, B, N, maxD = 64, 256, 256, 6, 3, 24, 4
    w, dev, cond = _wrapper_model(R, S, A, L, B, N)
    libc = ctypes.CDLL("libc.so.6")
    model, cond_dev = NVWaveNet(**dev), cond.cuda()
    # the HIP runtime draws from libc rand() when it first loads a code object, so warm every
    # kernel of this path up before seeding (the reference is exposed to its runtime the same way)
    model.infer(cond_dev, Impl.PERSISTENT)
    torch.cuda.synchronize()
    libc.srand(1234)
    y = model.infer(cond_dev, Impl.PERSISTENT).cpu().numpy()
    assert y.shape == (B, N) and y.dtype == np.int32
    # the same draws for the oracle: Matrix(B,N).randomize(0.5,1.0) order (wavenet_infer.cu:92-94)
    O._lib("oracle").nvw_srand(1234)
    sel = np.zeros((N, B), dtype=np.float32)
    O._lib("oracle").nvw_randomize(sel.ctypes.data_as(ctypes.POINTER(ctypes.c_float)), B, N,
                                   ctypes.c_float(0.5), ctypes.c_float(1.0))
    o = _wrapper_oracle(w, cond, R, S, A, L, B, N, maxD, sel)
    assert np.array_equal(y, o.run(N))
    o.close()


def test_reference_pybind_extension_on_this_library():
    """The drop-in claim at the binding level: the reference's own pybind extension (pytorch/wavenet_infer_wrapper.cpp, compiled by
    oracle/build_ref_binding.py against libwavenet_infer.so -- a compiled .so under oracle/_ref) generates through this engine when
    it is handed what the reference's NVWaveNet class hands it, and the samples equal both the oracle's (same libc rand() draws)
    and the ctypes module's.  The class driving it here is this repo's mirror (nv_wavenet_amd/nv_wavenet.py); the reference's own
    Python file does not travel to the GPU box in any form -- tests/test_capi_cpu.py::
    test_reference_python_wrapper_prepares_the_same_tensors imports it where it lies (authoring container) and holds the mirror
    to it tensor by tensor."""
    import ctypes
    import os
    import sys
    import torch
    from oracle import oracle as O
    import nv_wavenet_amd.nv_wavenet as mirror_mod
    from nv_wavenet_amd.nv_wavenet import NVWaveNet, Impl
    refdir = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "_ref")
    if not os.path.exists(os.path.join(refdir, "nv_wavenet_ext.so")):
        pytest.skip("oracle/_ref binding not built (needs the reference tree at build time)")
    assert not os.path.exists(os.path.join(refdir, "nv_wavenet_ref.pyc")), "bytecode of the reference's Python must not travel"
    sys.path.insert(0, refdir)
    try:
        import nv_wavenet_ext as ref_ext            # the REFERENCE's extension module (compiled C++)
    finally:
        sys.path.remove(refdir)
    assert ref_ext.__file__.startswith(refdir)
    assert (ref_ext.num_res_channels(), ref_ext.num_skip_channels(), ref_ext.num_out_channels()) == (64, 256, 256)
    R, S, A, L, B, N, maxD = 64, 256, 256, 6, 3, 24, 4
    w, dev, cond = _wrapper_model(R, S, A, L, B, N)
    libc = ctypes.CDLL("libc.so.6")
    cond_dev = cond.cuda()
    ours = mirror_mod.nv_wavenet_ext
    try:
        mirror_mod.nv_wavenet_ext = ref_ext         # the mirror class now calls the reference's pybind entry point
        model = NVWaveNet(**dev)
        model.infer(cond_dev, Impl.PERSISTENT)      # warm-up: the HIP runtime draws from rand() when it loads code
        torch.cuda.synchronize()
        libc.srand(1234)
        y = model.infer(cond_dev, Impl.PERSISTENT)
        torch.cuda.synchronize()
    finally:
        mirror_mod.nv_wavenet_ext = ours
    y = y.cpu().numpy()
    O._lib("oracle").nvw_srand(1234)
    sel = np.zeros((N, B), dtype=np.float32)
    O._lib("oracle").nvw_randomize(sel.ctypes.data_as(ctypes.POINTER(ctypes.c_float)), B, N,
                                   ctypes.c_float(0.5), ctypes.c_float(1.0))
    o = _wrapper_oracle(w, cond, R, S, A, L, B, N, maxD, sel)
    assert np.array_equal(y, o.run(N)), "the reference's binding on this engine disagrees with the oracle"
    o.close()
    mirror = NVWaveNet(**dev)                       # ... and through this package's ctypes module of the same name
    libc.srand(1234)
    y2 = mirror.infer(cond_dev, Impl.PERSISTENT).cpu().numpy()
    assert np.array_equal(y, y2)


def _native_uniform(tensor_id, n):
    """wn_test_uniform of tests/cpp/api_surface.hip (splitmix64 finaliser of (tensor id, index)), elementwise."""
    with np.errstate(over="ignore"):
        i = np.arange(1, n + 1, dtype=np.uint64)
        z = i * np.uint64(0x9E3779B97F4A7C15) + np.uint64(tensor_id) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return (z >> np.uint64(40)).astype(np.float32) * np.float32(1.0 / 16777216.0)


def _native_inputs(R, S, A, L, B, N):
    """The seeded model of `api_surface run` (generate<> there), in the oracle's TestInputs layout."""
    from oracle import oracle as O
    t = O.TestInputs(R, S, A, L, B, N)
    f32 = np.float32

    def fill(a, tid, scale):
        a.reshape(-1)[:] = (f32(2.0) * _native_uniform(tid, a.size) - f32(1.0)) * f32(scale)
    sR, sS, sA = f32(np.sqrt(f32(3.0) / f32(R))), f32(np.sqrt(f32(3.0) / f32(S))), f32(np.sqrt(f32(3.0) / f32(A)))
    fill(t.embP, 1, 1.0), fill(t.embC, 2, 1.0)
    for l in range(L):
        tid = 100 + 7 * l
        for k, (arr, sc) in enumerate(((t.Wprev[l], sR), (t.Wcur[l], sR), (t.Bh[l], 0.1), (t.Wres[l], sR), (t.Bres[l], 0.1),
                                       (t.Wskip[l], sR), (t.Bskip[l], 0.1))):
            fill(arr, tid + k, sc)
    fill(t.Wzs, 3, sS), fill(t.Bzs, 4, 0.1), fill(t.Wza, 5, f32(4.0) * sA), fill(t.Bza, 6, 0.1), fill(t.Lh, 7, 0.5)
    t.sel.reshape(-1)[:] = _native_uniform(8, t.sel.size)
    return t


@pytest.mark.parametrize("precision,impl,L,maxD,B,N", [(32, 1, 6, 8, 5, 40), (32, 3, 7, 4, 16, 33), (16, 1, 6, 8, 20, 40), (16, 3, 8, 16, 4, 50)])
def test_native_host_program_against_the_c_abi(precision, impl, L, maxD, B, N, tmp_path):
    """A C++ host written against the CLASS (tests/cpp/api_surface.hip, compiled by __graft_entry__.build(); role of
    nv_wavenet_test.cu:331-395 as a native binary) is EXECUTED on the GPU: first the whole public surface (every member, the
    reference's defaults, lambda consumers, both precisions), then a seeded generation whose yOut must equal the C-ABI run of the
    same tensors (nv_wavenet_test.cu:302-304: identical samples) and, in fp32, the oracle's; inside the binary run() and
    run_chunks() with a lambda consumer must agree and the dumped distribution must sum to one."""
    import os
    import subprocess
    exe = os.path.join(os.path.dirname(os.path.abspath(__file__)), "cpp", "api_surface")
    assert os.path.exists(exe), "tests/cpp/api_surface is built by __graft_entry__.build()"
    if (precision, impl) == (32, 1):
        r = subprocess.run([exe], capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, "surface drive failed (%d): %s" % (r.returncode, r.stderr[-2000:])
    out = str(tmp_path / "y.bin")
    r = subprocess.run([exe, "run", str(precision), str(impl), str(L), str(maxD), str(B), str(N), out], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, "native run failed (%d): %s" % (r.returncode, r.stderr[-2000:])
    y_native = np.fromfile(out, dtype=np.int32).reshape(B, N)
    R, S, A = 64, 128, 256
    t = _native_inputs(R, S, A, L, B, N)
    case = cases.Case("native", 0, [], cases.Shape(R, S, A, L, B, N, maxD), impl, 1, N)
    e = util.make_engine(case, t, precision=precision)
    y = np.full((B, N), -1, dtype=np.int32)
    assert e.run(N, B, y, 1, False)
    e.synchronize()
    e.close()
    assert np.array_equal(y_native, y), "the class driven natively and the C ABI disagree"
    assert all(len(np.unique(y[b])) > 4 for b in range(B))          # (a seeded model, not a constant)
    if precision == 32:
        o = util.make_oracle(case, t)
        assert np.array_equal(y, o.run(N)), "fp32 samples differ from the oracle"
        o.close()


def test_persistent_python_wrapper_seeded_audio():
    """NVWaveNetEngine keeps the engine (and the uploaded weights) alive across
    infer() calls, takes R/S/A from the tensors (here an instantiation wavenet_infer() does not
    offer), accepts the conditioning already in the engine's layout on the device, draws selectors
    in-kernel from a seed and returns int16 audio. fp32: indices identical to the oracle's under
    the same Philox selectors, on every call."""
    import torch
    from nv_wavenet_amd.nv_wavenet import NVWaveNetEngine, Impl
    R, S, A, L, B, N, maxD = 64, 128, 256, 6, 5, 40, 4
    w, dev, cond = _wrapper_model(R, S, A, L, B, N, seed=11)
    table = util.load_golden("mulaw_pcm")["pcm_%d" % A]
    model = NVWaveNetEngine(**dev, precision=32)
    cond_nlbc = cond.permute(3, 2, 1, 0).contiguous().cuda()
    for call, seed in enumerate((5, 5, 99)):
        o = _wrapper_oracle(w, cond, R, S, A, L, B, N, maxD, util.O.philox_selectors(seed, N, B))
        y_ref = o.run(N)
        o.close()
        if call == 0:
            y, audio = model.infer(cond.cuda(), Impl.MANYBLOCK, seed=seed, return_audio=True)
        else:
            y, audio = model.infer(cond_nlbc, Impl.MANYBLOCK, seed=seed, return_audio=True, layout="NLBC")
        assert np.array_equal(y.cpu().numpy(), y_ref), "call %d" % call
        assert np.array_equal(audio.cpu().numpy(), table[y_ref])
    assert len(model._engines) == 1, "the engine must be reused across calls"
    # another utterance LENGTH: same engine (capacity bucket), the shorter conditioning runs as a prefix
    N2 = 27
    cond2 = cond[:, :, :, :N2].contiguous()
    o = _wrapper_oracle(w, cond2, R, S, A, L, B, N2, maxD, util.O.philox_selectors(77, N2, B))
    y_ref2 = o.run(N2)
    o.close()
    y2, audio2 = model.infer(cond2.cuda(), Impl.MANYBLOCK, seed=77, return_audio=True)
    assert y2.shape == (B, N2) and np.array_equal(y2.cpu().numpy(), y_ref2)
    assert np.array_equal(audio2.cpu().numpy(), table[y_ref2])
    assert len(model._engines) == 1, "a new utterance length must not build a new engine"
    # under a non-default stream the launches follow the caller's stream
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        y3 = model.infer(cond2.cuda(), Impl.MANYBLOCK, seed=77)
    side.synchronize()
    assert np.array_equal(y3.cpu().numpy(), y_ref2)
    # torch-drawn selectors (no seed): plausible output, engine still reused
    y = model.infer(cond_nlbc, Impl.MANYBLOCK, layout="NLBC", generator=torch.Generator(device="cuda").manual_seed(3))
    assert y.shape == (B, N) and int(y.min()) >= 0 and int(y.max()) < A and len(model._engines) == 1
    model.close()


@pytest.mark.parametrize("n_cond,L,B,frames,win,stride", [(80, 20, 48, 2, 1024, 256), (40, 3, 16, 5, 32, 8), (100, 2, 32, 5, 8, 4)])
def test_fused_conditioning_producer_against_the_torch_operations(n_cond, L, B, frames, win, stride):
    """csrc/cond_producer.hip (nvw_produce_conditioning_f16): the model's conditioning convolution computed by MFMAs straight into
    the engine's fragment order.  Held to get_cond_input's torch operations on the same inputs (which the engine consumes bit for
    bit like its own packed copy, test_conditioning_produced_in_fragment_order) to fp16 rounding (the torch path rounds the scaled
    weights, the bias and the sum separately: three units in the last place) and to a float64 evaluation of the convolution; samples that
    are no multiple of the kernel's block of 8, features that are no multiple of 32, buffer contents beyond the produced samples."""
    import torch
    from nv_wavenet_amd.nv_wavenet import get_cond_input
    R = 64
    g = torch.Generator(device="cuda").manual_seed(11)
    rnd = lambda *s, sc=1.0: (torch.rand(*s, device="cuda", generator=g) - 0.5) * sc
    h = torch.float16
    up_w, up_b = rnd(n_cond, n_cond, win, sc=0.1).to(h), rnd(n_cond, sc=0.2).to(h)
    cw, cb = rnd(2 * R * L, n_cond, 1, sc=0.6).to(h), rnd(2 * R * L).to(h)
    f = rnd(B, n_cond, frames, sc=2.0).to(h)
    tiles, N = B // 16, frames * stride
    a = torch.full((N + 1, L, tiles, 2 * R // 32, 4, 16, 8), 7.0, dtype=h, device="cuda")
    b = torch.zeros_like(a)
    ra = get_cond_input(f, up_w, up_b, stride, cw, cb, L, layout="packed", precision=16, tiles=tiles, out=a[:N])            # fused (default)
    get_cond_input(f, up_w, up_b, stride, cw, cb, L, layout="packed", precision=16, tiles=tiles, out=b[:N], fused=False)
    torch.cuda.synchronize()
    assert ra.data_ptr() == a.data_ptr()
    assert bool((a[N] == 7.0).all()), "the producer wrote beyond the samples it was given"
    af, bf = a[:N].float(), b[:N].float()
    assert float(af.abs().max()) > 1.0 and len(torch.unique(a[:N])) > 1000
    # (the torch path rounds the scaled weights, the scaled bias and the matrix product to fp16 separately, so its error is relative
    #  to the terms, not to their sum: a loose bar here, the tight one against float64 below)
    err = (af - bf).abs()
    k = int(err.argmax())
    assert bool((err <= 3.0 * 2.0 ** -10 * (bf.abs() + 4.0)).all()), (float(err.max()), float(af.flatten()[k]), float(bf.flatten()[k]), k)
    # float64 on the CPU, from the very fp16 operands
    c64 = get_cond_input(f.double().cpu(), up_w.double().cpu(), up_b.double().cpu(), stride, cw.double().cpu(), cb.double().cpu(), L,
                         layout="NLBC", via_gemm=True)                                    # [N][L][B][2R], reference channel order
    from nv_wavenet_amd.nv_wavenet import pack_cond_input
    want = pack_cond_input(c64.float(), 16, tiles)[:N].float()                            # packed, pre-scaled, rounded once
    errw = (af.cpu() - want).abs()
    kw = int(errw.argmax())
    print("fused producer: max |fused - torch| %.2e, max |fused - float64| %.2e at %d (%.4f vs %.4f), max |torch - float64| %.2e" %
          (float(err.max()), float(errw.max()), kw, float(af.flatten()[kw]), float(want.flatten()[kw]), float((bf.cpu() - want).abs().max())))
    assert bool((errw <= 2.0 ** -9 * want.abs() + 4e-3).all()), float(errw.max())        # (the upsampled features are rounded to fp16 in between)


def test_python_wrapper_fp16_conditioning_in_place_and_get_cond_input():
    """SURVEY.md 8f rank 1 with T_data conditioning: get_cond_input(..., layout="NLBC", dtype=torch.float16) emits the upsampled
    conditioning in the fp16 engine's own element type and layout, NVWaveNetEngine(precision=16).infer reads that tensor in
    place (RAW=2 kernels, no fp32 detour, no packed copy), and the samples equal those from the fp32 tensor holding the
    same (fp16-representable) values."""
    import torch
    from nv_wavenet_amd.nv_wavenet import NVWaveNetEngine, Impl, get_cond_input, pack_cond_input
    R, S, A, L, B, frames, stride, n_cond = 64, 256, 256, 6, 5, 6, 8, 20
    w, dev, _ = _wrapper_model(R, S, A, L, B, 8, seed=13)
    gen = torch.Generator().manual_seed(3)
    rnd = lambda *s, sc=1.0: ((torch.rand(*s, generator=gen) - 0.5) * sc).cuda()
    feats = rnd(B, n_cond, frames)
    up_w, up_b = rnd(n_cond, n_cond, 2 * stride, sc=0.5), rnd(n_cond, sc=0.1)
    cw, cb = rnd(2 * R * L, n_cond, 1, sc=2.0), rnd(2 * R * L, sc=0.5)
    c16 = get_cond_input(feats, up_w, up_b, stride, cw, cb, L, layout="NLBC", dtype=torch.float16)
    N = frames * stride
    assert c16.dtype == torch.float16 and tuple(c16.shape) == (N, L, B, 2 * R) and c16.is_contiguous()
    ref = get_cond_input(feats, up_w, up_b, stride, cw, cb, L, layout="CBLN")           # the reference's view, fp32
    assert torch.equal(ref.permute(3, 2, 1, 0).half(), c16)
    model = NVWaveNetEngine(**dev, precision=16)
    y16 = model.infer(c16, Impl.SINGLE_BLOCK, seed=11, layout="NLBC")
    e = next(iter(model._engines.values()))
    assert "RAW=2" in e.kernelInfo(B, False), e.kernelInfo(B, False)
    y32 = model.infer(c16.float(), Impl.SINGLE_BLOCK, seed=11, layout="NLBC")
    assert "RAW=1" in e.kernelInfo(B, False)
    assert torch.equal(y16, y32) and y16.shape == (B, N) and int(torch.unique(y16).numel()) > 8
    yc = model.infer(c16, Impl.PERSISTENT, seed=11, layout="NLBC")                   # the chain reads it in place as well
    assert torch.equal(yc, y16)
    # round 3: the conditioning convolution emits the engine's FRAGMENT order itself (channel permutation and gate pre-scale
    # folded into its weights): the packed path of the generation kernels runs on that tensor, no in-place conversion
    tiles = model.cond_tiles(B, N, Impl.SINGLE_BLOCK)
    cp = get_cond_input(feats, up_w, up_b, stride, cw, cb, L, layout="packed", precision=16, tiles=tiles)
    assert cp.dtype == torch.float16 and cp.is_contiguous() and cp.shape[0] == N + 1 and cp.shape[2] == tiles
    assert cp.numel() == (N + 1) * L * tiles * 16 * 2 * R and float(cp[N].abs().max()) == 0.0
    assert torch.equal(cp, pack_cond_input(ref.permute(3, 2, 1, 0), 16, tiles)) or \
        float((cp.float() - pack_cond_input(ref.permute(3, 2, 1, 0), 16, tiles).float()).abs().max()) <= 2.0 ** -9 * float(cp.abs().max())
    yp = model.infer(cp, Impl.SINGLE_BLOCK, seed=11, layout="packed", batch_size=B)
    assert "RAW=0" in e.kernelInfo(B, False), e.kernelInfo(B, False)
    assert yp.shape == (B, N) and float((yp == y16).float().mean()) > 0.5      # same network, conditioning equal to an fp16 ulp
    model.close()


def test_bench_c5_path_runs_on_one_gpu():
    """BASELINE configs[4] (C3 shape, global batch 64 sharded over the ranks, RCCL gather of the samples) cannot be
    measured on a one-GPU box; its code path -- shard_range, the per-rank engine, the gather bookkeeping with world 1, the
    JSON line -- is at least executed on a GPU here."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--config", "c5", "--no-extras",
                        "--no-cpu-baseline", "--steps", "2", "--warmup", "1", "--samples", "256"], capture_output=True, text=True,
                       timeout=900, cwd=root)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
    j = json.loads(line)
    assert j["n_gpus"] == 1 and j["config"]["global_batch"] == 64 and j["config"]["batch_per_gpu"] == 64
    assert j["value"] > 0 and j["khz_per_utterance"] > 24.0 and j["distinct_samples_in_last_step"] > 8, j


@pytest.mark.parametrize("mode", MODES)
def test_in_kernel_selectors_and_pcm_out(mode):
    """SURVEY.md 8f rank 2. With setSelectorSeed the engine draws its selectors in-kernel
    (Philox4x32-10); the oracle fed nvw_philox_selectors(seed) must produce the same fp32 indices,
    bit for bit, with no selector matrix uploaded. setAudioOut adds int16 PCM = the reference's
    mu_law_decode_numpy + int16 cast (fixture tests/golden/mulaw_pcm.npz) of those indices, through
    run() and through run_chunks()."""
    case = cases.BY_NAME["C3_R64S256A256_L20_B21"]      # ragged batch, 2 tiles
    s = case.shape
    seed = 0x1234ABCD5678EF01
    t = util.gen_inputs(case)
    o = util.make_oracle(case, t)
    sel = util.O.philox_selectors(seed, s.N, s.B)
    o.set_inputs(t.Lh, sel)
    y_ref = o.run(s.N)
    table = util.load_golden("mulaw_pcm")["pcm_%d" % s.A]
    assert len(np.unique(y_ref)) > 8

    e = util.make_engine(case, t, precision=32, mode=mode)   # uploads t.sel, which must NOT be used
    e.setConditioning(t.Lh)
    e.setSelectorSeed(seed)
    y = np.full((s.B, s.N), -1, dtype=np.int32)
    pcm = np.full((s.B, s.N), 7, dtype=np.int16)
    e.setAudioOut(pcm)
    assert e.run(s.N, s.B, y, 1, False)
    e.synchronize()
    assert np.array_equal(y, y_ref), "in-kernel Philox selectors differ from the oracle's"
    assert np.array_equal(pcm, table[y_ref])

    # chunked: the consumer sees finished PCM for its chunk
    e.setConditioning(t.Lh)
    y2 = np.full((s.B, s.N), -1, dtype=np.int32)
    pcm2 = np.zeros((s.B, s.N), dtype=np.int16)
    e.setAudioOut(pcm2)
    seen = []

    def consume(yo, first, count):
        seen.append(np.array_equal(pcm2[:, first:first + count], table[y_ref[:, first:first + count]]))
    assert e.run_chunks(16, consume, s.N, s.B, y2, 1)
    e.synchronize()
    assert seen and all(seen)
    assert np.array_equal(y2, y_ref) and np.array_equal(pcm2, table[y_ref])

    # back to the uploaded table
    e.setAudioOut(None)
    e.setInputs(t.Lh, t.sel)
    o.set_inputs(t.Lh, t.sel)
    y3 = np.full((s.B, s.N), -1, dtype=np.int32)
    assert e.run(s.N, s.B, y3, 1, False)
    e.synchronize()
    assert np.array_equal(y3, o.run(s.N))
    e.close(), o.close()


@pytest.mark.parametrize("B", [4112, 8208])
def test_full_chip_batches_by_replication(B):
    """Size-independent property at the batch sizes where the engine changes organisation by itself
    (257 tiles: two tiles per workgroup; 513 tiles: loader/consumer kernel on a 256-CU GPU): utterances
    are independent, so a batch that repeats the 19 utterances of a small case cyclically must repeat
    that case's fp32 samples, which are pinned to the oracle and the reference fixture."""
    case = cases.BY_NAME["R64S128A256_L7_B19_oddL"]
    s = case.shape
    g = util.load_golden(case.name)
    t = util.gen_inputs(case)
    idx = np.arange(B) % s.B
    from nv_wavenet_amd import WavenetEngine
    e = WavenetEngine(s.R, s.S, s.A, s.L, s.maxD, B, s.N, impl=case.impl, tanhEmbed=True, precision=32)
    e.setEmbeddings(t.embP, t.embC)
    for l in range(s.L):
        e.setLayerWeights(l, t.Wprev[l], t.Wcur[l], t.Bh[l], t.Wres[l], t.Bres[l], t.Wskip[l], t.Bskip[l])
    e.setOutWeights(t.Wzs, t.Bzs, t.Wza, t.Bza)
    Lh = np.ascontiguousarray(t.Lh[:, :, idx, :])            # [N][L][B][2R]
    sel = np.ascontiguousarray(t.sel[:, idx])                # [N][B]
    e.setInputs(Lh, sel)
    y = np.full((B, s.N), -1, dtype=np.int32)
    assert e.run(s.N, B, y, 1, False)
    e.synchronize()
    assert np.array_equal(y, g["yOut"][0][idx]), "utterance %d differs" % int(np.argwhere((y != g["yOut"][0][idx]).any(axis=1))[0, 0])
    e.close()


@pytest.mark.parametrize("B,impl", [(4112, 1), (8208, 1), (12304, 1), (16400, 1), (12304, 0), (16368, 0), (16400, 0), (32752, 0)])
def test_full_chip_batches_by_replication_fp16(B, impl):
    """The same property for the fp16 production path (dump-free kernels, engine's own choice of organisation at
    full-chip batch sizes; O(1) inputs): the big batch must repeat, bit for bit, what the one-tile kernel generates for
    the 19 utterances alone (one to four tiles per workgroup perform the same arithmetic per utterance; beyond
    four tiles per CU the launch has more workgroups than CUs)."""
    case = cases.BY_NAME["R64S128A256_L7_B19_oddL"]
    s = case.shape
    t = util.gen_o1(case, half=True)
    e0 = _engine_o1(case, t, 16, "wg")
    y0 = np.full((s.B, s.N), -1, dtype=np.int32)
    assert e0.run(s.N, s.B, y0, 1, False)
    e0.synchronize()
    e0.close()
    assert len(np.unique(y0)) > 100
    idx = np.arange(B) % s.B
    # impl 1 = SINGLE_BLOCK: wavenet_wg whatever the batch; impl 0 = AUTO: beyond three tiles per CU the throughput organisation
    case = cases.Case(case.name, case.seed, case.prior, case.shape, impl, case.iters, case.chunk)
    e = _engine_o1(case, t, 16, None, B=B, Lh=np.ascontiguousarray(t.Lh[:, :, idx, :]), sel=np.ascontiguousarray(t.sel[:, idx]))
    # the engine reports what it launches: dump-free kernels; two / three / four tiles per workgroup beyond one / two / three tiles per CU
    import torch
    ncu = torch.cuda.get_device_properties(0).multi_processor_count
    tiles = (B + 15) // 16
    info = e.kernelInfo(B, False)
    assert "DUMP=0" in info and "fp16" in info, info
    # (beyond three tiles per CU: four tiles per workgroup where they save a round of workgroups, nvWavenetInfer::wgTiles)
    four = tiles > 3 * ncu and -(-tiles // (4 * ncu)) * 47 <= -(-tiles // (3 * ncu)) * 36
    want = "BT=4" if four else "BT=3" if tiles > 2 * ncu else "BT=2" if tiles > ncu else "BT=1"
    assert want in info and "wavenet_wg<" in info, (info, ncu)
    y = np.full((B, s.N), -1, dtype=np.int32)
    assert e.run(s.N, B, y, 1, False)
    e.synchronize()
    assert np.array_equal(y, y0[idx])
    e.close()


@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("precision", [32, 16])
def test_no_tanh_on_the_embedding(mode, precision):
    """tanhEmbed = false is what the PyTorch path uses (WaveNet.export_weights sets use_embed_tanh False,
    pytorch/wavenet.py:186) although the reference's CPU class always applies the tanh
    (nv_wavenet_reference.cpp:52); the oracle restatement carries the flag.  O(1) inputs (|embedding sum| up to ~1.5, so
    the tanh matters).  fp32: exact indices and the reference harness's activation bars; fp16: util.fp16_bars."""
    case = cases.Case("C3_o1_B21_notanh", 31, [], cases.Shape(64, 256, 256, 20, 21, 40, 16), 3, 1, 16)
    s = case.shape
    t = util.gen_o1(case, half=(precision == 16))
    e = _engine_o1(case, t, precision, mode, tanh_embed=False)
    got = _run_dumped(e, case, chunk=s.N)
    ref = _teacher_forced_ref(case, t, got["y"], tanh_embed=False)
    with_tanh = _teacher_forced_ref(case, t, got["y"], tanh_embed=True)      # sanity: the flag changes the network well beyond the bars
    assert np.abs(with_tanh["Xout"] - ref["Xout"]).max() > 0.05 * np.abs(ref["Xout"]).max()
    assert (with_tanh["y"] == got["y"]).mean() < 0.9
    if precision == 32:
        assert np.array_equal(got["y"], ref["y"])
        util.compare_activations(ref, got, atol_eps=32)
    else:
        util.fp16_bars(ref, got, t.sel.T, "tanhEmbed=0/" + mode)
    e.close()


def test_perf_cli_is_flag_compatible_with_the_reference_harness():
    """scripts/nv_wavenet_perf.py mirrors nv_wavenet_perf.cu:203-281: same flags (-l -r -s -a -b -c -n -d -m -p -t),
    same report lines, "Sample rate: %f kHz" at the end; the Implementation value selects the organisation."""
    import os
    import re
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for mode, kernel in ((1, "wavenet_wg<"), (3, "wavenet_chain<")):
        r = subprocess.run([sys.executable, os.path.join(root, "scripts", "nv_wavenet_perf.py"), "-l", "6", "-r", "64", "-s", "128",
                            "-a", "256", "-b", "4", "-c", "2", "-n", "512", "-d", "8", "-m", str(mode), "-p", "16", "-t", "128"],
                           capture_output=True, text=True, timeout=600, cwd=root)
        assert r.returncode == 0, r.stderr[-2000:]
        out = r.stdout
        for line in ("R: 64", "S: 128", "A: 256", "num layers: 6", "max dilation: 8", "batch size: 4", "batch size per block: 2",
                     "num samples: 512", "precision: fp16"):
            assert line in out, (line, out)
        assert kernel in out, out
        m = re.search(r"Sample rate: ([0-9.]+) kHz", out)
        assert m and float(m.group(1)) > 1.0, out
