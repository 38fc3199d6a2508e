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


MODES = ["wg", "stream"]   # the two kernel organisations (wn_kernels.hpp / wn_stream.hpp)


@pytest.mark.parametrize("mode", MODES)
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


@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("name", ["R64S256A256_impl3", "R64S128A256_impl1", "R32S128A256_impl1", "R128S256A256_impl3",
                                  "R64S128A512_impl3", "R128S256A1024_impl3", "R256S256A256_L6_B5"])
def test_fp16_engine_against_fp32_oracle(name, mode):
    """fp16 parity is unpinned by the reference (no test runs half). Stated tolerance: with every
    weight / bias / embedding / conditioning value rounded to fp16 and fed to BOTH sides, the fp16
    engine (fp16 MFMA operands, fp32 accumulation) must give logits within 2e-2*|ref| + 2e-3 of the
    fp32 oracle, probabilities within 2%, and >= 90% of the utterances must produce exactly the
    oracle's indices over the 8-sample horizon."""
    case = cases.BY_NAME[name]
    s = case.shape
    t = util.gen_inputs(case, half=True)
    o = util.make_oracle(case, t)
    e = util.make_engine(case, t, precision=16, mode=mode)
    y_ref = o.run(s.N)
    y = np.full((s.B, s.N), -1, dtype=np.int32)
    assert e.run_chunks(7, None, s.N, s.B, y, _bspb(s.B))
    e.synchronize()
    ref, got = o.getters(), util.engine_getters(e, s.L)
    same = np.all(y == y_ref, axis=1)
    assert same.mean() >= 0.9, "only %.0f%% of utterances reproduce the oracle's samples" % (100 * same.mean())
    ok = same  # activations of diverged utterances legitimately differ
    za_err = np.abs(got["Za"][ok] - ref["Za"][ok])
    assert np.all(za_err <= 2e-2 * np.abs(ref["Za"][ok]) + 2e-3), "logit error %g" % za_err.max()
    assert np.all(np.abs(got["P"][ok] / ref["P"][ok] - 1) <= 2e-2)
    assert np.all(np.abs(got["Xout"][:, ok] - ref["Xout"][:, ok]) <= 2e-2 * np.abs(ref["Xout"][:, ok]) + 2e-3)
    e.close(), o.close()


@pytest.mark.parametrize("mode", MODES)
def test_fp16_teacher_forced_agreement(mode, record_property):
    """fp16 sample agreement is measured teacher-forced, not asserted exact over a long
    free run (one differing pick changes every later sample). The fp16 engine generates freely over
    256 samples at the C3 shape; the fp32 oracle (same fp16-rounded parameters) is then FED the
    engine's samples and asked for its own pick at every step. Stated bar: >= 99.5% of all
    (utterance, step) picks identical, and every differing pick is an edge case: at most two bins
    away, with the draw within 2e-3 of the CDF edge of the oracle's own pick (an fp16-sized shift of
    the cumulative distribution)."""
    case = cases.Case("C3_fp16_teacher_forced", 30, [], cases.Shape(64, 256, 256, 20, 16, 256, 32), 3, 1, 64)
    s = case.shape
    t = util.O.gen_test_inputs(case.seed, case.prior, s, "oracle")   # seeded recipe; no fixture for this statistic
    t.round_to_half()
    o = util.make_oracle(case, t)
    e = util.make_engine(case, t, precision=16, mode=mode)
    y = np.full((s.B, s.N), -1, dtype=np.int32)
    assert e.run(s.N, s.B, y, 1, False)
    e.synchronize()
    y_own, lo, hi = o.run(s.N, forced=y, edges=True)
    agree = float((y_own == y).mean())
    record_property("fp16_teacher_forced_agreement", agree)
    print("fp16 teacher-forced agreement (%s): %.4f over %d picks" % (mode, agree, y.size))
    assert agree >= 0.995, "teacher-forced agreement %.4f" % agree
    sel_bn = t.sel.T
    worst = 0.0
    for b, n in np.argwhere(y_own != y):
        near = min(abs(float(sel_bn[b, n]) - float(lo[b, n])), abs(float(sel_bn[b, n]) - float(hi[b, n])))
        worst = max(worst, near)
        assert abs(int(y_own[b, n]) - int(y[b, n])) <= 2 and near <= 2e-3, (b, n, y_own[b, n], y[b, n], near)
    print("  largest distance of a differing draw from the oracle's CDF edge: %.2e" % worst)
    e.close(), o.close()


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
    gen = torch.Generator().manual_seed(7)
    rnd = lambda *s, sc=0.1: (torch.rand(*s, generator=gen) - 0.5) * sc
    w = dict(embedding_prev=rnd(A, R), embedding_curr=rnd(A, R), conv_out_weight=rnd(A, S, 1),
             conv_end_weight=rnd(A, A, 1), dilate_weights=[rnd(2 * R, R, 2) for _ in range(L)],
             dilate_biases=[rnd(2 * R) for _ in range(L)], max_dilation=maxD,
             res_weights=[rnd(R, R, 1) for _ in range(L - 1)], res_biases=[rnd(R) for _ in range(L - 1)],
             skip_weights=[rnd(S, R, 1) for _ in range(L)], skip_biases=[rnd(S) for _ in range(L)],
             use_embed_tanh=True)
    cond = rnd(2 * R, B, L, N)
    dev = {k: ([x.cuda() for x in v] if isinstance(v, list) else v.cuda() if torch.is_tensor(v) else v)
           for k, v in w.items()}
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
    o = O.Oracle(L, B, N, R, S, A, maxD)
    f = lambda x: np.ascontiguousarray(x.numpy(), dtype=np.float32)
    cm = lambda x: f(x.squeeze(-1) if x.dim() == 3 else x).T.copy()
    lib = o.lib
    fp = lambda a: a.ctypes.data_as(ctypes.POINTER(ctypes.c_float))
    ep, ec = f(w["embedding_prev"]), f(w["embedding_curr"])
    lib.nvw_oracle_set_embeddings(o.h, fp(ep), fp(ec))
    zR, zRR = np.zeros(R, np.float32), np.zeros((R, R), np.float32)
    for l in range(L):
        dw = w["dilate_weights"][l]
        a = [cm(dw[:, :, 0]), cm(dw[:, :, 1]), f(w["dilate_biases"][l]),
             cm(w["res_weights"][l]) if l < L - 1 else zRR, f(w["res_biases"][l]) if l < L - 1 else zR,
             cm(w["skip_weights"][l]), f(w["skip_biases"][l])]
        lib.nvw_oracle_set_layer_weights(o.h, l, *[fp(x) for x in a])
    zA = np.zeros(A, np.float32)
    wzs, wza = cm(w["conv_out_weight"]), cm(w["conv_end_weight"])
    lib.nvw_oracle_set_out_weights(o.h, fp(wzs), fp(zA), fp(wza), fp(zA))
    Lh = np.ascontiguousarray(cond.permute(3, 2, 1, 0).numpy(), dtype=np.float32)
    o.set_inputs(Lh, sel)
    y_ref = o.run(N)
    assert np.array_equal(y, y_ref)
    o.close()
