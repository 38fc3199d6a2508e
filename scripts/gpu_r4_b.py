#!/usr/bin/env python3
"""round 4: first contact of wn::wavenet_bcast with the GPU -- parity against the oracle / wavenet_wg on a few shapes, then
steady-state timings next to wavenet_wg's.  usage: gpu_r4_b.py [parity] [time]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np

what = sys.argv[1:] or ["parity", "time"]


def log(*a):
    print(*a, flush=True)


if "parity" in what:
    import cases
    import util
    import test_parity_gpu as T

    def run(case, t, prec, mode, dump, B=None):
        e = T._engine_o1(case, t, prec, mode, B=B)
        info = e.kernelInfo(case.shape.B if B is None else B, dump)
        if dump:
            got = T._run_dumped(e, case, B=B)
        else:
            y = np.full((case.shape.B if B is None else B, case.shape.N), -1, dtype=np.int32)
            assert e.run(case.shape.N, case.shape.B if B is None else B, y, 1, False)
            e.synchronize()
            got = {"y": y}
        e.close()
        return got, info

    # 1. fp32, C3 shape, O(1) inputs: exact samples and activations against the oracle
    case = cases.Case("C3_o1_first", 30, [], cases.Shape(64, 256, 256, 20, 16, 48, 32), 3, 1, 48)
    t = util.gen_o1(case, half=False)
    got, info = run(case, t, 32, "bcast", True)
    log("fp32:", info)
    ref = util.teacher_forced_oracle(case, t, got["y"])
    ok = np.array_equal(got["y"], ref["y"])
    log("fp32 bcast samples == oracle:", ok, "first rows", got["y"][0, :8], ref["y"][0, :8])
    if ok:
        util.compare_activations(ref, got, atol_eps=32)
        log("fp32 bcast activations ok")
    else:
        bad = np.argwhere(got["y"] != ref["y"])
        log("first mismatches (b, t):", bad[:10].tolist())
        gw, _ = run(case, t, 32, "wg", True)
        log("wg == oracle:", np.array_equal(gw["y"], ref["y"]))
        for k in ("Xout", "skipOut", "Zs", "Za"):
            d = np.abs(got[k] - gw[k])
            log(k, "max |bcast - wg|", float(d.max()), "at", np.unravel_index(d.argmax(), d.shape))

    # 2. fp16, C3: bars against the oracle, dump / no-dump variants, identical to wavenet_wg
    case = T.O1_CASES["C3"]
    t = util.gen_o1(case, half=True)
    gb, info = run(case, t, 16, "bcast", True)
    log("fp16:", info)
    gw, _ = run(case, t, 16, "wg", True)
    same = np.array_equal(gb["y"], gw["y"])
    log("fp16 C3 bcast (dump) == wg:", same)
    if not same:
        bad = np.argwhere(gb["y"] != gw["y"])
        log(" first mismatches:", bad[:8].tolist(), "of", len(bad))
        for k in ("Xout", "skipOut", "Zs", "Za"):
            d = np.abs(gb[k] - gw[k])
            log(" ", k, "max |bcast - wg|", float(d.max()), "at", np.unravel_index(d.argmax(), d.shape))
    gn, info = run(case, t, 16, "bcast", False)
    log("fp16:", info, " no-dump == wg:", np.array_equal(gn["y"], gw["y"]))
    try:
        st = util.fp16_bars(util.teacher_forced_oracle(case, t, gb["y"]), gb, t.sel.T, "C3/bcast")
        log("fp16 bars:", {k: round(v, 4) for k, v in st.items()})
    except AssertionError as ex:
        log("fp16 bars FAILED:", ex)

    # 3. ragged multi-workgroup batch
    case = cases.Case("C3_o1_b200", 31, [], cases.Shape(64, 256, 256, 20, 200, 40, 32), 3, 1, 40)
    t = util.gen_o1(case, half=True)
    gw, _ = run(case, t, 16, "wg", False)
    for mode in ("bcast",):
        gb, info = run(case, t, 16, mode, False)
        same = np.array_equal(gb["y"], gw["y"])
        log("B=200", info, "== wg:", same)
        if not same:
            bad = np.argwhere(gb["y"] != gw["y"])
            log(" mismatching utterances:", sorted(set(bad[:, 0].tolist()))[:20], "first t:", int(bad[:, 1].min()))

    # 4. C2 over 1100 samples (every ring wraps, d = 512 live twice), S = 128 shape
    case = T.O1_CASES["C2"]
    t = util.gen_o1(case, half=True)
    gw, _ = run(case, t, 16, "wg", False)
    gb, info = run(case, t, 16, "bcast", False)
    log("C2 N=1100", info, "== wg:", np.array_equal(gb["y"], gw["y"]))
    case = T.O1_CASES["oddL_ragged"]
    t = util.gen_o1(case, half=True)
    gw, _ = run(case, t, 16, "wg", False)
    gb, info = run(case, t, 16, "bcast", False)
    same = np.array_equal(gb["y"], gw["y"])
    log("odd L, ragged", info, "== wg:", same)
    if not same:
        bad = np.argwhere(gb["y"] != gw["y"])
        log(" first mismatch t:", int(bad[:, 1].min()), "utterances", sorted(set(bad[:, 0].tolist())))

if "time" in what:
    import torch
    import bench
    w = bench.make_weights()
    ncu = torch.cuda.get_device_properties(0).multi_processor_count
    pts = [("wg3", 4, 48 * ncu), ("bcast1", 8, 64 * ncu), ("auto", 0, 96 * ncu), ("bcast1", 8, 64)]
    if os.environ.get("R4_POINTS"):       # e.g. "bcast1:8:64,wg3:4:12288" (name : organisation code : utterances)
        pts = [(a, int(b), int(c)) for a, b, c in (x.split(":") for x in os.environ["R4_POINTS"].split(","))]
    for name, org, B in pts:
        t0 = time.time()
        e, N, keep = bench.steady_engine(w, B, 128, organisation=org)
        e.setClockProbe(True)
        info = e.kernelInfo(B, False)
        ms = bench.time_range(e, bench.STEADY_FROM, 128, N, B)
        ms = bench.time_range(e, bench.STEADY_FROM, 128, N, B, reps=3)
        ghz = e.lastLaunchClockGHz()
        us = 1e3 * ms / 128
        log("%-7s B=%6d  %.2f us/sample  %.2f kHz  %.1f M samples/s  clock %.3f GHz  %.0f clk/sample  [%s] (%.0f s)" %
            (name, B, us, 1e3 / us, B / us, ghz, us * 1e3 * ghz, info.split(" ")[0], time.time() - t0))
        e.close()
        del keep
        torch.cuda.empty_cache()
