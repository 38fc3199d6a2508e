#!/usr/bin/env python3
"""round-3 debugging aid: per-layer errors of an O(1) case against the teacher-forced oracle."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import cases, util
import test_parity_gpu as T

def show(name, mode, prec, chunk=None, N=None):
    case = T.O1_CASES[name]
    if N: case = case._replace(shape=case.shape._replace(N=N))
    s = case.shape
    t = util.gen_o1(case, half=(prec == 16))
    e = T._engine_o1(case, t, prec, mode)
    got = T._run_dumped(e, case, chunk=chunk)
    info = e.kernelInfo()
    e.close()
    ref = util.teacher_forced_oracle(case, t, got["y"])
    u = util.FP16_U if prec == 16 else 1.19e-7
    xe = [float(np.abs(got["Xout"][l] - ref["Xout"][l]).max() / (u * np.abs(ref["Xout"][l]).max())) for l in range(s.L)]
    ke = [float(np.abs(got["skipOut"][l] - ref["skipOut"][l]).max() / (u * np.abs(ref["skipOut"][l]).max())) for l in range(s.L)]
    print("%s %s fp%d chunk=%s N=%d  %s" % (name, mode, prec, chunk or case.chunk, s.N, info.split(" ")[0]))
    print("  agreement %.4f" % (got["y"] == ref["y"]).mean())
    print("  Xout units/layer:", " ".join("%.1f" % v for v in xe))
    print("  skip units/layer:", " ".join("%.1f" % v for v in ke))
    for k in ("Zs", "Za"):
        print("  %s units: %.1f" % (k, np.abs(got[k] - ref[k]).max() / (u * np.abs(ref[k]).max())))
    bad = np.argwhere(np.abs(got["Xout"][0] - ref["Xout"][0]) > 50 * u * np.abs(ref["Xout"][0]).max())
    if len(bad):
        print("  Xout[0] bad (utt, ch) first 12:", bad[:12].tolist(), "utts:", sorted(set(bad[:, 0].tolist()))[:20], "chs:", sorted(set(bad[:, 1].tolist()))[:40])

for a in sys.argv[1:]:
    name, mode, prec, chunk, N = (a.split(":") + ["", ""])[:5]
    show(name, mode, int(prec), int(chunk) if chunk else None, int(N) if N else None)
