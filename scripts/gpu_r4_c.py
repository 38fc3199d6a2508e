#!/usr/bin/env python3
"""round 4 debug: wavenet_bcast vs wavenet_wg, per-layer dumps after 1 sample"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import cases, util
import test_parity_gpu as T

def run(case, t, prec, mode, n):
    e = T._engine_o1(case, t, prec, mode)
    y = np.full((case.shape.B, case.shape.N), -1, dtype=np.int32)
    assert e.run(n, case.shape.B, y, 1, True)
    e.synchronize()
    got = util.engine_getters(e, case.shape.L)
    got["y"] = y[:, :n].copy()
    e.close()
    return got

np.set_printoptions(linewidth=200, precision=3, suppress=True)
for prec in (32,):
    case = cases.Case("C3_dbg", 30, [], cases.Shape(64, 256, 256, 20, 16, 8, 32), 3, 1, 8)
    t = util.gen_o1(case, half=(prec == 16))
    for variant in ("plain", "nocond", "nocur"):
        tt = util.gen_o1(case, half=(prec == 16))
        if variant == "nocond":
            tt.Lh[:] = 0
        if variant == "nocur":
            tt.Wcur[:] = 0
        a = run(case, tt, prec, "wg", 1)
        b = run(case, tt, prec, "bcast", 1)
        for l in (0, 1, 2):
            d = np.abs(a["Xout"][l] - b["Xout"][l])
            print(variant, "layer", l, "Xout diff max %.3g; by 16-row block:" % d.max(), [round(float(d[:, i*16:(i+1)*16].max()), 3) for i in range(4)],
                  "by utterance:", [round(float(v), 2) for v in d.max(axis=1)])
        sys.stdout.flush()
