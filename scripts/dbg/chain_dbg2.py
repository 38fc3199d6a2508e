import sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import cases, util
np.set_printoptions(linewidth=200, precision=4)
case = cases.BY_NAME["R64S256A256_impl3"]
s = case.shape
t = util.gen_inputs(case)
o = util.make_oracle(case, t)
res = {}
for mode in ("wg", "chain"):
    e = util.make_engine(case, t, precision=32, mode=mode)
    y = np.full((s.B, s.N), -1, dtype=np.int32)
    assert e.run(s.N, s.B, y, 1, True)
    e.synchronize()
    res[mode] = util.engine_getters(e, s.L)
    e.close()
o.run(s.N)
ref = o.getters()
o2 = util.make_oracle(case, t); o2.run(s.N - 1); refm1 = o2.getters()
for k in ("Xout", "skipOut", "Zs", "Za"):
    a, b, r = res["wg"][k], res["chain"][k], ref[k]
    print(k, "wg==chain exact:", np.array_equal(a, b), " max|wg-ref|", np.abs(a - r).max(), " max|chain-ref|", np.abs(b - r).max(),
          " max|chain-ref(t-1)|", np.abs(b - refm1[k]).max())
g, r = res["chain"]["Xout"], ref["Xout"]
print("layer0 utt0 ref ", r[0, 0, :16]); print("layer0 utt0 got ", g[0, 0, :16]); print("layer0 utt0 t-1 ", refm1["Xout"][0, 0, :16])
good = np.abs(g - r) <= 1e-2 * np.abs(r) + 1e-7
print("good pattern layer0 utt0:", good[0, 0].astype(int))
print("good pattern layer0 utt1:", good[0, 1].astype(int))
g, r = res["chain"]["skipOut"], ref["skipOut"]
print("skip layer4 utt0 ref ", r[4, 0, :12]); print("skip layer4 utt0 got ", g[4, 0, :12]); print("skip layer4 t-1", refm1["skipOut"][4, 0, :12])
