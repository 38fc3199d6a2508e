import sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import cases, util

def run(name, mode, prec=32, chunk=None, N=None):
    case = cases.BY_NAME[name]
    s = case.shape
    t = util.gen_inputs(case, half=(prec == 16))
    o = util.make_oracle(case, t)
    e = util.make_engine(case, t, precision=prec, mode=mode)
    print("==", name, mode, prec, "chunk", chunk, e.kernelInfo())
    n = N or s.N
    y_ref = o.run(n)
    y = np.full((s.B, s.N), -1, dtype=np.int32)
    if chunk:
        assert e.run_chunks(chunk, None, n, s.B, y, 1)
    else:
        assert e.run(n, s.B, y, 1, True)
    e.synchronize()
    print("status", hex(e.chainStatus()))
    print("y equal:", np.array_equal(y[:, :n], y_ref[:, :n]), "first diff per utt:",
          [int(np.argmax(y[b, :n] != y_ref[b, :n])) if (y[b, :n] != y_ref[b, :n]).any() else -1 for b in range(s.B)])
    ref, got = o.getters(), util.engine_getters(e, s.L)
    for k in ("Xout", "skipOut"):
        r, g = ref[k], got[k]
        bad = np.abs(g - r) > 1e-2 * np.abs(r) + 1e-6
        print(k, "bad per layer:", bad.reshape(s.L, -1).sum(1).tolist())
        if bad.any():
            l = int(np.argwhere(bad)[0][0])
            print("  layer", l, "bad per utt:", bad[l].sum(1).tolist())
            print("  layer", l, "bad per channel-group(16):", bad[l].reshape(s.B, -1, 16).sum((0, 2)).tolist())
    for k in ("Zs", "Za", "P"):
        r, g = ref[k], got[k]
        print(k, "max rel err", float(np.max(np.abs(g - r) / (np.abs(r) + 1e-6))))
    e.close(); o.close()

if __name__ == "__main__":
    run("R64S256A256_impl3", "chain")
    run("R64S256A256_impl3", "chain", chunk=7)
    run("R32S128A256_impl1", "chain")
    run("R32S128A256_impl1", "chain", chunk=7)
    run("R64S256A256_impl3", "chain", prec=16)
