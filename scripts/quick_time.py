#!/usr/bin/env python3
"""Quick kernel timing on the GPU box: python scripts/quick_time.py R S A L maxD prec B N [B ...]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from nv_wavenet_amd import WavenetEngine

def main():
    R, S, A, L, maxD, prec = [int(x) for x in sys.argv[1:7]]
    N = int(sys.argv[7])
    Bs = [int(x) for x in sys.argv[8:]]
    rng = np.random.default_rng(1)
    u = lambda *s, sc=1.0: ((rng.random(s, dtype=np.float32) - 0.5) * sc).astype(np.float32)
    for B in Bs:
        e = WavenetEngine(R, S, A, L, maxD, B, N, impl=3, precision=prec)
        e.setEmbeddings(u(A, R, sc=0.5 / R), u(A, R, sc=0.5 / R))
        for l in range(L):
            e.setLayerWeights(l, u(R, 2 * R, sc=0.25 / R), u(R, 2 * R, sc=0.25 / R), u(2 * R, sc=0.25 / R),
                              u(R, R, sc=0.5 / R), u(R, sc=0.5 / R), u(R, S, sc=0.5 / S), u(S, sc=0.5 / S))
        e.setOutWeights(u(S, A, sc=0.5 / R), u(A, sc=0.5 / R), u(A, A, sc=0.5 / R), u(A, sc=0.5 / R))
        Lh = u(N, L, B, 2 * R, sc=0.5 / R)
        sel = rng.random((N, B), dtype=np.float32)
        e.setInputs(Lh, sel)
        e.time_runs(1, min(N, 64), B)  # warm-up
        ms = e.time_runs(1, N, B)
        y = np.zeros((B, N), dtype=np.int32)
        e.run(N, B, y); e.synchronize()
        print("R%d S%d A%d L%d maxD%d fp%d B=%d N=%d: %.3f ms -> %.2f us/sample, %.2f kHz/utt, %.3f Msamples/s  y[0,:8]=%s hist=%d"
              % (R, S, A, L, maxD, prec, B, N, ms, 1e3 * ms / N, N / ms, B * N / ms / 1e3, y[0, :8], len(np.unique(y))),
              flush=True)
        e.close()

if __name__ == "__main__":
    main()
