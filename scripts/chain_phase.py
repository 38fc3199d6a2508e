#!/usr/bin/env python3
"""Phase timing of the multi-CU chain (experiment build: make -C nv_wavenet_amd/csrc with EXTRA=-DWN_CHAIN_TIMING
into another library, loaded through NVW_LIB).  Prints, per stage, the mean time (us) between the wall-clock
stamps of wn_chain.hpp over samples 8..23 of a launch, and the hop latencies between stages."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nv_wavenet_amd import WavenetEngine

def main():
    R, S, A, L, B = [int(x) for x in sys.argv[1:6]]
    org = int(sys.argv[6]) if len(sys.argv) > 6 else 5
    prec = int(sys.argv[7]) if len(sys.argv) > 7 else 16
    N = 64
    rng = np.random.default_rng(1)
    u = lambda sc, *shape: ((rng.random(shape, dtype=np.float32) - 0.5) * sc).astype(np.float32)
    e = WavenetEngine(R, S, A, L, 512, B, N, impl=0, precision=prec, organisation=org)
    info = e.kernelInfo(B, True)
    print(info)
    K = int(info.split("stages=")[1].split()[0])
    e.setEmbeddings(u(0.5 / R, A, R), u(0.5 / R, A, R))
    for l in range(L):
        e.setLayerWeights(l, u(0.25 / R, R, 2 * R), u(0.25 / R, R, 2 * R), u(0.25 / R, 2 * R), u(0.5 / R, R, R),
                          u(0.5 / R, R), u(0.5 / S, R, S), u(0.5 / S, S))
    e.setOutWeights(u(0.5 / R, S, A), u(0.5 / R, A), u(0.5 / R, A, A), u(0.5 / R, A))
    e.setInputs(u(0.5 / R, N, L, B, 2 * R), rng.random((N, B), dtype=np.float32))
    for it in range(2):
        e.setInputs(u(0.5 / R, N, L, B, 2 * R), rng.random((N, B), dtype=np.float32))
        assert e.run(N, B, None, 1, False)
        e.synchronize()
    assert e.chainStatus() == 0
    ncu = torch.cuda.get_device_properties(0).multi_processor_count
    chains = min(ncu // K, (B + 15) // 16)
    nq = -(-((B + 15) // 16) // chains)                   # tiles of chain 0
    allq = e.getP().view(np.uint64).reshape(-1)[:K * nq * 8 * 16].reshape(K, nq, 8, 16).astype(np.int64)
    us = lambda a: a * 0.01
    if nq > 1:
        # several tiles per chain: when does every stage see the tiles of chain 0, relative to tile 0's arrival (sample 8..15 mean)
        for s in range(K):
            ev = 1 if s == K - 1 else 2                   # head: skip received; layer stage: x received
            line = "  ".join("q%d +%.2f (busy %.2f)" % (q, us((allq[s, q, :, ev] - allq[s, 0, :, ev]).mean()),
                                                        us((allq[s, q, :, 4 if s == K - 1 else 7] - allq[s, q, :, 0]).mean())) for q in range(nq))
            print("stage %2d arrivals: %s" % (s, line))
    raw = allq[:, 0]
    names = ["idle work", "wait x", "layers", "send x", "wait skip", "skip gemm", "send skip"]
    tot = us(np.diff(raw[K - 1, :, 3]).mean())
    print("sample period (head pick to pick): %.2f us" % tot)
    for s in range(K - 1):
        d = us(np.diff(raw[s], axis=1)[:, :7].mean(0))
        print("stage %2d: " % s + "  ".join("%s %.2f" % (n, v) for n, v in zip(names, d)))
    fine = ["(x recv->)xb read", "cur gemm+cond", "gate+put h", "barrier h", "get hb", "res gemm", "put x+barrier"]
    for s in range(0, K - 1, max(1, (K - 1) // 4)):
        r = raw[s]
        ev = np.stack([r[:, 2], r[:, 8], r[:, 9], r[:, 10], r[:, 11], r[:, 12], r[:, 13], r[:, 14]], 1)
        print("stage %2d layer 0 fine: " % s + "  ".join("%s %.3f" % (n, v) for n, v in zip(fine, us(np.diff(ev, axis=1).mean(0)))))
    h = raw[K - 1]
    print("head: wait skip %.2f  zs+za %.2f  softmax %.2f  embed+send %.2f" % tuple(us(np.diff(h, axis=1)[:, :4].mean(0))))
    # hops: x sent by stage s (event 4) -> received by stage s+1 (event 2); head x0 sent (4) -> stage 0 received (2) of next sample
    for s in range(K - 2):
        print("hop x %d->%d: %.2f us (send done -> received)" % (s, s + 1, us((raw[s + 1, :, 2] - raw[s, :, 4]).mean())))
    print("hop skip last->head: %.2f us" % us((raw[K - 1, :, 1] - raw[K - 2, :, 7]).mean()))
    print("hop head->0: %.2f us" % us((raw[0, 1:, 2] - raw[K - 1, :-1, 4]).mean()))
    print("critical path per sample: stage residence (x received -> x sent): " +
          " ".join("%.2f" % us((raw[s, :, 4] - raw[s, :, 2]).mean()) for s in range(K - 2)))
    print("last layer stage: x received -> skip sent %.2f" % us((raw[K - 2, :, 7] - raw[K - 2, :, 2]).mean()))
    print("head residence: skip received -> x0 sent %.2f" % us((raw[K - 1, :, 4] - raw[K - 1, :, 1]).mean()))
    e.close()

if __name__ == "__main__":
    main()
