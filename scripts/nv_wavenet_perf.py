#!/usr/bin/env python3
"""nv_wavenet_perf -- flag-compatible equivalent of the reference's perf harness
(/root/reference/nv_wavenet_perf.cu:203-281): same options, same output lines, same metric
("Sample rate: %f kHz" = num_samples / elapsed_ms per utterance, measured around run_chunks
including the per-chunk copies of the samples to the host).

  -l layers  -r R  -s S  -a A  -b batch  -c batch_size_per_block  -n samples  -d max_dilation
  -m mode (0 AUTO 1 SINGLE 2 DUAL 3 PERSISTENT 4 MANYBLOCK)  -p precision (16|32)
  -t samples_per_chunk  -f device
Extension: -o organisation (0 = from -m and the batch size, 1..6 and 10 see include/nv_wavenet_c.h: nvw_create_ex);
the line "kernel: ..." reports the device code that ran.
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np


def main():
    ap = argparse.ArgumentParser(add_help=False)
    for flag, name, default in (("-l", "num_layers", 20), ("-r", "r", 64), ("-s", "s", 128), ("-a", "a", 256),
                                ("-b", "batch_size", 1), ("-c", "batch_size_per_block", 1),
                                ("-n", "num_samples", 16384), ("-d", "max_dilation", 512), ("-m", "mode", 0),
                                ("-p", "precision", 16), ("-t", "num_samples_per_chunk", 2048), ("-f", "device", 0),
                                ("-o", "organisation", 0)):
        ap.add_argument(flag, dest=name, type=int, default=default)
    o = ap.parse_args()
    import torch
    torch.cuda.set_device(o.device)
    from nv_wavenet_amd import WavenetEngine
    print("R: %d\nS: %d\nA: %d\nnum layers: %d\nmax dilation: %d\nbatch size: %d\nbatch size per block: %d\nnum samples: %d"
          % (o.r, o.s, o.a, o.num_layers, o.max_dilation, o.batch_size, o.batch_size_per_block, o.num_samples))
    print("mode: %s" % ["AUTO", "SINGLE_block", "DUAL_block", "PERSISTENT", "MANYBLOCK"][o.mode])
    print("precision: fp%d" % o.precision)
    rng = np.random.default_rng(1)   # the reference seeds srand(1)
    R, S, A, L, B, N = o.r, o.s, o.a, o.num_layers, o.batch_size, o.num_samples
    u = lambda sc, *shape: ((rng.random(shape, dtype=np.float32) - 0.5) * sc).astype(np.float32)
    e = WavenetEngine(R, S, A, L, o.max_dilation, B, N, impl=o.mode, precision=o.precision, organisation=o.organisation)
    print("kernel: %s" % e.kernelInfo(B, False))
    # the reference uploads uniform [-0.5,0.5] weights and leaves embeddings / conditioning
    # uninitialised (nv_wavenet_perf.cu:40-63); here the parity recipe's scales keep values finite
    e.setEmbeddings(u(0.5 / R, A, R), u(0.5 / R, A, R))
    for l in range(L):
        e.setLayerWeights(l, u(0.25 / R, R, 2 * R), u(0.25 / R, R, 2 * R), u(0.25 / R, 2 * R), u(0.5 / R, R, R),
                          u(0.5 / R, R), u(0.5 / S, R, S), u(0.5 / S, S))
    e.setOutWeights(u(0.5 / R, S, A), u(0.5 / R, A), u(0.5 / R, A, A), u(0.5 / R, A))
    g = torch.Generator(device="cuda").manual_seed(1)
    Lh = torch.empty(N, L, B, 2 * R, dtype=torch.float32, device="cuda").uniform_(-0.25 / R, 0.25 / R, generator=g)
    sel = torch.rand(N, B, dtype=torch.float32, device="cuda", generator=g)
    e.setInputs(Lh, sel)
    y = torch.zeros(B, N, dtype=torch.int32).pin_memory()
    e.run(min(N, 64), B)            # warm-up (code objects, clocks)
    e.synchronize()
    e.setInputs(Lh, sel)            # history back to silence, like a fresh utterance
    del Lh
    t0 = time.perf_counter()
    ok = e.run_chunks(o.num_samples_per_chunk, None, N, B, y.numpy(), o.batch_size_per_block)
    e.synchronize()
    ms = 1e3 * (time.perf_counter() - t0)
    st = e.chainStatus()
    if st:
        print("multi-CU hand-off timed out: code 0x%x" % st)
    print("Sample rate: %f kHz" % (N / ms if ok and not st else 0.0))


if __name__ == "__main__":
    main()
