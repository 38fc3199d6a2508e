#!/usr/bin/env python3
"""round 4: the fused conditioning producer: per-chunk time beside the generation launch, and bench.py's with_producer sweep"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from nv_wavenet_amd.nv_wavenet import get_cond_input
w = bench.make_weights()
R, L = bench.R, bench.L
h = torch.float16
for B in (12288, 8192):
    tiles = B // 16
    g = torch.Generator(device="cuda"); g.manual_seed(1)
    feats = (torch.rand(B, 80, 1, device="cuda", generator=g) - 0.5).to(h)
    up_w = ((torch.rand(80, 80, 1024, device="cuda", generator=g) - 0.5) * 0.05).to(h); up_b = torch.zeros(80, device="cuda", dtype=h)
    cw = ((torch.rand(2 * R * L, 80, 1, device="cuda", generator=g) - 0.5) * 0.5).to(h); cb = torch.zeros(2 * R * L, device="cuda", dtype=h)
    frags = torch.zeros(257, L, tiles, 4, 4, 16, 8, dtype=h, device="cuda")
    from nv_wavenet_amd.nv_wavenet import _upsample_trimmed_gemm, cond_producer_weights, produce_cond_packed
    def timed(fn, reps=3):
        fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(reps): fn()
        torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e3
    xcl = _upsample_trimmed_gemm(feats, up_w, up_b, 256, pad_to=32)
    wfrag, bpos, KF, NWF = cond_producer_weights(cw, cb, L, 16)
    print("  B=%d: upsample %.2f ms, weights %.2f ms, kernel alone %.2f ms" % (B, timed(lambda: _upsample_trimmed_gemm(feats, up_w, up_b, 256, pad_to=32)),
          timed(lambda: cond_producer_weights(cw, cb, L, 16)), timed(lambda: produce_cond_packed(xcl, wfrag, bpos, frags[:256], tiles))), flush=True)
    for fused in (True, False):
        get_cond_input(feats, up_w, up_b, 256, cw, cb, L, layout="packed", precision=16, tiles=tiles, out=frags[:256], fused=fused)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(3):
            get_cond_input(feats, up_w, up_b, 256, cw, cb, L, layout="packed", precision=16, tiles=tiles, out=frags[:256], fused=fused)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / 3 * 1e3
        print("producer B=%d fused=%d: %.2f ms per 256-sample chunk (%.1f GB written: %.2f TB/s)" % (B, fused, ms, frags[:256].numel() * 2 / 1e9, frags[:256].numel() * 2 / ms / 1e9), flush=True)
    del frags
    torch.cuda.empty_cache()
for B in (12288, 9216, 8192, 6144):
    print("with_producer B=%d: %.2f kHz" % (B, bench.with_producer_khz(w, B)), flush=True)
