#!/usr/bin/env python3
"""round 4 debug: wavenet_bcast at full chip on the benchmarked sequence vs the one-tile wavenet_wg: where do they differ?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import cases, util, bench
from nv_wavenet_amd import WavenetEngine
n_timed, seed = 64, 111
N = int(os.environ.get("R4_N", bench.STEADY_FROM + n_timed))
case = cases.Case("C3_dbg", 30, [], cases.Shape(64, 256, 256, 20, 16, N, 512), 3, 1, 128)
s = case.shape
t = util.gen_o1(case, half=True)
block = np.ascontiguousarray(t.Lh[:bench.COND_BLOCK])
def sequence(B, org):
    e = WavenetEngine(s.R, s.S, s.A, s.L, s.maxD, B, N, impl=0, tanhEmbed=True, precision=16, organisation=org)
    e.setEmbeddings(t.embP, t.embC)
    for l in range(s.L):
        e.setLayerWeights(l, t.Wprev[l], t.Wcur[l], t.Bh[l], t.Wres[l], t.Bres[l], t.Wskip[l], t.Bskip[l])
    e.setOutWeights(t.Wzs, t.Bzs, t.Wza, t.Bza)
    idx = torch.arange(B, device="cuda") % s.B
    blk = torch.from_numpy(block).cuda()[:, :, idx, :].contiguous()
    e.setSelectorSeed(seed); e.resetHistory()
    for first in range(0, N, bench.COND_BLOCK):
        e.packConditioning(blk, first, min(bench.COND_BLOCK, N - first))
    torch.cuda.synchronize()
    info = e.kernelInfo(B, False)
    assert e.run_partial_chunk(0, N, N, B)
    e.synchronize()
    y = torch.full((B, N), -1, dtype=torch.int32, device="cuda")
    e.getYOut(y, 0, N); e.synchronize(); e.close()
    return y.cpu().numpy(), info
ncu = torch.cuda.get_device_properties(0).multi_processor_count
ref, _ = sequence(4096, 2)
for org, B in ((8, 64 * ncu), (9, 128 * ncu), (8, 4096)):
    for rep in range(2):
        y, info = sequence(B, org)
        d = (y[:4096] != ref)
        bad = np.argwhere(d.any(axis=1))[:, 0]
        first_t = [int(np.argmax(d[b])) for b in bad]
        print(info.split(" ")[0], "B", B, "rep", rep, ": differing utterances (of first 4096):", len(bad), " first t min/median:",
              (min(first_t), int(np.median(first_t))) if first_t else None, " tiles:", sorted(set((bad // 16).tolist()))[:12], flush=True)
