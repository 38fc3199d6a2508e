#!/usr/bin/env python3
"""wn::wavenet_split against wn::wavenet_wg on the GPU: bit-identical free-running fp16 samples on the O(1) inputs (the
organisations share arithmetic and summation order), then steady-state timing.
usage: split_check.py [check] [time] [points]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import cases, util


def engine(case, t, mode, B, Lh=None):
    from nv_wavenet_amd import WavenetEngine
    s = case.shape
    e = WavenetEngine(s.R, s.S, s.A, s.L, s.maxD, B, s.N, impl=case.impl, tanhEmbed=True, precision=16, organisation=util.MODE_ORG[mode])
    e.setEmbeddings(t.embP, t.embC)
    for l in range(s.L):
        e.setLayerWeights(l, t.Wprev[l], t.Wcur[l], t.Bh[l], t.Wres[l], t.Bres[l], t.Wskip[l], t.Bskip[l])
    e.setOutWeights(t.Wzs, t.Bzs, t.Wza, t.Bza)
    return e


def free_run(case, t, mode, B, chunk=None, in_place=False, rng=False):
    s = case.shape
    e = engine(case, t, mode, B)
    keep = None
    if in_place:
        keep = torch.from_numpy(np.ascontiguousarray(t.Lh)).cuda().to(torch.float16)
        e.setConditioningDirect(keep)
        e.setSelectors(t.sel)
    else:
        e.setInputs(t.Lh, t.sel)
    if rng:
        e.setSelectorSeed(1234)
    y = np.full((B, s.N), -1, dtype=np.int32)
    if chunk:
        for first in range(0, s.N, chunk):
            assert e.run_partial_chunk(first, min(chunk, s.N - first), s.N, B)
        e.synchronize()
        yd = torch.zeros(B, s.N, dtype=torch.int32, device="cuda")
        e.getYOut(yd, 0, s.N)
        e.synchronize()
        y = yd.cpu().numpy()
    else:
        assert e.run(s.N, B, y, 1, False)
        e.synchronize()
    info = e.kernelInfo(B, False)
    e.close()
    return y, info


def check():
    S = cases.Shape
    runs = [
        # name, shape(R,S,A,L,B,N,maxD), split mode, reference mode, kwargs
        ("C3 one tile", S(64, 256, 256, 20, 16, 48, 8), "split1", "wg", {}),
        ("C3 two tiles ragged", S(64, 256, 256, 20, 21, 40, 16), "split2", "wg", {}),
        ("C3 three tiles", S(64, 256, 256, 20, 48, 70, 32), "split3", "wg", {}),
        ("C3 three tiles, 2 wgs ragged, chunks", S(64, 256, 256, 20, 75, 45, 4), "split3", "wg", {"chunk": 13}),
        ("C3 in place fp16", S(64, 256, 256, 20, 40, 33, 8), "split3", "wg", {"in_place": True}),
        ("C3 rng selectors", S(64, 256, 256, 20, 37, 33, 8), "split2", "wg", {"rng": True}),
        ("C2 maxD512 long", S(64, 128, 256, 20, 4, 1100, 512), "split1", "wg", {"chunk": 300}),
        ("L=4", S(64, 128, 256, 4, 16, 24, 2), "split1", "wg", {}),
    ]
    bad = 0
    for name, sh, ms, mr, kw in runs:
        case = cases.Case(name, 30, [], sh, 3, 1, sh.N)
        t = util.gen_o1(case, half=True)
        t0 = time.time()
        ys, infos = free_run(case, t, ms, sh.B, **kw)
        yr, infor = free_run(case, t, mr, sh.B, **kw)
        same = np.array_equal(ys, yr)
        print("%-40s %s  [%s | %s] %.1fs" % (name, "IDENTICAL" if same else "DIFFERENT", infos.split(" ")[0], infor.split(" ")[0], time.time() - t0), flush=True)
        if not same:
            bad += 1
            d = np.argwhere(ys != yr)
            firsts = {}
            for b, n in d:
                firsts.setdefault(int(b), int(n))
            print("   first differing sample per utterance (first 24):", sorted(firsts.items())[:24])
            print("   utterances differing: %d of %d; distinct picks split %d ref %d; split range %d..%d" %
                  (len(firsts), sh.B, len(np.unique(ys)), len(np.unique(yr)), ys.min(), ys.max()))
            b0 = sorted(firsts.items())[0][0]
            print("   utt %d split:" % b0, ys[b0, :12].tolist(), " ref:", yr[b0, :12].tolist())
    print("CHECK", "OK" if bad == 0 else "FAILED (%d)" % bad, flush=True)
    return bad == 0


def timing(points):
    import bench
    w = bench.make_weights()
    ncu = torch.cuda.get_device_properties(0).multi_processor_count
    out = []
    for p in points:
        kind, bt = p[0], int(p[1])
        org = {"w": 1 + bt, "s": 7 + bt}[kind] if len(p) == 2 else {"W": 1, "S": 7}[kind]
        B = 16 * bt if len(p) == 2 else 16 * bt * ncu
        khz, info = bench.measure_steady_khz(w, B, 256, organisation=org)
        out.append("%s %.2f us (%s)" % (p, 1e3 / khz if khz else -1, info.split(" ")[0][:40]))
        print(out[-1], flush=True)


if __name__ == "__main__":
    args = sys.argv[1:] or ["check", "time"]
    ok = True
    if "check" in args:
        ok = check()
    if "time" in args and ok:
        pts = [a for a in args if a not in ("check", "time")] or ["s1", "w1", "s2", "w2", "s3", "w3", "S1g", "W1g", "S2g", "W2g", "S3g", "W3g"]
        timing(pts)
