#!/usr/bin/env python3
"""Round 6: one lever, one line.  For the library NVW_LIB selects (scripts/build_variant.sh builds), the steady-state launch at the
given batches: us per sample, kHz per utterance, the clock the launch ran at, socket watts and uJ per utterance-sample (the SMU's
gpu_metrics table while the launch repeats for `--seconds`), and a checksum of the samples of a small O(1) run (an experiment build
must reproduce the shipped library's unless it says why not).  Appends JSON lines to gpurun_out/r6_ab.jsonl.

usage: NVW_LIB=... gpu_r6_ab.py <tag> [--batches 12288,16384] [--mode packed|features] [--org 0] [--seconds 3]
"""
import argparse
import json
import os
import sys
import zlib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "scripts"))


def checksum(modes=("wg3",)):
    import numpy as np
    import cases
    import util
    import test_parity_gpu as T
    case = cases.Case("ab", 33, [], cases.Shape(64, 256, 256, 20, 48, 600, 512), 3, 1, 600)
    t = util.gen_o1(case, half=True)
    out = {}
    for mode in modes:
        e = T._engine_o1(case, t, 16, mode)
        y = np.full((48, 600), -1, dtype=np.int32)
        assert e.run(600, 48, y, 1, False)
        e.synchronize()
        e.close()
        out[mode] = "%08x" % zlib.crc32(y.tobytes())
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("tag")
    ap.add_argument("--batches", default="12288")
    ap.add_argument("--mode", default="packed")
    ap.add_argument("--org", type=int, default=0)
    ap.add_argument("--seconds", type=float, default=3.0)
    ap.add_argument("--no-power", action="store_true")
    ap.add_argument("--crc-modes", default="wg3")
    ap.add_argument("--ring", type=int, default=0, help="nvw_set_ring_in_lds mode of the timed launches (1: as many short-dilation layers as fit)")
    args = ap.parse_args()
    import torch
    import bench
    w = bench.make_weights()
    crc = checksum(tuple(args.crc_modes.split(","))) if args.crc_modes else {}
    for B in [int(b) for b in args.batches.split(",")]:
        n = 256
        e, N, keep = bench.steady_engine(w, B, n, 11, None if args.mode == "packed" else args.mode, organisation=args.org)
        e.setClockProbe(True)
        e.setRingInLds(args.ring)
        ms = min(bench.time_range(e, bench.STEADY_FROM, n, N, B) for _ in range(3))
        ghz = e.lastLaunchClockGHz()
        info = e.kernelInfo(B, False)
        rec = dict(tag=args.tag, ring=args.ring, lib=os.environ.get("NVW_LIB", "shipped"), B=B, mode=args.mode, us_per_sample=round(1e3 * ms / n, 3), khz=round(n / ms, 3),
                   msamples_per_s=round(B * n / ms / 1e3, 1), clock_ghz=round(ghz, 3), cycles_per_sample=round(1e3 * ms / n * ghz * 1e3), kernel=info, crc=crc)
        if not args.no_power:
            stream = torch.cuda.current_stream().cuda_stream
            pw = bench.power_reading(e, n, N, B, stream, seconds=args.seconds)
            if pw and "socket_w" in pw:
                rec.update(socket_w=pw["socket_w"], sclk_mhz=pw["sclk_mhz"], uj_per_utterance_sample=round(pw["uj_per_utterance_sample"], 4),
                           us_per_sample_under_power=round(1e3 * pw["ms_per_launch"] / n, 3))
            else:
                rec["power"] = pw
        e.close()
        del keep
        torch.cuda.empty_cache()
        print(json.dumps(rec), flush=True)
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        open(os.path.join(ROOT, "gpurun_out", "r6_ab.jsonl"), "a").write(json.dumps(rec) + "\n")


if __name__ == "__main__":
    main()
