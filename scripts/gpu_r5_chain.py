"""Round 5: the multi-CU chain with several tiles per chain (C4 by default): kHz per utterance by the reference's definition
(run_chunks, per-chunk copies to pinned memory, wall clock) against tiles per chain.  usage: gpu_r5_chain.py [C4|C3|C2] [tpc,tpc,...]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import torch  # noqa: E402


def main():
    sh0 = {"C4": bench.C4, "C3": bench.C3, "C2": bench.C2}[sys.argv[1] if len(sys.argv) > 1 else "C4"]
    tpcs = [int(v) for v in sys.argv[2].split(",")] if len(sys.argv) > 2 else [1, 2, 4, 6, 8]
    ncu = torch.cuda.get_device_properties(0).multi_processor_count
    out = []
    for tpc in tpcs:
        probe = bench.reference_definition_khz(bench.Shape(sh0.name, sh0.R, sh0.S, sh0.A, sh0.L, sh0.maxD, 16), 3, N=256, chunk=256)
        import re
        stages = int(re.search(r"stages=(\d+)", probe["kernel"]).group(1))
        chains = ncu // stages
        B = 16 * chains * tpc
        N = 4096
        while N > 512 and N * sh0.L * B * 2 * sh0.R * 4 > 70e9:      # (the harness hands over the whole fp32 conditioning tensor)
            N //= 2
        sh = bench.Shape(sh0.name, sh0.R, sh0.S, sh0.A, sh0.L, sh0.maxD, B)
        r = bench.reference_definition_khz(sh, 3, N=N, chunk=N // 2)
        # ... and the steady state (samples 640 .. 640 + 2048, conditioning packed block by block: no big tensor)
        khz, info = bench.measure_steady_khz(bench.make_weights(sh, seed=1), B, 2048, sh=sh, impl=3)
        r["steady_khz"] = round(khz, 2)
        r["tpc"] = tpc
        out.append(r)
        print(json.dumps(r), flush=True)
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "r5_chain_%s.json" % sh0.name)
    os.makedirs(os.path.dirname(path), exist_ok=True)
    json.dump(out, open(path, "w"), indent=1)


if __name__ == "__main__":
    main()
