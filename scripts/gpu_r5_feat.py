"""Round 5, first measurement: the headline launch with the conditioning packed (round 4's path) against the same launch with the
conditioning computed in the kernel from the features (wavenet_wg<.., RAW=3>); steady state, kHz per utterance and the clock."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import torch  # noqa: E402


def one(w, B, mode, n_timed=256, reps=3):
    e, N, keep = bench.steady_engine(w, B, n_timed, 11, mode)
    e.setClockProbe(True)
    ms = min(bench.time_range(e, bench.STEADY_FROM, n_timed, N, B) for _ in range(reps))
    ghz = e.lastLaunchClockGHz()
    info = e.kernelInfo(B, False)
    e.close()
    del keep
    torch.cuda.empty_cache()
    return dict(B=B, mode=str(mode), us_per_sample=round(1e3 * ms / n_timed, 2), khz=round(n_timed / ms, 2), msamples=round(B * n_timed / ms / 1e3, 1),
                clock_ghz=round(ghz, 3), kernel=info.split(" ")[0])


def main():
    w = bench.make_weights()
    out = []
    batches = [int(b) for b in sys.argv[1].split(",")] if len(sys.argv) > 1 else [12288]
    modes = sys.argv[2].split(",") if len(sys.argv) > 2 else ["packed", "features"]
    for B in batches:
        for mode in modes:
            r = one(w, B, None if mode == "packed" else mode)
            out.append(r)
            print(json.dumps(r), flush=True)
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "r5_feat.json")
    os.makedirs(os.path.dirname(path), exist_ok=True)
    json.dump(out, open(path, "a"), indent=1)


if __name__ == "__main__":
    main()
