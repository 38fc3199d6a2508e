"""Round 5: the small-batch latency figures of the BASELINE configs by the reference's definition (run_chunks of 16 384 samples, copies to
pinned memory, wall clock): C2 (B = 4), C3 (B = 16), C4 (B = 8) on the multi-CU chain.  usage: gpu_r5_latency.py [C2,C3,C4]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

for name in (sys.argv[1].split(",") if len(sys.argv) > 1 else ["C2", "C3", "C4"]):
    sh = {"C2": bench.C2, "C3": bench.C3, "C4": bench.C4}[name]
    r = [bench.reference_definition_khz(sh, 3)["khz_per_utterance"] for _ in range(3)]
    print(json.dumps({"config": name, "batch": sh.B, "khz_per_utterance": [round(v, 2) for v in r]}), flush=True)
