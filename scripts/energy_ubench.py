#!/usr/bin/env python3
"""What an operation costs in joules on this MI355X: runs scripts/ubench/energy (one kind of operation back to
back on N CUs, one wave per SIMD like the headline kernel) while polling the SMU's gpu_metrics table (scripts/clock_probe.py:
socket power, energy accumulator, XCD clocks), and prices every kind at its MARGINAL energy

    E(op) = (P_mode - P_loop) / (operations per second)        P_loop: the same launch shape running an empty scalar loop

Output: gpurun_out/r06_energy_ubench.json (raw points + prices) and .txt (the table); copy both to profiles/.
bench.py prices the algorithm's operation counts with profiles/r06_energy_ubench.json (`roofline.energy`).

usage: energy_ubench.py [--wgs 256,128] [--seconds 4] [--modes loop,mfma16,...]
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "scripts"))
import clock_probe  # noqa: E402

EXE = os.path.join(ROOT, "scripts", "ubench", "energy")
UNITS = {"loop": "loop iteration (4 x s_nop 15)", "mfma16": "v_mfma_f32_16x16x32_f16 (8 192 MAC)", "mfma32": "v_mfma_f32_32x32x16_f16 (16 384 MAC)",
         "valu": "v_fma_f32 wave instruction (64 lanes)", "trans": "v_exp_f32 wave instruction (64 lanes)", "lds": "KiB read from LDS (ds_read_b128)",
         "l2": "KiB L2 -> CU (buffer_load_dwordx4, L2-resident)", "hbm": "KiB read from HBM (buffer_load_dwordx4 nt)",
         "hbmw": "KiB written to HBM (buffer_store_dwordx4 nt)"}


def build():
    src = os.path.join(ROOT, "scripts", "ubench", "energy.hip")
    if not os.path.exists(EXE) or os.path.getmtime(EXE) < os.path.getmtime(src):
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", src, "-o", EXE])


def run_point(mode, wgs, seconds):
    smp = clock_probe.Sampler("metrics")
    smp.start()
    r = subprocess.run([EXE, mode, str(wgs), str(seconds)], capture_output=True, text=True, timeout=120 + 4 * seconds)
    smp.stop = True
    smp.join()
    if r.returncode != 0:
        return {"mode": mode, "wgs": wgs, "error": r.stderr[-300:]}
    pt = json.loads(r.stdout.strip().split("\n")[-1])
    busy = clock_probe.summarise(smp.samples, pt["t0"], pt["t1"])      # (CLOCK_MONOTONIC on both sides)
    pt.update({"socket_w": busy.get("power_w"), "socket_w_from_energy_accumulator": busy.get("power_w_from_energy_accumulator"),
               "sclk_mhz": busy.get("sclk_mhz"), "hotspot_c": busy.get("hotspot_c_max"), "polls": busy.get("polls")})
    return pt


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--wgs", default="256,128")
    ap.add_argument("--seconds", type=float, default=4.0)
    ap.add_argument("--modes", default="loop,mfma16,mfma32,valu,trans,lds,l2,hbm,hbmw")
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "r06_energy_ubench"))
    args = ap.parse_args()
    build()
    if "power_w" not in clock_probe.read_metrics():
        sys.exit("no gpu_metrics telemetry on this box")
    doc = {"source": "scripts/energy_ubench.py: scripts/ubench/energy under rocm-smi --showmetrics (gpu_metrics: current_socket_power, energy_accumulator, "
                     "current_gfxclks); one workgroup of four waves per CU (one wave per SIMD)", "seconds_per_point": args.seconds, "points": [], "prices": {}}
    idle = clock_probe.read_metrics()
    doc["idle"] = {"socket_w": idle.get("power_w"), "sclk_mhz": idle.get("sclk_mhz")}
    lines = ["%-7s %4s %10s %9s %14s %12s   %s" % ("mode", "wgs", "socket W", "sclk MHz", "ops / s", "pJ / op", "op")]
    for wgs in [int(x) for x in args.wgs.split(",")]:
        base = None
        for mode in args.modes.split(","):
            pt = run_point(mode, wgs, args.seconds)
            doc["points"].append(pt)
            if "error" in pt or pt.get("socket_w") is None:
                lines.append("%-7s %4d  failed: %s" % (mode, wgs, pt.get("error", "no telemetry")))
                continue
            # the accumulator-derived power integrates over the window; the instantaneous readings are the fallback
            p = pt.get("socket_w_from_energy_accumulator") or pt["socket_w"]
            pt["power_used_w"] = p
            if mode == "loop":
                base = p
            pj = None
            if base is not None and mode != "loop":
                pj = (p - base) / pt["ops_per_s"] * 1e12
                pt["pj_per_op"] = pj
                doc["prices"].setdefault(str(wgs), {})[mode] = {"pj_per_op": pj, "sclk_mhz": pt["sclk_mhz"], "socket_w": p, "baseline_w": base, "op": UNITS[mode]}
            elif mode == "loop":
                doc["prices"].setdefault(str(wgs), {})["loop"] = {"socket_w": p, "sclk_mhz": pt["sclk_mhz"], "op": UNITS[mode]}
            lines.append("%-7s %4d %10.1f %9.0f %14.4e %12s   %s" % (mode, wgs, p, pt["sclk_mhz"] or 0, pt["ops_per_s"], "-" if pj is None else "%.1f" % pj, UNITS[mode]))
            print(lines[-1], file=sys.stderr, flush=True)
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    json.dump(doc, open(args.out + ".json", "w"), indent=1)
    hdr = ["# Marginal energy per operation on MI355X, one wave per SIMD (scripts/energy_ubench.py; VERDICT r5 #2)",
           "# E(op) = (socket power of the mode - socket power of the empty loop at the same workgroup count) / operations per second",
           "# idle socket: %s W at %s MHz" % (doc["idle"]["socket_w"], doc["idle"]["sclk_mhz"]), ""]
    open(args.out + ".txt", "w").write("\n".join(hdr + lines) + "\n")
    print("\n".join(lines))


if __name__ == "__main__":
    main()
