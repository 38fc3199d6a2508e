#!/usr/bin/env python3
"""Why does the full chip run the headline kernel at ~1.8 GHz?  

For a list of launch shapes -- N workgroups of three tiles at steady state (samples 640 ...), N = 1 .. CUs, and a few
smaller tilings -- this runs the generation kernel back to back for `--seconds` of wall time while a sampler thread polls
the board's power / clock / throttle telemetry (amd-smi / rocm-smi / sysfs hwmon, whatever the box offers), and records
  * us per sample and kHz per utterance,
  * the clock the launch actually ran at, from the kernel itself: shader-clock ticks of s_memtime over ticks of the
    constant-rate s_memrealtime between the first and the last instruction of workgroup 0 (nvw_set_clock_probe),
  * shader cycles per sample (the product: what the kernel costs in clocks, independent of the DVFS state),
  * telemetry averages over the busy window.
Output: one JSON document (stdout, or --out FILE).  NVW_LIB selects an experiment build of the library.

usage: clock_probe.py [--wgs 1,64,128,192,256] [--seconds 2.0] [--out FILE]
"""
import argparse
import glob
import json
import os
import re
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def read_sysfs():
    """power (W), sclk (MHz) from the amdgpu hwmon / pp_dpm files when the container exposes them"""
    out = {}
    for hw in glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*"):
        for name, key, scale in (("power1_average", "power_w", 1e-6), ("power1_input", "power_w", 1e-6), ("freq1_input", "sclk_mhz", 1e-6)):
            p = os.path.join(hw, name)
            if os.path.exists(p) and key not in out:
                try:
                    out[key] = float(open(p).read().strip()) * scale
                except Exception:
                    pass
        if out:
            break
    return out


def read_smi():
    """one amd-smi / rocm-smi poll -> dict (slow: a process start per poll)"""
    out = {}
    try:
        r = subprocess.run(["/opt/rocm/bin/amd-smi", "metric", "-g", "0", "--power", "--clock", "--json"], capture_output=True, text=True, timeout=10)
        if r.returncode == 0 and r.stdout.strip():
            j = json.loads(r.stdout)
            j = j[0] if isinstance(j, list) else j
            j = j.get("gpu_data", [j])[0] if isinstance(j, dict) and "gpu_data" in j else j
            pw = j.get("power", {})
            for k in ("socket_power", "current_socket_power", "average_socket_power"):
                v = pw.get(k)
                if isinstance(v, dict) and isinstance(v.get("value"), (int, float)):
                    out["power_w"] = float(v["value"])
                    break
            th = pw.get("throttle_status")
            if th is not None:
                out["throttle_status"] = th
            ck = j.get("clock", {})
            gfx = [v for k, v in ck.items() if k.startswith("gfx")]
            mhz = [float(g["clk"]["value"]) for g in gfx if isinstance(g, dict) and isinstance(g.get("clk"), dict) and isinstance(g["clk"].get("value"), (int, float))]
            if mhz:
                out["sclk_mhz"] = sum(mhz) / len(mhz)
                out["sclk_mhz_min"] = min(mhz)
            return out
    except Exception as e:
        out["amd_smi_error"] = str(e)[:100]
    try:
        r = subprocess.run(["/opt/rocm/bin/rocm-smi", "--showpower", "--showclocks", "--json"], capture_output=True, text=True, timeout=10)
        if r.returncode == 0 and r.stdout.strip():
            j = json.loads(r.stdout)
            c = j.get("card0", {})
            for k, v in c.items():
                if "Power" in k and "W" in k:
                    try:
                        out["power_w"] = float(v)
                    except Exception:
                        pass
                if k.startswith("sclk clock speed"):
                    m = re.search(r"(\d+)\s*Mhz", str(v), re.I)
                    if m:
                        out["sclk_mhz"] = float(m.group(1))
    except Exception as e:
        out["rocm_smi_error"] = str(e)[:100]
    return out


def read_metrics():
    """one `rocm-smi --showmetrics` poll: the SMU's gpu_metrics table (round 5: unlike the hwmon files, which read 295 W and 2.36 GHz
    idle or busy on these boxes, this table moves with the load) -> socket power (W), the eight XCDs' gfx clocks (MHz), the energy
    accumulator (J) and the hotspot temperature.  ~0.5 s per poll (a process start)."""
    out = {}
    try:
        r = subprocess.run(["/opt/rocm/bin/rocm-smi", "--showmetrics"], capture_output=True, text=True, timeout=20)
        for ln in r.stdout.split("\n"):
            m = re.match(r"GPU\[0\]\s*:\s*([a-z_]+)[^:]*:\s*(.*)$", ln)
            if not m:
                continue
            k, v = m.group(1), m.group(2).strip()
            try:
                if k == "current_socket_power":
                    out["power_w"] = float(v)
                elif k == "energy_accumulator":
                    out["energy_j"] = float(v) * 15.259e-6
                elif k == "current_gfxclks":
                    mhz = [float(x) for x in re.findall(r"\d+", v)]
                    if mhz:
                        out["sclk_mhz"], out["sclk_mhz_min"] = sum(mhz) / len(mhz), min(mhz)
                elif k == "temperature_hotspot":
                    out["hotspot_c"] = float(v)
                elif k in ("throttle_status", "indep_throttle_status") and v != "N/A":
                    out[k] = v
            except ValueError:
                pass
    except Exception as e:
        out["rocm_smi_error"] = str(e)[:100]
    return out


class Sampler(threading.Thread):
    def __init__(self, source="auto"):
        super().__init__(daemon=True)
        self.stop = False
        self.samples = []
        self.source = source
        if source == "auto":
            self.source = "metrics" if "power_w" in read_metrics() else ("sysfs" if read_sysfs() else "smi")
        self.fast = self.source == "sysfs"

    def run(self):
        while not self.stop:
            s = {"metrics": read_metrics, "sysfs": read_sysfs, "smi": read_smi}[self.source]()
            s["t"] = time.perf_counter()          # (stamped when the poll returns: the table is read at the end of the process start)
            self.samples.append(s)
            time.sleep(0.02 if self.fast else 0.05)


def summarise(samples, t0, t1):
    win = [s for s in samples if t0 + 0.3 * (t1 - t0) <= s["t"] <= t1]       # skip the ramp
    out = {"polls": len(win)}
    for k in ("power_w", "sclk_mhz", "sclk_mhz_min"):
        v = [s[k] for s in win if k in s]
        if v:
            out[k] = sum(v) / len(v)
            out[k + "_max"] = max(v)
            out[k + "_min"] = min(v)
    en = [(s["t"], s["energy_j"]) for s in win if "energy_j" in s]
    if len(en) >= 2 and en[-1][0] > en[0][0]:
        out["power_w_from_energy_accumulator"] = (en[-1][1] - en[0][1]) / (en[-1][0] - en[0][0])
    hs = [s["hotspot_c"] for s in win if "hotspot_c" in s]
    if hs:
        out["hotspot_c_max"] = max(hs)
    th = [s["throttle_status"] for s in win if "throttle_status" in s]
    if th:
        out["throttle_status"] = sorted(set(str(x) for x in th))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--wgs", default="1,32,64,128,192,256")
    ap.add_argument("--bt", default="3", help="tiles per workgroup of the sweep (organisation wg1 / wg2 / wg3 / wg4)")
    ap.add_argument("--seconds", type=float, default=2.0)
    ap.add_argument("--samples", type=int, default=256)
    ap.add_argument("--out", default="")
    ap.add_argument("--telemetry", default="auto", help="auto | metrics (rocm-smi --showmetrics) | sysfs (hwmon) | smi (amd-smi / rocm-smi --showpower)")
    ap.add_argument("--mode", default="packed", help="packed | features (conditioning computed in the kernel)")
    args = ap.parse_args()
    import torch
    import bench
    w = bench.make_weights()
    ncu = torch.cuda.get_device_properties(0).multi_processor_count
    doc = {"device": torch.cuda.get_device_name(0), "cus": ncu, "library": os.environ.get("NVW_LIB", "nv_wavenet_amd/libwavenet_infer.so"),
           "static": {}, "points": []}
    try:
        r = subprocess.run(["/opt/rocm/bin/amd-smi", "static", "-g", "0", "--limit", "--json"], capture_output=True, text=True, timeout=20)
        doc["static"]["amd_smi_limit"] = json.loads(r.stdout) if r.returncode == 0 and r.stdout.strip() else r.stderr[:300]
    except Exception as e:
        doc["static"]["amd_smi_limit"] = str(e)[:200]
    try:
        r = subprocess.run(["/opt/rocm/bin/rocm-smi", "--showmaxpower", "--showperflevel", "--showclocks", "--json"], capture_output=True, text=True, timeout=20)
        doc["static"]["rocm_smi"] = json.loads(r.stdout) if r.returncode == 0 and r.stdout.strip() else r.stderr[:300]
    except Exception as e:
        doc["static"]["rocm_smi"] = str(e)[:200]
    bt = int(args.bt)
    for nwg in [int(x) for x in args.wgs.split(",")]:
        nwg = min(nwg, ncu)
        B = 16 * bt * nwg
        n = args.samples
        e, N, keep = bench.steady_engine(w, B, n, 11, None if args.mode == "packed" else args.mode, organisation={1: 2, 2: 3, 3: 4, 4: 10}[bt])
        e.setClockProbe(True)
        info = e.kernelInfo(B, False)
        ms = bench.time_range(e, bench.STEADY_FROM, n, N, B)          # warm
        reps = max(1, int(args.seconds * 1e3 / ms))
        smp = Sampler(args.telemetry)
        smp.start()
        time.sleep(0.3 if smp.fast else 2.5)                         # idle telemetry first
        t0 = time.perf_counter()
        ms = bench.time_range(e, bench.STEADY_FROM, n, N, B, reps=reps)
        t1 = time.perf_counter()
        ghz = e.lastLaunchClockGHz()
        smp.stop = True
        smp.join()
        idle = summarise(smp.samples, smp.samples[0]["t"] - 1.0, t0) if smp.samples else {}
        busy = summarise(smp.samples, t0, t1)
        us = 1e3 * ms / n
        pt = {"workgroups": nwg, "tiles_per_wg": bt, "batch": B, "kernel": info, "launches": reps, "samples_per_launch": n,
              "us_per_sample": us, "khz_per_utterance": 1e3 / us, "samples_per_sec": B * 1e6 / us,
              "shader_clock_ghz": ghz, "shader_cycles_per_sample": us * 1e3 * ghz,
              "telemetry_source": {"sysfs": "sysfs hwmon", "metrics": "rocm-smi --showmetrics (gpu_metrics table)", "smi": "amd-smi / rocm-smi"}[smp.source],
              "telemetry_idle": idle, "telemetry_busy": busy}
        doc["points"].append(pt)
        print("wgs %3d  %.2f us/sample  %.3f GHz  %.0f clk/sample  power %s W  sclk %s MHz" %
              (nwg, us, ghz, pt["shader_cycles_per_sample"], busy.get("power_w"), busy.get("sclk_mhz")), file=sys.stderr, flush=True)
        e.close()
        del keep
        torch.cuda.empty_cache()
    s = json.dumps(doc, indent=1)
    if args.out:
        open(args.out, "w").write(s)
    else:
        print(s)


if __name__ == "__main__":
    main()
