"""bench.py's own launcher: `python bench.py --gpus N` must start N ranks by itself, prove through the collective
backend that N distinct processes took part, and refuse a world that does not match --gpus.  No GPU: the
--selftest-dist mode exercises launcher + rendezvous + nv_wavenet_amd.sharding.gather_samples over gloo."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env=None):
    e = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    e.update(env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True,
                          timeout=600, cwd=ROOT, env=e)


def test_bench_spawns_its_own_ranks():
    r = _run(["--gpus", "2", "--backend", "gloo", "--selftest-dist"])
    assert r.returncode == 0, r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    out = json.loads(line)
    assert out["n_gpus"] == 2 and out["ok"] and out["gathered_rows"] == 16


def test_bench_refuses_a_world_that_does_not_match_gpus():
    r = _run(["--gpus", "2", "--backend", "gloo", "--selftest-dist"], env={"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and "--gpus 2" in (r.stderr + r.stdout)


def test_single_rank_selftest_needs_no_launcher():
    r = _run(["--selftest-dist"])
    assert r.returncode == 0, r.stderr[-2000:]
    assert json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])["n_gpus"] == 1


def test_socket_telemetry_reader_parses_the_gpu_metrics_table(monkeypatch):
    """roofline.power of bench.py and scripts/clock_probe.py read `rocm-smi --showmetrics`; the parser on a canned table (lines as a
    GPU box printed them in round 5), the energy-accumulator power over two polls, and 'no telemetry' when the tool prints nothing."""
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    import clock_probe
    table = "\n".join([
        "GPU[0]\t\t: temperature_hotspot (C): 55",
        "GPU[0]\t\t: average_socket_power (W): N/A",
        "GPU[0]\t\t: energy_accumulator (15.259uJ (2^-16)): 10231786994058",
        "GPU[0]\t\t: current_gfxclk (MHz): 1990",
        "GPU[0]\t\t: throttle_status: N/A",
        "GPU[0]\t\t: current_socket_power (W): 1375",
        "GPU[0]\t\t: current_gfxclks (MHz): [1990, 1997, 1957, 1967, 1948, 1990, 2001, 2010]",
    ])

    class R:
        returncode = 0
        stdout = table
        stderr = ""
    monkeypatch.setattr(clock_probe.subprocess, "run", lambda *a, **k: R)
    m = clock_probe.read_metrics()
    assert m["power_w"] == 1375.0 and m["hotspot_c"] == 55.0 and "throttle_status" not in m
    assert abs(m["sclk_mhz"] - 1982.5) < 1e-9 and m["sclk_mhz_min"] == 1948.0
    assert abs(m["energy_j"] - 10231786994058 * 15.259e-6) < 1.0
    a = dict(m, t=10.0)
    b = dict(m, t=12.0, energy_j=m["energy_j"] + 2750.0)
    s = clock_probe.summarise([a, b], 9.0, 12.0)                 # (the window skips its first 30 %: both polls are inside)
    assert abs(s["power_w_from_energy_accumulator"] - 1375.0) < 1e-6 and s["polls"] == 2
    R.stdout = ""
    assert "power_w" not in clock_probe.read_metrics()
