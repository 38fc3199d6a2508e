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


def test_energy_roofline_prices_the_algorithms_counts(tmp_path, monkeypatch):
    """roofline.energy of bench.py (round 6): SURVEY.md 8(d)'s operation counts per utterance-sample priced with the marginal energies of
    profiles/r06_energy_ubench.json, against the socket power of the timed launch minus the power of resident idle waves.  On the
    committed price list and a canned power reading: the parts add up to the floor, the ring is priced separately, fewer ring layers in
    HBM cost less, four tiles per workgroup stream fewer weight bytes per utterance than three, and the fraction is floor / achieved."""
    import bench
    power = {"socket_w": 1380.0, "ms_per_launch": 9.5, "shader_clock_ghz": 2.0, "limit_w": 1400.0}
    B, N = 12288, 256
    e3 = bench.energy_roofline(power, 9.5, B, N, 3)
    assert e3 and "error" not in e3, e3
    assert abs(sum(e3["floor_parts_uj"].values()) - e3["floor_uj"]) < 1e-3
    assert abs(e3["floor_with_ring_uj"] - e3["floor_uj"] - e3["ring_uj"]) < 1e-3
    assert abs(e3["frac"] - e3["floor_uj"] / e3["achieved_uj"]) < 1e-9 and 0.3 < e3["frac"] < 1.0
    want = (1380.0 - e3["idle_resident_waves_w"]) * 9.5e-3 / (B * N) * 1e6
    assert abs(e3["achieved_uj"] - want) < 1e-9 and e3["achieved_total_uj"] > e3["achieved_uj"]
    c = e3["counts_per_utterance_sample"]
    assert c["mfma_16x16x32"] == 106.0 and c["hbm_compulsory_bytes"] == 5128 and c["weight_stream_bytes"] == bench.HEAD.weight_bytes / 48.0
    e4 = bench.energy_roofline(power, 9.5, B, N, 4)
    assert e4["floor_parts_uj"]["l2_weight_stream"] < e3["floor_parts_uj"]["l2_weight_stream"]
    e18 = bench.energy_roofline(power, 9.5, B, N, 3, ring_layers_in_hbm=18)
    assert abs(e18["ring_uj"] - 0.9 * e3["ring_uj"]) < 1e-3
    # a byte of HBM costs far more than a MAC: what the round's levers were priced with
    p = e3["prices_nj"]
    assert p["hbm"] / 1024 > 50 * p["mfma16"] / 8192 and p["mfma32"] / 16384 > 0.9 * p["mfma16"] / 8192
    assert bench.energy_roofline(None, 9.5, B, N, 3) is None
