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
