"""Pre-flight of bench.py's launch shapes on ONE GPU, so that the first multi-GPU run cannot die on plumbing
(VERDICT r2 item 9): the collective path forced at world size 1, and the plain launch (`python bench.py --gpus N` with no
launcher around it) through its parent -> child route, each asserting the one-line JSON contract."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SMALL = ["--gpus", "1", "--steps", "2", "--warmup", "1", "--rows", "200000", "--e2e-images", "4096", "--no-sweep",
         "--embed-steps", "2", "--concurrent-queries", "128", "--cpu-sample-rows", "50000"]


def _run(extra_env):
    env = dict(os.environ, **extra_env)
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    env.pop("LOCAL_RANK", None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + SMALL, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       env=env, timeout=1500)
    assert p.returncode == 0, p.stderr.decode(errors="replace")[-3000:]
    lines = [ln for ln in p.stdout.decode().splitlines() if ln.strip()]
    assert len(lines) == 1, lines  # exactly ONE line on stdout: the JSON
    return json.loads(lines[0])


def _check_schema(out, distributed):
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                "dtype", "data", "config", "roofline", "cpu_baseline", "launch"):
        assert key in out, key
    assert out["n_gpus"] == 1 and out["steps"] == 2 and out["warmup"] == 1 and out["value"] > 0
    assert "workload" in out["config"] and out["dtype"] == "u8" and out["vs_baseline"] is None
    r = out["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    c = out["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] == 1 and c["host_cores"] >= 1 and c["value"] > 0
    assert out["embed"]["roofline"]["bound"] == "mfma" and out["embed"]["cpu_baseline"]["host_cores"] >= 1
    e = out["end_to_end"]
    assert e["parity"]["parity_ok"] == e["parity"]["parity_checked"] > 0
    if distributed:
        assert "weak_scaling" in out and out["launch"]["uses_rccl"] and out["launch"]["n_ranks_seen"] == 1
        assert out["launch"]["shard_rows_per_rank"] == [200000]


@pytest.mark.gpu
def test_bench_collective_path_at_world_size_one():
    out = _run({"PIXELBOX_FORCE_DIST": "1"})
    _check_schema(out, distributed=True)


@pytest.mark.gpu
def test_bench_plain_launch_spawns_its_ranks_and_the_in_library_leg():
    out = _run({"PIXELBOX_FORCE_SPAWN": "1", "PIXELBOX_FORCE_DIST": "1"})
    _check_schema(out, distributed=True)
    assert out["launch"]["spawned_by_plain_launch"] is True
    lib = out["in_library_sharded"]
    assert lib["value"] > 0 and lib["n_shards"] == 1 and lib["rows_total"] == 200000 and lib["shard_rows"] == [200000]
    # the product form answers the same first query with the same first hit as the rank form
    assert lib["check"]["first_result_id"] == out["check"]["first_result_id"]
