"""CPU: `python bench.py --gpus N` starts its own N ranks (driver contract) -- rehearsed on gloo with MST_BENCH_DRYRUN=1
(no kernels, labelled as a dry run), plus the refusal paths: a world size that differs from --gpus never prints a line."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, **env):
    e = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    e.update(env)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=e, capture_output=True, text=True,
                          timeout=300)


def test_gpus_n_spawns_n_ranks_and_prints_one_line():
    r = _run(["--gpus", "2", "--steps", "3", "--warmup", "1"], MST_BENCH_DRYRUN="1")
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["world"] == 2 and out["rccl_ranks_seen"] == 2 and out["dry_run"] is True
    assert out["steps"] == 3 and out["value"] == 0.0 and "DRY RUN" in out["metric"]


def test_world_size_mismatch_is_refused():
    r = _run(["--gpus", "4", "--steps", "1"], MST_BENCH_DRYRUN="1", WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    assert r.returncode == 3 and "WORLD_SIZE=1" in r.stderr and not r.stdout.strip()


def test_too_few_gpus_is_refused_before_any_rank_starts():
    r = _run(["--gpus", "64", "--steps", "1"])          # no dry run: the device count is checked first
    assert r.returncode == 2 and "GPU(s) visible" in r.stderr and not r.stdout.strip()
