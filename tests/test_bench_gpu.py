"""GPU: bench.py end to end on small shapes -- the contract line's keys, BASELINE configs[1] (`--encoder torch`, which
crashed after timing in round 1), and `--gpus 2` launching its own two ranks (both on the box's single GPU, gloo for the
exchange: RCCL refuses two ranks on one device; the control flow, sharding and gather are the same)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SMALL = ["--steps", "2", "--warmup", "1", "--triplets", "2", "--seconds", "2", "--no-cpu-baseline"]


def _bench(args, **env):
    e = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    e.update(env)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=e, capture_output=True, text=True,
                       timeout=900)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    return json.loads(lines[0])


def test_contract_line_hip_encoder():
    out = _bench(SMALL + ["--verify-b1"])
    assert out["n_gpus"] == 1 and out["world"] == 1 and out["unit"] == "triplets/s" and out["value"] > 0
    rf = out["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "traffic_source", "stage_a_hbm_frac", "stage_b_tflops",
              "stage_b_tflops_algorithmic", "fused_hbm_frac"):
        assert k in rf, k
    assert rf["bound"] == "mfma" and 0 < rf["frac"] < 1 and 0 < rf["stage_a_hbm_frac"] < 1
    assert rf["stage_b_tflops"] <= rf["stage_b_tflops_algorithmic"] < 157.3       # executed <= reference's flop count
    assert "alt" in out and isinstance(out["alt"], list)
    # the timed workload is checked after timing: every clip bit-equal to its B = 1 run (no oracle here: --no-cpu-baseline)
    v = out["config"]["verified"]
    assert v["ok"] and v["batch_independence"]["bit_equal"] == v["clips"] == out["config"]["clips_per_gpu"]
    assert v["batch_independence"]["re_runs"] == ["reversed", "rotated by 29", "every clip alone (B = 1)"]
    assert out["ms_per_step_median"] > 0
    for row in out["alt"]:
        assert "error" not in row, row
        assert row["verified"]["ok"] and row["embedding_error_vs_fp32_kernels"]["normwise"] < (1e-5 if "f16x3" in row["mode"] else 2e-3)
        assert 0 < row["roofline"]["conv1_frac"] < 1


def test_contract_line_with_cpu_baseline_checks_against_the_oracle():
    out = _bench(["--steps", "2", "--warmup", "1", "--triplets", "2", "--seconds", "2"])
    v = out["config"]["verified"]
    assert v["ok"] and v["oracle"]["beyond_tol"] == 0 and v["oracle"]["max_rel"] <= 1e-4 and len(v["oracle"]["clips"]) == 3
    assert out["cpu_baseline"]["kind"] == "port" and out["cpu_baseline"]["value"] > 0


def test_configs1_torch_encoder_line():
    out = _bench(SMALL + ["--encoder", "torch"])
    assert "configs[1]" in out["config"]["workload"] and out["config"]["encoder_backend"] == "torch"
    assert out["roofline"]["frac"] > 0 and out["roofline"]["traffic"] is None


def test_gpus_2_launches_two_ranks_itself():
    out = _bench(["--gpus", "2"] + SMALL, MST_BENCH_ONE_GPU="1", MST_BENCH_BACKEND="gloo")
    assert out["n_gpus"] == 2 and out["world"] == 2 and out["rccl_ranks_seen"] == 2
    assert out["config"]["parallelism"] == "clip-sharded x2" and "cpu_baseline" not in out
    mg = out["multi_gpu"]
    assert len(mg["per_rank_ms_per_step_wall"]) == 2 and mg["all_gather_ms"]["max"] >= 0 and 0 <= mg["all_gather_share_of_step"] < 1
    assert out["config"]["verified"]["ok"] and out["config"]["verified"]["ok_all_ranks"]


@pytest.mark.parametrize("precision", ["f16x3", "amp"])
def test_train_step_line_by_precision(precision):
    """`bench.py --train --train-precision ...` (extra measurement, not the contract line): the hand-written trunk in split
    precision, and the reference's --use_amp step (autocast + GradScaler around the float16-operand trunk)."""
    out = _bench(["--train", "--train-precision", precision, "--steps", "2", "--warmup", "1", "--triplets", "3", "--seconds", "2"])
    assert out["config"]["train_precision"] == precision and out["config"]["train_backend"] == "hip"
    assert out["value"] > 0 and out["config"]["loss"] == out["config"]["loss"] and "NOT THE CONTRACT LINE" in out["config"]["workload"]


def test_gpus_2_training_step_with_overlapped_gradient_reduction():
    """`bench.py --train --gpus 2`: two ranks (both on the box's GPU, gloo), the hand-written trunk launching its conv2-side and
    conv1-side gradient buckets itself during the backward (mst_amd/dist.GradientReducer), the pooling head / FiLM MLP buckets from
    autograd hooks; with cross-rank BatchNorm statistics on top (`--sync-bn`)."""
    for extra in ([], ["--sync-bn"]):
        out = _bench(["--gpus", "2", "--train", "--steps", "2", "--warmup", "1", "--triplets", "2", "--seconds", "2"] + extra,
                     MST_BENCH_ONE_GPU="1", MST_BENCH_BACKEND="gloo")
        assert out["n_gpus"] == 2 and out["world"] == 2 and out["value"] > 0
        assert out["config"]["sync_bn"] == bool(extra) and out["config"]["loss"] == out["config"]["loss"]
