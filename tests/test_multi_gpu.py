"""Two ranks over RCCL ("nccl") on a box with >= 2 GPUs: the bench's own launcher, its shard-parity check (all_gather of the
ranks' outputs, each compared bit for bit with rank 0's run of that shard) and `rccl_ranks_seen`.  Skipped on the 1-GPU
boxes the round's `-m gpu` run uses; the same plumbing runs on gloo in tests/test_dist_cpu.py."""
import json
import os
import subprocess
import sys

import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("workload", ["opt_softmax1", "bert_gated", "opt_int8"])
def test_bench_two_ranks_over_rccl(workload):
    if torch.cuda.device_count() < 2:
        pytest.skip("needs 2 GPUs")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--workload", workload],
                       env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
    rec = json.loads(line)
    assert rec["n_gpus"] == 2 and rec["rccl_ranks_seen"] == 2 and rec["scaling"] == "weak"
    assert rec["shard_check"]["ranks"] == 2 and rec["shard_check"]["bitwise_equal"], rec["shard_check"]


def test_bench_two_ranks_plumbing_on_one_gpu():
    """The N > 1 code path of bench.py on a 1-GPU box: its own launcher, two ranks that both use cuda:0 with gloo collectives
    (OEH_BENCH_SHARE_ONE_GPU=1; a plumbing test, never a measurement): barriers, max-over-ranks, `rccl_ranks_seen`, and the
    shard check - rank 0 regenerating rank 1's inputs and reproducing its output bit for bit."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    env["OEH_BENCH_SHARE_ONE_GPU"] = "1"
    for workload in ("opt_softmax1", "bert_gated", "opt_int8", "opt_int8_i8"):   # (bert_gated: B = 32 per rank, BASELINE config 5's shard)
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--workload", workload],
                           env=env, capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stderr[-3000:]
        rec = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
        assert rec["n_gpus"] == 2 and rec["rccl_ranks_seen"] == 2 and "PLUMBING TEST" in rec["config"]["parallelism"]
        assert rec["shard_check"]["ranks"] == 2 and rec["shard_check"]["bitwise_equal"], rec["shard_check"]
        assert "cpu_baseline" not in rec
