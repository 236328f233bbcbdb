"""N > 1 plumbing with world_size-2 gloo on CPU: contiguous batch sharding, max-over-ranks timing, parity gather.
The per-shard computation here is a stand-in row-independent function - the GPU kernel itself is covered by
tests/test_attn_gpu.py::test_bit_reproducible_and_batch_shard_invariant (a shard's rows are bit-identical)."""
import os

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from outeffhop_amd.dist import gather_batch, max_over_ranks, run_sharded, shard_bounds


def test_shard_bounds_cover_batch_exactly():
    for B in (1, 7, 16, 256, 257):
        for W in (1, 2, 3, 8):
            spans = [shard_bounds(B, W, r) for r in range(W)]
            assert spans[0][0] == 0 and spans[-1][1] == B
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1
    assert [shard_bounds(256, 8, r) for r in (0, 7)] == [(0, 32), (224, 256)]  # BASELINE config 5: 8 x 32
    with pytest.raises(ValueError):
        shard_bounds(4, 2, 2)


def _worker(rank, world, port, B):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        g = torch.Generator().manual_seed(0)
        q, k = torch.randn(B, 3, 5, generator=g), torch.randn(B, 3, 5, generator=g)
        fn = lambda a, b: torch.softmax(a @ b.transpose(1, 2), dim=-1)  # noqa: E731  row-independent stand-in
        full = fn(q, k)
        local = run_sharded(fn, [q, k])
        lo, hi = shard_bounds(B, world, rank)
        assert torch.equal(local, full[lo:hi])  # no cross-sample dependence: a shard equals the full batch's rows
        assert torch.equal(run_sharded(fn, [q, k], gather=True), full)
        assert torch.equal(gather_batch(local, B), full)
        assert max_over_ranks(1.0 + rank) == float(world)
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("B", [8, 5])
def test_two_rank_gloo(B):
    port = 29500 + (os.getpid() % 1000) + B
    mp.spawn(_worker, args=(2, port, B), nprocs=2, join=True)


def test_launch_ranks_starts_one_process_per_rank(tmp_path):
    """`python bench.py --gpus N` starts its own N ranks through outeffhop_amd.dist.launch_ranks (it used
    to fall through to a 1-GPU measurement).  Here: 2 ranks on CPU/gloo, rendezvous on 127.0.0.1."""
    import json
    import sys

    from outeffhop_amd.dist import launch_ranks

    out = tmp_path / "seen.json"
    child = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_dist_child.py")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    rc = launch_ranks(child, 2, [str(out)], env=env, timeout=300)
    assert rc == 0
    got = json.loads(out.read_text())
    assert got == {"world": 2, "seen": 2, "wall": 11.0, "parts": [0.0, 1.0], "master": "127.0.0.1"}


def test_bench_refuses_an_n_gpu_line_it_cannot_measure():
    """No launcher, fewer GPUs than asked: exit code 2 and no JSON line (never `n_gpus: 1` for `--gpus 8`); a launcher
    whose WORLD_SIZE disagrees with --gpus is refused too."""
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if torch.cuda.device_count() >= 8:
        pytest.skip("8 GPUs visible: the request would be served")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "8"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 2 and "refusing" in r.stderr and "n_gpus" not in r.stdout
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "4"], env={**env, "WORLD_SIZE": "2", "RANK": "0"},
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 2 and "WORLD_SIZE=2" in r.stderr and "n_gpus" not in r.stdout


def test_bench_eight_rank_protocol_on_cpu_stub():
    """`python bench.py --gpus 8` has never met 8 GPUs.  Its rank protocol - own launcher (one process per rank, started
    before anything touches a GPU), the WORLD_SIZE guard, barriers, max-over-ranks, ranks_seen, ONE JSON line from rank 0 - runs here with 8
    CPU ranks over gloo (OEH_BENCH_STUB_CPU=1: a no-op step, `value` null, the line says it is a stub), both through bench.py's own launcher
    and as the driver starts it (`python -m torch.distributed.run --nproc-per-node 8 ... bench.py --gpus 8`)."""
    import json
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    env.update(OEH_BENCH_STUB_CPU="1", OMP_NUM_THREADS="1")
    port = 29700 + (os.getpid() % 200)
    cmds = [[sys.executable, os.path.join(root, "bench.py"), "--gpus", "8", "--steps", "5", "--warmup", "2", "--no-check"],
            [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr", "127.0.0.1", "--master-port", str(port),
             os.path.join(root, "bench.py"), "--gpus", "8", "--steps", "5", "--warmup", "2", "--no-check"]]
    for cmd in cmds:
        r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-3000:]
        lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
        assert len(lines) == 1, r.stdout[-2000:]
        rec = json.loads(lines[0])
        assert rec["n_gpus"] == 8 and rec["rccl_ranks_seen"] == 8 and rec["steps"] == 5 and rec["warmup"] == 2 and rec["scaling"] == "weak"
        assert rec["stub"] is True and rec["value"] is None
    # a launcher whose WORLD_SIZE disagrees with --gpus is refused in stub mode too
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "8"], env={**env, "WORLD_SIZE": "2", "RANK": "0"}, capture_output=True, text=True, timeout=300)
    assert r.returncode == 2 and "WORLD_SIZE=2" in r.stderr and "n_gpus" not in r.stdout
