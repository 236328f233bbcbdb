"""Child program of tests/test_dist_cpu.py::test_launch_ranks_starts_one_process_per_rank: started by
outeffhop_amd.dist.launch_ranks as N ranks of torch.distributed.run (gloo on CPU), it exercises the helpers bench.py uses
around its timed region and writes what rank 0 saw to the file named in argv[1]."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch
import torch.distributed as dist

from outeffhop_amd.dist import gather_equal, max_over_ranks, ranks_seen


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo")
    try:
        seen = ranks_seen()
        parts = gather_equal(torch.full((3, 2), float(rank)))
        wall = max_over_ranks(10.0 + rank)
        if rank == 0:
            with open(sys.argv[1], "w") as f:
                json.dump({"world": world, "seen": seen, "wall": wall, "parts": [float(p[0, 0]) for p in parts],
                           "master": os.environ.get("MASTER_ADDR")}, f)
        dist.barrier()
    finally:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
