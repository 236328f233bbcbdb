#!/bin/bash
# round 5: the 64 x 192 projection tile with a ring of THREE LDS-DMA slots (two steps for a tile to land) against the ring of two (OEH_GEMM_RING=2)
set -u
ROOT=${GRAFT_REPO_ROOT:-$PWD}
cd $ROOT
mkdir -p gpurun_out/r05_ring
pb() { python tools/proj_bench.py --no-baseline opt_out_proj bert_qkv | python -c "import sys,json; [print(json.loads(l)['config']['workload'][:44], round(json.loads(l)['kernel_us'],2)) for l in sys.stdin if l.startswith('{')]"; }
{
python -m pytest tests/test_proj_gpu.py -m gpu -q 2>&1 | tail -2
for rep in 1 2 3; do
  echo "## ring of three (production)"; pb
  echo "## ring of two"; OEH_DEBUG_HOOKS=1 OEH_GEMM_RING=2 pb
done
python -m pytest tests/test_modules_gpu.py -m gpu -q 2>&1 | tail -2
} 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05_ring/out.txt
