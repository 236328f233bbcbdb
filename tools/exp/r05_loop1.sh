#!/bin/bash
# round 5: BERT-base q/k/v projections (M = 4096) on the one-workgroup-per-CU loop (128 x 288 tiles, oeh_gemm_kernel<A_F32,4,9,1>) against the 64 x 192 tile
set -u
ROOT=${GRAFT_REPO_ROOT:-$PWD}
cd $ROOT
mkdir -p gpurun_out/r05_loop1
pb() { python tools/proj_bench.py --no-baseline bert_qkv | python -c "import sys,json; [print(json.loads(l)['config']['workload'][:44], round(json.loads(l)['kernel_us'],2)) for l in sys.stdin if l.startswith('{')]"; }
{
python -m pytest tests/test_proj_gpu.py -m gpu -q 2>&1 | tail -3
for rep in 1 2 3; do
  echo "## production (LOOP == 1, 128 x 288, one workgroup per CU)"; pb
  echo "## 64 x 192 tile (OEH_GEMM_TILE=2)"; OEH_DEBUG_HOOKS=1 OEH_GEMM_TILE=2 pb
done
python -m pytest tests/test_modules_gpu.py -m gpu -q 2>&1 | tail -2
python tools/module_bench.py int8 2>&1 | grep "QuantizedBert"
} 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05_loop1/out.txt
