#!/bin/bash
# round 5: the pipelined K loop (oeh_gemm_kernel<A_F32,4,9,LOOP>) against the round-4 loop (OPT shape, OEH_GEMM_LOOP0=1) and the 64 x 192 tile (BERT shape, OEH_GEMM_TILE=2)
set -u
ROOT=${GRAFT_REPO_ROOT:-$PWD}
cd $ROOT
mkdir -p gpurun_out/r05_loop2
pb() { python tools/proj_bench.py --no-baseline $@ 2>/dev/null | python -c "import sys,json; print('   '.join(str(round(json.loads(l)['kernel_us'],2)) for l in sys.stdin if l.startswith('{')))"; }
{
echo "# kernel_us: opt_qkv | opt_qkv_novalues || bert_qkv"
for rep in 1 2 3; do
  echo -n "round-4 loop (OEH_GEMM_LOOP0=1) || 64 x 192 tile (OEH_GEMM_TILE=2) : "; OEH_DEBUG_HOOKS=1 OEH_GEMM_LOOP0=1 pb opt_qkv opt_qkv_novalues; echo -n "                                                                     || "; OEH_DEBUG_HOOKS=1 OEH_GEMM_TILE=2 pb bert_qkv
  echo -n "production (pipelined body: LOOP == 2 || LOOP == 1)                : "; pb opt_qkv opt_qkv_novalues bert_qkv
done
python -m pytest tests/test_proj_gpu.py tests/test_modules_gpu.py -m gpu -q 2>&1 | tail -2
python tools/module_bench.py int8 2>&1 | grep "Quantized"
} 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05_loop2/out.txt
