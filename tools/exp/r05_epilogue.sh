#!/bin/bash
# round 5: the projection GEMM's epilogue (segment descriptors in scalar registers, write-through index pieces, one form dispatch per column tile) and the
# one-workgroup-per-CU loop for <= 256 tiles, against the library of the commit before (outeffhop_amd/lib/prevgemm); graph replay, alternating rounds
set -u
ROOT=${GRAFT_REPO_ROOT:-$PWD}
cd $ROOT
mkdir -p gpurun_out/r05_epilogue
pb() { python tools/proj_bench.py --no-baseline opt_qkv opt_qkv_novalues opt_out_proj bert_qkv 2>/dev/null | python -c "import sys,json; print('   '.join(str(round(json.loads(l)['kernel_us'],2)) for l in sys.stdin if l.startswith('{')))"; }
{
echo "# kernel_us: opt_qkv | opt_qkv_novalues | opt_out_proj | bert_qkv"
for rep in 1 2 3; do
  echo -n "this commit   : "; pb
  echo -n "commit before : "; OEH_LIB=$ROOT/outeffhop_amd/lib/prevgemm/liboeh_hip.so pb
done
python -m pytest tests/test_proj_gpu.py tests/test_modules_gpu.py -m gpu -q 2>&1 | tail -2
python tools/module_bench.py int8 2>&1 | grep "Quantized"
} 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05_epilogue/out.txt
