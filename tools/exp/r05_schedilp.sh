#!/bin/bash
# round 5: the library built with -mllvm -amdgpu-sched-strategy=max-ilp ("other") against production, same process
set -u
ROOT=${GRAFT_REPO_ROOT:-$PWD}
cd $ROOT
L=$ROOT/outeffhop_amd/lib/ilp/liboeh_hip.so
mkdir -p gpurun_out/r05_schedilp
{
OEH_LIB=$L python -m pytest tests/test_attn_gpu.py tests/test_proj_gpu.py -m gpu -q 2>&1 | tail -2
python tools/microbench.py "B=16,H=12,S=512,D=64,causal=1,iters=300,ab=$L" "B=16,H=12,S=512,D=64,causal=1,iters=300,clip=1,ab=$L" "B=16,H=12,S=512,D=64,causal=1,iters=300,int8=1,ab=$L" \
  "B=16,H=12,S=512,D=64,causal=1,iters=300,i8=1,dtype=f16,ab=$L" "B=16,H=12,S=512,D=64,causal=1,iters=300,i8=1,dtype=f32,ab=$L" "B=32,H=12,S=128,D=64,pad=1,iters=400,ab=$L" \
  "B=32,H=12,S=128,D=64,pad=1,gmlp=16,iters=400,ab=$L" "B=32,H=12,S=128,D=64,pad=1,int8=1,iters=400,ab=$L" "B=32,H=12,S=128,D=64,pad=1,i8=1,dtype=f32,iters=400,ab=$L" \
  "B=16,H=12,S=512,D=64,causal=1,dtype=f32,iters=200,ab=$L" "B=16,H=12,S=512,D=64,causal=1,dtype=f32,int8=1,iters=200,ab=$L" "B=8,H=12,S=1024,D=64,causal=1,iters=200,ab=$L" \
  "B=8,H=16,S=512,D=128,causal=1,iters=200,ab=$L" "B=16,H=12,S=512,D=64,iters=300,ab=$L"
pb() { python tools/proj_bench.py --no-baseline opt_qkv opt_qkv_novalues opt_out_proj bert_qkv 2>/dev/null | python -c "import sys,json; print('   '.join(str(round(json.loads(l)['kernel_us'],2)) for l in sys.stdin if l.startswith('{')))"; }
for rep in 1 2; do echo -n "gemm production: "; pb; echo -n "gemm max-ilp   : "; OEH_LIB=$L pb; done
} 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05_schedilp/out.txt
