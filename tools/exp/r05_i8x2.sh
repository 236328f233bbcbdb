#!/bin/bash
# round 5: the 128-row form of the INT8-storage core (two blocks per wave, resident tiles) against the 64-row form (off=8192 forces it), one process per line pair
set -u
ROOT=${GRAFT_REPO_ROOT:-$PWD}
cd $ROOT
mkdir -p gpurun_out/r05_i8x2
{
python -m pytest tests/test_attn_gpu.py -m gpu -q -x -k "int8_storage or i8" 2>&1 | tail -5
for rep in 1 2; do
python tools/microbench.py "B=16,H=12,S=512,D=64,causal=1,i8=1,dtype=f32,iters=300,reps=5" "B=16,H=12,S=512,D=64,causal=1,i8=1,dtype=f32,iters=300,reps=5,off=8192" \
  "B=16,H=12,S=512,D=64,causal=1,i8=1,dtype=f16,iters=300,reps=5" "B=16,H=12,S=512,D=64,causal=1,i8=1,dtype=f16,iters=300,reps=5,off=8192" \
  "B=16,H=12,S=512,D=64,i8=1,dtype=f32,iters=300,reps=5" "B=16,H=12,S=512,D=64,i8=1,dtype=f32,iters=300,reps=5,off=8192" \
  "B=24,H=12,S=384,D=64,causal=1,i8=1,dtype=f32,iters=300,reps=5" "B=24,H=12,S=384,D=64,causal=1,i8=1,dtype=f32,iters=300,reps=5,off=8192" \
  "B=4,H=12,S=512,D=64,causal=1,i8=1,dtype=f32,iters=300,reps=5" "B=4,H=12,S=512,D=64,causal=1,i8=1,dtype=f32,iters=300,reps=5,off=8192"
done
} 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05_i8x2/ab.txt
