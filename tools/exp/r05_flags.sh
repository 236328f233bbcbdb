#!/bin/bash
# round 5: code-generation variants of the attention units ("other") against production, same process: $1 = alt library name
set -u
ROOT=${GRAFT_REPO_ROOT:-$PWD}
cd $ROOT
for n in "$@"; do
L=$ROOT/outeffhop_amd/lib/$n/liboeh_hip.so
echo "##### $n"
OEH_LIB=$L python -m pytest tests/test_attn_gpu.py -m gpu -q -x 2>&1 | tail -1
python tools/microbench.py "B=16,H=12,S=512,D=64,causal=1,iters=300,ab=$L" "B=16,H=12,S=512,D=64,causal=1,iters=300,clip=1,ab=$L" "B=16,H=12,S=512,D=64,causal=1,iters=300,int8=1,ab=$L" \
  "B=16,H=12,S=512,D=64,causal=1,iters=300,i8=1,dtype=f16,ab=$L" "B=32,H=12,S=128,D=64,pad=1,iters=400,ab=$L" "B=32,H=12,S=128,D=64,pad=1,gmlp=16,iters=400,ab=$L" \
  "B=32,H=12,S=128,D=64,pad=1,i8=1,dtype=f32,iters=400,ab=$L" "B=16,H=12,S=512,D=64,causal=1,dtype=f32,iters=200,ab=$L" "B=16,H=12,S=512,D=64,iters=300,ab=$L" 2>&1 | grep built/other | awk '{print $1, $(NF-8), $(NF-4), $NF}'
done
