#!/bin/bash
# round 6: wave issue priority by q tile (OEH_PRIO_MODE 1 / 2 / 3: oeh_attn_flash.inl) against the built library, same process
set -u
ROOT=${GRAFT_REPO_ROOT:-$PWD}
L=$ROOT/outeffhop_amd/lib
O=$ROOT/gpurun_out/r06_ab3
mkdir -p $O
cd $ROOT
SPECS=("B=16,H=12,S=512,D=64,causal=1,iters=300" "B=16,H=12,S=512,D=64,causal=1,iters=300,dtype=bf16" "B=8,H=12,S=1024,D=64,causal=1,iters=200"
  "B=4,H=12,S=2048,D=64,causal=1,iters=100" "B=32,H=12,S=256,D=64,causal=1,iters=300" "B=16,H=12,S=512,D=64,causal=1,gmlp=16,iters=300" "B=16,H=12,S=512,D=64,causal=1,iters=300")
{
for v in ${VARIANTS:-r6prio1 r6prio2 r6prio3}; do
  echo "# $v (built = the variant, other = the production library; built/other < 1: the variant wins)"
  args=()
  for s in "${SPECS[@]}"; do args+=("$s,ab=$L/liboeh_hip.so"); done
  OEH_LIB=$L/$v/liboeh_hip.so python tools/microbench.py "${args[@]}"
done
} 2>&1 | grep -v amdgpu.ids | tee $O/out.txt
