#!/bin/bash
# round 6: two placed orders of the steady-state tile's FIRST half (OEH_PIPE_QK = 1: scale step of sub-tile s between the score MFMAs of s + 1, one LDS-DMA
# piece behind each sub-tile; = 2: the same without moving the requests) against the built library, same process; tests on both
set -u
ROOT=${GRAFT_REPO_ROOT:-$PWD}
L=$ROOT/outeffhop_amd/lib
O=$ROOT/gpurun_out/r06_ab2
mkdir -p $O
cd $ROOT
SPECS=("B=16,H=12,S=512,D=64,causal=1,iters=300" "B=16,H=12,S=512,D=64,causal=1,iters=300,dtype=bf16" "B=16,H=12,S=512,D=64,iters=300" "B=8,H=12,S=1024,D=64,causal=1,iters=200"
  "B=4,H=12,S=2048,D=64,causal=1,iters=100" "B=32,H=12,S=256,D=64,causal=1,iters=300" "B=16,H=12,S=512,D=64,causal=1,gmlp=16,iters=300" "B=16,H=12,S=512,D=64,causal=1,iters=300")
ab() {
  echo "# $3 (built/other < 1: the first one wins)"
  local args=()
  for s in "${SPECS[@]}"; do args+=("$s,ab=$L/$2/liboeh_hip.so"); done
  if [ -n "$1" ]; then OEH_LIB=$L/$1/liboeh_hip.so python tools/microbench.py "${args[@]}"; else python tools/microbench.py "${args[@]}"; fi
}
{
for v in r6qk1 r6qk2; do
  echo "# tests on $v"; OEH_LIB=$L/$v/liboeh_hip.so python -m pytest tests/test_attn_gpu.py -m gpu -q -x -k "not outlier" 2>&1 | tail -2
done
ab r6qk1 "" "placed first half, LDS-DMA pieces behind the sub-tiles (OEH_PIPE_QK=1) vs the built library"
ab r6qk2 "" "placed first half, requests at the top of the tile (OEH_PIPE_QK=2) vs the built library"
ab "" r5head "the built library vs round-5 HEAD"
} 2>&1 | grep -v amdgpu.ids | tee $O/out.txt
