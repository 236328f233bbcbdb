#!/bin/bash
# round 6, one gpurun call: the one-pass kernel's instruction-count cuts (merged max tree, ones operand kept across the loop, partly masked tiles
# specialised) - tests on the built library, then same-process A/Bs against the round-5 library (lib/r5head) and between the round-6 builds
set -u
ROOT=${GRAFT_REPO_ROOT:-$PWD}
L=$ROOT/outeffhop_amd/lib
O=$ROOT/gpurun_out/r06_ab1
mkdir -p $O
cd $ROOT
SPECS=("B=16,H=12,S=512,D=64,causal=1,iters=300" "B=16,H=12,S=512,D=64,causal=1,iters=300,dtype=bf16" "B=16,H=12,S=512,D=64,iters=300" "B=8,H=12,S=1024,D=64,causal=1,iters=200"
  "B=32,H=12,S=256,D=64,causal=1,iters=300" "B=16,H=12,S=512,D=64,causal=1,gmlp=16,iters=300" "B=16,H=12,S=512,D=32,causal=1,iters=300" "B=8,H=12,S=704,D=64,pad=1,iters=200"
  "B=64,H=6,S=197,D=64,iters=300" "B=16,H=12,S=512,D=64,causal=1,iters=300")
ab() {  # ab <main lib dir or ''> <other lib dir> <label>
  echo "# $3 (built/other < 1: the first one wins)"
  local args=()
  for s in "${SPECS[@]}"; do args+=("$s,ab=$L/$2/liboeh_hip.so"); done
  if [ -n "$1" ]; then OEH_LIB=$L/$1/liboeh_hip.so python tools/microbench.py "${args[@]}"; else python tools/microbench.py "${args[@]}"; fi
}
{
python -m pytest tests/test_attn_gpu.py -m gpu -q -x 2>&1 | tail -3
ab "" r5head "built (max tree + ones kept + partial tiles for block 1 alone) vs round-5 HEAD"
ab r6ab r5head "max tree + ones kept only (no partial tiles) vs round-5 HEAD"
ab r6m3 r5head "all partial-tile bodies (spills at MQ=2) vs round-5 HEAD"
} 2>&1 | grep -v amdgpu.ids | tee $O/out.txt
