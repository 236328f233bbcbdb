cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r4f
python tools/exp/wide_check.py 2>&1 | grep -v amdgpu.ids | grep -E "FAIL|ALL" | tail -9
echo "== main (optimistic exp)"; python tools/dispatch_ab.py wide 2>&1 | grep -v amdgpu.ids | cut -c36-200 | tee gpurun_out/r4f/wide_ab.txt
echo "== wideA (previous)"; OEH_LIB=$GRAFT_REPO_ROOT/outeffhop_amd/lib/wideA/liboeh_hip.so python tools/dispatch_ab.py wide 2>&1 | grep -v amdgpu.ids | cut -c36-200 | tee gpurun_out/r4f/wide_abA.txt
