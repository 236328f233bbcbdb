cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r4f
python tools/exp/wide_check.py 2>&1 | grep -v amdgpu.ids | tail -3
python tools/dispatch_ab.py wide 2>&1 | grep -v amdgpu.ids | cut -c36-200 | tee gpurun_out/r4f/wide_ab.txt
