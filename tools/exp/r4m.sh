cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r4m
timeout 1200 python tools/stress_determinism.py 200 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r4m/stress.txt | tail -45
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -4
