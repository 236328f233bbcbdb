cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r4g
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/r4g/tests.log 2>&1; echo "tests rc=$?" >> gpurun_out/r4g/tests.log
grep -E "passed|failed|FAILED|rc=|Warning" gpurun_out/r4g/tests.log | tail -20
L=$GRAFT_REPO_ROOT/outeffhop_amd/lib/r03/liboeh_hip.so
python tools/microbench.py "B=16,H=12,S=512,D=64,causal=1,ab=$L" "B=8,H=12,S=1024,D=64,causal=1,clip=1,ab=$L" "B=8,H=12,S=1024,D=64,pad=1,clip=1,ab=$L" "B=8,H=12,S=1024,D=64,pad=1,base=0,ab=$L" 2>&1 | grep -v amdgpu.ids
