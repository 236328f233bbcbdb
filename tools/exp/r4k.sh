cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r4k
L=$GRAFT_REPO_ROOT/outeffhop_amd/lib/r03/liboeh_hip.so
python tools/microbench.py "B=16,H=12,S=512,D=64,causal=1,clip=1,ab=$L" "B=16,H=12,S=512,D=64,causal=1,int8=1,ab=$L" "B=32,H=12,S=128,D=64,pad=1,ab=$L" "B=32,H=12,S=128,D=64,pad=1,int8=1,ab=$L" "B=32,H=12,S=256,D=64,causal=1,clip=1,ab=$L" "B=16,H=12,S=512,D=64,causal=0,clip=1,ab=$L" "B=16,H=12,S=448,D=64,causal=1,int8=1,ab=$L" "B=16,H=12,S=384,D=64,causal=1,clip=1,ab=$L" "B=32,H=12,S=128,D=64,pad=1,gmlp=16,ab=$L" "B=16,H=8,S=512,D=128,causal=1,ab=$L" 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r4k/fast_ab.txt
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -4
