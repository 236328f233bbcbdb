cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
L=$GRAFT_REPO_ROOT/outeffhop_amd/lib/pf2/liboeh_hip.so
python tools/microbench.py "B=16,H=12,S=512,D=64,causal=1,i8=1,dtype=f32,ab=$L" "B=16,H=12,S=512,D=64,causal=1,i8=1,dtype=f16,ab=$L" "B=32,H=12,S=128,D=64,i8=1,dtype=f32,ab=$L" "B=32,H=12,S=256,D=64,causal=1,i8=1,dtype=f32,ab=$L" "B=16,H=12,S=512,D=64,causal=0,i8=1,dtype=f32,ab=$L" 2>&1 | grep -v amdgpu.ids
