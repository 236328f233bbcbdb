cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r4j
L=$GRAFT_REPO_ROOT/outeffhop_amd/lib/r03/liboeh_hip.so
python tools/microbench.py "B=16,H=12,S=512,D=64,causal=1,i8=1,dtype=f32,ab=$L" "B=16,H=12,S=512,D=64,causal=1,i8=1,dtype=f16,ab=$L" "B=32,H=12,S=128,D=64,i8=1,dtype=f32,ab=$L" "B=32,H=12,S=256,D=64,causal=1,i8=1,dtype=f32,ab=$L" "B=16,H=12,S=512,D=64,causal=0,i8=1,dtype=f32,ab=$L" 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r4j/i8_ab.txt
timeout 900 python -m pytest tests/test_attn_gpu.py tests/test_modules_gpu.py -m gpu -q -x -k "int8_storage or i8 or quantised or int8_modules or int8" 2>&1 | tail -4
