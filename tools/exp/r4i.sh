cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r4i
L=$GRAFT_REPO_ROOT/outeffhop_amd/lib/r03/liboeh_hip.so
python tools/microbench.py "B=16,H=12,S=512,D=64,causal=1,ab=$L" "B=16,H=12,S=512,D=64,causal=0,ab=$L" "B=8,H=12,S=1024,D=64,causal=1,ab=$L" "B=4,H=12,S=2048,D=64,causal=1,ab=$L" "B=16,H=12,S=512,D=64,causal=1,dtype=bf16,ab=$L" "B=16,H=12,S=512,D=128,causal=1,ab=$L" 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r4i/prescale_ab.txt
python __graft_entry__.py smoke 2>&1 | tail -1
timeout 900 python -m pytest tests/test_attn_gpu.py -m gpu -q -x -k "opt_order or core_nomask or full_size_opt_cfg3 or geometries or reproducible or full_size_properties or fp32_output" 2>&1 | tail -5
