#!/bin/bash
# round 5, one gpurun call: same-process A/Bs of the experiment builds (make -C outeffhop_amd/csrc alt NAME=... DEFS=...) against the production library
set -u
ROOT=${GRAFT_REPO_ROOT:-$PWD}
L=$ROOT/outeffhop_amd/lib
O=$ROOT/gpurun_out/r05_ab
mkdir -p $O
cd $ROOT
H="B=16,H=12,S=512,D=64,causal=1,iters=300"
{
echo "# OEH_EARLY_O: block 0's epilogue before the workgroup's last tile (built = production, other = experiment; other/built < 1: the experiment wins)"
python tools/microbench.py "$H,ab=$L/early/liboeh_hip.so" "B=16,H=12,S=512,D=64,iters=300,ab=$L/early/liboeh_hip.so" "B=8,H=12,S=1024,D=64,causal=1,iters=200,ab=$L/early/liboeh_hip.so" "B=4,H=12,S=2048,D=64,causal=1,iters=100,ab=$L/early/liboeh_hip.so" "B=32,H=12,S=256,D=64,causal=1,iters=300,ab=$L/early/liboeh_hip.so" "$H,dtype=bf16,ab=$L/early/liboeh_hip.so"
} > $O/early.txt 2>&1
{
echo "# OEH_F32_KDMA: fp32 K tiles by LDS-DMA, (hi, lo) split at fragment-read time (built = production: register-staged K)"
python tools/microbench.py "$H,dtype=f32,ab=$L/kdma/liboeh_hip.so" "B=16,H=12,S=512,D=64,dtype=f32,iters=200,ab=$L/kdma/liboeh_hip.so" "B=8,H=12,S=1024,D=64,causal=1,dtype=f32,iters=150,ab=$L/kdma/liboeh_hip.so" "B=64,H=12,S=128,D=64,dtype=f32,iters=200,ab=$L/kdma/liboeh_hip.so"
} > $O/kdma.txt 2>&1
for n in 16 48; do
{
echo "# OEH_STAGGER=$n: first-round workgroups of a CU start 64 * $n cycles apart (full-row and INT8-storage kernels)"
python tools/microbench.py "$H,clip=1,ab=$L/stag$n/liboeh_hip.so" "$H,int8=1,ab=$L/stag$n/liboeh_hip.so" "$H,i8=1,dtype=f32,ab=$L/stag$n/liboeh_hip.so" "$H,int8=1,dtype=f32,ab=$L/stag$n/liboeh_hip.so" "B=32,H=12,S=128,D=64,pad=1,iters=300,ab=$L/stag$n/liboeh_hip.so" "B=32,H=12,S=128,D=64,pad=1,int8=1,iters=300,ab=$L/stag$n/liboeh_hip.so"
} > $O/stag$n.txt 2>&1
done
# correctness of the two candidates that change results' path (same tests as the production library runs)
OEH_LIB=$L/early/liboeh_hip.so python -m pytest tests/test_attn_gpu.py -m gpu -q -x -k "full_size or one_pass or core_nomask or opt_order_causal or snake or dispatch_rule or bit_repro" 2>&1 | tail -4 > $O/early_tests.txt
OEH_LIB=$L/kdma/liboeh_hip.so python -m pytest tests/test_attn_gpu.py tests/test_modules_gpu.py -m gpu -q -x -k "fp32 or f32 or float32" 2>&1 | tail -4 > $O/kdma_tests.txt
cat $O/early.txt $O/kdma.txt $O/stag16.txt $O/stag48.txt $O/early_tests.txt $O/kdma_tests.txt
