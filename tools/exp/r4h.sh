cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r4h
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r4h/bench.json 2> gpurun_out/r4h/bench.err; tail -3 gpurun_out/r4h/bench.err
python -c "
import json; r=json.loads(open('gpurun_out/r4h/bench.json').read().strip().splitlines()[-1])
print({k:r[k] for k in ('metric','value','ms_per_step','n_gpus')}); print(r['roofline']); print(r['cpu_baseline'].get('value'), r['cpu_baseline'].get('cores')); print(r['cpu_baseline'].get('fp16_vs_reference'))"
python __graft_entry__.py smoke 2>&1 | tail -1
