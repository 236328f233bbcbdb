#!/bin/bash
# usage: ab_bench.sh "<workload> ..." <other liboeh_hip.so> ...   - bench.py per workload, the built library and other builds
# alternating (separate processes inside one gpurun call; 2 rounds); prints the per-launch time of every run
WL=$1; shift
for w in $WL; do
  for r in 1 2; do
    for lib in built "$@"; do
      if [ "$lib" = built ]; then unset OEH_LIB; else export OEH_LIB=$lib; fi
      python bench.py --workload $w --steps 100 --warmup 20 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    l=l.strip()
    if l.startswith('{'):
        d=json.loads(l); print('$w', '$lib'.split('/')[-2] if '/' in '$lib' else '$lib', 'kernel_us', round(d['ms_per_step']*1e3/12,2), d['config'].get('variant'))"
    done
  done
done
