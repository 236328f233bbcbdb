#!/bin/bash
# The headline launch's knock-out ladder ("commit profiles/r05_headline_floor.txt with a knock-out ladder on the final
# kernel, each as kernel_us from --kernel-trace").  Builds: make -C outeffhop_amd/csrc knockout KO=1..4 (oeh_attn_flash.inl: OEH_KO) ->
# outeffhop_amd/lib/ko<n>/liboeh_hip.so; the production library is rung 5.  Per rung: bench.py (12 layers x 20 steps + ramp) under
# rocprofv3 --kernel-trace --stats, the mean / min duration of oeh_attn_flash_kernel<64,f16,MQ=2>; two passes over the rungs, interleaved.
#   gpurun: bash tools/floor_ladder.sh > gpurun_out/r05_floor/ladder.txt
set -u
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/r05_floor
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for pass in 1 2; do
  for rung in ko1 ko2 ko3 ko4 prod; do
    if [ $rung = prod ]; then unset OEH_LIB; else export OEH_LIB=$ROOT/outeffhop_amd/lib/$rung/liboeh_hip.so; fi
    d=$OUT/$rung.$pass
    rocprofv3 --kernel-trace --stats --output-format csv -d "$d" -- python3 "$ROOT/bench.py" --steps 20 --warmup 5 --no-cpu-baseline > "$d.json" 2> "$d.log"
    f=$(find "$d" -name '*kernel_stats.csv' | head -1)
    python3 - "$rung" "$pass" "$f" "$d.json" <<'PY'
import csv, json, sys
rung, ps, f, jf = sys.argv[1:5]
row = [r for r in csv.reader(open(f)) if "oeh_attn_flash_kernel" in r[0]][0]
try:
    rec = json.loads([ln for ln in open(jf) if ln.startswith("{")][-1])
    ev = rec["roofline"]["kernel_us"]
except Exception:
    ev = float("nan")
print(f"{rung:5s} pass {ps}: trace mean {float(row[3]) / 1e3:7.2f} us  min {float(row[5]) / 1e3:6.2f}  calls {row[1]:>6s}   HIP-event mean in bench.py {ev:6.2f} us", flush=True)
PY
    rm -rf "$d"
  done
done
