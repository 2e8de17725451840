#!/bin/bash
# same-box A/B of the C2 step: the round's baseline build (build/base_repo, a checkout of the previous round's HEAD) against the
# working tree; usage: tools/gpu_jobs/ab_base.sh [pairs] [config]
pairs=${1:-2}; cfg=${2:-c2}
summ='import sys,json; l=json.loads(sys.stdin.read()); s=l["stage_ms"]; print(sys.argv[1], round(l["ms_per_step"],4), round(l["ms_per_step_median"],4), " ".join("%s=%.0f"%(k.split(".")[1],x*1e3) for k,x in s.items() if x>0.003))'
for i in $(seq $pairs); do
  (cd build/base_repo && timeout -k 10 300 python bench.py --config $cfg --steps 50 --warmup 10 --no-cpu-baseline 2>/dev/null | python -c "$summ" base)
  timeout -k 10 300 python bench.py --config $cfg --steps 50 --warmup 10 --no-cpu-baseline 2>/dev/null | python -c "$summ" new
done
