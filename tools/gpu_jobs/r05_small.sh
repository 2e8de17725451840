#!/bin/bash
# strong-scaling shards of C2 on one GPU: the weight-stationary forward pair against the one-kernel forward, eager and captured
summ='import sys,json; l=json.loads(sys.stdin.read()); s=l["stage_ms"]; print(sys.argv[1], round(l["ms_per_step"],4), round(l["ms_per_step_median"],4), " ".join("%s=%.0f"%(k.split(".")[1],x*1e3) for k,x in s.items() if x>0.008))'
for rows in 2048 4096 8192 16384 32768; do
  for g in "" "--graph"; do
    timeout -k 10 200 python bench.py --rows $rows --steps 100 --warmup 20 --no-cpu-baseline $g 2>/dev/null | python -c "$summ" "rows=$rows default $g"
    AECF_DEBUG=fused_fwd timeout -k 10 200 python bench.py --rows $rows --steps 100 --warmup 20 --no-cpu-baseline $g 2>/dev/null | python -c "$summ" "rows=$rows fused_fwd $g"
  done
done
