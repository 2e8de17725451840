#!/bin/bash
# tools/gpu_jobs/ab_libs.sh ROUNDS CFG name1 name2 ...: bench CFG (EXTRA = more bench.py flags) with each aecf_amd/lib/var/<name> library ("main" = the in-tree build),
# interleaved ROUNDS times on this box
rounds=$1; cfg=$2; shift 2
summ='import sys,json; l=json.loads(sys.stdin.read()); s=l["stage_ms"]; print(sys.argv[1], round(l["ms_per_step"],4), round(l["ms_per_step_median"],4), " ".join("%s=%.0f"%(k.split(".")[1],x*1e3) for k,x in s.items() if x>0.008))'
for i in $(seq $rounds); do
  for n in "$@"; do
    if [ "$n" = main ]; then lib=""; else lib=$PWD/aecf_amd/lib/var/$n/libaecf_hip.so; fi
    AECF_LIB_PATH=$lib timeout -k 10 300 python bench.py --config $cfg --steps 50 --warmup 10 --no-cpu-baseline $EXTRA 2>/dev/null | python -c "$summ" $n
  done
done
