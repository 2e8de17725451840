#!/bin/bash
out=gpurun_out; mkdir -p $out
summ='import sys,json; l=json.loads(sys.stdin.read()); s=l["stage_ms"]; print(sys.argv[1], round(l["ms_per_step"],4), round(l["ms_per_step_median"],4), " ".join("%s=%.0f"%(k.split(".")[1],x*1e3) for k,x in s.items() if x>0.008))'
for n in main wstream8 wstream32 main; do
  if [ "$n" = main ]; then lib=""; else lib=$PWD/build/var/$n/libaecf_hip.so; fi
  AECF_DEBUG=dsu_var=3 AECF_LIB_PATH=$lib timeout -k 10 300 python bench.py --steps 50 --warmup 10 --no-cpu-baseline 2>/dev/null | python -c "$summ" "slab+$n"
done | tee $out/o_fusedbwd.txt
echo "== driver flags"; python bench.py --steps 20 --warmup 5 2>/dev/null | python -c "import sys,json; l=json.loads(sys.stdin.read()); print(l['value']/1e6, l['ms_per_step'], l['ms_per_step_median'], l['roofline']['frac'], l['cpu_baseline'])"
