#!/bin/bash
out=gpurun_out; mkdir -p $out
python tools/debug/hilo_errors.py 8192 2>&1 | grep -v amdgpu.ids | tee $out/m_hilo.txt
summ='import sys,json; l=json.loads(sys.stdin.read()); s=l["stage_ms"]; print(sys.argv[1], round(l["ms_per_step"],4), round(l["ms_per_step_median"],4), " ".join("%s=%.0f"%(k.split(".")[1],x*1e3) for k,x in s.items() if x>0.008))'
for i in 1 2; do
python bench.py --steps 50 --warmup 10 --no-cpu-baseline 2>/dev/null | python -c "$summ" default | tee -a $out/m_hilo.txt
python bench.py --steps 50 --warmup 10 --no-cpu-baseline --hilo 2>/dev/null | python -c "$summ" hilo | tee -a $out/m_hilo.txt
done
