#!/bin/bash
# round-5 first run: the GPU suite + the C2 step in its default form, with float32 master parameters (hi/lo weight gradients on by
# itself) and the N > 1 step rehearsed on a one-rank RCCL group, eager and captured (same box)
out=gpurun_out; mkdir -p $out
bash tools/gpu_jobs/suite.sh > $out/r05_base_suite.txt 2>&1
tail -40 $out/r05_base_suite.txt
summ='import sys,json; l=json.loads(sys.stdin.read()); s=l["stage_ms"]; print(sys.argv[1], round(l["ms_per_step"],4), round(l["ms_per_step_median"],4), l.get("collective_ms"), " ".join("%s=%.0f"%(k.split(".")[1],x*1e3) for k,x in s.items() if x>0.003))'
for i in 1 2; do
timeout -k 10 300 python bench.py --steps 50 --warmup 10 --no-cpu-baseline 2>/dev/null | tee $out/r05_base_c2_$i.json | python -c "$summ" c2
timeout -k 10 300 python bench.py --steps 50 --warmup 10 --no-cpu-baseline --f32-params 2>/dev/null | tee $out/r05_base_c2_f32p_$i.json | python -c "$summ" c2_f32params_hilo
done
timeout -k 10 300 python bench.py --steps 50 --warmup 10 --no-cpu-baseline --force-dp 2>$out/r05_forcedp.err | tee $out/r05_base_c2_forcedp.json | python -c "$summ" c2_forcedp
timeout -k 10 300 python bench.py --steps 50 --warmup 10 --no-cpu-baseline --force-dp --graph 2>$out/r05_forcedp_graph.err | tee $out/r05_base_c2_forcedp_graph.json | python -c "$summ" c2_forcedp_graph
timeout -k 10 400 python bench.py --steps 20 --warmup 5 2>$out/r05_c2_full.err | tee $out/r05_base_c2_full.json | python -c "import sys,json; l=json.loads(sys.stdin.read()); print(json.dumps(l['cpu_baseline'])); print(l['stage_pass'])"
