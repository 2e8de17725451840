#!/bin/bash
# tools/gpu_jobs/driver_like.sh: what the driver runs at round end -- the GPU suite, smoke(), the default bench line
out=gpurun_out; mkdir -p $out
bash tools/gpu_jobs/suite.sh > $out/driver_like_suite.txt 2>&1
grep -E "^==|passed|failed|^FAILED" $out/driver_like_suite.txt
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -5
timeout -k 10 600 python bench.py --gpus 1 --steps 20 --warmup 5 2>$out/driver_like_bench.err > $out/driver_like_bench.json; echo "bench rc=$?"; wc -l $out/driver_like_bench.json
python - <<'PY'
import json
l=json.loads(open("gpurun_out/driver_like_bench.json").read().strip().splitlines()[-1])
print(l["metric"], l["value"], l["ms_per_step"], l["roofline"]["frac"], l["roofline"]["traffic"], l["weight_grad_mode"])
print(l["cpu_baseline"]["value"], l["cpu_baseline"]["sample"][:120], l["cpu_baseline"]["seconds"])
print(l["stage_pass"])
PY
