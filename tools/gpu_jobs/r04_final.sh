#!/bin/bash
out=gpurun_out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
python bench.py --steps 20 --warmup 5 > $out/final_bench_driver_flags.json 2>$out/final_bench_err.txt; python -c "
import json; l=json.load(open('$out/final_bench_driver_flags.json')); print('driver flags', l['ms_per_step'], l['value']/1e6, l['roofline'], l['cpu_baseline'])"
python -m pytest tests -x -q -m gpu 2>&1 | tail -4
