#!/bin/bash
# end-of-round artefacts that changed after the first profile pass: the float32-master step (one-launch casts), hi/lo timing record
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out; mkdir -p $out
timeout -k 10 600 python -m pytest tests/test_dp_gpu.py -m gpu -q -x 2>&1 | grep -v "^\[W\|amdgpu.ids" | tail -4
f=$out/r05_c2_hilo_time.txt
echo "# float32-stored parameter gradients of the bf16 kernels at the headline shape [65536, 3, 512], 8 heads: error against fp32 math" > $f
echo "# (tools/debug/hilo_errors.py 65536: default products, then hi + lo products) and the step both ways on one box, three rounds" >> $f
echo "# (bench.py --steps 50 --warmup 10; f32_params_hilo = --f32-params: float32 master parameters, hi/lo weight-gradient products on" >> $f
echo "# by themselves, master-weight casts as one launch; default = bf16 parameters, the benchmarked combination)" >> $f
timeout -k 10 300 python tools/debug/hilo_errors.py 65536 2>/dev/null | tail -2 >> $f
summ='import sys,json; l=json.loads(sys.stdin.read()); s=l["stage_ms"]; print(sys.argv[1], round(l["ms_per_step"],4), round(l["ms_per_step_median"],4), " ".join("%s=%.0f"%(k.split(".")[1],x*1e3) for k,x in s.items() if x>0.008))'
for i in 1 2 3; do
timeout -k 10 300 python bench.py --steps 50 --warmup 10 --no-cpu-baseline 2>/dev/null | python -c "$summ" default >> $f
timeout -k 10 300 python bench.py --steps 50 --warmup 10 --no-cpu-baseline --f32-params 2>/dev/null | tee $out/r05_c2_f32params_bench.json | python -c "$summ" f32_params_hilo >> $f
done
cat $f
rocprofv3 --kernel-trace --stats -d $out/ksh -o r -- python3 bench.py --settle-seconds 0 --steps 20 --warmup 5 --no-cpu-baseline --f32-params > /dev/null 2>&1
python3 tools/kernel_stats_from_db.py $out/ksh/r_results.db $out/r05_c2_f32params_kernel_stats.csv | head -12
rm -rf $out/ksh
