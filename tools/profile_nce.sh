#!/bin/bash
# Artefacts of the contrastive side at configs[2] size (8192 x 65536 x 768) -> gpurun_out/<tag>_c3nce_*:
#   kernel trace of `bench.py --config c3 --contrastive`, the bench line, SQ / TCC counters of the loss-side kernels alone.
set -o pipefail
tag=${1:-r03}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out
mkdir -p $out
rocprofv3 --kernel-trace --stats -d $out/ks_nce -o r -- python3 bench.py --settle-seconds 0 --config c3 --contrastive --steps 20 --warmup 5 --no-cpu-baseline > /dev/null 2>&1
python3 tools/kernel_stats_from_db.py $out/ks_nce/r_results.db $out/${tag}_c3nce_kernel_stats.csv | head -12
rm -rf $out/ks_nce
python3 bench.py --config c3 --contrastive --steps 50 --warmup 10 > $out/${tag}_c3nce_bench.json 2> /dev/null
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VMEM" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE TCC_HIT_sum TCC_MISS_sum SQ_INSTS_VALU SQ_INSTS_LDS GRBM_GUI_ACTIVE" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace -d $out/pmc_nce$i -o pmc -- python3 tools/nce_time.py sym 8192 65536 768 2 > /dev/null 2>&1
done
python3 tools/pmc_summary.py $out/pmc_nce1 $out/pmc_nce2 $out/pmc_nce3 $out/pmc_nce4 | grep -E "^kernel|nce_" > $out/${tag}_c3nce_sq_counters.csv
rm -rf $out/pmc_nce1 $out/pmc_nce2 $out/pmc_nce3 $out/pmc_nce4
cut -c1-300 $out/${tag}_c3nce_sq_counters.csv
python3 -c "
import json; l=json.load(open('$out/${tag}_c3nce_bench.json')); r=l['roofline']
print(round(l['ms_per_step'],4), round(l['value']/1e6,3), 'M/s', r['kernel'], round(r['frac'],3), round(l['path_mfma_frac'],3))"
