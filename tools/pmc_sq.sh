#!/bin/bash
# SQ stall-attribution counters of the c2 step (separate passes; BENCH_FLAGS = more bench.py flags, e.g. --f32-params), summarised
# per kernel: gpurun_out/pmc_sq.csv
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/sq; rm -rf $out; mkdir -p $out
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
           "SQ_INSTS_VALU SQ_INSTS_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM SQ_INSTS_SALU SQ_ACTIVE_INST_SCA" \
           "GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_WAIT_INST_ANY" ; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace -d $out/p$i -o pmc -- python3 bench.py --settle-seconds 0 --steps 3 --warmup 2 --no-cpu-baseline $BENCH_FLAGS > $out/p$i.log 2>&1 || tail -3 $out/p$i.log
done
python3 tools/pmc_summary.py $out/p1 $out/p2 $out/p3 > gpurun_out/pmc_sq.csv
head -12 gpurun_out/pmc_sq.csv | cut -c1-400
