#!/bin/bash
# round-4 records: kernel stats, FETCH/WRITE traffic (c2, c3, c5), SQ counters, bench lines of the three shard configurations
out=gpurun_out; mkdir -p $out
tools/profile_round.sh r04 2>&1 | tee $out/r04_profile_round.log
tools/pmc_sq.sh 2>&1 | tail -14
cp $out/pmc_sq.csv $out/r04_c2_sq_counters.csv
