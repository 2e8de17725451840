#!/bin/bash
# the GPU test suite file by file (an abort in one file does not lose the others' results); summary lines on stdout
out=gpurun_out; mkdir -p $out; : > $out/suite.txt
for f in tests/test_*.py; do
  timeout -k 10 900 python -m pytest $f -m gpu -q > $out/suite_part.txt 2>&1
  echo "== $f rc=$?" >> $out/suite.txt; grep -E "^FAILED|^ERROR|passed|failed|Fatal|Aborted|fault" $out/suite_part.txt >> $out/suite.txt
  grep -B2 -A12 "^E  " $out/suite_part.txt | head -60 >> $out/suite.txt
done
grep -E "^==|passed|failed|^FAILED|^ERROR|Fatal" $out/suite.txt
