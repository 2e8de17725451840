#!/bin/bash
out=gpurun_out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
cat > /tmp/cal.py <<'PY'
import torch
dev = torch.device("cuda:0")
n = 64 * 1024 * 1024           # elements
a32 = torch.randn(n, device=dev)                  # 268 MB
b32 = torch.empty_like(a32)
a16 = a32.to(torch.bfloat16)                      # 134 MB
b16 = torch.empty_like(a16)
torch.cuda.synchronize()
for _ in range(3):
    b32.copy_(a32)          # fp32 copy: reads 268.4 MB, writes 268.4 MB
    b16.copy_(a16)          # bf16 copy: reads 134.2, writes 134.2
    b16.copy_(a32)          # convert: reads 268.4, writes 134.2
    a32.add_(1.0)           # in place: reads 268.4, writes 268.4
torch.cuda.synchronize()
PY
for ctr in FETCH_SIZE WRITE_SIZE; do
  rm -rf $out/cal_$ctr
  rocprofv3 --pmc $ctr --kernel-trace -d $out/cal_$ctr -o pmc -- python3 /tmp/cal.py > /dev/null 2>&1
  python3 tools/pmc_all_kernels.py $out/cal_$ctr $ctr
  rm -rf $out/cal_$ctr
done
