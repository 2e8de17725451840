#!/bin/bash
bash tools/profile_nce.sh r05 > gpurun_out/r05_nce_profile.txt 2>&1; tail -22 gpurun_out/r05_nce_profile.txt
summ='import sys,json; l=json.loads(sys.stdin.read()); print(sys.argv[1], "ok", round(l["ms_per_step"],4), l["n_gpus"], l.get("collective_ms"), (l.get("strong_scaling") or {}).get("ms_per_step"))'
AECF_DIST_BACKEND=gloo timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --config tiny --steps 5 --warmup 2 2>gpurun_out/v9.err > gpurun_out/v9.out; echo "torchrun rc=$?"; wc -l gpurun_out/v9.out; python -c "$summ" torchrun2_gloo < gpurun_out/v9.out
