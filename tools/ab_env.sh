#!/bin/bash
# usage: [CFG=c5] tools/ab_env.sh VAR v1 v2 ...   -- bench a config with VAR set to each value, print step / median / stage times (us)
var=$1; shift
for v in "$@"; do
  env $var=$v timeout -k 10 200 python bench.py --config ${CFG:-c2} --steps 40 --warmup 10 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json; l=json.loads(sys.stdin.read()); s=l['stage_ms']
short={'fwd.gate':'gate','fwd.vproj':'vp','fwd.outproj':'op','bwd.dout':'do','bwd.dw_out':'dwo','bwd.dscore':'dsu','bwd.dx':'dx','bwd.dw_v':'dwv','bwd.u':'u','bwd.finalize':'fin'}
print('$var=$v', round(l['ms_per_step'],4), round(l['ms_per_step_median'],4), ' '.join('%s=%.0f'%(short[k],x*1e3) for k,x in s.items() if k in short and x>0.01))"
done
