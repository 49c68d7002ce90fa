#!/bin/bash
# GPU-box timing experiment: k_stage with parts switched off (CDL_FUSED_DEBUG bits, see cdl_fused2d.hip)
for d in 0 1 2 16 32 64 31 127; do
  echo "== CDL_FUSED_DEBUG=$d"
  CDL_FUSED_DEBUG=$d timeout -k 10 100 python tools/bench_kernels.py 2>/dev/null | grep -E "k_iter_fwd\[split3\]\"|k_stage<BWD>\[split3\]|first" | python -c "
import sys, json
for l in sys.stdin:
    r = json.loads(l); print('   %-32s %.3f ms' % (r['kernel'], r['ms']))"
done
