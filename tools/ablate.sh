#!/bin/bash
# GPU-box timing experiment: fused kernels with 4-byte (default) vs 16-byte (CDL_FUSED_WIDE=1) fat accesses
for n in 0 1; do
  echo "== CDL_FUSED_WIDE=$n"
  if [ $n -eq 1 ]; then export CDL_FUSED_WIDE=1; else unset CDL_FUSED_WIDE; fi
  timeout -k 10 100 python tools/bench_kernels.py 2>/dev/null | grep -E "k_iter_fwd|k_stage<BWD>|k_wgrad2d|yardstick" | python -c "
import sys, json
for l in sys.stdin:
    r = json.loads(l); print('   %-34s %.3f ms  %.0f GB/s' % (r['kernel'], r['ms'], r.get('GBps', 0)))"
done
