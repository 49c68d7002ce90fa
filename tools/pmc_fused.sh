#!/bin/bash
# GPU-box: HBM traffic of the fused 2-D kernels of THIS source revision (separate rocprofv3 --pmc passes, no trace options),
# folded into gpurun_out/<tag>_pmc_traffic.json (copy it to profiles/ so that bench.py quotes it).
#   tools/pmc_fused.sh <tag> <layout: nchw|blocked|blocked_bf16>
set -e -o pipefail
tag=$1; lay=${2:-blocked}
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
out=$root/gpurun_out/pmc_$tag
mkdir -p "$out"
export TMPDIR=/tmp BK_LAYOUTS=$lay BK_ROUNDS=1 BK_REPS=2
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d "$out/$c" -o p -- python3 "$root/tools/bench_kernels.py" > "$out/$c.log" 2>&1
  echo "[$tag] $c pass done"
done
python3 "$root/tools/pmc_summary.py" "$(find "$out/FETCH_SIZE" -name '*counter_collection.csv' | head -1)" \
        "$(find "$out/WRITE_SIZE" -name '*counter_collection.csv' | head -1)" "$root/gpurun_out/${tag}_pmc_traffic.json" "$lay" > /dev/null
rm -rf "$out/FETCH_SIZE" "$out/WRITE_SIZE"
echo "[$tag] -> gpurun_out/${tag}_pmc_traffic.json"
