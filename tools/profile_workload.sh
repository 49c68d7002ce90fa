#!/bin/bash
# GPU-box: rocprofv3 kernel table + separate PMC passes (HBM bytes, LDS conflicts, MFMA busy) of one workload.
#   tools/profile_workload.sh <tag> <python script> [args...]   ->  gpurun_out/<tag>_kernel_stats.csv, <tag>_profile.json
# The program goes directly after `--` (no env/bash hop); counters are collected without any trace option.
set -e -o pipefail
tag=$1; shift
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
out=$root/gpurun_out/prof_$tag
mkdir -p "$out"
export TMPDIR=/tmp
cd /tmp
script=$root/$1; shift
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace" -o t -- python3 "$script" "$@" > "$out/trace.log" 2>&1
echo "[$tag] kernel trace done"
i=0
for pmc in "FETCH_SIZE" "WRITE_SIZE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS"; do
  rocprofv3 --pmc $pmc --output-format csv -d "$out/pmc$i" -o p -- python3 "$script" "$@" > "$out/pmc$i.log" 2>&1
  echo "[$tag] pmc pass $i done"
  i=$((i+1))
done
stats=$(find "$out/trace" -name '*kernel_stats.csv' | head -1)
cp "$stats" "$root/gpurun_out/${tag}_kernel_stats.csv"
python3 "$root/tools/pmc_fold.py" "$tag" "$stats" $(find "$out"/pmc* -name '*counter_collection.csv') > "$root/gpurun_out/${tag}_profile.json"
rm -rf "$out"/pmc*/ "$out/trace"
echo "[$tag] folded -> gpurun_out/${tag}_profile.json"
