# kernel names a pytest selection launches: tools/trace_tests.sh <out-tag> <pytest args...>   (on the GPU box)
export TMPDIR=/tmp
root=$GRAFT_REPO_ROOT
tag=$1; shift
mkdir -p $root/gpurun_out/$tag
cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $root/gpurun_out/$tag/tr -o t -- python3 -m pytest "$@" > $root/gpurun_out/$tag/out.log 2>&1
rc=$?
f=$(find $root/gpurun_out/$tag/tr -name '*kernel_stats.csv' 2>/dev/null | head -1)
if [ -n "$f" ]; then cp "$f" $root/gpurun_out/$tag/kernel_stats.csv; fi
rm -rf $root/gpurun_out/$tag/tr
grep -E "passed|failed|error" $root/gpurun_out/$tag/out.log | tail -3
exit $rc
