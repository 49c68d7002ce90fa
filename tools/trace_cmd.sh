# per-kernel times of one python command: tools/trace_cmd.sh <out-tag> <script.py> [args...]   (on the GPU box)
export TMPDIR=/tmp
root=$GRAFT_REPO_ROOT
tag=$1; shift
mkdir -p $root/gpurun_out/$tag
script=$root/$1; shift
cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $root/gpurun_out/$tag/tr -o t -- python3 $script "$@" > $root/gpurun_out/$tag/out.log 2>&1
rc=$?
f=$(find $root/gpurun_out/$tag/tr -name '*kernel_stats.csv' 2>/dev/null | head -1)
if [ -n "$f" ]; then cp "$f" $root/gpurun_out/$tag/kernel_stats.csv; fi
rm -rf $root/gpurun_out/$tag/tr
grep '^{' $root/gpurun_out/$tag/out.log | cut -c1-400
exit $rc
