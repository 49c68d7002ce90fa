export TMPDIR=/tmp
root=$GRAFT_REPO_ROOT
mkdir -p $root/gpurun_out/r3ae
cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $root/gpurun_out/r3ae/tr -o t -- python3 $root/tools/bench_configs.py args3dmri > $root/gpurun_out/r3ae/out.log 2>&1
cp $(find $root/gpurun_out/r3ae/tr -name '*kernel_stats.csv' | head -1) $root/gpurun_out/r3ae/args3dmri_kernel_stats.csv
rm -rf $root/gpurun_out/r3ae/tr
head -12 $root/gpurun_out/r3ae/args3dmri_kernel_stats.csv | cut -c1-150
