#!/bin/bash
# GPU-box: the round's profile set -> gpurun_out/r03/ (copy what is to be judged into profiles/)
set -o pipefail
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
out=$root/gpurun_out/r03
mkdir -p "$out"
cd "$root"
python3 bench.py > "$out/bench_fp32.json" 2> "$out/bench_fp32.err" && cut -c1-300 "$out/bench_fp32.json"
echo "[r03] bench done"
export TMPDIR=/tmp
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d "$out/rocprof_bench" -o b -- python3 "$root/bench.py" --no-cpu-baseline > "$out/bench_under_rocprof.json" 2> "$out/bench_under_rocprof.err")
cp "$out"/rocprof_bench/*/b_kernel_stats.csv "$out/bench_kernel_stats.csv" 2>/dev/null || cp $(find "$out/rocprof_bench" -name '*kernel_stats.csv' | head -1) "$out/bench_kernel_stats.csv"
rm -rf "$out/rocprof_bench"
echo "[r03] bench under rocprof done"
bash tools/pmc_fused.sh r03 blocked > "$out/pmc_fused.log" 2>&1; cp gpurun_out/r03_pmc_traffic.json "$out/" 2>/dev/null
echo "[r03] pmc fused done"
bash tools/profile_workload.sh r03_s2030_strip tools/probe_s2030.py 3 > "$out/profile_s2030.log" 2>&1
cp gpurun_out/r03_s2030_strip_kernel_stats.csv gpurun_out/r03_s2030_strip_profile.json "$out/" 2>/dev/null
echo "[r03] s2030 strip profile done"
python3 tools/bench_configs.py cfg1 cfg1-b10 cfg2 cfg3 cfg4 cfg5 s2030-arch args3dmri args3dmri-b8 2>/dev/null | grep '^{' > "$out/configs.jsonl"
cut -c1-24,100-230 "$out/configs.jsonl"
echo "[r03] configs done"
./tools/probes/probe_stream 9 > "$out/stream_shapes.jsonl" 2>&1
./tools/probes/probe_stream 9 rw > "$out/stream_rw.jsonl" 2>&1
echo "[r03] stream probes done"
