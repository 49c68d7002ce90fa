#!/bin/bash
# GPU-box: the round's closing measurements -> gpurun_out/final/
set -o pipefail
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
out=$root/gpurun_out/final
mkdir -p "$out"
cd "$root"
python3 bench.py > "$out/bench_fp32.json" 2> "$out/bench_fp32.err" && tail -1 "$out/bench_fp32.json" | cut -c1-400
python3 bench.py --code-storage bf16 > "$out/bench_bf16.json" 2> "$out/bench_bf16.err" && tail -1 "$out/bench_bf16.json" | cut -c1-300
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/rocprof_bench" -o b -- python3 "$root/bench.py" > "$out/bench_under_rocprof.json" 2> "$out/bench_under_rocprof.err"
cp "$out/rocprof_bench/b_kernel_stats.csv" "$out/bench_kernel_stats.csv"
rm -rf "$out/rocprof_bench"
cd "$root"
python3 tools/bench_configs.py cfg1 cfg1-b10 cfg2 cfg2-b16 cfg3 cfg4 cfg5 s2030-arch csr-f2 2>/dev/null | grep '^{' > "$out/configs.jsonl"
cut -c1-20,100-230 "$out/configs.jsonl"
