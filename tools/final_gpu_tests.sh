set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/final
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/final/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/final/pytest_gpu.log
tail -3 gpurun_out/final/pytest_gpu.log
