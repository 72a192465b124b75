# usage (on the GPU box): bash tools/prof_gpu_tests.sh [pytest -k expression]
# Kernel trace of the GPU parity tests themselves (every kernel instantiation the tests reach); the summary of the
# whole suite is kept as profiles/<round>_gpu_tests_kernel_stats.csv.
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/gt
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/gt -- python3 -m pytest tests -m gpu -q -x --timeout 200 --timeout-method thread -k "${1:-test}" > gpurun_out/gt.log 2>&1
grep -v "^E2026\|^W2026" gpurun_out/gt.log | tail -n 2
