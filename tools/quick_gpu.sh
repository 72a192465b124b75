# usage (on the GPU box): bash tools/quick_gpu.sh <tag>  -- extractor tests, then a short bench without the kNN / CPU legs
set -e
tag=${1:-q}
timeout -k 10 300 python -u -m pytest tests/test_extract_gpu.py tests/test_chain_gpu.py -m gpu -x -q --timeout 120 --timeout-method thread > gpurun_out/pytest_$tag.log 2>&1 || { tail -30 gpurun_out/pytest_$tag.log; exit 1; }
tail -1 gpurun_out/pytest_$tag.log
timeout -k 10 200 python bench.py --knn 0 --cpu-frames 0 --host-reps 0 > gpurun_out/bench_$tag.json 2> gpurun_out/bench_$tag.err
python - <<PY
import json
d = json.load(open("gpurun_out/bench_$tag.json"))
print(d["value"], d["stages_ms_per_step"], d.get("latency_ms_single_frame"))
PY
