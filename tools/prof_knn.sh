# usage: bash tools/prof_knn.sh <tag> <counters...>   (one PMC pass over the 1M x 1M kNN-2 leg; run on the GPU box)
set -e
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
tag=$1; shift
timeout -s KILL 170 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d gpurun_out/pmc_$tag -- python3 bench.py --steps 1 --warmup 0 --cpu-frames 0 --batch 8 > gpurun_out/pmc_$tag.log 2>&1
python3 tools/pmc_summary.py gpurun_out/pmc_$tag | grep knn
