# usage: bash tools/prof_pmc.sh <tag> <counters...>   (one PMC pass over the 256-frame device-resident step; run on the GPU box)
set -e
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
tag=$1; shift
timeout -s KILL 120 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d gpurun_out/pmc_$tag -- python3 bench.py --steps 2 --warmup 1 --knn 0 --cpu-frames 0 --host-reps 0 --batch 256 > gpurun_out/pmc_$tag.log 2>&1
python3 tools/pmc_summary.py gpurun_out/pmc_$tag
