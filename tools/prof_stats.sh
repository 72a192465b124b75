set -e
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r1 -- python3 bench.py --steps 10 --warmup 2 --knn 0 --cpu-frames 0 > gpurun_out/prof_r1_bench.log 2>&1
ls -R gpurun_out/prof_r1 | head -30
