# usage (on the GPU box): bash tools/prof_round.sh <tag>     e.g. r2
# Collects everything profiles/<tag>_* is made from, each rocprofv3 pass a run of its own:
#   kernel trace + stats of bench.py (extraction + matcher + 1M x 1M kNN-2, both kNN kernels),
#   four PMC passes over the extraction step, one PMC pass over the kNN-2 leg.
set -e
tag=${1:-r3}
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
out=gpurun_out/prof_$tag
rm -rf $out && mkdir -p $out
# kernels alone: one stream, whole batches (the times `roofline` is computed from) ...
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 bench.py --steps 10 --warmup 2 --cpu-frames 0 --host-reps 0 --knn-reps 3 --split 0 --overlap 0 > $out/stats_bench.log 2>&1
# ... and the benchmark as the driver runs it: two staggered half batches, the matcher beside the next extraction
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_default -- python3 bench.py --steps 10 --warmup 2 --cpu-frames 0 --host-reps 0 --knn 0 > $out/stats_default_bench.log 2>&1
pass() {  # name, counters, bench args
  name=$1; ctr=$2; shift 2
  timeout -s KILL 170 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $out/pmc_$name -- python3 bench.py "$@" > $out/pmc_$name.log 2>&1
  python3 tools/pmc_summary.py $out/pmc_$name > $out/pmc_${name}_sum.log
}
EX="--steps 2 --warmup 1 --knn 0 --cpu-frames 0 --host-reps 0 --batch 256 --split 0"
pass sq1 "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" $EX
pass sq2 "SQ_INSTS_VMEM SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_THREAD_CYCLES_VALU" $EX
pass fetch "FETCH_SIZE" $EX
pass write "WRITE_SIZE" $EX
pass grbm "GRBM_GUI_ACTIVE" $EX
pass knn "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_LDS SQ_BUSY_CYCLES SQ_INSTS_LDS SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE" --steps 1 --warmup 0 --cpu-frames 0 --host-reps 0 --batch 8 --knn-reps 1
tail -3 $out/stats_bench.log | cut -c1-300
python3 tools/pmc_summary.py $out/pmc_knn max > $out/pmc_knn_sum.log
