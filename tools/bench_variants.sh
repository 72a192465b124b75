# usage (on the GPU box): bash tools/bench_variants.sh [bench.py args]
# The bench's device-resident step (extraction + matcher, no kNN / CPU / host legs) for orb_slam2_amd/liborbhip.so
# and every tmp_exp/*.so build, on the same box (ORBHIP_LIB names the library the package loads).
cd $GRAFT_REPO_ROOT
for so in orb_slam2_amd/liborbhip.so tmp_exp/*.so; do
  [ -f "$so" ] || continue
  ORBHIP_LIB=$PWD/$so timeout -k 10 120 python3 bench.py --knn 0 --cpu-frames 0 --host-reps 0 "$@" 2>/dev/null |
    python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('%-28s' % '$so', d['value'], d['ms_per_step'], d['stages_ms_per_step'])" || exit 1
done
