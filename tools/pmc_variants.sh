# usage (on the GPU box): bash tools/pmc_variants.sh "<counters>" [args for stage_times.py]
# One rocprofv3 --pmc pass per library build (orb_slam2_amd/liborbhip.so and tmp_exp/*.so); prints the
# mean counter values per launch of every extractor kernel.
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
ctr="$1"; shift
for so in orb_slam2_amd/liborbhip.so tmp_exp/*.so; do
  [ -f "$so" ] || continue
  tag=$(basename $so .so)
  rm -rf gpurun_out/pmcv_$tag
  timeout -s KILL 120 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d gpurun_out/pmcv_$tag -- python3 tools/stage_times.py "$so" --steps 2 --check 0 "$@" > gpurun_out/pmcv_$tag.log 2>&1 || { tail -5 gpurun_out/pmcv_$tag.log; exit 1; }
  echo "== $tag"; python3 tools/pmc_summary.py gpurun_out/pmcv_$tag | grep -i "fast"
done
