# usage (on the GPU box): bash tools/run_variants.sh [args for stage_times.py]
# Times the extractor stages for orb_slam2_amd/liborbhip.so and every tmp_exp/*.so build.
cd $GRAFT_REPO_ROOT
for so in orb_slam2_amd/liborbhip.so tmp_exp/*.so; do
  [ -f "$so" ] || continue
  timeout -k 10 120 python3 tools/stage_times.py "$so" "$@" 2>&1 | grep -v amdgpu.ids || exit 1
done
