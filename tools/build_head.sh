# usage: bash tools/build_head.sh [rev]   -- builds the library of a committed revision (default HEAD) as tmp_exp/<rev>.so,
# for timing the working tree against it on the same box (tools/run_variants.sh)
set -e
rev=${1:-HEAD}
root=$(cd "$(dirname "$0")/.." && pwd)
work=/tmp/orb_rev_$rev
rm -rf $work && mkdir -p $work $root/tmp_exp
git -C $root archive $rev orb_slam2_amd/csrc include | tar -x -C $work
make -C $work/orb_slam2_amd/csrc -j8 OUT=$root/tmp_exp/$rev.so 2>&1 | grep -E "error|warning" || true
ls -la $root/tmp_exp/$rev.so
