# usage: bash tools/build_variant.sh <name> <sed expression for orb_extract_kernels.hip> [more make args]
# Builds tmp_exp/<name>.so from a copy of the sources with the expression applied (experiments only).
set -e
name=$1; expr=$2; shift 2
root=$(cd "$(dirname "$0")/.." && pwd)
work=/tmp/orb_variant_$name
rm -rf $work && mkdir -p $work/orb_slam2_amd $work/include $root/tmp_exp
cp -r $root/orb_slam2_amd/csrc $work/orb_slam2_amd/ && cp $root/include/orbhip.h $work/include/
rm -rf $work/orb_slam2_amd/csrc/_obj
sed -i "$expr" $work/orb_slam2_amd/csrc/orb_extract_kernels.hip
make -C $work/orb_slam2_amd/csrc -j8 OUT=$root/tmp_exp/$name.so "$@" 2>&1 | grep -E "error|warning" || true
ls -la $root/tmp_exp/$name.so
