# FETCH_SIZE / WRITE_SIZE calibration passes (run on the GPU box)
set -e
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for c in FETCH_SIZE WRITE_SIZE "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_BUBBLE_sum"; do
  tag=$(echo $c | cut -d' ' -f1)
  timeout -s KILL 60 rocprofv3 --kernel-trace --pmc $c --output-format csv -d gpurun_out/calib_$tag -- ./tools/ubench/hbm_calib > gpurun_out/calib_$tag.log 2>&1
done
python3 - <<'PY'
import csv, glob
for fn in sorted(glob.glob("gpurun_out/calib_*/**/*counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(fn)):
        print(r["Kernel_Name"][:60], r["Counter_Name"], r["Counter_Value"])
PY
