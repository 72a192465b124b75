# usage (on the GPU box): bash tools/kernel_times.sh [args for stage_times.py]
# rocprofv3 kernel trace of a few extractor steps; prints the per-launch durations of one step.
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/ktimes
timeout -k 10 120 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/ktimes -- python3 tools/stage_times.py orb_slam2_amd/liborbhip.so --steps 4 --check 0 "$@" > gpurun_out/ktimes.log 2>&1 || { tail -5 gpurun_out/ktimes.log; exit 1; }
python3 - <<'PY'
import csv, glob
rows = list(csv.DictReader(open(glob.glob("gpurun_out/ktimes/*/*kernel_trace.csv")[0])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "pyr_level0" in r["Kernel_Name"]]
i0 = idx[-2]
t0 = int(rows[i0]["Start_Timestamp"])
for r in rows[i0:idx[-1]]:
    d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    name = r["Kernel_Name"].replace("orbhip::", "").replace("void ", "")[:28]
    print(f"{name:28s} {d/1000:8.1f} us  start+{(int(r['Start_Timestamp'])-t0)/1000:8.1f}  grid {r.get('Grid_Size_X')}x{r.get('Grid_Size_Y')}x{r.get('Grid_Size_Z')} lds {r.get('LDS_Block_Size')}")
PY
