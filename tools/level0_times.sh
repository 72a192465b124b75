# usage (on the GPU box): bash tools/level0_times.sh  -- per-launch durations of the level-0 kernel over a bench run
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/l0times
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/l0times -- python3 bench.py --steps 10 --warmup 2 --knn 0 --cpu-frames 0 --host-reps 0 > gpurun_out/l0times.log 2>&1 || { tail -5 gpurun_out/l0times.log; exit 1; }
python3 - <<'PY'
import csv, glob
rows = list(csv.DictReader(open(glob.glob("gpurun_out/l0times/*/*kernel_trace.csv")[0])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
prev = None
for r in rows:
    d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    name = r["Kernel_Name"].replace("orbhip::", "")[:24]
    if "pyr_level0" in name or ("pyr_resize" in name and prev and "level0" in prev[0]):
        gap = (int(r["Start_Timestamp"]) - prev[2]) / 1000 if prev else 0
        print(f"{name:24s} {d/1000:8.1f} us  after {prev[0] if prev else '-':24s} ({prev[1]/1000 if prev else 0:8.1f} us) gap {gap:7.1f} grid {r.get('Grid_Size_X')}x{r.get('Grid_Size_Y')}")
    prev = (name, d, int(r["End_Timestamp"]))
PY
