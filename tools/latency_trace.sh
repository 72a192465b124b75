# usage (on the GPU box): bash tools/latency_trace.sh  -- kernel timeline of one single-frame host-to-host extraction
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/ltrace
timeout -k 10 120 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d gpurun_out/ltrace -- python3 tools/latency_trace.py 40 > gpurun_out/ltrace.log 2>&1 || { tail -5 gpurun_out/ltrace.log; exit 1; }
tail -1 gpurun_out/ltrace.log
python3 - <<'PY'
import csv, glob
ev = []
for r in csv.DictReader(open(glob.glob("gpurun_out/ltrace/*/*kernel_trace.csv")[0])):
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].replace("orbhip::", "").replace("void ", "")[:26]))
mc = glob.glob("gpurun_out/ltrace/*/*memory_copy_trace.csv")
if mc:
    for r in csv.DictReader(open(mc[0])):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "copy " + r.get("Direction", "")[:20]))
ev.sort()
idx = [i for i, e in enumerate(ev) if "pyr_level0" in e[2]]
i0 = idx[-2]
while i0 > 0 and ev[i0 - 1][2].startswith("copy") and ev[i0][0] - ev[i0 - 1][1] < 50000:
    i0 -= 1
t0 = ev[i0][0]
prev = t0
for s, e, name in ev[i0:idx[-1]]:
    print(f"{name:28s} start+{(s-t0)/1000:7.1f}  dur {(e-s)/1000:6.1f}  gap {(s-prev)/1000:6.1f}")
    prev = e
PY
