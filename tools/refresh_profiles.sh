# Copies the summaries of the last `bash tools/prof_round.sh <tag>` run (gpurun_out/prof_<tag>/) into profiles/.
# usage: bash tools/refresh_profiles.sh <tag>       e.g. r2
set -e
tag=${1:-r3}
src=gpurun_out/prof_$tag
python3 - "$tag" "$src" <<'PY'
import csv, glob, sys
tag, src = sys.argv[1], sys.argv[2]
import os
for sub, name, head in [
        ('stats', 'kernel_stats', "# rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 10 --warmup 2 --cpu-frames 0 --host-reps 0 --knn-reps 3 --split 0 --overlap 0 "
         "(every kernel alone on the GPU, 256 frames per launch: the times bench.py's roofline is computed from; the calls include "
         "the benchmark's 100 ms of untimed spin-up steps; 1M x 1M kNN-2: "
         "knn2_mfma_kernel x3 + warm-ups, knn2_kernel x1; MI355X). pyr_level01_kernel makes levels 0 and 1, pyr_resize_tile_kernel "
         "takes six launches per step (levels 2-7)\n"),
        ('stats_default', 'kernel_stats_default', "# rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 10 --warmup 2 --cpu-frames 0 --host-reps 0 --knn 0 "
         "(the benchmark as the driver runs it: every step as two half batches of 128 frames on two streams, the matcher beside the "
         "next step's extraction, so kernels share the GPU and their durations overlap; the profiling pass after the timed region and "
         "the first extraction add launches of 256 frames)\n")]:
    fns = glob.glob(src + '/' + sub + '/*/*kernel_stats.csv')
    if not fns:
        continue
    fn = max(fns, key=os.path.getmtime)  # (earlier runs' files may still lie there)
    rows = [r for r in csv.reader(open(fn))]
    out = open('profiles/%s_%s.csv' % (tag, name), 'w')
    out.write(head)
    w = csv.writer(out); w.writerow(rows[0])
    for r in rows[1:]:
        if 'orbhip' in r[0]:
            r[0] = r[0].replace('(anonymous namespace)::', '').split('(')[0].replace('void ', ''); w.writerow(r)
    out.close()
PY
( echo "# rocprofv3 --pmc passes (one run per pass) over python3 bench.py --steps 2 --warmup 1 --knn 0 --cpu-frames 0 --host-reps 0 --batch 256 --split 0; mean per launch"
  for p in sq1 sq2 grbm; do echo "# pass $p"; cat $src/pmc_${p}_sum.log; done
  echo "# pass fetch (FETCH_SIZE in KiB at 64 B per request: double it for bytes read on gfx950, see r1_hbm_counter_calibration.txt)"; cat $src/pmc_fetch_sum.log
  echo "# pass write (WRITE_SIZE in KiB)"; cat $src/pmc_write_sum.log
  echo "# pass knn: python3 bench.py --steps 1 --warmup 0 --cpu-frames 0 --host-reps 0 --batch 8 --knn-reps 1 (1M x 1M; GRBM_GUI_ACTIVE is summed over the 8 XCDs)"; grep -i knn $src/pmc_knn_sum.log ) > profiles/${tag}_pmc_counters.txt
python3 tools/pmc_to_traffic.py $src/pmc_fetch $src/pmc_write profiles/${tag}_traffic.json 256 $src/pmc_sq1 > /dev/null
cut -c1-120 profiles/${tag}_kernel_stats.csv
