# Copies the summaries of the last tools/prof_stats.sh + tools/prof_pmc.sh runs (gpurun_out/) into profiles/.
# usage: bash tools/refresh_profiles.sh <tag>       e.g. r1
set -e
tag=${1:-r1}
python3 - "$tag" <<'PY'
import csv, glob, sys
tag = sys.argv[1]
fn = glob.glob('gpurun_out/prof_r1/*/*kernel_stats.csv')[0]
rows = [r for r in csv.reader(open(fn))]
out = open('profiles/%s_kernel_stats.csv' % tag, 'w')
out.write("# rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 10 --warmup 2 --knn 0 --cpu-frames 0 (256 frames per launch, MI355X)\n")
w = csv.writer(out); w.writerow(rows[0])
for r in rows[1:]:
    if 'orbhip' in r[0]:
        r[0] = r[0].split('(')[0].replace('void ', ''); w.writerow(r)
PY
( echo "# rocprofv3 --pmc passes (one pass per line group) over python3 bench.py --steps 2 --warmup 1 --knn 0 --cpu-frames 0 --batch 256; mean per launch"
  echo "# pass sq1"; cat gpurun_out/pmc_sq1_sum.log
  echo "# pass sq2"; cat gpurun_out/pmc_sq2_sum.log
  echo "# pass fetch (FETCH_SIZE in KiB at 64 B per request: double it for bytes read on gfx950, see r1_hbm_counter_calibration.txt)"; cat gpurun_out/pmc_fetch_sum.log
  echo "# pass write (WRITE_SIZE in KiB)"; cat gpurun_out/pmc_write_sum.log ) > profiles/${tag}_pmc_counters.txt
python3 tools/pmc_to_traffic.py gpurun_out/pmc_fetch gpurun_out/pmc_write profiles/${tag}_traffic.json 256 > /dev/null
cut -c1-110 profiles/${tag}_kernel_stats.csv
