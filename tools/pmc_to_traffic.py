#!/usr/bin/env python3
"""Turn the FETCH_SIZE and WRITE_SIZE passes of tools/prof_pmc.sh into the per-launch HBM
traffic table bench.py reports (profiles/<tag>_traffic.json).

FETCH_SIZE on gfx950 tallies the 128-byte read requests of the L2's memory side at 64 bytes
(calibrated with tools/ubench/hbm_calib.hip for 1, 4 and 16 bytes per lane: the counter
shows exactly half of the bytes read), so it is doubled; WRITE_SIZE is exact.  Both are in KiB.
"""
import csv
import glob
import json
import sys
from collections import defaultdict


def per_kernel(d, counter):
    """kernel -> (mean per launch, launches)"""
    acc = defaultdict(list)
    for fn in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(fn)):
            if row["Counter_Name"] == counter:
                acc[row["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")].append(float(row["Counter_Value"]))
    return {k: (sum(v) / len(v), len(v)) for k, v in acc.items()}


def main():
    fetch_dir, write_dir, out, frames = sys.argv[1], sys.argv[2], sys.argv[3], int(sys.argv[4])
    fc, wc = per_kernel(fetch_dir, "FETCH_SIZE"), per_kernel(write_dir, "WRITE_SIZE")
    # optional fifth argument: the pass that holds SQ_INSTS_VALU / SQ_ACTIVE_INST_VALU (wave-level vector instructions
    # and the cycles the vector units were busy with them, summed over all SIMDs)
    ic = per_kernel(sys.argv[5], "SQ_INSTS_VALU") if len(sys.argv) > 5 else {}
    ac = per_kernel(sys.argv[5], "SQ_ACTIVE_INST_VALU") if len(sys.argv) > 5 else {}
    # steps of the profiled run = launches of a kernel that runs once per step (the quad-tree); a kernel
    # with several launches per step (the resize: one per level; FAST: two view heights) gets their sum too
    steps = max([n for k, (_, n) in fc.items() if "octree_kernel" in k] or [1])
    table = {}
    for k in sorted(fc):
        if not k.startswith("orbhip"):
            continue
        f, n = fc[k]
        w = wc.get(k, (0.0, 0))[0]
        rd, wr = 2.0 * f * 1024.0, w * 1024.0
        per_step = n / steps
        table[k.replace("orbhip::", "")] = {"fetch_size_kib_raw": round(f, 1), "write_size_kib": round(w, 1),
                                            "read_bytes": round(rd), "write_bytes": round(wr), "hbm_bytes_per_launch": round(rd + wr),
                                            "launches_per_step": round(per_step, 3), "hbm_bytes_per_step": round((rd + wr) * per_step)}
        if k in ic:
            table[k.replace("orbhip::", "")]["valu_insts_per_launch"] = round(ic[k][0])
            table[k.replace("orbhip::", "")]["valu_active_cycles_per_launch"] = round(ac.get(k, (0.0, 0))[0])
    json.dump({"frames_per_launch": frames, "correction": "read = 2 x FETCH_SIZE KiB, write = WRITE_SIZE KiB",
               "kernels": table}, open(out, "w"), indent=1)
    print(json.dumps(table, indent=1))


if __name__ == "__main__":
    main()
