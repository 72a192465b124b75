#!/usr/bin/env python3
"""Summarise a rocprofv3 --pmc CSV: per kernel, mean counter value per dispatch."""
import csv
import glob
import sys
from collections import defaultdict

d = sys.argv[1]
acc = defaultdict(lambda: defaultdict(list))
for fn in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(fn)):
        name = row["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")[:40]
        acc[name][row["Counter_Name"]].append(float(row["Counter_Value"]))
for name, cs in sorted(acc.items()):
    if not name.startswith("orbhip"):
        continue
    print(name, " ".join("%s=%.4g(n=%d)" % (c, sum(v) / len(v), len(v)) for c, v in sorted(cs.items())))
    if len(sys.argv) > 2 and sys.argv[2] == "max":  # launches of very different sizes (a warm-up among them): the largest one
        print(name, "[largest launch]", " ".join("%s=%.4g" % (c, max(v)) for c, v in sorted(cs.items())))
