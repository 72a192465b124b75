"""Per-basic-block instruction mix of one kernel in an assembly listing (hipcc -S --cuda-device-only).
usage: python tools/asm_blocks.py file.s kernel_name_substring [block_to_print ...]"""
import collections
import re
import sys

path, key = sys.argv[1], sys.argv[2]
show = set(sys.argv[3:])
lines = open(path).read().split("\n")
start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\S*%s\S*:" % key, l))
end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
blocks, cur = [], ["entry", []]
for l in lines[start + 1:end + 1]:
    t = l.strip()
    if not t or t.startswith(";"):
        continue
    if re.match(r"^\.?[A-Za-z_0-9]+:", t):
        blocks.append(cur)
        cur = [t.split(":")[0], []]
        continue
    if t.startswith("."):
        continue
    cur[1].append(t)
blocks.append(cur)
for name, ins in blocks:
    c = collections.Counter()
    for i in ins:
        op = i.split()[0]
        c["V" if op.startswith("v_") else "S" if op.startswith("s_") else "L" if op.startswith("ds_") else "M"] += 1
    br = [i.split()[-1] for i in ins if i.startswith(("s_cbranch", "s_branch"))]
    print(f"{name:12s} n={len(ins):4d} V={c['V']:3d} S={c['S']:3d} L={c['L']:3d} M={c['M']:2d} -> {br}")
    if name in show:
        for i in ins:
            print("      ", i[:100])
