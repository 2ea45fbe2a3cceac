"""Per-basic-block instruction counts of one kernel in a hipcc -S listing.  usage: isa_blocks.py file.s <mangled-substring> [min]"""
import re
import sys

path, key = sys.argv[1], sys.argv[2]
minimum = int(sys.argv[3]) if len(sys.argv) > 3 else 20
lines = open(path).read().split("\n")
start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\S*" + re.escape(key) + r"\S*:", l))
end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
blocks, cur = [], ["entry", 0, 0, 0, 0, 0, start]
blocks.append(cur)
for n in range(start + 1, end + 1):
    t = lines[n].strip()
    if re.match(r"^\.LBB\d+_\d+:", t):
        cur = [t.split(":")[0], 0, 0, 0, 0, 0, n + 1]
        blocks.append(cur)
        continue
    if not t or t.startswith(";") or t.startswith("."):
        continue
    op = t.split()[0]
    k = 1 if op.startswith("v_") else 2 if op.startswith("s_") else 3 if op.startswith("ds_") else 4 if op.split("_")[0] in ("global", "buffer", "flat", "scratch") else 5
    cur[k] += 1
print("kernel lines", start + 1, "-", end + 1, " totals valu/salu/ds/vmem/other:", [sum(b[i] for b in blocks) for i in range(1, 6)])
for b in blocks:
    if sum(b[1:5]) >= minimum:
        print(f"{b[0]:12s} line {b[6]:5d}  valu {b[1]:4d} salu {b[2]:4d} ds {b[3]:3d} vmem {b[4]:3d}")
