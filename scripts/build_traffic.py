"""Per-kernel HBM bytes of the BVH build from the passes of scripts/gpu_build_pmc.sh: SUM over all dispatches of a kernel in ONE
build (the run makes REPS builds: totals are divided by it), next to the kernel's time per build and the 364 B/triangle budget
of SURVEY.md 8(d).  HBM bytes = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 (MI355X_MICROARCH.md: KiB units, FETCH_SIZE counts
128-byte requests as 64 on gfx950).   usage: build_traffic.py <gpurun_out/tag> <N>"""
import collections, csv, glob, os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from pmc_summary import short

tag, n = sys.argv[1], float(sys.argv[2])
reps = int(os.environ.get("REPS", "2"))


def sums(path, counter):
    acc = collections.defaultdict(float)
    for f in glob.glob(path + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                acc[short(r["Kernel_Name"]).split("<")[0]] += float(r["Counter_Value"])
    return acc


fetch, write = sums(tag + "/pmc_fetch", "FETCH_SIZE"), sums(tag + "/pmc_write", "WRITE_SIZE")
times = collections.defaultdict(float)
for f in glob.glob(tag + "/stats/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        times[short(r["Name"]).split("<")[0]] += float(r["TotalDurationNs"])
rows = []
for k in sorted(set(fetch) | set(write), key=lambda k: -(2 * fetch[k] + write[k])):
    rd, wr = 2.0 * fetch[k] * 1024 / reps, write[k] * 1024 / reps
    ms = times.get(k, 0.0) / reps / 1e6
    rows.append((k, rd, wr, ms))
tot_r, tot_w, tot_ms = sum(r[1] for r in rows), sum(r[2] for r in rows), sum(r[3] for r in rows)
print(f"BVH build of {n:.0f} triangles, per build (mean of {reps}); HBM bytes = (2 x FETCH_SIZE + WRITE_SIZE) x 1024")
print(f"{'kernel':28s} {'read GB':>9s} {'write GB':>9s} {'B/tri':>8s} {'ms':>8s} {'TB/s':>6s}")
for k, rd, wr, ms in rows:
    if rd + wr < 1e6:
        continue
    print(f"{k:28s} {rd/1e9:9.2f} {wr/1e9:9.2f} {(rd+wr)/n:8.1f} {ms:8.3f} {((rd+wr)/ms/1e9 if ms else 0):6.2f}")
print(f"{'total':28s} {tot_r/1e9:9.2f} {tot_w/1e9:9.2f} {(tot_r+tot_w)/n:8.1f} {tot_ms:8.3f} {(tot_r+tot_w)/tot_ms/1e9:6.2f}")
print(f"algorithmic budget (SURVEY.md 8(d)): 364 B/triangle = {364*n/1e9:.1f} GB; measured / budget = {(tot_r+tot_w)/(364*n):.2f}")
