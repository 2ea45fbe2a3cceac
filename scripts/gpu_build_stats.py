"""Kernel-level summary of a rocprofv3 --kernel-trace --stats run of scripts/gpu_build_only.py."""
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/**/*kernel_stats.csv', recursive=True)[0]
for r in csv.DictReader(open(f)):
    print(f"{r['Name'][:60]:60s} calls {r['Calls']:>3s} total {float(r['TotalDurationNs'])/1e6:8.3f} ms avg {float(r['AverageNs'])/1e6:8.3f}")
