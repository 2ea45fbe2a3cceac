"""profiles/traffic_paths.json and profiles/valu_paths.json from the rocprofv3 --pmc passes of scripts/gpu_round_profile.sh.

Every pass is its own run of `bench.py --steps 16 --warmup 8 --no-cpu-baseline`; the bench line of a pass gives the rays
per launch of exactly that run, so every constant is expressed PER RAY and compares with bench.py's bytes_per_ray whatever
the launch length.
  HBM bytes  = (2 x FETCH_SIZE + WRITE_SIZE) x 1024: the counters are in KiB, and on gfx950 FETCH_SIZE reports half the
               bytes of wide coalesced reads (MI355X_MICROARCH.md, section HBM); WRITE_SIZE is exact for 16-B stores.
  VALU       = SQ_INSTS_VALU (wave instructions) per ray; lane utilisation = SQ_THREAD_CYCLES_VALU / (64 x SQ_ACTIVE_INST_VALU);
               clock held = GRBM_GUI_ACTIVE / 8 / kernel wall time (the counter sums over the 8 XCDs).
usage: make_pmc_json.py <gpurun_out/tag> <kernel name as rocprofv3 prints it | auto> <commit> [suffix]
  auto   = the k_paths instantiation that ran in the pass (the one with the most VALU instructions)
  suffix = "" for the headline workload, _viking / _teapot / _soup for bench.py --scene ...: the files are
           profiles/valu_paths<suffix>.json and traffic_paths<suffix>.json, and they carry the bench line's
           "profile_workload" so that bench.py only prices the workload they were taken on."""
import csv
import glob
import hashlib
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from pmc_summary import load, short


KERNEL_SOURCES = ("pt_paths.hip", "pt_wave_traverse.h", "pt_device_common.h", "pt_trace.h")


def kernel_source_hash(root):
    """sha256 over the sources of k_paths: bench.py recomputes it and treats the constants of a profile taken on other
    sources as stale."""
    h = hashlib.sha256()
    for name in KERNEL_SOURCES:
        with open(os.path.join(root, "pathtracer_amd", "csrc", name), "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def line_of(path):
    for ln in open(path):
        if ln.startswith('{"metric"'):
            return json.loads(ln)
    raise SystemExit(f"no bench line in {path}")


def rays_per_launch(line):
    r = line.get("roofline_hbm") or line["roofline"]
    return float(r["rays_per_launch"]), float(r["launch_ms"])


def mean(v):
    return sum(v) / len(v)


def wall_ns(pass_dir, kernel):
    out = []
    for f in glob.glob(pass_dir + "/**/*counter_collection.csv", recursive=True):
        seen = set()
        for r in csv.DictReader(open(f)):
            if short(r["Kernel_Name"]) == kernel and r["Dispatch_Id"] not in seen:
                seen.add(r["Dispatch_Id"])
                out.append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
    return out


def main():
    tag_dir, kernel, commit = sys.argv[1], sys.argv[2], sys.argv[3]
    suffix = sys.argv[4] if len(sys.argv) > 4 else ""
    prof = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles")
    if kernel == "auto":
        cand = {k: mean(v["SQ_INSTS_VALU"]) * len(v["SQ_INSTS_VALU"]) for k, v in load(tag_dir + "/pmc_sq1").items()
                if k.startswith("k_paths") and "SQ_INSTS_VALU" in v}
        kernel = max(cand, key=cand.get)
    l0 = line_of(tag_dir + "/pmc_sq1.log")
    workload = l0.get("profile_workload")
    cmd = ("rocprofv3 --pmc <counters> -- python3 bench.py --steps %d --warmup %d --no-cpu-baseline%s (one run per counter set; mean "
           "over the launches of the kernel in the timed region and the warm-up)" % (l0["steps"], l0["warmup"], " " + l0.get("bench_args", "") if l0.get("bench_args") else ""))
    # ---- HBM traffic ----
    f = load(tag_dir + "/pmc_fetch")[kernel]["FETCH_SIZE"]
    w = load(tag_dir + "/pmc_write")[kernel]["WRITE_SIZE"]
    rf, _ = rays_per_launch(line_of(tag_dir + "/pmc_fetch.log"))
    rw, _ = rays_per_launch(line_of(tag_dir + "/pmc_write.log"))
    per_ray = 2.0 * mean(f) * 1024.0 / rf + mean(w) * 1024.0 / rw
    sha = kernel_source_hash(os.path.dirname(prof))
    d = {"kernel": kernel, "commit": commit, "kernel_sources_sha": sha, "workload": workload,
         "source": "profiles/%s_pmc_{fetch,write}_size.csv" % os.path.basename(tag_dir.rstrip("/")), "launches_sampled": len(f), "FETCH_SIZE_KiB_mean": mean(f), "WRITE_SIZE_KiB_mean": mean(w),
         "rays_per_launch": rf, "correction": "FETCH_SIZE x2 (gfx950 counts 128-B requests as 64 B), WRITE_SIZE x1",
         "hbm_read_bytes_per_ray": 2.0 * mean(f) * 1024.0 / rf, "hbm_write_bytes_per_ray": mean(w) * 1024.0 / rw,
         "hbm_bytes_per_ray": per_ray, "hbm_bytes_per_launch": per_ray * rf, "command": cmd}
    json.dump(d, open(os.path.join(prof, "traffic_paths%s.json" % suffix), "w"), indent=1)
    print("traffic:", {k: d[k] for k in ("hbm_bytes_per_ray", "rays_per_launch", "launches_sampled")})
    # ---- VALU issue ----
    a = load(tag_dir + "/pmc_sq1")[kernel]
    b = load(tag_dir + "/pmc_sq2")[kernel]
    ra, _ = rays_per_launch(line_of(tag_dir + "/pmc_sq1.log"))
    wall = wall_ns(tag_dir + "/pmc_sq2", kernel)
    gui = b.get("GRBM_GUI_ACTIVE")
    clock = mean(gui) / 8.0 / mean(wall) if gui and wall else None  # cycles per ns = GHz
    tcc = load(tag_dir + "/pmc_tcc").get(kernel) if os.path.isdir(tag_dir + "/pmc_tcc") else None
    v = {"kernel": kernel, "commit": commit, "kernel_sources_sha": sha, "workload": workload, "source": "profiles/%s_pmc_sq.log" % os.path.basename(tag_dir.rstrip("/")),
         "rays_per_launch": ra,
         "valu_wave_insts_per_ray": mean(a["SQ_INSTS_VALU"]) / ra,
         "salu_insts_per_ray": mean(a["SQ_INSTS_SALU"]) / ra,
         "lds_insts_per_64_rays": 64.0 * mean(a["SQ_INSTS_LDS"]) / ra,
         "vmem_rd_insts_per_64_rays": 64.0 * mean(a["SQ_INSTS_VMEM_RD"]) / ra,
         "vmem_wr_insts_per_64_rays": 64.0 * mean(a["SQ_INSTS_VMEM_WR"]) / ra,
         "valu_lane_util": round(mean(b["SQ_THREAD_CYCLES_VALU"]) / (64.0 * mean(b["SQ_ACTIVE_INST_VALU"])), 4)
         if "SQ_ACTIVE_INST_VALU" in b else None,
         "lds_bank_conflict_share": round(mean(b["SQ_LDS_BANK_CONFLICT"]) / mean(b["SQ_ACTIVE_INST_LDS"]), 4)
         if "SQ_ACTIVE_INST_LDS" in b else None,
         "wait_any_share": round(mean(b["SQ_WAIT_ANY"]) / mean(b["SQ_WAVE_CYCLES"]), 4) if "SQ_WAIT_ANY" in b else None,
         "wait_inst_any_share": round(mean(b["SQ_WAIT_INST_ANY"]) / mean(b["SQ_WAVE_CYCLES"]), 4) if "SQ_WAIT_INST_ANY" in b else None,
         "l2_hit_rate": round(mean(tcc["TCC_HIT_sum"]) / max(1.0, mean(tcc["TCC_HIT_sum"]) + mean(tcc["TCC_MISS_sum"])), 4) if tcc else None,
         "l2_requests_per_ray": round(mean(tcc["TCC_REQ_sum"]) / ra, 3) if tcc else None,
         "clock_ghz": round(clock, 4) if clock else None, "kernel_wall_ms_profiled": round(mean(wall) / 1e6, 4) if wall else None,
         "raw_pass1": {k: mean(x) for k, x in a.items()}, "raw_pass2": {k: mean(x) for k, x in b.items()}, "command": cmd}
    if tcc:
        v["raw_tcc"] = {k: mean(x) for k, x in tcc.items()}
    json.dump(v, open(os.path.join(prof, "valu_paths%s.json" % suffix), "w"), indent=1)
    print("valu:", {k: v[k] for k in ("kernel", "workload", "valu_wave_insts_per_ray", "valu_lane_util", "wait_any_share", "l2_hit_rate",
                                      "clock_ghz", "lds_bank_conflict_share")})


if __name__ == "__main__":
    main()
