#!/usr/bin/env python3
"""profiles/rNN/mlp_hbm_traffic.json from the passes of tools/collect_profiles.sh: HBM bytes per point of the MLP kernel
(no-warp path from the cfg2 passes, indexed path from the cfg3 passes) and the per-frame bytes of the cfg2 compositor
kernels.  FETCH_SIZE x 2 (gfx950: wide coalesced reads are tallied at half their size, MI355X_MICROARCH.md, HBM section),
WRITE_SIZE as is, both in KiB.   Usage: traffic_json.py <prof_dir> <commit> <frames> > mlp_hbm_traffic.json"""
import collections, csv, glob, json, os, sys

prof, commit, frames = sys.argv[1], sys.argv[2], int(sys.argv[3])


def pmc(d, counter):
    acc = collections.defaultdict(float)
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if row["Counter_Name"] == counter:
                acc[row["Kernel_Name"].split("(")[0]] += float(row["Counter_Value"])
    return acc


def line(wl):
    for l in open(os.path.join(prof, f"{wl}_trace.json")):
        if l.startswith("{"):
            return json.loads(l)
    raise SystemExit(f"no bench line in {wl}_trace.json")


sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench                                                    # (for the hash of the kernel sources bench.py compares with)

out = {"commit": commit, "kernel_sources_sha": bench.kernel_sources_sha(), "kernel_sources": list(bench.TRAFFIC_SOURCES),
       "command": "bash tools/collect_profiles.sh (rocprofv3 --pmc FETCH_SIZE --kernel-trace / --pmc WRITE_SIZE --kernel-trace, separate "
                  f"passes, on python3 bench.py --workload cfgN --no-extras --cpu-rays 0 --no-psnr --steps {frames - 1} --warmup 1: {frames} frames)",
       "fetch_correction": 2.0,
       "note": "FETCH_SIZE on gfx950 reports half of the bytes of wide coalesced reads (MI355X_MICROARCH.md section HBM): doubled; "
               "WRITE_SIZE as is; KiB"}
for wl, key in (("cfg2", "bytes_per_point"), ("cfg3", "bytes_per_point_indexed")):
    rd, wr = pmc(os.path.join(prof, wl + "_fetch"), "FETCH_SIZE"), pmc(os.path.join(prof, wl + "_write"), "WRITE_SIZE")
    r = line(wl)["roofline"]
    per_frame = r["points"] / (frames - 1)                      # the line counts the timed steps; the passes saw all frames
    k = max((k for k in rd if "mlp_kernel" in k), key=lambda k: rd[k] * 2 + wr.get(k, 0))     # (not the fp32 check launches)
    pts = per_frame * frames
    out[key] = (rd[k] * 2 + wr[k]) * 1024 / pts
    out[key + "_detail"] = {"kernel": k, "points": int(pts), "FETCH_SIZE_KB": rd[k], "WRITE_SIZE_KB": wr[k]}
    if wl == "cfg2":
        out["hbm_kernels_cfg2"] = {k: {"read_GB_per_frame": rd[k] * 2 * 1024 / frames / 1e9, "write_GB_per_frame": wr.get(k, 0) * 1024 / frames / 1e9}
                                   for k in rd if "composite" in k}
print(json.dumps(out, indent=1))
