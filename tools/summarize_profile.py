#!/usr/bin/env python3
"""Condense the rocprofv3 CSVs written by tools/profile_bench.sh into profiles/<tag>_*.{csv,json}.

HBM traffic follows MI355X_MICROARCH.md "HBM": FETCH_SIZE and WRITE_SIZE collected in separate --pmc
passes; on gfx950 FETCH_SIZE reads 1/2 of a coalesced stream, so it is calibrated on the known-byte
k_pack launch of the same run (bench.py --calib: 1 GiB f64 + 0.5 GiB int32 read, 1 GiB written, 8 B/lane)
and the resulting factor is applied to the apply kernel.  WRITE_SIZE is checked the same way.

usage: summarize_profile.py <gpurun_out/prof dir> <tag> [traffic json to write for bench.py]"""
import collections
import csv
import json
import os
import sys

src = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/prof"
tag = sys.argv[2] if len(sys.argv) > 2 else "r01"
out_traffic = sys.argv[3] if len(sys.argv) > 3 else None
os.makedirs("profiles", exist_ok=True)


def rows(name):
    return list(csv.DictReader(open(os.path.join(src, name, name + "_counter_collection.csv"))))


def short(k):
    return k.split("(")[0].replace("void ", "")


def gsize(r):
    return int(r["Grid_Size"]) if "Grid_Size" in r else int(r["Grid_Size_X"])


# 1. kernel stats (top kernels) + per-dispatch durations of the apply kernel (full launches = largest grid)
stats = list(csv.DictReader(open(os.path.join(src, "stats", "stats_kernel_stats.csv"))))
with open("profiles/%s_kernel_stats.csv" % tag, "w") as f:
    w = csv.writer(f)
    w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
    for r in stats[:25]:
        w.writerow([short(r["Name"]), r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"], r["MinNs"], r["MaxNs"]])
trace = list(csv.DictReader(open(os.path.join(src, "stats", "stats_kernel_trace.csv"))))
ap = [r for r in trace if "k_apply3" in r["Kernel_Name"]]
# One Regrid = one launch -- except the staged level-fast kernel (round 6), launched once per CLASS of tile-list length: the
# instantiations k_apply3_lfu<..., NPF> of one Regrid are the "parts" of a launch; per part the full-size dispatches are those
# with its largest grid (the bench's one-field self-checks launch smaller ones), and a launch is the SUM of its parts.
parts = collections.defaultdict(list)
for r in ap:
    parts[short(r["Kernel_Name"])].append(r)
by_time = {k: sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in v) for k, v in parts.items()}
lead = max(by_time, key=by_time.get) if by_time else None
family = lead.split("<")[0] if lead else None
multi = bool(lead) and "k_apply3_lfu" in lead
prefix = lead.rsplit(",", 1)[0] if lead else None     # the same element types and flags, any NPF (the bench's float64 self-check launches another family)
names = sorted(k for k in parts if (multi and k.rsplit(",", 1)[0] == prefix) or k == lead)
dur_avg = dur_min = 0.0
nfull = 0
for k in names:
    g = max(gsize(r) for r in parts[k])
    d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in parts[k] if gsize(r) == g]
    dur_avg += sum(d) / len(d)
    dur_min += min(d)
    nfull = max(nfull, len(d))
res = {"kernel": lead if not multi else "%s: %d class launches per Regrid (%s)" % (family, len(names), ", ".join(n.split(",")[-1].strip(" >") for n in names)),
       "full_launches": nfull, "kernel_ms_avg_rocprof": dur_avg if names else None, "kernel_ms_min_rocprof": dur_min if names else None}


# 2. PMC passes
def per_kernel(name, counter):
    d = collections.defaultdict(list)
    for r in rows(name):
        if r["Counter_Name"] == counter:
            d[(short(r["Kernel_Name"]), int(r["Grid_Size"]))].append(float(r["Counter_Value"]))
    return d


def pick(d, key_sub):
    ks = [k for k in d if key_sub in k[0]]
    if not ks:
        return None
    if key_sub == "k_apply3" and names:       # the Regrid's launch = the sum of its parts (see above), each at its largest grid
        tot = 0.0
        for n in names:
            kk = [k for k in ks if k[0] == n]
            if not kk:
                return None
            v = d[max(kk, key=lambda k: k[1])]
            tot += sum(v) / len(v)
        return tot
    v = d[max(ks, key=lambda k: k[1])]
    return sum(v) / len(v)


fetch, write = per_kernel("fetch", "FETCH_SIZE"), per_kernel("write", "WRITE_SIZE")
cal_f, cal_w = pick(fetch, "k_pack"), pick(write, "k_pack")
known_r, known_w = (2 ** 30 + 2 ** 29) / 1024.0, 2 ** 30 / 1024.0  # KiB
ff = known_r / cal_f if cal_f else None
fw = known_w / cal_w if cal_w else None
ap_f, ap_w = pick(fetch, "k_apply3"), pick(write, "k_apply3")
res.update({"calibration": {"k_pack_FETCH_SIZE_KiB": cal_f, "k_pack_WRITE_SIZE_KiB": cal_w, "known_read_KiB": known_r,
                            "known_write_KiB": known_w, "fetch_factor": ff, "write_factor": fw},
            "apply_FETCH_SIZE_KiB_raw": ap_f, "apply_WRITE_SIZE_KiB_raw": ap_w})
if ap_f and ap_w and ff and fw:
    rb, wb = ap_f * 1024 * ff, ap_w * 1024 * fw
    res.update({"hbm_read_bytes_per_launch": rb, "hbm_write_bytes_per_launch": wb, "hbm_bytes_per_launch": rb + wb})
try:
    tcc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in rows("tcc"):
        tcc[(short(r["Kernel_Name"]), int(r["Grid_Size"]))][r["Counter_Name"]].append(float(r["Counter_Value"]))
    t = collections.defaultdict(float)
    for n in names:
        k = max([k for k in tcc if k[0] == n], key=lambda k: k[1])
        for c, v in tcc[k].items():
            t[c] += sum(v) / len(v)
    t = dict(t)
    t["l2_hit_rate"] = t["TCC_HIT_sum"] / (t["TCC_HIT_sum"] + t["TCC_MISS_sum"])
    res["tcc"] = t
except Exception:  # noqa
    res["tcc"] = None
json.dump(res, open("profiles/%s_pmc_summary.json" % tag, "w"), indent=1)
print(json.dumps(res, indent=1))
if out_traffic and "hbm_bytes_per_launch" in res:
    json.dump({"hbm_bytes_per_launch": res["hbm_bytes_per_launch"], "source": "profiles/%s_pmc_summary.json" % tag},
              open(out_traffic, "w"))
