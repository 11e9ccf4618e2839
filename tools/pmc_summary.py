#!/usr/bin/env python3
"""Per-kernel digest of tools/pmc_passes.sh: for every kernel (by short name) of the traced command the launch count, the
average duration, registers / LDS / scratch from the kernel trace, and the counters of the four --pmc passes averaged over
its launches, with a few derived ratios:
  valu_busy = SQ_ACTIVE_INST_VALU / SQ_BUSY_CYCLES / (4 SIMDs) ... reported as the fraction of wave-cycles instead, which
  needs no per-chip constants:  wait_frac = SQ_WAIT_ANY / SQ_WAVE_CYCLES (waves parked on s_waitcnt / barrier),
  issue_stall_frac = SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES, active_frac = SQ_ACTIVE_INST_ANY / SQ_WAVE_CYCLES,
  valu / vmem / lds_frac = SQ_ACTIVE_INST_x / SQ_WAVE_CYCLES, occupancy_waves = SQ_LEVEL_WAVES / SQ_BUSY_CYCLES-ish is
  not portable -> waves_per_launch and the trace's resource numbers are given instead;
  l2_hit = TCC_HIT / (TCC_HIT + TCC_MISS);  lds_conflict = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE.
usage: pmc_summary.py <gpurun_out/pmc_tag> <tag>"""
import collections
import csv
import glob
import json
import os
import sys

out, tag = sys.argv[1], sys.argv[2]


def short(k):
    k = k.replace("void ", "")
    return k.split("(")[0]


def find(d, pat):
    f = glob.glob(os.path.join(out, d, "**", pat), recursive=True)
    return f[0] if f else None


kern = collections.OrderedDict()
tr = find("trace", "*kernel_trace.csv")
if tr:
    for r in csv.DictReader(open(tr)):
        k = short(r["Kernel_Name"])
        e = kern.setdefault(k, {"launches": 0, "ns": 0.0})
        e["launches"] += 1
        e["ns"] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        for src, dst in (("VGPR_Count", "vgpr"), ("Accum_VGPR_Count", "agpr"), ("SGPR_Count", "sgpr"), ("LDS_Block_Size", "lds_bytes"),
                         ("Scratch_Size", "scratch_bytes"), ("Workgroup_Size", "wg_size"), ("Grid_Size", "grid")):
            if src in r and r[src] not in ("", None):
                e[dst] = max(e.get(dst, 0), int(float(r[src])))
for name in ("sq1", "sq2", "sq3", "mem"):
    f = find(name, "*counter_collection.csv")
    if not f:
        continue
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        acc[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, cs in acc.items():
        e = kern.setdefault(k, {"launches": 0, "ns": 0.0})
        for c, v in cs.items():
            e[c] = sum(v) / len(v)
rows = []
for k, e in kern.items():
    if e.get("launches"):
        e["avg_ms"] = e["ns"] / e["launches"] / 1e6
        e["total_ms"] = e["ns"] / 1e6
    wc = e.get("SQ_WAVE_CYCLES")
    if wc:
        for key, c in (("wait_frac", "SQ_WAIT_ANY"), ("issue_stall_frac", "SQ_WAIT_INST_ANY"), ("active_frac", "SQ_ACTIVE_INST_ANY"),
                       ("valu_frac", "SQ_ACTIVE_INST_VALU"), ("vmem_frac", "SQ_ACTIVE_INST_VMEM"), ("lds_frac", "SQ_ACTIVE_INST_LDS")):
            if c in e:
                e[key] = e[c] / wc
    if e.get("SQ_BUSY_CYCLES") and wc:
        e["waves_per_busy_cycle"] = wc / e["SQ_BUSY_CYCLES"]     # mean resident waves per busy SQ (per XCD-slice unit; relative between kernels)
    if e.get("TCC_HIT_sum") is not None and (e.get("TCC_HIT_sum", 0) + e.get("TCC_MISS_sum", 0)) > 0:
        e["l2_hit"] = e["TCC_HIT_sum"] / (e["TCC_HIT_sum"] + e["TCC_MISS_sum"])
    if e.get("SQ_LDS_IDX_ACTIVE"):
        e["lds_conflict"] = e.get("SQ_LDS_BANK_CONFLICT", 0.0) / e["SQ_LDS_IDX_ACTIVE"]
    if e.get("SQ_THREAD_CYCLES_VALU") and e.get("SQ_ACTIVE_INST_VALU"):
        e["valu_lane_util"] = e["SQ_THREAD_CYCLES_VALU"] / (64.0 * e["SQ_ACTIVE_INST_VALU"])    # active lanes per VALU cycle / 64 (divergence)
    if wc:
        for key, c in (("ta_addr_fifo_full_frac", "SQ_VMEM_TA_ADDR_FIFO_FULL"), ("ta_cmd_fifo_full_frac", "SQ_VMEM_TA_CMD_FIFO_FULL"),
                       ("ta_wrdata_fifo_full_frac", "SQ_VMEM_WR_TA_DATA_FIFO_FULL")):
            if c in e:
                e[key] = e[c] / wc
    if e.get("SQ_WAVES") and e.get("SQ_INSTS_VALU") is not None:
        e["valu_insts_per_wave"] = e["SQ_INSTS_VALU"] / e["SQ_WAVES"]
        e["vmem_rd_per_wave"] = e.get("SQ_INSTS_VMEM_RD", 0.0) / e["SQ_WAVES"]
        e["vmem_wr_per_wave"] = e.get("SQ_INSTS_VMEM_WR", 0.0) / e["SQ_WAVES"]
        e["lds_insts_per_wave"] = e.get("SQ_INSTS_LDS", 0.0) / e["SQ_WAVES"]
    rows.append((e.get("total_ms", 0.0), k))
rows.sort(reverse=True)
json.dump({"tag": tag, "kernels": kern}, open(os.path.join(out, "summary.json"), "w"), indent=1)
cols = ["launches", "avg_ms", "vgpr", "lds_bytes", "scratch_bytes", "wg_size", "wait_frac", "issue_stall_frac", "active_frac", "valu_frac", "vmem_frac",
        "lds_frac", "waves_per_busy_cycle", "valu_lane_util", "ta_addr_fifo_full_frac", "ta_wrdata_fifo_full_frac", "l2_hit", "lds_conflict", "valu_insts_per_wave", "vmem_rd_per_wave", "vmem_wr_per_wave", "lds_insts_per_wave"]
with open(os.path.join(out, "summary.md"), "w") as f:
    f.write("# %s: counters per kernel (tools/pmc_passes.sh; fractions are of SQ_WAVE_CYCLES)\n\n" % tag)
    f.write("| kernel | " + " | ".join(cols) + " |\n|---|" + "---|" * len(cols) + "\n")
    for _, k in rows[:24]:
        e = kern[k]
        f.write("| %s | " % k[:70] + " | ".join(("%.3g" % e[c]) if isinstance(e.get(c), float) else str(e.get(c, "")) for c in cols) + " |\n")
print(open(os.path.join(out, "summary.md")).read())
