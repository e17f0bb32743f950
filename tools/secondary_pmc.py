"""Counters of the dominant kernels of the secondary configurations (tools/r06_secondary_pmc.sh) -> markdown table + json.

    python tools/secondary_pmc.py gpurun_out/secpmc [profiles/r06_secondary_pmc]

Per case the four passes (FETCH_SIZE | WRITE_SIZE | SQ group 1 | SQ group 2) are reduced to per-launch averages of the kernel that
takes the most time in the case's kernel trace.  Derived columns:
  traffic      = 2 x FETCH_SIZE KB + WRITE_SIZE KB (gfx950 correction of MI355X_MICROARCH.md: FETCH_SIZE reports half of 16-B/lane reads)
  HBM rate     = traffic / kernel duration (kernel trace of the FETCH pass)
  VALU issue   = SQ_INSTS_VALU x 2 cycles / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs): every VALU instruction at the 2-cycle wave64 issue cost of a
                 SIMD-32 -- a LOWER bound (a packed v_pk_* costs 4: tools/probes/valu_probe.hip); "quad" = the same with 4 cycles each, what
                 SQ_ACTIVE_INST_VALU's quad-cycle unit suggests: the upper bound (exact for kernels made of packed instructions)
  issue stall  = SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES;  parked = SQ_WAIT_ANY / SQ_WAVE_CYCLES (s_waitcnt / barrier)
  LDS conflict = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE
"""
import csv, glob, json, os, sys, collections

src = sys.argv[1]
dst = sys.argv[2] if len(sys.argv) > 2 else None
# case -> (label, algorithmic bytes of the dominant kernel's launch or None, kernel-name filter or None = busiest kernel)
CASES = collections.OrderedDict([
    ("c3", ("2-D dwt db4 8192^2 f32 (headline)", 2 * 8192 * 8192 * 4, "k_fwd2d_pair")),
    ("c2", ("C2: 1-D dwt db4 2^24 f32", 2 * (1 << 24) * 4, "k_fwd1d_multi<float, 8, 1>")),
    ("c4", ("C4: 1-D dwt cdf9/7 lifting 2^24 f32", 2 * (1 << 24) * 4, "k_lift1d_fwd3<float, 0, 1>")),
    ("dwt3d", ("3-D dwt db4 512^3 f32 (level 1: the one-pass kernel)", 2 * 512 ** 3 * 4, "k_fwd3d_one<float, 4, 8, 2>")),
    ("idwt3d", ("3-D idwt db4 512^3 f32 (level 1: the one-pass inverse kernel)", 2 * 512 ** 3 * 4, "k_inv3d_one<float, 4, 8, 2>")),
    ("lift2d", ("2-D dwt cdf9/7 lifting 8192^2 f32", 2 * 8192 * 8192 * 4, None)),
    ("lift2d_inv", ("2-D idwt cdf9/7 lifting 8192^2 f32", 2 * 8192 * 8192 * 4, None)),
    ("sym8_fwd", ("2-D dwt sym8 (16 taps) 8192^2 f32", 2 * 8192 * 8192 * 4, None)),
    ("sym8_inv", ("2-D idwt sym8 (16 taps) 8192^2 f32", 2 * 8192 * 8192 * 4, None)),
    ("modwt", ("1-D modwt db4 2^24 f32, 8 levels", None, None)),
    ("batt6", ("2-D dwt batt6 (59 taps) 8192^2 f32", None, None)),
])


def trace_rows(d):
    hits = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
    return [r for h in hits for r in csv.DictReader(open(h)) if ("wl::" in r["Kernel_Name"] or "k_modwt" in r["Kernel_Name"] or "anonymous namespace" in r["Kernel_Name"])]


def counter_rows(d):
    hits = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    return [r for h in hits for r in csv.DictReader(open(h)) if ("wl::" in r["Kernel_Name"] or "k_modwt" in r["Kernel_Name"] or "anonymous namespace" in r["Kernel_Name"])]


def short(name):
    return name.replace("void wl::", "").replace("void (anonymous namespace)::", "").split("(")[0]


out = []
for case, (label, alg, filt) in CASES.items():
    tr = trace_rows(os.path.join(src, f"{case}_g1"))
    if not tr:
        continue
    tot = collections.Counter()
    cnt = collections.Counter()
    for r in tr:
        tot[r["Kernel_Name"]] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        cnt[r["Kernel_Name"]] += 1
    if filt:
        cand = [k for k in tot if filt in k]
        if not cand:
            continue
        kern = max(cand, key=lambda k: tot[k])
    else:
        kern = max(tot, key=lambda k: tot[k])
    share = tot[kern] / sum(tot.values())
    dur_us = tot[kern] / cnt[kern] / 1e3
    c = {}
    for g in (1, 2, 3, 4):
        acc = collections.defaultdict(list)
        for r in counter_rows(os.path.join(src, f"{case}_g{g}")):
            if r["Kernel_Name"] == kern:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, v in acc.items():
            c[k] = sum(v) / len(v)
    row = {"case": case, "workload": label, "kernel": short(kern), "share_of_case_kernel_time": round(share, 3), "launch_us_under_pmc": round(dur_us, 1),
           "launches_per_pass": cnt[kern], "counters": {k: round(v, 1) for k, v in sorted(c.items())}}
    if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
        traffic = (2 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024
        row["traffic_bytes"] = int(traffic)
        row["hbm_TBps_under_pmc"] = round(traffic / (dur_us * 1e-6) / 1e12, 2)
        if alg:
            row["algorithmic_bytes"] = alg
            row["traffic_over_algorithmic"] = round(traffic / alg, 3)
    if "SQ_INSTS_VALU" in c and c.get("GRBM_GUI_ACTIVE"):
        row["valu_issue_2cyc"] = round(c["SQ_INSTS_VALU"] / (64.0 * c["GRBM_GUI_ACTIVE"]), 3)
        row["valu_busy"] = round(c["SQ_INSTS_VALU"] / (32.0 * c["GRBM_GUI_ACTIVE"]), 3)
    if c.get("SQ_WAVE_CYCLES"):
        if "SQ_WAIT_INST_ANY" in c:
            row["issue_stall"] = round(c["SQ_WAIT_INST_ANY"] / c["SQ_WAVE_CYCLES"], 3)
    if c.get("SQ_LDS_IDX_ACTIVE"):
        row["lds_conflict"] = round(c.get("SQ_LDS_BANK_CONFLICT", 0.0) / c["SQ_LDS_IDX_ACTIVE"], 3)
    # group 4 has its own SQ_WAVE_CYCLES-free ratios: parked waves relative to active instruction cycles
    if "SQ_WAIT_ANY" in c and "SQ_ACTIVE_INST_ANY" in c and c["SQ_ACTIVE_INST_ANY"]:
        row["parked_over_active"] = round(c["SQ_WAIT_ANY"] / c["SQ_ACTIVE_INST_ANY"], 2)
    # the bound the counters point at
    vb, hb = row.get("valu_busy", 0.0), row.get("hbm_TBps_under_pmc", 0.0)
    lo = row.get("valu_issue_2cyc", 0.0)
    row["bound"] = "hbm" if hb >= 4.5 else ("valu" if lo >= 0.45 else ("valu / latency" if vb >= 0.6 else "latency"))
    out.append(row)

lines = ["| case | dominant kernel | share | launch µs (PMC) | traffic MB | × algorithmic | HBM TB/s | VALU issue (2-cycle … quad) | issue stall | parked / active | LDS conflict | reads as |",
         "|---|---|---|---|---|---|---|---|---|---|---|---|"]
for r in out:
    lines.append("| {} | `{}` | {:.0%} | {} | {} | {} | {} | {} | {} | {} | {} | {} |".format(
        r["workload"], r["kernel"][:44], r["share_of_case_kernel_time"], r["launch_us_under_pmc"],
        round(r["traffic_bytes"] / 1e6, 1) if "traffic_bytes" in r else "–", r.get("traffic_over_algorithmic", "–"), r.get("hbm_TBps_under_pmc", "–"),
        "{} … {}".format(r.get("valu_issue_2cyc", "–"), r.get("valu_busy", "–")), r.get("issue_stall", "–"), r.get("parked_over_active", "–"), r.get("lds_conflict", "–"), r["bound"]))
text = "\n".join(lines)
print(text)
if dst:
    json.dump(out, open(dst + ".json", "w"), indent=1)
    head = ("# r06 -- counters of the dominant kernel of the secondary configurations (tools/r06_secondary_pmc.sh)\n\n"
            "Four rocprofv3 passes per case, `--kernel-trace` only: FETCH_SIZE | WRITE_SIZE | SQ group 1 | SQ group 2.  Per-launch averages of the kernel\n"
            "with the largest share of the case's kernel time (for 2-D / 3-D transforms the average mixes the levels that kernel serves: level 1\n"
            "of the 2-D cdf9/7 transform alone is 116 us, level 2 41 us).  traffic = 2 x FETCH_SIZE + WRITE_SIZE (KB -> bytes; gfx950 correction of\n"
            "MI355X_MICROARCH.md); HBM TB/s = traffic / launch time under the profiler; VALU issue = SQ_INSTS_VALU x 2 cycles (lower bound: every\n"
            "instruction at the wave64 issue cost of a SIMD-32) ... x 4 (upper bound: exact for kernels made of packed v_pk_* instructions) over\n"
            "1024 SIMDs x GRBM_GUI_ACTIVE / 8; issue stall = SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES; parked = SQ_WAIT_ANY / SQ_ACTIVE_INST_ANY (s_waitcnt,\n"
            "barriers); LDS conflict = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE.  `reads as`: hbm from 4.5 TB/s, valu from 0.45 of the lower\n"
            "bound, else latency.  Raw per-launch counter averages: r06_secondary_pmc.json.\n\n")
    open(dst + "_table.md", "w").write(head + text + "\n")
