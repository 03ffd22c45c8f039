#!/usr/bin/env python3
"""gpurun_out/prof/<prec>/ (tools/profile_bench.sh) -> profiles/<tag>_kernel_stats_<prec>.csv, <tag>_pmc_<prec>.csv,
<tag>_bench_<prec>.json and profiles/pmc_traffic.json (per-launch HBM bytes read by bench.py).

HBM bytes per launch = (2 * FETCH_SIZE + WRITE_SIZE) KB * 1024, per-launch MEDIAN (the first pass of a run sizes the buffers: its
tile kernel stops early and would pull a mean down).  MI355X_MICROARCH.md (HBM section): on gfx950 FETCH_SIZE reports half of the
bytes of a wide coalesced streaming read and "other access widths and WRITE_SIZE are uncalibrated: calibrate on a known byte count in
your own access pattern".  tools/hbm_calib.hip does (profiles/r4/hbm_counter_calibration.txt): FETCH_SIZE is one half for 16-byte
streaming loads, for 32-byte records read as two 16-byte loads per lane (the pile records) and for lone 8-byte gathers (64 B reported
per 128-byte line touched); WRITE_SIZE is exact for streaming 16-byte stores and counts 32 B for a lone 16-byte record.  Both count
traffic between the L2s and the fabric: Infinity-Cache hits are included, so they bound HBM traffic from above."""
import collections
import csv
import glob
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def short(name):
    if "rocprim" in name:
        return "rocprim_radix_sort"
    if "k_prep<" in name or "k_prepIL" in name:
        return "k_prep_write" if ("k_prep<true>" in name or "k_prepILb1" in name) else "k_prep_count"
    m = re.search(r"k_lstm2_mx|k_lstm2_w8|k_lstm1_rs|k_lstm|k_[a-z0-9_]+", name)
    if not m:
        return name[:40]
    k = m.group(0)
    if k == "k_lstm1_rs":
        return "k_lstm1"
    if k in ("k_lstm2_w8", "k_lstm2_mx"):
        return "k_lstm2"
    if k == "k_lstm":
        return "k_lstm2" if ("ILi256E" in name or "Li160E" in name or re.search(r"k_lstm<256,", name)) else "k_lstm1"
    return k


def counters(d):
    per = collections.defaultdict(lambda: collections.defaultdict(list))
    # (gpurun merges every call's files into gpurun_out/: only the newest collection of a sub-directory belongs to this round's build)
    files = sorted(glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True), key=os.path.getmtime)
    for f in files[-1:]:
        for r in csv.DictReader(open(f)):
            per[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return per


def main():
    prec = sys.argv[1] if len(sys.argv) > 1 else "f16x3"
    tag = sys.argv[2] if len(sys.argv) > 2 else "r1_end"
    src = os.path.join(ROOT, "gpurun_out", "prof", prec)
    out = os.path.join(ROOT, "profiles")
    # which build the run was: the commit the profiles were taken at (argv[3]; default: HEAD of this checkout — the box runs a snapshot of the tree)
    commit = sys.argv[3] if len(sys.argv) > 3 else os.popen("git -C %s rev-parse --short HEAD 2>/dev/null" % ROOT).read().strip() or "unknown"
    dirty = os.popen("git -C %s status --porcelain -- clair3_rna_amd bench.py 2>/dev/null" % ROOT).read().strip()
    stamp = "# build: commit %s%s\n" % (commit, " + uncommitted changes" if dirty else "")
    stamp += ("# C3R_DEEP_SERIAL=1: k_fused_deep BEHIND k_fused_tiles, so that every traced duration is the kernel's own (in production the two run side by side on two\n"
              "# streams and a trace shows the deep kernel with the duration of the kernel it waits beside)\n")
    cal = ("# (the --stats averages of the network kernels include ONE 2,048-window calibration launch each, made by c3r_load_weights' precision guard before the passes:\n"
           "#  read MaxNs for a full launch, or %s_last_pass_%s.csv, which holds one pass without it)\n" % (tag, prec))
    # 1. kernel stats
    st = sorted(glob.glob(os.path.join(src, "stats", "**", "*kernel_stats.csv"), recursive=True), key=os.path.getmtime)
    if st:
        text = open(st[-1]).read()
        open(os.path.join(out, "%s_kernel_stats_%s.csv" % (tag, prec)), "w").write(
            "# rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --steps 2 --warmup 1 --no_cpu_baseline --no_profile "
            "--no_fast --no_f32 --no_resident --no_overlap --no_strong --precision %s   (MI355X; durations in ns; one context: 1 priming + 1 warm-up + 2 timed passes, "
            "c3r_load_reads inside every pass)\n" % prec + stamp + cal + text)
    sx = sorted(glob.glob(os.path.join(src, "stats_extra", "**", "*kernel_stats.csv"), recursive=True), key=os.path.getmtime)
    if sx:
        open(os.path.join(out, "%s_kernel_stats_extra_%s.csv" % (tag, prec)), "w").write(
            "# rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --steps 2 --warmup 1 --no_cpu_baseline --no_profile "
            "--no_fast --no_f32 --no_resident --no_overlap --no_strong --precision %s   (MI355X; ns): the chr20 passes AND the additional configurations "
            "(phased_1gpu: the <30> instantiations; stress_500x; depth_cap_20000x; realistic_expr), one context each\n" % prec + stamp + cal + open(sx[-1]).read())
    # 1b. the LAST pass of the same trace, kernel by kernel (the --stats averages above include the first pass, which sizes the buffers
    # with a tile kernel that stops early and then repeats it): a pass starts at its k_prep<false> launch
    tr = sorted(glob.glob(os.path.join(src, "stats", "**", "*kernel_trace.csv"), recursive=True), key=os.path.getmtime)
    if tr:
        ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"])) for r in csv.DictReader(open(tr[-1])))
        starts = [i for i, e in enumerate(ev) if e[2] == "k_prep_count"]
        if starts:
            last = ev[starts[-1]:]
            per = collections.OrderedDict()
            for a, b, k in last:
                n, t = per.get(k, (0, 0))
                per[k] = (n + 1, t + b - a)
            net = ("k_lstm1", "k_lstm2", "k_heads_mfma", "k_heads", "k_fc4")
            with open(os.path.join(out, "%s_last_pass_%s.csv" % (tag, prec)), "w") as f:
                f.write("# the last timed pass of the trace behind %s_kernel_stats_%s.csv: every launch from its k_prep<false> on (ns)\n" % (tag, prec) + stamp)
                f.write("kernel,launches,total_ns\n")
                for k, (n, t) in per.items():
                    f.write("%s,%d,%d\n" % (k, n, t))
                tb = [(n, t) for k, (n, t) in per.items() if k not in net and k.startswith("k_")]
                f.write("# tensor build (all non-network kernels): %d launches, %.3f ms; network: %.3f ms\n"
                        % (sum(n for n, _ in tb), sum(t for _, t in tb) / 1e6, sum(t for k, (n, t) in per.items() if k in net) / 1e6))
    # 2. PMC
    rows = {}
    for sub in ("pmc_fetch", "pmc_write", "pmc_sq"):
        for k, cs in counters(os.path.join(src, sub)).items():
            for c, v in cs.items():
                rows.setdefault(k, {})[c] = (sorted(v)[len(v) // 2], len(v))
    hdr = ["FETCH_SIZE", "WRITE_SIZE", "SQ_VALU_MFMA_BUSY_CYCLES", "GRBM_GUI_ACTIVE", "SQ_WAVE_CYCLES", "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY", "SQ_ACTIVE_INST_ANY"]
    traffic = {}
    with open(os.path.join(out, "%s_pmc_%s.csv" % (tag, prec)), "w") as f:
        f.write(stamp + "# rocprofv3 --pmc <set> --kernel-trace, separate passes (FETCH_SIZE | WRITE_SIZE | SQ_*), same bench command; per-launch medians.\n"
                "# hbm_bytes = (2 * FETCH_SIZE + WRITE_SIZE) KB * 1024: FETCH_SIZE reports half of the bytes fetched on gfx950 (MI355X_MICROARCH.md; calibrated on this path's own access patterns by tools/hbm_calib.hip, profiles/r4/hbm_counter_calibration.txt); mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / "
                "(1024 SIMDs * GRBM_GUI_ACTIVE / 8 XCDs); clock_GHz needs the kernel duration and is quoted in DESIGN.md.\n")
        f.write("kernel,launches," + ",".join(hdr) + ",hbm_bytes_per_launch,mfma_busy\n")
        for k in sorted(rows):
            r = rows[k]
            vals = [r.get(h, (0.0, 0))[0] for h in hdr]
            n = max(v[1] for v in r.values())
            ff = 2.0
            hbm = (ff * vals[0] + vals[1]) * 1024
            busy = vals[2] / (1024 * vals[3] / 8) if vals[3] else 0.0
            traffic[k] = int(hbm)
            f.write("%s,%d,%s,%d,%.3f\n" % (k, n, ",".join("%.4g" % v for v in vals), hbm, busy))
    tf = os.path.join(out, "pmc_traffic.json")
    allt = json.load(open(tf)) if os.path.exists(tf) else {}
    allt[prec] = traffic
    json.dump(allt, open(tf, "w"), indent=1, sort_keys=True)
    b = os.path.join(src, "bench.json")
    if os.path.exists(b) and os.path.getsize(b):
        open(os.path.join(out, "%s_bench_%s.json" % (tag, prec)), "w").write(open(b).read())
    print("wrote", tag, prec, sorted(rows))


if __name__ == "__main__":
    main()
