#!/usr/bin/env python3
"""gpurun_out/prof/<prec>/ (tools/profile_bench.sh) -> profiles/<tag>_kernel_stats_<prec>.csv, <tag>_pmc_<prec>.csv,
<tag>_bench_<prec>.json and profiles/pmc_traffic.json (per-launch HBM bytes read by bench.py).

HBM bytes per launch = (fetch_factor * FETCH_SIZE + WRITE_SIZE) KB * 1024.  MI355X_MICROARCH.md (HBM section): on gfx950
FETCH_SIZE reports exactly half of the bytes of a wide coalesced streaming read (16 B per lane, global_load and buffer_load..lds
alike) and has to be doubled; other access patterns and WRITE_SIZE are uncalibrated.  The split-f16 layer-2 kernel streams the
y1 planes exactly that way (LDS-DMA, 1 KiB contiguous per wave-instruction): fetch_factor = 2 there — 2 x 7.5 GB = 15.0 GB
against 13.65 GB of algorithmic reads (the planes once per direction) plus the weight/L4 slices that miss L2.  Every other
kernel is reported raw (fetch_factor = 1) and flagged uncalibrated."""
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
    # 1. kernel stats
    st = sorted(glob.glob(os.path.join(src, "stats", "**", "*kernel_stats.csv"), recursive=True), key=os.path.getmtime)
    if st:
        text = open(st[-1]).read()
        open(os.path.join(out, "%s_kernel_stats_%s.csv" % (tag, prec)), "w").write(
            "# rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --steps 2 --warmup 1 --no_cpu_baseline --no_profile "
            "--no_fast --no_resident --no_overlap --precision %s   (MI355X; durations in ns; one context: 1 priming + 1 warm-up + 2 timed passes, "
            "c3r_load_reads inside every pass)\n" % prec + text)
    # 2. PMC
    rows = {}
    for sub in ("pmc_fetch", "pmc_write", "pmc_sq"):
        for k, cs in counters(os.path.join(src, sub)).items():
            for c, v in cs.items():
                rows.setdefault(k, {})[c] = (sum(v) / len(v), len(v))
    hdr = ["FETCH_SIZE", "WRITE_SIZE", "SQ_VALU_MFMA_BUSY_CYCLES", "GRBM_GUI_ACTIVE", "SQ_WAVE_CYCLES", "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY", "SQ_ACTIVE_INST_ANY"]
    traffic = {}
    with open(os.path.join(out, "%s_pmc_%s.csv" % (tag, prec)), "w") as f:
        f.write("# rocprofv3 --pmc <set> --kernel-trace, separate passes (FETCH_SIZE | WRITE_SIZE | SQ_*), same bench command; per-launch averages.\n"
                "# hbm_bytes = (fetch_factor * FETCH_SIZE + WRITE_SIZE) KB * 1024; fetch_factor = 2 for the split-f16 k_lstm2 (wide coalesced streaming reads, MI355X_MICROARCH.md), 1 = raw/uncalibrated elsewhere (tools/summarize_profiles.py); mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / "
                "(1024 SIMDs * GRBM_GUI_ACTIVE / 8 XCDs); clock_GHz needs the kernel duration and is quoted in DESIGN.md.\n")
        f.write("kernel,launches," + ",".join(hdr) + ",hbm_bytes_per_launch,mfma_busy\n")
        for k in sorted(rows):
            r = rows[k]
            vals = [r.get(h, (0.0, 0))[0] for h in hdr]
            n = max(v[1] for v in r.values())
            ff = 2.0 if (k == "k_lstm2" and prec in ("f16x3", "f16+f8")) else 1.0      # (precision 2 streams y1 the same way)
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
