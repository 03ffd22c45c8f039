#!/usr/bin/env python3
"""Summarise a tools/pmc_sq.sh counter directory: per LSTM kernel the effective clock, matrix-pipe busy fraction and the
wave-cycle split (SQ counters are summed over the chip; SQ_WAVE_CYCLES & co. count quad-cycles)."""
import collections
import csv
import glob
import re
import sys

out = sys.argv[1]
per = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(dict)
for f in glob.glob(out + "/**/*_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        m = re.search(r"k_[a-z0-9_]+", r["Kernel_Name"])
        if not m:
            continue
        k = m.group(0)
        per[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        dur[k][r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
for k in sorted(per):
    if "lstm" not in k:
        continue
    c = {n: sum(v) / len(v) for n, v in per[k].items()}
    d = sum(dur[k].values()) / len(dur[k])
    clk = c["GRBM_GUI_ACTIVE"] / 8 / (d * 1e-3) / 1e9
    simd_cycles = c["GRBM_GUI_ACTIVE"] / 8 * 256 * 4
    wc = c["SQ_WAVE_CYCLES"]
    print("%-14s %.2f ms  clk %.2f GHz  mfma_busy %.3f | wave-cycles: wait_inst %.3f wait_any %.3f active %.3f lds_wait %.3f | VALU insts %.3g"
          % (k, d, clk, c["SQ_VALU_MFMA_BUSY_CYCLES"] / simd_cycles, c["SQ_WAIT_INST_ANY"] / wc, c["SQ_WAIT_ANY"] / wc,
             c["SQ_ACTIVE_INST_ANY"] / wc, c.get("SQ_WAIT_INST_LDS", 0) / wc, c.get("SQ_INSTS_VALU", 0)))
