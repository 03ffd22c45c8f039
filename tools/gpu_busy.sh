#!/bin/bash
# how busy is the GPU during the pipelined bench?  kernel trace -> union of kernel intervals over the steady-state span
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/busy
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --output-format csv -d $OUT -- python3 $R/bench.py --steps 12 --warmup 3 --no_cpu_baseline --no_profile --no_fast --no_strong --no_extra > $OUT/log.txt 2>&1
python3 - <<PY
import csv,glob
rows=[]
for f in glob.glob("$OUT/**/*_kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
# steady state: 50 % .. 95 % of the trace's span (--no_fast: the headline f16x3 passes only)
t0=rows[0][0]; t1=max(r[1] for r in rows)
a=t0+(t1-t0)*0.5; b=t1-(t1-t0)*0.05
iv=[(max(s,a),min(e,b)) for s,e,_ in rows if e>a and s<b]
iv.sort()
busy=0; cs,ce=iv[0]
for s,e in iv[1:]:
    if s>ce: busy+=ce-cs; cs,ce=s,e
    else: ce=max(ce,e)
busy+=ce-cs
print("steady-state span %.1f ms, GPU busy (union of kernels) %.1f ms = %.1f %%" % ((b-a)/1e6, busy/1e6, 100*busy/(b-a)))
# idle gaps > 50 us
gaps=[]; cs,ce=iv[0]
for s,e in iv[1:]:
    if s>ce:
        if s-ce>50000: gaps.append((s-ce)/1e3)
        cs,ce=s,e
    else: ce=max(ce,e)
print("gaps > 50 us:", len(gaps), "total %.2f ms" % (sum(gaps)/1e3), sorted(gaps)[-5:])
lstm=sum(min(e,b)-max(s,a) for s,e,n in rows if e>a and s<b and "lstm" in n)
print("sum of LSTM kernel time in span %.1f ms (%.1f %% of span; >100 %% = they overlap each other)" % (lstm/1e6, 100*lstm/(b-a)))
PY
