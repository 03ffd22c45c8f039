#!/usr/bin/env python3
"""Does ONE rank of an 8-GPU run get enough host?  On a 1-GPU box: the whole-sample driver (22 contigs, a quarter of their length, BAM ->
output.vcf.gz) and `bench.py --scaling strong --genome_scale 0.25`, each once with the whole host and once confined to the slice
shard.host_budget gives rank 0 of 8 (C3R_HOST_SLICE=0/8: affinity mask = 1/8 of the GPU's NUMA node or of all CPUs, C3R_THREADS,
fetch / inflate / compression threads cut to match).  Writes the table the 8-GPU claim rests on:
    gpurun -- python tools/host_slice.py > profiles/rN/host_slice.txt"""
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run(cmd, slice_):
    env = dict(os.environ)
    env.pop("C3R_THREADS", None); env.pop("OMP_NUM_THREADS", None); env.pop("C3R_FETCH_INFLATE", None)
    if slice_:
        env["C3R_HOST_SLICE"] = slice_
    else:
        env.pop("C3R_HOST_SLICE", None)
    p = subprocess.run([sys.executable] + cmd, cwd=ROOT, env=env, capture_output=True, text=True)
    if p.returncode:
        sys.exit("FAILED %s\n%s" % (cmd, p.stderr[-3000:]))
    return p.stdout


def main():
    sys.path.insert(0, ROOT)
    from clair3_rna_amd import shard
    os.environ["C3R_HOST_SLICE"] = "0/8"
    n_thr, cpus = shard.host_budget(apply=False)
    os.environ.pop("C3R_HOST_SLICE")
    n_all = len(os.sched_getaffinity(0))
    print("host: %d usable CPUs; the slice of rank 0 of 8: %d CPUs (%s...), %d worker threads" % (n_all, len(cpus), ",".join(map(str, cpus[:4])), n_thr))
    rows = []
    for name, slice_ in (("whole host", None), ("1/8 slice ", "0/8")):
        out = run(["tools/sample_e2e.py", "--contigs", "22", "--scale", "0.25", "--repeat", "4"], slice_)
        rates = [float(m) for m in re.findall(r"([0-9.]+) M sites/s host-inclusive", out)]
        secs = [float(m) for m in re.findall(r"output.vcf.gz in ([0-9.]+) s", out)]
        b = json.loads(run(["bench.py", "--scaling", "strong", "--genome_scale", "0.25", "--steps", "2", "--warmup", "1"], slice_).strip().splitlines()[-1])
        rows.append((name, rates, secs, b["value"], b["ms_per_step"]))
        print("%s | call_sample, 22 contigs x 0.25 (BAM -> output.vcf.gz), runs 2-4: %s s = %s M sites/s host-inclusive (first run of the process: %.2f s) | "
              "bench.py --scaling strong --genome_scale 0.25: %.2f M sites/s (%.1f ms per pass over the 24 contigs)"
              % (name, "/".join("%.2f" % s for s in secs[1:]), "/".join("%.2f" % r for r in rates[1:]), secs[0], b["value"] / 1e6, b["ms_per_step"]), flush=True)
    best = lambda r: max(r[1][1:])
    print("slice-confined / unconfined: call_sample %.2f (best of runs 2-4), bench --scaling strong %.2f" % (best(rows[1]) / best(rows[0]), rows[1][3] / rows[0][3]))


if __name__ == "__main__":
    main()
