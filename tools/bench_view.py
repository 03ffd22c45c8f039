# Short view of a bench.py JSON line:  python tools/bench_view.py gpurun_out/r5/bench.json
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("value %.0f sites/s  %.3f ms/step  streams %s  steps %s" % (d["value"], d["ms_per_step"], d["config"].get("streams"), d["steps"]))
if d.get("roofline"): print("roofline", {k: d["roofline"][k] for k in ("kernel", "achieved", "frac", "avg_launch_ms") if k in d["roofline"]})
tb = d.get("roofline_tensor_build")
if tb: print("tensor build: %.3f ms  %.1f GB/s  frac %.4f  bytes/site %s" % (tb["ms"], tb["achieved"], tb["frac"], tb["bytes_per_site"]))
if tb and tb.get("phase1"): print("   phase 1:", {x: tb["phase1"][x] for x in ("bytes", "aligned_positions", "achieved", "frac", "frac_on_covered_positions") if x in tb["phase1"]})
if d.get("f32_mfma"): print("f32_mfma", {x: d["f32_mfma"][x] for x in ("value", "ms_per_step", "network_ms", "network_frac_of_f32_mfma_peak", "error") if x in d["f32_mfma"]})
print("kernels", d.get("kernels_ms_per_step"))
if d.get("cpu_baseline"): print("cpu_baseline", d["cpu_baseline"]["value"], "|", d["cpu_baseline"]["sample"][-170:])
for k in ("resident_inputs", "fast_precision"):
    if d.get(k): print(k, d[k].get("value"), d[k].get("ms_per_step"))
for k in ("strong_1gpu", "phased_1gpu", "stress_500x", "depth_cap_20000x", "realistic_expr"):
    v = d.get(k)
    if not v: continue
    print(k, {x: v[x] for x in ("value", "ms_per_step", "sites_per_step", "reads", "reads_per_s", "error", "cap_cost_ms_per_step", "precision") if x in v})
    if v.get("two_contexts"): print("   two contexts", v["two_contexts"])
    if "kernels_ms_per_step" in v: print("   kernels", v["kernels_ms_per_step"])
    if v.get("roofline"): print("   roofline", {x: v["roofline"][x] for x in ("kernel", "bound", "achieved", "frac", "avg_launch_ms") if x in v["roofline"]})
    if v.get("roofline_tensor_build"): print("   tensor build", {x: v["roofline_tensor_build"][x] for x in ("achieved", "frac", "ms", "bytes_per_site")})
    if v.get("roofline_tensor_build", {}).get("phase1"): print("   phase 1", {x: v["roofline_tensor_build"]["phase1"][x] for x in ("bytes", "aligned_positions", "achieved", "frac") if x in v["roofline_tensor_build"]["phase1"]})
    if "without_cap" in v: print("   without cap", v["without_cap"] if "error" in v["without_cap"] else ({x: v["without_cap"][x] for x in ("ms_per_step", "sites_per_step", "value")}, v["without_cap"]["kernels_ms_per_step"]))
