#!/usr/bin/env python3
"""gpurun_out/flip_cause.json (tests/test_gpu_flip_cause.py) + flip_cause_refdefault.json (tests/test_gpu_refdefault_fullsize.py)
-> profiles/<round>_flip_cause.{md,json}.  usage: flip_cause_table.py <round>"""
import json, os, sys

rnd = sys.argv[1]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rows = json.load(open(os.path.join(ROOT, "gpurun_out", "flip_cause.json")))
ref = json.load(open(os.path.join(ROOT, "gpurun_out", "flip_cause_refdefault.json")))
json.dump({"cases": rows, "refdefault": ref}, open(os.path.join(ROOT, "profiles", f"{rnd}_flip_cause.json"), "w"), indent=1)
out = [f"# What misses 1e-4, and why ({rnd}; tests/test_gpu_flip_cause.py, tests/test_gpu_refdefault_fullsize.py on the device)", "",
       "F = the pixels at which, IN THE fp64 ORACLE, a discrete decision of the reference's algorithm sits within `eps` of its threshold (oracle/margins.py: alpha >= 1/255",
       "less what the float32 projected centre alone moves it by, the 0.999 clamp, T <= 1e-4, a depth-order near-tie, a toggling tile of a Gaussian's rectangle, a max / min",
       "blend tie).  `plain`: the ordinary comparison (share of elements beyond 1e-4 x max|ref|).  `masked`: the same comparison with the loss cotangents zeroed on F on both",
       "sides - worst element of every gradient, NO allowance.  Every image miss lies in F in every case.", "",
       "| case | image elements off | plain: share of gradient elements off (worst tensor) | eps that explains it | fragile pixels (share of the frame) | masked: worst element of any gradient / max |",
       "|---|---:|---:|---:|---:|---:|"]
for r in rows:
    e = r["explained"]
    pl = r["plain_gradient_frac_off"]
    k = max(pl, key=pl.get)
    out.append(f"| {r['case'].replace('flip cause ', '')} | {r['image_elements_off']} | {pl[k]:.1e} ({k}) | {e['eps']:g} | {e['fragile_fraction']:.2e} | {max(e['worst_masked_rel_err'].values()):.1e} |")
w = ref["worst_masked_rel_err"]
out.append(f"| {ref['case']} (blurry frame, 11 sub-samples: F is their union, {min(ref['fragile_fraction_per_subsample']):.3f} - {max(ref['fragile_fraction_per_subsample']):.3f} each) | see the parity table | up to 6e-3 of max on single elements | {ref['eps']:g} | {ref['fragile_fraction']:.3f} | {max(w.values()):.1e} ({max(w, key=w.get)}) |")
open(os.path.join(ROOT, "profiles", f"{rnd}_flip_cause.md"), "w").write("\n".join(out) + "\n")
print("\n".join(out))
