#!/usr/bin/env python3
"""gpurun_out/pmc_fetch.txt + pmc_write.txt (scripts/pmc_run.sh) -> profiles/pmc_traffic.json (KB per launch)."""
import json, re, sys
out = {}
for path, ctr in (("gpurun_out/pmc_fetch.txt", "FETCH_SIZE"), ("gpurun_out/pmc_write.txt", "WRITE_SIZE")):
    for ln in open(path):
        m = re.match(r"(.{42}) (\S+)\s+avg\s+([\d.]+)\s+n\s+(\d+)", ln)
        if not m or m.group(2) != ctr:
            continue
        k = m.group(1).strip().replace("void ", "").split("<")[0].split("(")[0]
        out.setdefault(k, {})[ctr] = round(float(m.group(3)), 1)
tag = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
doc = {tag: {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, kernel-trace only), "
                       "profiles/r01_j_pmc_hbm_cfg2.txt; KB per launch", "kernels": out}}
json.dump(doc, open("profiles/pmc_traffic.json", "w"), indent=1)
print(json.dumps(out, indent=1)[:1500])
