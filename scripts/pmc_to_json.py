#!/usr/bin/env python3
"""gpurun_out/pmc_<tag>_{fetch,write,sq}.txt (scripts/pmc_run.sh) ->
profiles/pmc_traffic.json (FETCH_SIZE / WRITE_SIZE, KB per launch, per kernel; one object per workload, merged across calls
as long as they describe the same libd4gs.so), profiles/<round>_pmc_sq_<cfg>.json (SQ / GRBM counters per launch),
profiles/<round>_lane_stats_<cfg>.json (scripts/lane_stats.py), profiles/pmc_current.json (which round / library bench.py quotes).
usage: pmc_to_json.py <tag> <round> [cfg]      (tag: the pmc_run.sh tags are <tag>_fetch, <tag>_write, <tag>_sq)"""
import hashlib, json, os, re, shutil, sys

tag, rnd = sys.argv[1], sys.argv[2]
cfg = sys.argv[3] if len(sys.argv) > 3 else "cfg2"


def parse(path):
    out = {}
    for ln in open(path):
        m = re.match(r"(.{42}) (\S+)\s+avg\s+([\d.]+)\s+n\s+(\d+)", ln)
        if not m:
            continue
        k = m.group(1).strip().replace("void ", "").split("<")[0].split("(")[0]
        k = re.sub(r"^(k_raster_(?:bwd_q|fwd_r))s?8?$", r"\1", k)  # the 8-waves-per-SIMD / depth-segment entry points of the same bodies
        out.setdefault(k, {})[m.group(2)] = round(float(m.group(3)), 1)
    return out


# the counters describe ONE build of the kernels: bench.py only quotes them while libd4gs.so still hashes to this
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "deblur4dgs_amd", "libd4gs.so")
lib_sha = hashlib.sha256(open(LIB, "rb").read()).hexdigest()
traffic = {}
for part in ("fetch", "write"):
    src = f"gpurun_out/pmc_{tag}_{part}.txt"
    for k, v in parse(src).items():
        traffic.setdefault(k, {}).update(v)
    shutil.copy(src, f"profiles/{rnd}_pmc_{part}_{cfg}.txt")
try:
    doc = json.load(open("profiles/pmc_traffic.json"))
except Exception:
    doc = {}
if doc.get("lib_sha256") != lib_sha:  # another build: its numbers must not sit beside this one's
    doc = {}
doc[cfg] = {"source": f"rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, kernel-trace only), "
                      f"profiles/{rnd}_pmc_fetch_{cfg}.txt / {rnd}_pmc_write_{cfg}.txt; KB per launch",
            "kernels": {k: v for k, v in traffic.items() if "FETCH_SIZE" in v and "WRITE_SIZE" in v}}
doc["lib_sha256"] = lib_sha
json.dump(doc, open("profiles/pmc_traffic.json", "w"), indent=1)
sq = parse(f"gpurun_out/pmc_{tag}_sq.txt")
shutil.copy(f"gpurun_out/pmc_{tag}_sq.txt", f"profiles/{rnd}_pmc_sq_{cfg}.txt")
if os.path.exists(f"gpurun_out/pmc_{tag}_sq2.txt"):  # PMC_MORE=1: a fourth pass (SALU / SMEM / VMEM / LDS-active / bank conflicts / MFMA)
    for k, v in parse(f"gpurun_out/pmc_{tag}_sq2.txt").items():
        sq.setdefault(k, {}).update(v)
    shutil.copy(f"gpurun_out/pmc_{tag}_sq2.txt", f"profiles/{rnd}_pmc_sq2_{cfg}.txt")
sq["lib_sha256"] = lib_sha
json.dump(sq, open(f"profiles/{rnd}_pmc_sq_{cfg}.json", "w"), indent=1)
json.dump({"lib_sha256": lib_sha, "round": rnd}, open("profiles/pmc_current.json", "w"), indent=1)
for lane in (f"gpurun_out/{tag}_lane_stats_{cfg}.json", f"gpurun_out/{tag.rsplit('_', 1)[0]}_lane_stats_{cfg}.json", f"gpurun_out/lane_stats_{cfg}.json"):  # scripts/lane_stats.py (stamped with the library hash itself)
    if os.path.exists(lane) and json.load(open(lane)).get("lib_sha256") == lib_sha:
        json.dump(json.load(open(lane)), open(f"profiles/{rnd}_lane_stats_{cfg}.json", "w"), indent=1)
        break
print(cfg, "kernels:", sorted(doc[cfg]["kernels"]))
