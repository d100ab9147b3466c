#!/usr/bin/env python3
"""profiles/<round>_pmc_sq_<cfg>.json + <round>_lane_stats_<cfg>.json + pmc_traffic.json -> profiles/<round>_counters.md: what the hardware
counters say binds the two composites (and the streaming kernels) on every benched configuration.  usage: pmc_table.py <round> [cfg ...]"""
import json, os, sys

rnd = sys.argv[1]
cfgs = sys.argv[2:] or ["cfg2", "refdefault", "cfg3", "cfg5"]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = lambda *a: os.path.join(ROOT, "profiles", *a)
traffic = json.load(open(P("pmc_traffic.json")))
if json.load(open(P(f"{rnd}_pmc_sq_{cfgs[0]}.json"))).get("lib_sha256") != traffic.get("lib_sha256"):
    traffic = {"lib_sha256": json.load(open(P(f"{rnd}_pmc_sq_{cfgs[0]}.json"))).get("lib_sha256", "?")}  # (another build: no HBM column)
out = [f"# Hardware counters per configuration ({rnd}; library sha256 {traffic.get('lib_sha256', '?')[:16]}...)", "",
       "`rocprofv3 --pmc` passes (own runs, kernel-trace only; `scripts/run.sh pmc`), averages per launch.  Clock = GRBM_GUI_ACTIVE / 8 XCDs at 2.4 GHz.",
       "SQ_WAVE_CYCLES / SQ_ACTIVE_INST_* / SQ_WAIT_INST_ANY are summed over waves in units of 4 clocks (SQ_WAVE_CYCLES x 4 / (clocks x 1024 SIMDs) reproduces the",
       "resident waves per SIMD).  `VALU in flight` = SQ_ACTIVE_INST_VALU x 4 / (clocks x 1024): the average number of waves per SIMD with a VALU instruction in flight",
       "(~1 = the vector pipe never idles; it exceeds 1 where issue and execution of different waves overlap).  `lanes lit` / `empty replays`: `scripts/lane_stats.py` on the",
       "benched scene (share of the 64 lanes of a replayed (quadrant, splat) pair that pass the alpha test; replays that light none).  HBM: FETCH_SIZE (x1 ... x2, the gfx950",
       "correction for wide coalesced reads) + WRITE_SIZE per launch.", ""]
for c in cfgs:
    sq = json.load(open(P(f"{rnd}_pmc_sq_{c}.json")))
    try:
        L = json.load(open(P(f"{rnd}_lane_stats_{c}.json")))
    except OSError:
        L = None
    out.append(f"## {c}" + (f" - {L['n_isect'] / 1e6:.2f} M intersections, {L['n_isect_at_or_before_the_tile_last_contributor'] / 1e6:.2f} M at or before their tile's last contributor"
                            f"{' (lazy far sort)' if L.get('lazy_sort') else ''}; backward: {L['bwd_quadrant_replays'] / 1e6:.2f} M replays, lanes lit {L['bwd_active_lane_fraction']:.3f}, "
                            f"empty replays {L['bwd_replays_with_no_valid_lane']:.3f}" if L else ""))
    out.append("")
    out.append("| kernel | us | waves / SIMD resident | VALU in flight | wave time: VALU active / issue-stalled | clocks per VALU inst per SIMD | VALU / LDS / SALU / MFMA wave-insts (M) | LDS bank-conflict share | HBM MB (fetch x1 ... x2 + write) |")
    out.append("|---|---:|---:|---:|---:|---:|---|---:|---|")
    ks = [k for k in sq if isinstance(sq[k], dict) and "GRBM_GUI_ACTIVE" in sq[k]]
    ks.sort(key=lambda k: -sq[k]["GRBM_GUI_ACTIVE"])
    for k in ks:
        v = sq[k]
        cyc = v["GRBM_GUI_ACTIVE"] / 8
        if cyc / 2400 < 8:
            continue
        wc = v.get("SQ_WAVE_CYCLES", 0) or 1
        tr = traffic.get(c, {}).get("kernels", {}).get(k)
        hbm = f"{tr['FETCH_SIZE'] / 1024:.0f} ... {2 * tr['FETCH_SIZE'] / 1024:.0f} + {tr['WRITE_SIZE'] / 1024:.0f}" if tr else "-"
        out.append(f"| `{k}` | {cyc / 2400:.0f} | {wc * 4 / (cyc * 1024):.2f} | {v['SQ_ACTIVE_INST_VALU'] * 4 / (cyc * 1024):.2f} | {v['SQ_ACTIVE_INST_VALU'] / wc:.2f} / {v.get('SQ_WAIT_INST_ANY', 0) / wc:.2f} | "
                   f"{cyc * 1024 / max(v['SQ_INSTS_VALU'], 1):.2f} | {v['SQ_INSTS_VALU'] / 1e6:.0f} / {v.get('SQ_INSTS_LDS', 0) / 1e6:.1f} / {v.get('SQ_INSTS_SALU', 0) / 1e6:.0f} / {v.get('SQ_INSTS_MFMA', 0) / 1e6:.1f} | "
                   f"{v.get('SQ_LDS_BANK_CONFLICT', 0) / max(v.get('SQ_ACTIVE_INST_LDS', 1), 1):.2f} | {hbm} |")
    out.append("")
    out.append("(kernels under 8 us omitted; where a configuration launches several instantiations of one kernel - cfg5's lazy passes - the row averages them.)")
    out.append("")
open(P(f"{rnd}_counters.md"), "w").write("\n".join(out) + "\n")
print("\n".join(out))
