"""Build profiles/*_render_sq_pmc.json (+ the shading subset) from the rocprofv3 --pmc passes of tools/run_render_profile.sh.
Per depth-sample setting ("S128", "S64") and per renderer kernel, means over the launches after the first three: wave-instruction
counts by class (vector, scalar, LDS, vector-memory, scalar-memory, matrix), the wave-cycle split active / issue-stalled / parked,
MFMA-pipe busy fraction, LDS bank-conflict share, duration and effective clock under the profiler.
usage: make_render_pmc_json.py <dir with sq_S*/ and mem_S*/> out_all.json out_shade.json"""
import collections, csv, glob, json, os, sys

root, out_all, out_shade = sys.argv[1:4]
KEEP = ("grid_query", "shade_pairs", "shade_points", "shade_rows", "ray_march", "ray_gen", "grid_build", "grid_")


def short(k):
    k = k.split("<")[0].split("(")[0].split("::")[-1].replace("void ", "")
    return k


def load(d):
    fs = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    dur = collections.defaultdict(list)
    if not fs:
        return acc, dur
    for r in csv.DictReader(open(fs[0])):
        k = r["Kernel_Name"]
        if not any(w in k for w in KEEP):
            continue
        k = short(k)
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        if r["Counter_Name"] in ("GRBM_GUI_ACTIVE", "SQ_INSTS_SALU") and "Start_Timestamp" in r:
            dur[k].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
    return acc, dur


res = {}
for S in (128, 64):
    a1, d1 = load(os.path.join(root, f"sq_S{S}"))
    a2, _ = load(os.path.join(root, f"mem_S{S}"))
    per = {}
    for k in sorted(set(a1) | set(a2)):
        m = {}
        for src in (a1.get(k, {}), a2.get(k, {})):
            for n, v in src.items():
                vv = v[3:] if len(v) > 3 else v
                m[n] = sum(vv) / len(vv)
        d = {"launches_profiled": len(next(iter((a1.get(k) or a2.get(k)).values()))), "counters_mean_per_launch": m}
        d["insts_valu"] = m.get("SQ_INSTS_VALU")
        parts = [m.get(n) for n in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_INSTS_SMEM", "SQ_INSTS_MFMA")]
        d["insts_all"] = sum(p for p in parts if p) if any(parts) else None
        gui = m.get("GRBM_GUI_ACTIVE")
        if gui and d1.get(k):
            dd = d1[k][3:] if len(d1[k]) > 3 else d1[k]
            d["duration_us_under_pmc"] = sum(dd) / len(dd) / 1e3
            # GRBM_GUI_ACTIVE / profiler duration is a clock only for long dispatches (MI355X_MICROARCH.md, DVFS give-back: reads
            # high below ~0.3 ms; the round-3 file carried 2.7-7 "GHz" for the 5-60 us kernels, 2.7 still at 54 us): not reported for short ones
            if d["duration_us_under_pmc"] >= 100.0:
                d["clock_ghz"] = gui / 8 / (sum(dd) / len(dd))
        if gui and m.get("SQ_VALU_MFMA_BUSY_CYCLES"):
            d["mfma_busy"] = m["SQ_VALU_MFMA_BUSY_CYCLES"] / (gui / 8 * 1024)
        if m.get("SQ_INSTS_MFMA"):
            d["valu_per_mfma"] = m.get("SQ_INSTS_VALU", 0) / m["SQ_INSTS_MFMA"]
        if m.get("SQ_WAVE_CYCLES"):
            w = m["SQ_WAVE_CYCLES"]
            d["wave_cycle_split"] = {"active": m.get("SQ_ACTIVE_INST_ANY", 0) / w, "issue_stalled": m.get("SQ_WAIT_INST_ANY", 0) / w, "parked": m.get("SQ_WAIT_ANY", 0) / w}
        if m.get("SQ_LDS_IDX_ACTIVE"):
            d["lds_bank_conflict_share"] = m.get("SQ_LDS_BANK_CONFLICT", 0) / m["SQ_LDS_IDX_ACTIVE"]
        per[k] = d
    res[f"S{S}"] = per
res["note"] = ("rocprofv3 --pmc, tools/run_render_profile.sh: one 128 x 128 view per call of the bench scene (512-point ellipsoid cloud, pose (30, 20), k = 8, "
               "M = 50), means per launch; SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles (ratios unit-free)")
# which kernel sources these counters belong to: bench.py flags the file as stale when the in-tree sources differ
import hashlib
csrc = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "neural-point-cloud-diffusion_amd", "csrc")
res["source_sha256"] = {f: hashlib.sha256(open(os.path.join(csrc, f), "rb").read()).hexdigest()
                        for f in ("geometry.hip", "shade.hip", "shade_common.h", "common.h") if os.path.exists(os.path.join(csrc, f))}
json.dump(res, open(out_all, "w"), indent=1)
json.dump({S: {k: v for k, v in per.items() if k.startswith("shade_")} for S, per in res.items() if S.startswith("S") and isinstance(per, dict) and S[1:].isdigit()} | {"note": res["note"], "source_sha256": res["source_sha256"]}, open(out_shade, "w"), indent=1)
for S, per in res.items():
    if not (S.startswith("S") and S[1:].isdigit()):
        continue
    for k, v in per.items():
        print(S, k, {x: (round(y, 3) if isinstance(y, float) else y) for x, y in v.items() if x not in ("counters_mean_per_launch", "wave_cycle_split")},
              {a: round(b, 2) for a, b in v.get("wave_cycle_split", {}).items()})
