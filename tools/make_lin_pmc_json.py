"""profiles/*_gemm_c_fc_pmc.json from the passes of tools/run_lin_pmc.sh: per kernel (the own lin_kernel and the library's kernel of the
same product) means per launch after the first two of every counter, MFMA-pipe busy, the wave-cycle split, LDS bank-conflict share,
L2 hit rate, HBM-side bytes (FETCH_SIZE x 2 + WRITE_SIZE, KiB -> bytes) against the product's algorithmic bytes.
usage: make_lin_pmc_json.py <dir> out.json"""
import collections, csv, glob, json, os, sys
root, out = sys.argv[1:3]
T, N, K = 32768, 4096, 1024
alg = (T * K + N * K + T * N) * 2 + N * 2


def load(d):
    fs = glob.glob(os.path.join(root, d, "**", "*counter_collection.csv"), recursive=True)
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    dur = collections.defaultdict(list)
    if not fs:
        return acc, dur
    for r in csv.DictReader(open(fs[0])):
        k = r["Kernel_Name"]
        if "lin_kernel" in k:
            k = "lin_kernel (own, csrc/gemm_nt.hip)"
        elif k.startswith("Cijk") or k.startswith("Custom_Cijk"):
            k = "library: " + k[:60]
        else:
            continue
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        if "Start_Timestamp" in r:
            dur[k].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
    return acc, dur


res = {}
for d in ("sq", "mem", "l2", "fetch", "write"):
    acc, dur = load(d)
    for k, c in acc.items():
        e = res.setdefault(k, {"counters_mean_per_launch": {}})
        ncount = len(c)
        for n, v in c.items():
            e["counters_mean_per_launch"][n] = sum(v[2:]) / max(1, len(v[2:]))
        if d == "sq" and dur[k]:
            dd = dur[k][2 * ncount:]
            e["duration_us_under_pmc"] = sum(dd) / max(1, len(dd)) / 1e3
for k, e in res.items():
    m = e["counters_mean_per_launch"]
    gui = m.get("GRBM_GUI_ACTIVE")
    if gui and m.get("SQ_VALU_MFMA_BUSY_CYCLES"):
        e["mfma_busy"] = m["SQ_VALU_MFMA_BUSY_CYCLES"] / (gui / 8 * 1024)
    if gui and e.get("duration_us_under_pmc"):
        e["clock_ghz"] = gui / 8 / (e["duration_us_under_pmc"] * 1e3)
    if m.get("SQ_INSTS_MFMA"):
        e["valu_per_mfma"] = m.get("SQ_INSTS_VALU", 0) / m["SQ_INSTS_MFMA"]
    if m.get("SQ_WAVE_CYCLES"):
        w = m["SQ_WAVE_CYCLES"]
        e["wave_cycle_split"] = {"active": m.get("SQ_ACTIVE_INST_ANY", 0) / w, "issue_stalled": m.get("SQ_WAIT_INST_ANY", 0) / w, "parked": m.get("SQ_WAIT_ANY", 0) / w}
    if m.get("SQ_LDS_IDX_ACTIVE"):
        e["lds_bank_conflict_share"] = m.get("SQ_LDS_BANK_CONFLICT", 0) / m["SQ_LDS_IDX_ACTIVE"]
    if m.get("TCC_HIT_sum") is not None and m.get("TCC_MISS_sum") is not None and m["TCC_HIT_sum"] + m["TCC_MISS_sum"] > 0:
        e["l2_hit_rate"] = m["TCC_HIT_sum"] / (m["TCC_HIT_sum"] + m["TCC_MISS_sum"])
    if m.get("FETCH_SIZE") is not None or m.get("WRITE_SIZE") is not None:
        e["hbm_bytes"] = 2 * m.get("FETCH_SIZE", 0) * 1024 + m.get("WRITE_SIZE", 0) * 1024
        e["hbm_over_algorithmic"] = e["hbm_bytes"] / alg
res["note"] = (f"c_fc-shaped product T = {T}, N = {N}, K = {K}, bf16 + bias -> bf16; algorithmic bytes {alg}; rocprofv3 --pmc passes of tools/run_lin_pmc.sh; "
               "SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* are quad-cycles (ratios unit-free); FETCH_SIZE doubled (gfx950)")
json.dump(res, open(out, "w"), indent=1)
for k, e in res.items():
    if k != "note":
        print(k, {a: (round(b, 3) if isinstance(b, float) else b) for a, b in e.items() if a != "counters_mean_per_launch"})
