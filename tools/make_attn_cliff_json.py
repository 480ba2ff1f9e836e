"""profiles/r4_attention_occupancy_cliff.json from tools/run_attn_cliff.sh: the forward kernel's two instantiations side by side --
registers / LDS / scratch of the dispatch (rocprofv3's dispatch columns), duration, SQ counters (means over the launches after the
first two), derived matrix-pipe busy share and resident waves per SIMD (SQ_LEVEL_WAVES / SQ_BUSY_CU_CYCLES-style ratios where present)."""
import collections, csv, glob, json, sys
root, out = sys.argv[1], sys.argv[2]
res = {}
for mode in ("default", "rowx"):
    d = {}
    for pas in ("sq", "occ"):
        fs = glob.glob(f"{root}/{pas}_{mode}/*/*counter_collection.csv") + glob.glob(f"{root}/{pas}_{mode}/*counter_collection.csv")
        if not fs:
            continue
        acc, dur, disp = collections.defaultdict(list), [], {}
        for r in csv.DictReader(open(fs[0])):
            if "attn_fwd_kernel" not in r["Kernel_Name"]:
                continue
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
            if r["Counter_Name"] == "GRBM_GUI_ACTIVE":
                dur.append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
            for k in ("VGPR_Count", "Accum_VGPR_Count", "SGPR_Count", "LDS_Block_Size", "Scratch_Size", "Workgroup_Size", "Grid_Size"):
                if k in r:
                    disp[k] = r[k]
            disp["kernel"] = r["Kernel_Name"][:100]
        m = {n: sum(v[2:]) / max(1, len(v[2:])) for n, v in acc.items()}
        d.setdefault("dispatch", disp)
        d.setdefault("counters_mean_per_launch", {}).update(m)
        if dur and pas == "sq":
            d["duration_us_under_pmc"] = sum(dur[2:]) / len(dur[2:]) / 1e3
    m = d.get("counters_mean_per_launch", {})
    gui = m.get("GRBM_GUI_ACTIVE")
    if gui and m.get("SQ_VALU_MFMA_BUSY_CYCLES"):
        d["mfma_busy"] = m["SQ_VALU_MFMA_BUSY_CYCLES"] / (gui / 8 * 1024)
    if m.get("SQ_WAVE_CYCLES"):
        w = m["SQ_WAVE_CYCLES"]
        d["wave_cycle_split"] = {"active": m.get("SQ_ACTIVE_INST_ANY", 0) / w, "issue_stalled": (m.get("SQ_WAIT_INST_ANY", 0) - 0) / w, "parked": m.get("SQ_WAIT_ANY", 0) / w}
    if m.get("SQ_WAVE_CYCLES") and m.get("SQ_BUSY_CYCLES"):
        # SQ_WAVE_CYCLES counts wave-resident quad-cycles over the chip, SQ_BUSY_CYCLES the cycles an SQ (one per SE slice) is busy: their
        # ratio per SIMD is the mean number of resident waves while busy (unit-free comparison between the two builds)
        d["resident_wave_cycles_per_busy_cycle"] = m["SQ_WAVE_CYCLES"] / m["SQ_BUSY_CYCLES"]
    res[mode] = d
res["note"] = ("attn_fwd_kernel<BF16, ROWX> at B 64, H 16, n 513: `default` = ROWX false (the shipped form), `rowx` = NPCD_ATTN_ROWX32=1 (the last query row "
               "split over eight waves: a few more live registers).  The register count decides 3 or 2 waves per SIMD (512 / 3 = 170.67 -> 168 allocatable)")
json.dump(res, open(out, "w"), indent=1)
for k in ("default", "rowx"):
    d = res[k]
    print(k, d.get("dispatch"), "us", d.get("duration_us_under_pmc"), "busy", d.get("mfma_busy"), d.get("wave_cycle_split"), "waves/busy", d.get("resident_wave_cycles_per_busy_cycle"))
