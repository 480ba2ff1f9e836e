"""Build profiles/*_attention_sq_pmc.json from a rocprofv3 --pmc + --kernel-trace csv pass over tools/probes/gpu_dev_attn_only.py.
Counters (one pass: 8 SQ slots + 1 GRBM): SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY
SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES GRBM_GUI_ACTIVE.
Derived per kernel (means over the launches after the first two):
  mfma_busy   = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs * 1024 SIMDs)     (the gfx94x MfmaUtil expression of rocprofv3;
                the busy counter counts cycles, 32 per v_mfma_f32_32x32x16, summed over the chip's SIMDs)
  clock_ghz   = GRBM_GUI_ACTIVE / 8 / duration_ns                                     (MI355X_MICROARCH.md, DVFS give-back)
  valu_per_mfma, and the wave-cycle split  active / issue-stalled / parked  (SQ_ACTIVE_INST_ANY, SQ_WAIT_INST_ANY, SQ_WAIT_ANY over
  SQ_WAVE_CYCLES; quad-cycle units, the ratios are unit-free).
usage: make_sq_pmc_json.py <dir with *_counter_collection.csv> out.json [kernel-substring ...]"""
import collections, csv, glob, json, sys
f = glob.glob(sys.argv[1] + "/*counter_collection.csv")[0]
want = sys.argv[3:] or ["attn_"]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"]
    if not any(w in k for w in want):
        continue
    k = k.split("<")[0].split("(")[0].split("::")[-1].replace("void ", "")
    if k.startswith("_Z") and "attn_bwd_edge_kernel" in k:          # (rocprofv3 leaves the __bf16 instantiation mangled)
        k = "attn_bwd_edge_kernel"
    acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    if r["Counter_Name"] == "GRBM_GUI_ACTIVE" and "Start_Timestamp" in r:
        dur[k].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
out = {}
for k, c in acc.items():
    m = {n: sum(v[2:]) / max(1, len(v[2:])) for n, v in c.items()}
    d = {"counters_mean_per_launch": m, "launches": len(next(iter(c.values())))}
    gui = m.get("GRBM_GUI_ACTIVE")
    if gui and "SQ_VALU_MFMA_BUSY_CYCLES" in m:
        d["mfma_busy"] = m["SQ_VALU_MFMA_BUSY_CYCLES"] / (gui / 8 * 1024)
    if gui and dur[k]:
        dd = dur[k][2:]
        d["duration_us_under_pmc"] = sum(dd) / len(dd) / 1e3
        d["clock_ghz"] = gui / 8 / (sum(dd) / len(dd))
    if m.get("SQ_INSTS_MFMA"):
        d["valu_per_mfma"] = m.get("SQ_INSTS_VALU", 0) / m["SQ_INSTS_MFMA"]
        d["mfma_flops_executed"] = m["SQ_INSTS_MFMA"] * 32768
    if m.get("SQ_WAVE_CYCLES"):
        d["wave_cycle_split"] = {"active": m.get("SQ_ACTIVE_INST_ANY", 0) / m["SQ_WAVE_CYCLES"], "issue_stalled": m.get("SQ_WAIT_INST_ANY", 0) / m["SQ_WAVE_CYCLES"],
                                 "parked": m.get("SQ_WAIT_ANY", 0) / m["SQ_WAVE_CYCLES"]}
    out[k] = d
import os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from source_hashes import source_hashes
if any("attn" in w for w in want):
    out["_meta"] = {"source_sha256": source_hashes("attention.hip", "common.h")}
json.dump(out, open(sys.argv[2], "w"), indent=1)
out.pop("_meta", None)
print(json.dumps({k: {x: (round(y, 3) if isinstance(y, float) else y) for x, y in v.items() if x != "counters_mean_per_launch"} for k, v in out.items()}, indent=1))
