"""Build profiles/*_hbm_traffic_pmc.json from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs, --kernel-trace only)
of tools/probes/gpu_dev_attn_time.py (attention kernels) or tools/probes/gpu_dev_ew_time.py (elementwise kernels).  FETCH_SIZE / WRITE_SIZE are
reported in KiB; FETCH_SIZE is doubled on gfx950 (MI355X_MICROARCH.md, HBM: 128-B requests are tallied at 64 B).
usage: make_traffic_json.py attn|ew fetch.csv write.csv out.json"""
import csv, collections, json, sys
kind = sys.argv[1]
B, n, H, d = 64, 513, 16, 64
T, W = 32832, 1024
if kind == "attn":
    alg = {"attn_fwd_kernel": 4 * B * n * H * d * 2 + B * H * n * 4,
           "attn_bwd_dq_kernel": 6 * B * n * H * d * 2 + 4 * B * H * n * 4,       # q,k,v,out,dout read + dq written; lse read, 2 row-constant planes written
           "attn_bwd_dkdv_kernel": 6 * B * n * H * d * 2 + 2 * B * H * n * 4}     # q,k,v,dout read + dk,dv written; 2 row-constant planes read
    edge = B * H * (n >> 7) * 4 * 192 * 4 if (n & 127) == 1 and n > 128 else 0      # per-wave partial sums of the edge token's three gradient rows
    alg["attn_bwd_dq_kernel"] += edge * 2 // 3
    alg["attn_bwd_dkdv_kernel"] += edge // 3
    if edge:
        alg["attn_bwd_edge_kernel"] = edge + 3 * B * H * d * 2 + 5 * B * H * d * 2  # partials read, 3 rows written, 5 rows read
    note = "per launch, B=64 H=16 n=513 d=64 bf16 (tools/probes/gpu_dev_attn_time.py)"
else:
    alg = {"add_ln_fwd_kernel": T * W * (4 + 2 + 4 + 2), "ln_bwd_kernel": T * W * (2 + 4 + 4 + 4 + 2), "gelu_fwd_kernel": T * 4 * W * 4,
           "colsum_kernel<true": T * 4 * W * 6, "colsum_kernel<false": T * 4 * W * 2}
    note = "per launch, T=32832 tokens, W=1024 (tools/probes/gpu_dev_ew_time.py)"
def mean_counter(path, name):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != name or ("npcd" not in r["Kernel_Name"] and "attn_bwd_edge" not in r["Kernel_Name"]):
            continue
        k = r["Kernel_Name"]                      # demangled ("void npcd::colsum_kernel<true>(...)") or mangled ("_ZN4npcd13colsum_kernelILb1E...")
        def hit(a):
            base = a.split("<")[0]
            if base not in k:
                return False
            if "<" not in a:
                return True
            flag = a.split("<")[1]
            return (base + "<" + flag) in k or (base + "ILb" + ("1" if flag == "true" else "0")) in k
        key = next((a for a in alg if hit(a)), None)
        if key is None and "colsum_kernel<bool _Accum" in k:
            # rocprofv3 (ROCm 7.2) leaves this template's arguments undemangled ("colsum_kernel<bool _Accum, bool, E, 1>") for BOTH
            # instantiations; the probe is run with NPCD_EW_ONLY_GELU_COLSUM=1 for these passes, so that the only one in the trace
            # is the GELU backward + column sums
            key = "colsum_kernel<true"
        if key:
            acc[key].append(float(r["Counter_Value"]))
    return {k: sum(v[2:]) / len(v[2:]) for k, v in acc.items() if len(v) > 2}
f, w = mean_counter(sys.argv[2], "FETCH_SIZE"), mean_counter(sys.argv[3], "WRITE_SIZE")
out = {}
for k in alg:
    if k not in f or k not in w:
        continue
    fb, wb = f[k] * 1024 * 2, w[k] * 1024
    name = k + (">" if "<" in k else "")
    out[name] = {"fetch_bytes_corrected": fb, "write_bytes": wb, "hbm_bytes": fb + wb, "algorithmic_bytes": alg[k], "ratio": (fb + wb) / alg[k],
                 "note": "FETCH_SIZE x2 correction per MI355X_MICROARCH.md (gfx950 counts 128-B requests at 64 B); " + note}
import os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from source_hashes import source_hashes
out["_meta"] = {"source_sha256": source_hashes(*(("attention.hip", "common.h") if kind == "attn" else ("elementwise.hip", "common.h")))}
json.dump(out, open(sys.argv[4], "w"), indent=1)
out.pop("_meta")
print(json.dumps({k: round(v["ratio"], 3) for k, v in out.items()}))
