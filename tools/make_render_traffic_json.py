"""profiles/*_render_hbm_traffic_pmc.json from the FETCH_SIZE / WRITE_SIZE passes of tools/run_render_profile.sh: per renderer kernel
and depth-sample setting, HBM-side bytes per launch (FETCH_SIZE doubled: on gfx950 a wide coalesced read is tallied at half its
bytes, MI355X_MICROARCH.md section HBM; both counters are in KiB) next to the kernel's ALGORITHMIC bytes for the bench scene
(one 128 x 128 view, 512-point cloud with 32 features, k = 8, M = 50; P shading points and Q pairs from the probe's own line).
usage: make_render_traffic_json.py <dir with fetch_S*/ write_S*/ stats_S*.log> out.json"""
import collections, csv, glob, json, os, re, sys

root, out = sys.argv[1:3]
R, N, F, K, M = 128 * 128, 512, 32, 8, 50


def mean(d, name):
    fs = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    acc = collections.defaultdict(list)
    if not fs:
        return acc
    for r in csv.DictReader(open(fs[0])):
        if r["Counter_Name"] == name:
            k = r["Kernel_Name"].split("<")[0].split("(")[0].split("::")[-1].replace("void ", "")
            acc[k].append(float(r["Counter_Value"]))
    return {k: sum(v[3:]) / len(v[3:]) if len(v) > 3 else sum(v) / len(v) for k, v in acc.items()}


res = {}
for S in (128, 64):
    log = os.path.join(root, f"stats_S{S}.log")
    m = re.search(r"P=(\d+) Q=(-?\d+)", open(log).read()) if os.path.exists(log) else None
    if not m:
        continue
    P, Q = int(m.group(1)), int(m.group(2))
    row = K * 4 + 12                                           # one shading point of the lists: k indices + a position
    alg = {
        # rays (origin, direction, limits) in; per ray count / selected count / slot mask + the valid rows to the staging area out
        "grid_query_wave_kernel": R * 32 + N * 16 + R * 16 + P * row,
        # staged rows and counts in, ordered lists + bases out
        "compact_ordered_kernel": P * row + R * 4 + P * row + R * 4,
        # lists in, the gathered rows of the (L2-resident) point table are not HBM traffic; aggregated hidden features out (fp16)
        "shade_pairs_kernel": P * row + N * (F * 4 + 12) + P * 256 * 2,
        "shade_points_kernel": P * 256 * 2 + P * 16,
        "ray_march_wave_kernel": P * 16 + P * 12 + R * (8 + 4 + 24 + 4) + R * 20,
        "ray_gen_kernel": R * 32,
    }
    f, w = mean(os.path.join(root, f"fetch_S{S}"), "FETCH_SIZE"), mean(os.path.join(root, f"write_S{S}"), "WRITE_SIZE")
    per = {}
    for k, a in alg.items():
        if k not in f and k not in w:
            continue
        hbm = 2 * f.get(k, 0.0) * 1024 + w.get(k, 0.0) * 1024
        per[k] = {"algorithmic_bytes": a, "fetch_bytes_x2": 2 * f.get(k, 0.0) * 1024, "write_bytes": w.get(k, 0.0) * 1024, "hbm_bytes": hbm,
                  "ratio_to_algorithmic": hbm / a}
    per["_scene"] = {"shading_points": P, "pairs": Q}
    res[f"S{S}"] = per
res["note"] = ("rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes (tools/run_render_profile.sh), means per launch after the first three; FETCH_SIZE x 2 "
               "(gfx950 tallies 128-B requests at 64 B); every buffer of a view fits the 256-MB Infinity Cache, whose hits these fabric-side counters "
               "appear to include: the ratio says how many bytes the kernels MOVE per algorithmic byte, not how many came from DRAM.  The weights "
               "(1.2 MB) and the point table are re-read from L2 by every workgroup and are not algorithmic HBM bytes.")
import hashlib
_csrc = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "neural-point-cloud-diffusion_amd", "csrc")
res["source_sha256"] = {f: hashlib.sha256(open(os.path.join(_csrc, f), "rb").read()).hexdigest() for f in ("geometry.hip", "shade.hip", "shade_common.h", "common.h")}
json.dump(res, open(out, "w"), indent=1)
for S, per in res.items():
    if S not in ("note", "source_sha256"):
        for k, v in per.items():
            if not k.startswith("_"):
                print(S, k, f"alg {v['algorithmic_bytes'] / 1e6:.2f} MB  hbm {v['hbm_bytes'] / 1e6:.2f} MB  ratio {v['ratio_to_algorithmic']:.2f}")
