"""Dev probe: A/B/C.. of library builds on the attention kernels, interleaved (ABCABC...) so that clock drift cancels.
usage: python3 tools/probes/gpu_dev_attn_ab.py n rounds lib_a.so lib_b.so ...   ("default" = the in-tree library; one child process per library and round)"""
import sys, os, subprocess, re, statistics
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
n, rounds, libs = sys.argv[1], int(sys.argv[2]), sys.argv[3:]
res = {l: {} for l in libs}
for _ in range(rounds):
    for l in libs:
        env = dict(os.environ)
        if l != "default":
            env["NPCD_HIP_LIB"] = l
        r = subprocess.run([sys.executable, os.path.join(R, "tools", "probes", "gpu_dev_attn_time.py"), "40", n], env=env, capture_output=True, text=True)
        if r.returncode != 0:
            sys.exit(f"child failed for {l} (rc {r.returncode}):\n{r.stderr[-2000:]}")
        out = r.stdout
        for m in re.finditer(r"^(\w+): median ([\d.]+) us", out, re.M):
            res[l].setdefault(m.group(1), []).append(float(m.group(2)))
for l in libs:
    print(f"{os.path.basename(l):28s}", {k: round(statistics.median(v), 1) for k, v in res[l].items()}, {k: [round(x) for x in v] for k, v in res[l].items()})
