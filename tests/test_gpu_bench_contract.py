"""bench.py prints ONE JSON line with the contract keys; __graft_entry__.smoke() passes."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def test_bench_json_contract():
    env = dict(os.environ)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"],
                         capture_output=True, text=True, env=env, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                "dtype", "data", "config", "roofline", "render"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1 and d["higher_is_better"] is True
    assert d["value"] == pytest.approx(1000.0 / d["ms_per_step"], rel=1e-6)
    assert d["dtype"] == "bf16" and d["data"] == "synthetic" and d["vs_baseline"] is None
    r = d["roofline"]
    assert r["bound"] == "mfma" and r["unit"] == "TFLOP/s" and r["frac"] == pytest.approx(r["achieved"] / r["peak"])
    assert 0.02 < r["frac"] < 1.0
    assert "error" not in d["render"] and d["render"]["rays_per_s"] > 1e6


def test_smoke_entry():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "__graft_entry__.py"), "smoke"], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, (out.stdout[-1500:], out.stderr[-1500:])
    assert "smoke: render ok" in out.stdout and "smoke: denoiser step ok" in out.stdout
