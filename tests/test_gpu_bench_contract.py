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


def test_bench_multi_rank_code_path_dry_run():
    """The driver's N > 1 launch line (torch.distributed.run, one rank per GPU) on a ONE-GPU box: two ranks share cuda:0 over
    gloo (NPCD_BENCH_DRYRUN_ONE_GPU).  Checks the rank-0-only JSON line, the whole-job aggregation and strong scaling."""
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ, NPCD_BENCH_DRYRUN_ONE_GPU="1")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                          "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                          "--no-cpu-baseline"], capture_output=True, text=True, env=env, timeout=1200)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["config"]["global_batch"] == 64 and d["config"]["per_gpu_batch"] == 32
    assert d["config"]["parallelism"] == "dp2" and d["value"] == pytest.approx(1000.0 / d["ms_per_step"], rel=1e-6)
    assert "rays_per_s_all_gpus" in d["render"]


def test_bench_starts_its_own_ranks_without_a_launcher():
    """`python bench.py --gpus 2` with no WORLD_SIZE in the environment (the shape of the driver's N = 1 command at N > 1): the
    script starts torch.distributed.run as a child process before touching the GPU, relays exactly one JSON line and returns the
    launcher's exit code.  Two ranks share cuda:0 over gloo (NPCD_BENCH_DRYRUN_ONE_GPU)."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["NPCD_BENCH_DRYRUN_ONE_GPU"] = "1"
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--no-cpu-baseline",
                          "--no-render"], capture_output=True, text=True, env=env, timeout=1200)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["per_gpu_batch"] == 32 and d["steps"] == 2 and d["warmup"] == 1
    c = d["comm"]
    assert c["world_size_seen_by_backend"] == 2 and c["ranks_counted_by_all_reduce"] == 2 and c["backend"].startswith("gloo")
    assert c["parameters_identical_across_ranks"] is True and c["max_abs_parameter_difference_to_rank0"] == 0.0
    assert "starting 2 ranks" in out.stderr


@pytest.mark.parametrize("mode", ["bf16_wire", "all_reduce"])
def test_bench_multi_rank_dry_run_other_comm_modes(mode):
    """The same two-rank dry run through bench.py with the gradient buckets on the wire as bf16 (NPCD_COMM_BF16=1) and with the
    plain all-reduce + unsharded optimizer (NPCD_BENCH_NO_SHARD=1): after the run every rank must hold bit-identical parameters
    (`comm.parameters_identical_across_ranks`, computed from an all-gather of parameter checksums inside bench.py) and the line
    must say what went over the wire."""
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ, NPCD_BENCH_DRYRUN_ONE_GPU="1")
    env["NPCD_COMM_BF16" if mode == "bf16_wire" else "NPCD_BENCH_NO_SHARD"] = "1"
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                          "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
                          "--no-cpu-baseline", "--no-render"], capture_output=True, text=True, env=env, timeout=1200)
    assert out.returncode == 0, out.stderr[-3000:]
    d = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][0])
    c = d["comm"]
    assert c["parameters_identical_across_ranks"] is True
    n_bytes = sum(c["bucket_bytes_fp32"])
    if mode == "bf16_wire":
        assert c["mode"].startswith("reduce_scatter") and c["gradient_wire_dtype"] == "bf16"
        assert c["gradient_bytes_handed_to_collectives"] == n_bytes // 2 and c["parameter_all_gather_send_bytes"] == n_bytes // 2
    else:
        assert c["mode"] == "all_reduce" and d["native_path"]["sharded_optimizer"] is False
        assert c["gradient_bytes_handed_to_collectives"] == n_bytes and c["parameter_all_gather_send_bytes"] == 0


def test_smoke_entry():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "__graft_entry__.py"), "smoke"], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, (out.stdout[-1500:], out.stderr[-1500:])
    assert "smoke: render ok" in out.stdout and "smoke: denoiser step ok" in out.stdout
