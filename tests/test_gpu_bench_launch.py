"""`python bench.py --gpus 2` invoked plainly, with no launcher: the process must start its two ranks itself, rank 0
prints the one JSON line, and the line reports the ranks the process group really had.  The box has one GPU, so the
ranks share it and the merge is staged through the host (RT_BENCH_BACKEND=gloo); on a node every rank owns a GPU and
the same code path runs over RCCL."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(extra, timeout=300):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["RT_BENCH_BACKEND"] = "gloo"
    env["RT_BENCH_WATCHDOG"] = "240"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                        "--configs", "none", "--cpu-spp", "0"] + extra, env=env, capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


def test_plain_invocation_spawns_its_ranks(gpu_ctx):
    out = _run(["--width", "320", "--height", "180", "--spp", "8"])
    assert out["n_gpus"] == 2 and out["ranks"] == 2 and out["steps"] == 2 and out["scaling"] == "weak"
    # two ranks x two steps x 8 spp x 2 lambda-samples over 320 x 180 pixels
    assert round(out["samples_per_s"] * out["ms_per_step"] * 2 / 1e3) == 320 * 180 * 8 * 2 * 2 * 2
    assert out["value"] > 0 and out["roofline"]["kernel_ms"] > 0


def test_plain_invocation_strong_mode(gpu_ctx):
    out = _run(["--mode", "strong", "--workload", "3d", "--total-spp", "6", "--steps", "1"])
    assert out["n_gpus"] == 2 and out["ranks"] == 2 and out["scaling"] == "strong"


@pytest.mark.parametrize("extra,merge_text", [(["--width", "320", "--height", "180", "--spp", "8"], "XYZ projection + all-reduce + encode"),
                                              (["--width", "320", "--height", "180", "--spp", "8", "--merge", "spectral"], "spectral all-reduce + resolve"),
                                              (["--mode", "strong", "--workload", "3d", "--total-spp", "6", "--steps", "1"], "all-reduce of the f32[W*H][3] image")])
def test_the_merge_runs_over_rccl_with_one_rank(extra, merge_text, gpu_ctx):
    """The N > 1 road of bench.py with backend nccl (= RCCL) on the one GPU of the box: RT_BENCH_FORCE_DIST=1 creates
    the process group and sends the merge step through its all-reduce although there is one rank.  What a node adds
    is more ranks, not other code: device tensors on torch's stream, ordered behind the render kernels."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "RT_BENCH_BACKEND")}
    env["RT_BENCH_FORCE_DIST"] = "1"
    env["RT_BENCH_WATCHDOG"] = "240"
    import socket
    with socket.socket() as sock:                      # a port of its own: the cases run back to back
        sock.bind(("127.0.0.1", 0))
        env["MASTER_PORT"] = str(sock.getsockname()[1])
    env["MASTER_ADDR"] = "127.0.0.1"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1",
                        "--configs", "none", "--cpu-spp", "0"] + extra, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    assert out["n_gpus"] == 1 and out["ranks"] == 1 and merge_text in out["config"]["workload"]
    assert out["value"] > 0
