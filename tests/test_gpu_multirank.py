"""N>1 path on real hardware: two ranks (one process each, torch.distributed) drive the HIP decoder
on the one GPU of the box; their gathered 1-best records equal a single-rank decode.  Also the
bench's own rank spawning (`python bench.py --gpus 2` without torchrun)."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(cmd, env, timeout):
    """subprocess.run whose time limit also holds when the ranks (grandchildren) keep the pipes open: the whole process group goes."""
    import signal
    p = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, start_new_session=True)
    try:
        out, err = p.communicate(timeout=timeout)
    except subprocess.TimeoutExpired:
        os.killpg(p.pid, signal.SIGKILL)
        out, err = p.communicate()
        raise AssertionError("timed out after %d s:\n%s\n%s" % (timeout, out[-2000:], err[-2000:]))
    return subprocess.CompletedProcess(cmd, p.returncode, out, err)


def _port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_two_ranks_gather_hip_hypotheses(built):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(_port()), os.path.join(ROOT, "tests", "mr_worker.py")]
    out = _run(cmd, env, 240)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert "identical to the single-rank decode: True" in out.stdout


def test_bench_spawns_its_own_ranks(built):
    """`python bench.py --gpus 2` outside torchrun starts two ranks itself (here both on GPU 0 over
    gloo, JD_BENCH_SHARE_GPU=1 - the numbers are meaningless, the path is what is tested).  Several ranks run the
    headline's own path: batches through the resident slot kernel, nine announced ahead, the steps' 1-best records in ONE
    all_gather behind jd_dec_quiesce at the end of the timed region; --gather-every 1 keeps a collective per step and
    two batches in flight."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", JD_BENCH_SHARE_GPU="1")
    env.pop("WORLD_SIZE", None)
    base = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--utts-per-gpu", "6",
            "--arcs", "60000", "--no-cpu-baseline", "--no-extra-legs", "--pipeline-slots", "8"]
    out = _run(base, env, 400)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 2 and d["config"]["gathered_hyps"] == 12 and d["value"] > 0
    assert d["config"]["pipeline"] and d["config"]["batches_in_flight"] == 10 and d["config"]["pipeline_error"] is None, d["config"]
    assert "ONE all_gather" in d["config"]["gather"] and d["roofline"]["kernel"] == "k_slot"
    assert d["frames_timed"] > 0 and d["single_batch"]["serial_order"]["ms"] > 0 and d["single_batch"]["one_ahead"]["ms"] > 0
    out = _run(base + ["--gather-every", "1"], env, 400)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    d = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert d["config"]["pipeline"] is None and d["config"]["batches_in_flight"] == 2 and d["config"]["gathered_hyps"] == 12
    assert d["frames_timed"] == 3 * d["config"]["frames_per_step"]


def test_bench_strong_scaling_mode(built):
    """`--total-utts N` (BASELINE.json configs[2] with N = 512): ONE batch dealt over the ranks by length; every
    utterance comes back exactly once and the line says "strong"."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", JD_BENCH_SHARE_GPU="1")
    env.pop("WORLD_SIZE", None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--total-utts", "11",
           "--arcs", "60000", "--no-cpu-baseline", "--no-extra-legs"]
    out = _run(cmd, env, 240)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    d = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["config"]["gathered_hyps"] == 11 and d["value"] > 0
