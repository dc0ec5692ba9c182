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
    if p.returncode != 0:                                    # (what the ranks said, kept where a gpurun call brings it home; the assertion shows a tail)
        try:
            d = os.path.join(ROOT, "gpurun_out", "test_logs")
            os.makedirs(d, exist_ok=True)
            with open(os.path.join(d, "multirank_%d.err" % os.getpid()), "a") as f:
                f.write("==== %s\n%s\n%s\n" % (" ".join(cmd), out[-20000:], err[-60000:]))
        except OSError:
            pass
        said = [l for l in err.splitlines() if "bench.py" in l or "juicer_amd error" in l or "Error" in l]
        err = err[-1500:] + "\n-- lines that name the failure:\n" + "\n".join(said[-20:])
    return subprocess.CompletedProcess(cmd, p.returncode, out, err)


def _free_this_process():
    """Several ranks are about to share this box's ONE GPU with the pytest process itself, which by now holds what earlier tests left
    cached on the device - the arena slab of the last decoder (most of the HBM: jd_release_cached_memory) and torch's allocator pool.
    On a node with a GPU per rank none of this matters; here the ranks would size their arenas on what is left and run out."""
    import gc
    import torch
    from juicer_amd import capi
    gc.collect()
    capi.lib().jd_release_cached_memory(0)
    torch.cuda.empty_cache()


def _port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_two_ranks_gather_hip_hypotheses(built):
    _free_this_process()
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
    _free_this_process()
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
    _free_this_process()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", JD_BENCH_SHARE_GPU="1")
    env.pop("WORLD_SIZE", None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--total-utts", "11",
           "--arcs", "60000", "--no-cpu-baseline", "--no-extra-legs"]
    out = _run(cmd, env, 240)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    d = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["config"]["gathered_hyps"] == 11 and d["value"] > 0


def test_bench_eight_ranks_share_one_gpu(built):
    """`python bench.py --gpus 8` on the headline's own path - every rank a resident slot kernel (eight slots each here), nine batches
    announced ahead, ONE all_gather of all steps' records behind jd_dec_quiesce - with all eight ranks on this box's one GPU over gloo
    (JD_BENCH_SHARE_GPU=1: the numbers mean nothing, eight processes' kernels, arenas and collectives side by side do).  What the
    driver's first real 8-GPU run will do is this, with RCCL in gloo's place."""
    _free_this_process()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", JD_BENCH_SHARE_GPU="1")
    env.pop("WORLD_SIZE", None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "3", "--warmup", "1", "--utts-per-gpu", "4",
           "--arcs", "60000", "--no-cpu-baseline", "--no-extra-legs", "--pipeline-slots", "8"]
    out = _run(cmd, env, 600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    d = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert d["n_gpus"] == 8 and d["config"]["gathered_hyps"] == 32 and d["value"] > 0
    assert d["config"]["pipeline_error"] is None and d["config"]["batches_in_flight"] == 10 and d["roofline"]["kernel"] == "k_slot", d["config"]
    assert "wall-clock budget" in out.stderr                    # (what the run's time went to, for the driver's limit)


@pytest.mark.parametrize("phase", ["create", "warmup", "timed"])
def test_bench_falls_back_on_every_rank_when_one_pipeline_fails(built, phase):
    """One rank's resident pipeline fails (forced: JD_BENCH_FAIL=rank:phase - at decoder creation, while the pipeline fills, inside the
    timed region).  The other ranks must not be left waiting in a collective: the failing rank keeps taking part (empty records), ALL
    ranks agree that the attempt is void and repeat the measurement with two batches in flight; the line says so."""
    _free_this_process()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", JD_BENCH_SHARE_GPU="1", JD_BENCH_FAIL="1:" + phase)
    env.pop("WORLD_SIZE", None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--utts-per-gpu", "6",
           "--arcs", "60000", "--no-cpu-baseline", "--no-extra-legs", "--pipeline-slots", "8"]
    out = _run(cmd, env, 400)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    d = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert d["n_gpus"] == 2 and d["config"]["gathered_hyps"] == 12 and d["value"] > 0
    assert d["config"]["pipeline"] is None and d["config"]["batches_in_flight"] == 2, d["config"]
    assert d["config"]["pipeline_error"] and "measuring with two batches in flight on every rank" in out.stderr, (d["config"], out.stderr[-600:])


def test_batch_test_fails_fast_without_the_devices(built):
    """`jd_batch_test -devices N` with more devices than the box has: an error with jd_last_error()'s text within seconds, not a hang
    in ncclCommInitAll (src/DecoderBatchTest.cpp:738-771 is a serial loop; the sharded counterpart must not be worse at failing)."""
    import tempfile
    import time
    import torch
    from juicer_amd import io as jio, synth
    am, net, feats, _ = synth.config_toy()
    n = torch.cuda.device_count() + 1
    with tempfile.TemporaryDirectory() as td:
        jio.write_fsm(os.path.join(td, "n.fsm"), net); jio.write_jdam(os.path.join(td, "m.jdam"), am); jio.write_jdf(os.path.join(td, "u.jdf"), feats[0])
        with open(os.path.join(td, "list"), "w") as f:
            f.write(os.path.join(td, "u.jdf") + "\n")
        t0 = time.time()
        out = subprocess.run([os.path.join(ROOT, "juicer_amd", "jd_batch_test"), "-fsmFName", os.path.join(td, "n.fsm"), "-modelsFName",
                              os.path.join(td, "m.jdam"), "-inputFName", os.path.join(td, "list"), "-mainBeam", "150", "-devices", str(n)],
                             capture_output=True, text=True, timeout=120)
    assert out.returncode != 0 and time.time() - t0 < 60
    assert "visible" in out.stderr or "device" in out.stderr, out.stderr[-400:]
