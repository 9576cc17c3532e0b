"""bench.py's other modes on the one GPU of the test box: the external-product microbenchmark
(BASELINE.json configs[1]) and the row-sharded path with ONE rank under torch.distributed.run — RCCL
all-gather / broadcast on device int32 buffers, ordered against the evaluator's stream with events
(fheram_stream_signal / fheram_stream_wait), i.e. exactly the code N > 1 ranks run."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def last_json(stdout):
    lines = [l for l in stdout.splitlines() if l.startswith("{")]
    assert lines, stdout[-2000:]
    return json.loads(lines[-1])


def test_bench_external_product_mode():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "ep", "--steps", "3", "--warmup", "1",
                        "--no-cpu-baseline"], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    out = last_json(r.stdout)
    assert out["value"] > 0 and 1.0 < out["latency_us_single_product"] < 200.0
    assert out["roofline"]["bound"] == "valu_fp64" and 0.0 < out["roofline"]["frac"] < 1.0 and 0.0 < out["roofline_hbm"]["frac"] < 1.0


def test_bench_sharded_one_rank_rccl_device_buffers():
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1",
                        "--master-addr", "127.0.0.1", "--master-port", "29553", os.path.join(ROOT, "bench.py"),
                        "--gpus", "1", "--mode", "sharded", "--log-max-addr", "14", "--steps", "3", "--warmup", "1",
                        "--no-cpu-baseline", "--no-kernel-timing"], capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    out = last_json(r.stdout)
    assert out["mode"] == "sharded" and out["value"] > 0


def test_bench_sharded_two_ranks_gloo_on_one_gpu():
    """N = 2 ranks of bench.py's default mode (ONE RAM sharded over the ranks), rehearsed on the one GPU with gloo and
    host exchange buffers: weak scaling (2^13 entries per rank) and strong scaling (--total-log-max-addr 14)."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for extra, scaling in (([], "weak"), (["--total-log-max-addr", "14"], "strong")):
        r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                            "--master-addr", "127.0.0.1", "--master-port", "29557", os.path.join(ROOT, "bench.py"),
                            "--gpus", "2", "--dist-backend", "gloo", "--all-ranks-device0", "--log-max-addr", "13",
                            "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-kernel-timing"] + extra,
                           capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
        out = last_json(r.stdout)
        assert out["mode"] == "sharded" and out["n_gpus"] == 2 and out["scaling"] == scaling and out["value"] > 0
        assert "MAX_ADDR=2^14" in out["config"]["workload"]


def test_bench_starts_its_own_ranks_when_no_launcher_is_around():
    """`python bench.py --gpus 2` WITHOUT torch.distributed.run (how the driver calls it): bench.py starts the two ranks
    itself as child processes — never a silent 1-GPU run."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dist-backend", "gloo", "--all-ranks-device0",
                        "--log-max-addr", "13", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-kernel-timing"],
                       capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    out = last_json(r.stdout)
    assert out["n_gpus"] == 2 and out["mode"] == "sharded" and out["value"] > 0
    # a launcher environment that contradicts --gpus is refused, not silently reinterpreted
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=120, cwd=ROOT, env=dict(env, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0"))
    assert r.returncode != 0 and "refusing" in (r.stdout + r.stderr)


def test_bench_native_group_mode_on_one_gpu():
    """--mode group: one process drives the sharded RAM through fheram_group_* (here 2 shards on GPU 0)"""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--mode", "group", "--all-ranks-device0",
                        "--log-max-addr", "13", "--steps", "2", "--warmup", "1"], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    out = last_json(r.stdout)
    assert out["mode"] == "group" and out["n_gpus"] == 2 and out["value"] > 0 and "MAX_ADDR=2^14" in out["config"]["workload"]


def test_scale_check_rehearsal_on_one_gpu():
    """tools/scale_check.sh — the one command for the first contact with an 8-GPU node (digests through the native group and
    through one process per GPU, then bench.py in both modes, N = 1/2/4/8) — rehearsed on the one GPU of the test box: every
    shard on device 0, gloo instead of RCCL, the 2^18 digests (the 2^21 ones through 8 shards: test_gpu_golden.py), the
    process-per-GPU legs up to N = 4 (the box admits 6 GPU processes)."""
    r = subprocess.run(["bash", os.path.join(ROOT, "tools", "scale_check.sh")],
                       env=dict(os.environ, REHEARSE="1", LOG="18", STEPS="2", WARM="1", RANKS_MAX="4"), capture_output=True, text=True, timeout=1500, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert "every digest reproduced at every N" in r.stdout
    lines = [l for l in r.stdout.splitlines() if l[:1].isdigit()]
    assert len(lines) == 4 + 3 and all("FAILED" not in l for l in lines), r.stdout
