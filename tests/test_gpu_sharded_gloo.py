"""Two PROCESSES, each with a HIP shard context on GPU 0, under fheram_amd.sharded.ShardedRam + TorchComm
over torch.distributed (gloo, host buffers): the product engine — not the oracle-built test engine of
tests/test_sharded_gloo.py — sits under the multi-process orchestration.  (RCCL refuses two ranks on one
device, so the device-buffer hand-over is covered in-process by tests/test_gpu_golden.py and with one
rank by tests/test_gpu_bench_modes.py.)"""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys
import numpy as np
import torch.distributed as dist
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import pyoracle as po
from _pkg import load_package
pkg = load_package()
from fheram_amd.sharded import ShardedRam, TorchComm

dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
max_addr, ws = 8 * 4096, 2
o = po.Oracle(po.OParams(max_addr=max_addr, word_size=ws))
sk = o.secret_gen(5)
evk = o.evk_gen(sk, 6, 7)
rng = np.random.default_rng(8)
data = rng.integers(0, 256, size=max_addr * ws, dtype=np.uint8)
rows = o.ram_encrypt(data, sk, 9, 10)                       # every rank derives the same RAM
idx = 5 * 4096 + 1234
addr_g = o.address_encrypt(idx, sk, 11, 12)
params = pkg.Parameters(max_addr=max_addr, word_size=ws)
engine = pkg.Ram(params, 0, shard=rank, n_shards=world)     # HIP context: rows r = rank (mod world)
engine.load_encrypted(rows[:, rank::world])
keys = pkg.EvaluationKeysPrepared.from_dict(evk)
addr = pkg.Address(params, list(addr_g))
ram = ShardedRam(engine, TorchComm(device_buffers=False))

out = {}
out["read"] = ram.read(addr, keys)
out["rpw"] = ram.read_prepare_write(addr, keys)
val = [3, 250]
w = np.stack([o.glwe_encrypt_coeff0(v, sk, 20 + i, 30 + i) for i, v in enumerate(val)])
ram.write(w if rank == 0 else None, addr, keys)
out["readback"] = ram.read(addr, keys)
np.save(os.path.join(OUT, f"rows_{rank}.npy"), engine.store_encrypted())
if rank == 0:
    np.savez(os.path.join(OUT, "root.npz"), tree=engine.tree(0), **out)
dist.destroy_process_group()
'''


def test_hip_engine_two_ranks_gloo(po, tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(f"ROOT = {ROOT!r}\nOUT = {str(tmp_path)!r}\n" + WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    subprocess.check_call([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                           "--master-addr", "127.0.0.1", "--master-port", "29547", str(script)], env=env, timeout=600)
    max_addr, ws = 8 * 4096, 2
    o = po.Oracle(po.OParams(max_addr=max_addr, word_size=ws))
    sk = o.secret_gen(5)
    keys = o.keys_prepare(o.evk_gen(sk, 6, 7))
    rng = np.random.default_rng(8)
    data = rng.integers(0, 256, size=max_addr * ws, dtype=np.uint8)
    rows = o.ram_encrypt(data, sk, 9, 10)
    addr = o.address_new(o.address_encrypt(5 * 4096 + 1234, sk, 11, 12))
    ram = o.ram_new()
    ram.load(rows)
    got = np.load(tmp_path / "root.npz")
    assert np.array_equal(got["read"], ram.read(addr, keys))
    assert np.array_equal(got["rpw"], ram.read_prepare_write(addr, keys))
    val = [3, 250]
    w = np.stack([o.glwe_encrypt_coeff0(v, sk, 20 + i, 30 + i) for i, v in enumerate(val)])
    ram.write(w, addr, keys)
    assert np.array_equal(got["readback"], ram.read(addr, keys))
    assert np.array_equal(got["tree"], ram.tree(0))
    full = ram.store()
    for rank in range(2):
        assert np.array_equal(np.load(tmp_path / f"rows_{rank}.npy"), full[:, rank::2]), f"rows of rank {rank}"
