"""world_size-2 test of the row-sharded orchestration over torch.distributed (gloo, CPU):
fheram_amd.sharded.ShardedRam + TorchComm with the oracle-built test engine must reproduce the
unsharded oracle bit for bit — partition by residue class, one all-gather per read, one broadcast
per write."""
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys
import numpy as np
import torch.distributed as dist
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import pyoracle as po
from _pkg import load_package
from _oracle_engine import OracleShardEngine
pkg = load_package()
from fheram_amd.sharded import ShardedRam, TorchComm

dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
max_addr, ws = 4 * 4096, 2
o = po.Oracle(po.OParams(max_addr=max_addr, word_size=ws))
sk = o.secret_gen(5)
keys = o.keys_prepare(o.evk_gen(sk, 6, 7))
rng = np.random.default_rng(8)
data = rng.integers(0, 256, size=max_addr * ws, dtype=np.uint8)
rows = o.ram_encrypt(data, sk, 9, 10)                       # every rank derives the same RAM
idx = 2 * 4096 + 1234
addr_g = o.address_encrypt(idx, sk, 11, 12)
params = pkg.Parameters(max_addr=max_addr, word_size=ws)
engine = OracleShardEngine(o, params, rows[:, rank::world], rank, world)   # rows r = rank (mod world)
ram = ShardedRam(engine, TorchComm(device_buffers=False))

out = {}
out["read"] = ram.read(addr_g, keys)
out["rpw"] = ram.read_prepare_write(addr_g, keys)
val = [3, 250]
w = np.stack([o.glwe_encrypt_coeff0(v, sk, 20 + i, 30 + i) for i, v in enumerate(val)])
ram.write(w if rank == 0 else None, addr_g, keys)
out["readback"] = ram.read(addr_g, keys)
np.save(os.path.join(OUT, f"rows_{rank}.npy"), engine.data)
if rank == 0:
    np.savez(os.path.join(OUT, "root.npz"), **out)
dist.destroy_process_group()
'''


def test_sharded_orchestration_two_ranks_gloo(po, tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(f"ROOT = {ROOT!r}\nOUT = {str(tmp_path)!r}\n" + WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    subprocess.check_call([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                           "--master-addr", "127.0.0.1", "--master-port", "29541", str(script)], env=env, timeout=600)
    # unsharded oracle on the same inputs
    max_addr, ws = 4 * 4096, 2
    o = po.Oracle(po.OParams(max_addr=max_addr, word_size=ws))
    sk = o.secret_gen(5)
    keys = o.keys_prepare(o.evk_gen(sk, 6, 7))
    rng = np.random.default_rng(8)
    data = rng.integers(0, 256, size=max_addr * ws, dtype=np.uint8)
    rows = o.ram_encrypt(data, sk, 9, 10)
    idx = 2 * 4096 + 1234
    addr = o.address_new(o.address_encrypt(idx, sk, 11, 12))
    ram = o.ram_new()
    ram.load(rows)
    got = np.load(tmp_path / "root.npz")
    assert np.array_equal(got["read"], ram.read(addr, keys))
    assert np.array_equal(got["rpw"], ram.read_prepare_write(addr, keys))
    val = [3, 250]
    w = np.stack([o.glwe_encrypt_coeff0(v, sk, 20 + i, 30 + i) for i, v in enumerate(val)])
    ram.write(w, addr, keys)
    assert np.array_equal(got["readback"], ram.read(addr, keys))
    full = ram.store()
    for rank in range(2):
        assert np.array_equal(np.load(tmp_path / f"rows_{rank}.npy"), full[:, rank::2]), f"rows of rank {rank}"
    for i in range(ws):
        v, nz = o.glwe_decrypt(got["readback"][i], o.cast_u8_to_signed(val[i], 3), sk)
        assert v == o.cast_u8_to_signed(val[i], 3) and nz < -4
