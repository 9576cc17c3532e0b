"""Worker of tools/scale_check.sh (test infrastructure: it regenerates the golden inputs with the oracle's setup side).
Runs the example flow of the committed digests (tests/golden/digests_n4096.json) on ONE RAM sharded over N GPUs and
checks every digest:
    --mode group : one process, N devices behind fheram_group_* (api.GroupRam)
    --mode ranks : one process per GPU under torch.distributed (fheram_amd.sharded.ShardedRam + TorchComm; backend nccl = RCCL
                   on device buffers, or gloo on host buffers); start with torch.distributed.run
Prints one JSON line (rank 0) and exits non-zero on a mismatch."""
import argparse
import hashlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a, dtype=np.int64).tobytes()).hexdigest()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mode", choices=["group", "ranks"], required=True)
    ap.add_argument("--n", type=int, default=1)
    ap.add_argument("--log-max-addr", type=int, default=21)
    ap.add_argument("--all-ranks-device0", action="store_true")
    ap.add_argument("--dist-backend", default="nccl")
    a = ap.parse_args()
    import pyoracle as po
    import make_golden
    from _pkg import load_package
    d = json.load(open(os.path.join(ROOT, "tests", "golden", "digests_n4096.json")))[str(1 << a.log_max_addr)]
    d.setdefault("max_addr", 1 << a.log_max_addr)
    d.setdefault("params", {})
    inp = make_golden.inputs(po.OParams(max_addr=d["max_addr"], word_size=d["word_size"], **d["params"]), d["seed"])
    assert {k: sha(v) for k, v in inp.items()} == d["inputs"], "setup side not reproducible on this machine"
    want = d["outputs"]
    ws, max_addr = d["word_size"], d["max_addr"]
    out = {}
    if a.mode == "group":
        pkg = load_package()
        params = pkg.Parameters(max_addr=max_addr, word_size=ws, **d["params"])
        grp = pkg.GroupRam(params, [0] * a.n if a.all_ranks_device0 else list(range(a.n)))
        keys = pkg.EvaluationKeysPrepared(inp["gal_els"], list(inp["atk_glwe"]), inp["atk_ggsw_inv"], inp["tsk"])
        addr = pkg.Address(params, list(inp["addr"]))
        grp.load_encrypted(inp["rows"])
        out["read"] = sha(grp.read(addr, keys))
        out["rpw"] = sha(grp.read_prepare_write(addr, keys))
        out["rows_after_rpw"] = sha(grp.store_encrypted())
        out["tree_after_rpw"] = sha(grp.tree(0))
        grp.write(inp["w"], addr, keys)
        out["rows_after_write"] = sha(grp.store_encrypted())
        out["readback"] = sha(grp.read(addr, keys))
        ok = out == want
        print(json.dumps({"mode": "group", "n": a.n, "log_max_addr": a.log_max_addr, "digests_ok": ok, "peer_direct": grp.peer_info(),
                          "mismatch": [k for k in want if out.get(k) != want[k]]}))
        raise SystemExit(0 if ok else 4)
    import torch
    import torch.distributed as dist
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    local = 0 if a.all_ranks_device0 else int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    kw = {"device_id": torch.device("cuda", local)} if a.dist_backend == "nccl" else {}
    dist.init_process_group(a.dist_backend, rank=rank, world_size=world, **kw)
    pkg = load_package()
    from fheram_amd.sharded import ShardedRam, TorchComm
    params = pkg.Parameters(max_addr=max_addr, word_size=ws, **d["params"])
    engine = pkg.Ram(params, local, shard=rank, n_shards=world)
    engine.load_encrypted(inp["rows"][:, rank::world])
    keys = pkg.EvaluationKeysPrepared(inp["gal_els"], list(inp["atk_glwe"]), inp["atk_ggsw_inv"], inp["tsk"])
    addr = pkg.Address(params, list(inp["addr"]))
    ram = ShardedRam(engine, TorchComm(device_buffers=a.dist_backend == "nccl"))
    r = ram.read(addr, keys)
    q = ram.read_prepare_write(addr, keys)
    rows_rpw = engine.store_encrypted()
    tree_rpw = engine.tree(0) if rank == 0 else None
    ram.write(inp["w"] if rank == 0 else None, addr, keys)
    rows_w = engine.store_encrypted()
    rb = ram.read(addr, keys)

    def whole(mine):   # every rank's rows -> rank 0, interleaved back into [ws][rows][GLWE]
        full = np.zeros((ws, params.rows(), params.glwe_len()), dtype=np.int64) if rank == 0 else None
        dev = "cuda" if a.dist_backend == "nccl" else "cpu"
        t = torch.from_numpy(np.ascontiguousarray(mine)).to(dev)
        parts = [torch.empty_like(t) for _ in range(world)] if rank == 0 else None
        dist.gather(t, parts, dst=0)
        if rank == 0:
            for g in range(world):
                full[:, g::world] = parts[g].cpu().numpy()
        return full
    f_rpw, f_w = whole(rows_rpw), whole(rows_w)
    rc = 0
    if rank == 0:
        out = {"read": sha(r), "rpw": sha(q), "rows_after_rpw": sha(f_rpw), "tree_after_rpw": sha(tree_rpw),
               "rows_after_write": sha(f_w), "readback": sha(rb)}
        ok = out == want
        print(json.dumps({"mode": "ranks", "backend": a.dist_backend, "n": world, "log_max_addr": a.log_max_addr, "digests_ok": ok,
                          "mismatch": [k for k in want if out.get(k) != want[k]]}))
        rc = 0 if ok else 4
    dist.barrier()
    dist.destroy_process_group()
    raise SystemExit(rc)


if __name__ == "__main__":
    main()
