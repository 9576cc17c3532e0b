"""HIP path against the committed golden digests (tests/golden/digests_n4096.json): inputs are
regenerated from the recorded seeds by the oracle's setup side, the GPU runs the example flow, and
the SHA-256 of every output must equal the committed one."""
import hashlib
import json
import os
import sys

import numpy as np
import pytest

from _pkg import load_package

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a, dtype=np.int64).tobytes()).hexdigest()


@pytest.mark.parametrize("max_addr", [1 << 12, 1 << 14])
def test_hip_flow_matches_committed_digests(po, max_addr):
    sys.path.insert(0, GOLD)
    import make_golden
    pkg = load_package()
    d = json.load(open(os.path.join(GOLD, "digests_n4096.json")))[str(max_addr)]
    ws = d["word_size"]
    inp, _, _ = make_golden.flow(po.OParams(max_addr=max_addr, word_size=ws), d["seed"])
    assert {k: sha(v) for k, v in inp.items()} == d["inputs"], "setup side not reproducible on this machine"
    ram = pkg.Ram.new_from_ram_params(ws, [3, 3, 3, 3], max_addr)
    keys = pkg.EvaluationKeysPrepared(inp["gal_els"], list(inp["atk_glwe"]), inp["atk_ggsw_inv"], inp["tsk"])
    addr = pkg.Address(ram.params, list(inp["addr"]))
    ram.load_encrypted(inp["rows"])
    out = {"read": ram.read(addr, keys), "rpw": ram.read_prepare_write(addr, keys), "rows_after_rpw": ram.store_encrypted()}
    if max_addr > 4096:
        out["tree_after_rpw"] = ram.tree(0)
    ram.write(inp["w"], addr, keys)
    out["rows_after_write"] = ram.store_encrypted()
    out["readback"] = ram.read(addr, keys)
    assert {k: sha(v) for k, v in out.items()} == d["outputs"]
