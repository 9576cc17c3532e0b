"""Known-answer vectors exported from REAL Poulpy by tools/poulpy_kat/poulpy_kat.rs (a cargo example for the reference
crate; see tools/poulpy_kat/README.md).  They cannot be produced in the build image (no Rust toolchain, Poulpy
un-vendored), so this module skips cleanly until `tests/golden/poulpy_kat/manifest.json` exists (or FHERAM_POULPY_KAT
names another directory).  When the vectors are there, every Poulpy-level op of the path — and, if the rows were exported,
the whole example flow — is checked on the oracle (CPU) and on the HIP path (-m gpu): the day these pass, the parity
statement of DESIGN.md may drop "unpinned"."""
import json
import os

import numpy as np
import pytest

from _pkg import load_package

KAT = os.environ.get("FHERAM_POULPY_KAT") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "poulpy_kat")
HAVE = os.path.exists(os.path.join(KAT, "manifest.json"))
pytestmark = pytest.mark.skipif(not HAVE, reason="no Poulpy KAT vectors (tools/poulpy_kat/README.md: needs a Rust toolchain + Poulpy 0.3.2)")
N = 4096


def vec(name, shape=None):
    a = np.fromfile(os.path.join(KAT, name + ".i64"), dtype="<i8")
    return a.reshape(shape) if shape else a


def have(*names):
    return all(os.path.exists(os.path.join(KAT, n + ".i64")) for n in names)


@pytest.fixture(scope="module")
def kat(po):
    m = json.load(open(os.path.join(KAT, "manifest.json")))
    assert m["n"] == N and m["base2k"] == 17
    o = po.Oracle(po.OParams(max_addr=m["max_addr"], word_size=m["word_size"]))
    evk = {"gal_els": vec("gal_els"), "atk_glwe": np.stack([vec(f"atk_{i}") for i in range(12)]), "atk_ggsw_inv": vec("atk_inv"),
           "tsk": vec("tsk")}
    return {"m": m, "o": o, "evk": evk, "okeys": o.keys_prepare(evk)}


def _engines(kat, gpu):
    """(name, ops) pairs: the oracle, and the HIP path when asked for"""
    o, okeys = kat["o"], kat["okeys"]
    if not gpu:
        return {"ep": lambda a, g: o.glwe_external_product(a, g), "auto": lambda gal, a: o.glwe_automorphism(okeys, gal, 0, a),
                "trace": lambda a: o.glwe_trace(okeys, 0, 12, a), "pack": lambda cts, order, present: o.glwe_pack(okeys, cts[order], present),
                "inv": lambda g: o.ggsw_automorphism_inv(okeys, g)}
    pkg = load_package()
    ram = pkg.Ram.new_from_ram_params(kat["m"]["word_size"], [3, 3, 3, 3], kat["m"]["max_addr"])
    keys = pkg.EvaluationKeysPrepared.from_dict(kat["evk"])
    return {"ep": lambda a, g: ram.glwe_external_product(a, g)[0], "auto": lambda gal, a: ram.glwe_automorphism(keys, gal, 0, a)[0],
            "trace": lambda a: ram.glwe_trace(keys, 0, 12, a)[0], "pack": lambda cts, order, present: ram.glwe_pack(keys, cts),
            "inv": lambda g: ram.ggsw_automorphism_inv(keys, g), "ram": ram, "keys": keys, "pkg": pkg}


def _check_ops(kat, po, gpu):
    e = _engines(kat, gpu)
    checked = []
    if have("ep_a", "ep_ggsw", "ep_res"):
        assert np.array_equal(e["ep"](vec("ep_a"), vec("ep_ggsw")), vec("ep_res")), "glwe_external_product (SURVEY.md A.4)"
        checked.append("ep")
    for tag, gal in (("m1", -1), ("5", 5)):
        if have(f"auto_{tag}_in", f"auto_{tag}_out"):
            assert np.array_equal(e["auto"](gal, vec(f"auto_{tag}_in")), vec(f"auto_{tag}_out")), f"glwe_automorphism g={gal} (A.6: phi before/after KS)"
            checked.append(f"auto{gal}")
    if have("trace_in", "trace_out"):
        assert np.array_equal(e["trace"](vec("trace_in")), vec("trace_out")), "GLWE::trace (A.7; rsh rounding A.5)"
        checked.append("trace")
    if have("pack_in", "pack_out"):
        cts = vec("pack_in").reshape(4, -1)
        present, order = np.zeros(N, dtype=np.uint8), []
        for j in range(N):
            jr = int(po.lib().fo_reverse_bits_msb(j, 12))
            if jr < 4:
                present[j] = 1
                order.append(jr)
        assert np.array_equal(e["pack"](cts, order, present), vec("pack_out")), "GLWEPacker (A.7 combine schedule)"
        checked.append("pack")
    if have("ggsw_inv_in", "ggsw_inv_out"):
        assert np.array_equal(e["inv"](vec("ggsw_inv_in")), vec("ggsw_inv_out")), "GGSW::automorphism(-1) + tensor key (A.8)"
        checked.append("ggsw_inv")
    assert checked, "manifest present but no op-level vector found"
    return e


def test_oracle_reproduces_poulpy_op_vectors(kat, po):
    _check_ops(kat, po, gpu=False)


def test_oracle_reproduces_poulpy_flow(kat, po):
    if not have("rows", "addr", "w", "read", "rpw", "readback"):
        pytest.skip("flow vectors need the SubRam::data accessor patch (tools/poulpy_kat/poulpy_kat.rs header)")
    o, m = kat["o"], kat["m"]
    ram = o.ram_new()
    ram.load(vec("rows").reshape(m["word_size"], -1, 3 * 2 * N))
    addr = o.address_new(vec("addr").reshape(-1, o.p.ggsw_len))
    assert np.array_equal(ram.read(addr, kat["okeys"]).ravel(), vec("read")), "Ram::read"
    assert np.array_equal(ram.read_prepare_write(addr, kat["okeys"]).ravel(), vec("rpw")), "Ram::read_prepare_write"
    ram.write(vec("w").reshape(m["word_size"], -1), addr, kat["okeys"])
    assert np.array_equal(ram.read(addr, kat["okeys"]).ravel(), vec("readback")), "Ram::write + read-back"


@pytest.mark.gpu
def test_hip_reproduces_poulpy_op_vectors(kat, po):
    _check_ops(kat, po, gpu=True)


@pytest.mark.gpu
def test_hip_reproduces_poulpy_flow(kat, po):
    if not have("rows", "addr", "w", "read", "rpw", "readback"):
        pytest.skip("flow vectors need the SubRam::data accessor patch (tools/poulpy_kat/poulpy_kat.rs header)")
    e = _engines(kat, True)
    ram, keys, pkg, m = e["ram"], e["keys"], e["pkg"], kat["m"]
    ram.load_encrypted(vec("rows").reshape(m["word_size"], -1, 3 * 2 * N))
    addr = pkg.Address(ram.params, list(vec("addr").reshape(-1, ram.params.ggsw_len())))
    assert np.array_equal(ram.read(addr, keys).ravel(), vec("read"))
    assert np.array_equal(ram.read_prepare_write(addr, keys).ravel(), vec("rpw"))
    ram.write(vec("w").reshape(m["word_size"], -1), addr, keys)
    assert np.array_equal(ram.read(addr, keys).ravel(), vec("readback"))
