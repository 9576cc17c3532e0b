"""The loader of tests/test_poulpy_kat.py must work the day real Poulpy vectors arrive: here it is fed vectors in the
exporter's file format that the ORACLE produced (so this says nothing about parity — it checks the harness), in a
temporary directory, and the KAT module is run against them in a child pytest."""
import json
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N = 4096


def test_kat_loader_accepts_the_exporter_format(po, tmp_path):
    o = po.Oracle(po.OParams(max_addr=1 << 14))
    sk = o.secret_gen(1)
    evk = o.evk_gen(sk, 2, 3)
    okeys = o.keys_prepare(evk)
    rng = np.random.default_rng(4)
    files = {}

    def put(name, a):
        np.ascontiguousarray(a, dtype="<i8").tofile(tmp_path / f"{name}.i64")
        files[name] = list(np.shape(a))
    put("sk", sk)
    put("gal_els", evk["gal_els"])
    for i in range(12):
        put(f"atk_{i}", evk["atk_glwe"][i])
    put("atk_inv", evk["atk_ggsw_inv"])
    put("tsk", evk["tsk"])
    addr = o.address_encrypt(1234, sk, 5, 6)
    put("addr", addr)
    a = o.glwe_encrypt_coeff0(3, sk, 7, 8)
    put("ep_a", a)
    put("ep_ggsw", addr[0])
    put("ep_res", o.glwe_external_product(a, addr[0]))
    for tag, gal in (("m1", -1), ("5", 5)):
        put(f"auto_{tag}_in", a)
        put(f"auto_{tag}_out", o.glwe_automorphism(okeys, gal, 0, a))
    put("trace_in", a)
    put("trace_out", o.glwe_trace(okeys, 0, 12, a))
    leaves = np.stack([o.glwe_encrypt_coeff0(int(v), sk, 20 + i, 30 + i) for i, v in enumerate(rng.integers(0, 8, 4))])
    present, order = np.zeros(N, dtype=np.uint8), []
    for j in range(N):
        jr = int(po.lib().fo_reverse_bits_msb(j, 12))
        if jr < 4:
            present[j] = 1
            order.append(jr)
    put("pack_in", leaves)
    put("pack_out", o.glwe_pack(okeys, leaves[order], present))
    put("ggsw_inv_in", addr[0])
    put("ggsw_inv_out", o.ggsw_automorphism_inv(okeys, addr[0]))
    json.dump({"poulpy": "none: oracle-made vectors, harness check only", "backend": "oracle", "n": N, "base2k": 17, "max_addr": 1 << 14,
               "word_size": 4, "files": files}, open(tmp_path / "manifest.json", "w"))
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-m", "not gpu", os.path.join(ROOT, "tests", "test_poulpy_kat.py")],
                       env=dict(os.environ, FHERAM_POULPY_KAT=str(tmp_path)), capture_output=True, text=True, cwd=ROOT, timeout=600)
    assert r.returncode == 0 and "1 passed" in r.stdout and "1 skipped" in r.stdout, r.stdout[-1500:] + r.stderr[-1500:]
