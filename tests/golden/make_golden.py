#!/usr/bin/env python3
"""Generates the committed golden fixtures under tests/golden/.

What these vectors are — and are not.  The reference (phantomzone-org/fhe-ram) cannot be built or
run in this image (no Rust toolchain, Poulpy un-vendored) and holds no ciphertext-level vectors of
its own, so these fixtures are produced by THIS repository's CPU oracle (oracle/): they freeze the
oracle's bit-level behaviour (limb normalisation, rsh rounding, automorphism order, packer
schedule) so that neither the oracle nor the HIP path can drift silently.  Bit-parity with Poulpy
itself stays unpinned (oracle/README.md).

  flow_n16.npz / flow_n64.npz : full int64 input and output vectors of the reference example's
      flow (examples/fhe-ram.rs:97-176: read, read_prepare_write, write, read-back) at N = 16, 64.
  znx_kat.json : small known-answer vectors of the limb arithmetic (normalise, rsh, rotate,
      automorphism).
  digests_n4096.json : SHA-256 of inputs and outputs of the same flow at N = 4096
      (MAX_ADDR = 2^12, 2^14 and the full BASELINE.json sizes 2^18 and 2^21, WORDSIZE = 4), inputs
      regenerated from the recorded seeds.  The two full sizes run the oracle's all-core variant
      (same results as one thread, tests/test_oracle.py); 2^21 takes a few minutes on 8 cores.
      Keys "readme_*": the same flow with the parameter block of the reference's README (README.md:17-27:
      K_PT = 9, K_EVK = 85, i.e. 5-limb trace keys — what its published 450 / 1200 ms were taken with).
      `--readme-only` regenerates just those.

Run from the repo root:  python tests/golden/make_golden.py [--small-only]
"""
import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import pyoracle as po  # noqa: E402


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a, dtype=np.int64).tobytes()).hexdigest()


def inputs(params, seed, o=None):
    """Inputs of the example flow from the oracle's setup side (seeded): keys, RAM, address, word."""
    o = o or po.Oracle(params)
    p = o.p
    sk = o.secret_gen(seed)
    evk = o.evk_gen(sk, seed + 1, seed + 2)
    rng = np.random.default_rng(seed + 3)
    data = rng.integers(0, 256, size=p.max_addr * p.word_size, dtype=np.uint8)
    rows = o.ram_encrypt(data, sk, seed + 4, seed + 5)
    idx = int(rng.integers(0, p.max_addr))
    addr_g = o.address_encrypt(idx, sk, seed + 6, seed + 7)
    val = rng.integers(0, 256, size=p.word_size, dtype=np.uint8)
    w = np.stack([o.glwe_encrypt_coeff0(int(v), sk, seed + 8 + i, seed + 40 + i) for i, v in enumerate(val)])
    return {"sk": sk, "gal_els": evk["gal_els"], "atk_glwe": evk["atk_glwe"], "atk_ggsw_inv": evk["atk_ggsw_inv"],
            "tsk": evk["tsk"], "data": data.astype(np.int64), "rows": rows, "idx": np.array([idx], dtype=np.int64),
            "addr": addr_g, "val": val.astype(np.int64), "w": w}


def flow(params, seed, threads=1):
    """Runs the example flow on the oracle; returns (inputs, outputs) dicts of int64 arrays."""
    o = po.Oracle(params).set_threads(threads)
    p = o.p
    inp = inputs(params, seed, o)
    sk, rows, addr_g, w = inp["sk"], inp["rows"], inp["addr"], inp["w"]
    data, val, idx = inp["data"].astype(np.uint8), inp["val"].astype(np.uint8), int(inp["idx"][0])
    keys = o.keys_prepare({k: inp[k] for k in ("gal_els", "atk_glwe", "atk_ggsw_inv", "tsk")})
    addr = o.address_new(addr_g)
    ram = o.ram_new()
    ram.load(rows)
    out = {}
    out["read"] = ram.read(addr, keys)
    out["rpw"] = ram.read_prepare_write(addr, keys)
    out["rows_after_rpw"] = ram.store()
    t = ram.tree(0)
    if t is not None:
        out["tree_after_rpw"] = t
    ram.write(w, addr, keys)
    out["rows_after_write"] = ram.store()
    out["readback"] = ram.read(addr, keys)
    # decrypt checks (the reference's own assertion) so a fixture can never freeze a wrong answer
    newdata = data.copy()
    for i in range(p.word_size):
        want = o.expected_plain(int(data[i + p.word_size * idx]), p.k_glwe_pt)
        for key in ("read", "rpw"):
            v, nz = o.glwe_decrypt(out[key][i], want, sk)
            assert v == want and nz < -(p.k_glwe_pt + 1), (key, v, want, nz)
        newdata[i + p.word_size * idx] = val[i]
        want = o.expected_plain(int(val[i]), p.k_glwe_pt, written=True)
        v, nz = o.glwe_decrypt(out["readback"][i], want, sk)
        assert v == want and nz < -(p.k_glwe_pt + 1), ("readback", v, want, nz)
    return inp, out, o


def znx_kat():
    o = po.Oracle(po.OParams(log_n=4, max_addr=16, decomp_n=[2, 2]))
    rng = np.random.default_rng(7)
    kat = {"base2k": 17, "n": 16, "cases": []}
    for a_size, res_size in ((4, 3), (3, 3), (5, 4), (1, 1), (2, 3)):
        a = rng.integers(-(1 << 46), 1 << 46, size=(a_size, 16), dtype=np.int64)
        a[:, 0] = [(-1) ** j * (1 << 16) for j in range(a_size)]     # digit boundary
        a[:, 1] = -(1 << 16)
        a[:, 2] = (1 << 16) - 1
        kat["cases"].append({"op": "big_normalize", "res_size": res_size, "in": a.tolist(),
                             "out": o.big_normalize(a, res_size).tolist()})
    for k in (1, 2, 17, 18, 40):
        g = rng.integers(-(1 << 17), 1 << 17, size=3 * 2 * 16, dtype=np.int64)
        g[:6] = [1, -1, 65536, -65536, 65535, 3]
        kat["cases"].append({"op": "glwe_rsh", "k": k, "in": g.tolist(), "out": o.glwe_rsh(k, g).tolist()})
    for k in (1, -1, 5, 16, 17, -31):
        g = rng.integers(-(1 << 16), 1 << 16, size=3 * 2 * 16, dtype=np.int64)
        kat["cases"].append({"op": "glwe_rotate", "k": k, "in": g.tolist(), "out": o.glwe_rotate(k, g).tolist()})
    for gal in (-1, 5, 25, 17, 31):
        a = rng.integers(-(1 << 16), 1 << 16, size=16, dtype=np.int64)
        kat["cases"].append({"op": "poly_automorphism", "g": gal, "in": a.tolist(), "out": o.poly_automorphism(gal, a).tolist()})
    return kat


def main():
    for log_n, max_addr, dec, seed in ((4, 1 << 6, [2, 2], 1000), (6, 3 * 64, [3, 3], 2000)):
        inp, out, _ = flow(po.OParams(log_n=log_n, max_addr=max_addr, decomp_n=dec, word_size=2), seed)
        meta = np.array([log_n, max_addr, 2, seed] + dec, dtype=np.int64)
        np.savez_compressed(os.path.join(HERE, f"flow_n{1 << log_n}.npz"), meta=meta,
                            **{"in_" + k: v for k, v in inp.items()}, **{"out_" + k: v for k, v in out.items()})
    json.dump(znx_kat(), open(os.path.join(HERE, "znx_kat.json"), "w"))
    path = os.path.join(HERE, "digests_n4096.json")
    dig = json.load(open(path)) if os.path.exists(path) else {}
    # (key, max_addr, word_size, seed, cryptographic parameters other than the source constants)
    README = {"k_glwe_pt": 9, "k_evk_trace": 85}                # README.md:17-27: K_PT = 9, K_EVK = 5 * BASEK for every key
    sizes = [("4096", 1 << 12, 4, 3000, {}), ("16384", 1 << 14, 2, 4000, {}), ("readme_16384", 1 << 14, 4, 7000, README)]
    if "--small-only" not in sys.argv:
        sizes += [("262144", 1 << 18, 4, 5000, {}), ("2097152", 1 << 21, 4, 6000, {}),     # BASELINE.json configs[2..4]
                  ("readme_262144", 1 << 18, 4, 8000, README)]                              # the block the published timings used
    if "--readme-only" in sys.argv:
        sizes = [s_ for s_ in sizes if s_[0].startswith("readme")]
    for key, max_addr, ws, seed, crypto in sizes:
        inp, out, o = flow(po.OParams(max_addr=max_addr, word_size=ws, **crypto), seed, threads=os.cpu_count() or 1)
        dig[key] = {"max_addr": max_addr, "word_size": ws, "seed": seed, "params": crypto, "max_big_log2": float(np.log2(o.max_big())),
                    "inputs": {k: sha(v) for k, v in inp.items()}, "outputs": {k: sha(v) for k, v in out.items()}}
        json.dump(dig, open(path, "w"), indent=1)
        print("digests for", key, "done", flush=True)
    print("golden fixtures written to", HERE)


if __name__ == "__main__":
    main()
