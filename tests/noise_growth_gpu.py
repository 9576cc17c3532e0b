#!/usr/bin/env python3
"""BASELINE.json configs[3] ("write-back path, noise growth checked"; README.md:36: "at least ~40 mio read/write
without having to refresh the RAM"): many consecutive read_prepare_write + write cycles on ONE encrypted RAM on the GPU,
with the noise of what a read returns recorded along the way.

Every cycle picks a random address from a pool of encrypted addresses and a random word from a pool of encrypted words
(fresh encryptions per cycle would only time the host's sampler; the noise a cycle adds to the rows does not depend on
them being fresh), runs Ram::read_prepare_write + Ram::write (ram.rs:196-294) and updates a plaintext model.  Every
`sample_every` cycles a few Ram::read results — the address just written and untouched ones — are decrypted and the
reference's noise metric (examples/fhe-ram.rs:230-236: log2|decrypted - want * 2^log_scale| - k_ct) is recorded; every
sample must decrypt to the model's word with noise < -(k_pt + 1) (examples/fhe-ram.rs:109-114).

The only host-side ingredient is the sampler (the test-only oracle's seeded one), which is why this lives under tests/.

    python tests/noise_growth_gpu.py [--cycles 100000] [--sample-every 1000] [--log-max-addr 18] [--params source|readme] [--out FILE]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import pyoracle as po  # noqa: E402
from _pkg import load_package  # noqa: E402


def run(cycles=1000, sample_every=100, log_max_addr=14, params="source", pool=32, seed=7, progress=None, probes=6):
    pkg = load_package()
    crypto = {"k_glwe_pt": 9, "k_evk_trace": 85} if params == "readme" else {}
    max_addr = 1 << log_max_addr
    o = po.Oracle(po.OParams(max_addr=max_addr, **crypto))               # sampler + secret generation only
    P = pkg.Parameters(max_addr=max_addr, **crypto)
    ws, k_pt, k_ct = P.word_size(), P.k_glwe_pt(), P.k_glwe_ct()
    ram = pkg.Ram(P)
    sk = pkg.GLWESecret(ram, o.secret_gen(seed))
    keys = pkg.EvaluationKeysPrepared.encrypt_sk(ram, sk, o.source(seed + 1), o.source(seed + 2))
    rng = np.random.default_rng(seed + 3)
    data = rng.integers(0, 256, size=max_addr * ws, dtype=np.uint8)
    ram.encrypt_sk(data, sk, o.source(seed + 4), o.source(seed + 5))
    model = [pkg.expected_plain(int(v), k_pt) for v in data]             # what each byte decrypts to
    idxs = [int(v) for v in rng.integers(0, max_addr, size=pool)]
    addrs = [pkg.Address.encrypt_sk(ram, i, sk, o.source(seed + 100 + 2 * k), o.source(seed + 101 + 2 * k)) for k, i in enumerate(idxs)]
    # words that are NEVER written: every cycle still rotates their rows forth and back (ram.rs:502-504,644-646) and patches
    # coefficient 0 of every row (ram.rs:612-630) — their noise is the one that accumulates; a written word is as fresh as
    # its last write
    pidx = []
    while len(pidx) < probes:
        v = int(rng.integers(0, max_addr))
        if v not in idxs and v not in pidx:
            pidx.append(v)
    paddrs = [pkg.Address.encrypt_sk(ram, i, sk, o.source(seed + 300 + 2 * k), o.source(seed + 301 + 2 * k)) for k, i in enumerate(pidx)]
    vals = rng.integers(0, 256, size=(pool, ws), dtype=np.uint8)
    words = [ram.encrypt_word(sk, vals[k], o.source(seed + 500 + 2 * k), o.source(seed + 501 + 2 * k)) for k in range(pool)]
    bound = -(k_pt + 1.0)

    def read_noise(cycle, addr, idx):
        ct = ram.read(addr, keys)
        wants = [model[i + ws * idx] for i in range(ws)]
        out = []
        for (value, noise), want in zip(ram.decrypt_coeff(sk, ct, wants), wants):
            assert value == want, f"cycle {cycle}: address {idx} decrypts to {value}, model says {want}"
            assert noise < bound, f"cycle {cycle}: noise {noise} >= {bound} (examples/fhe-ram.rs:109-114)"
            out.append(noise)
        return out

    def sample(cycle, last):
        per = []
        for k in sorted(set([last, (last + 1) % pool, (last + 7) % pool, int(rng.integers(0, pool))])):
            per += read_noise(cycle, addrs[k], idxs[k])
        unt = []
        for a, i in zip(paddrs, pidx):
            unt += read_noise(cycle, a, i)
        return {"cycle": cycle, "max_noise_bits": max(per + unt), "mean_noise_bits": float(np.mean(per)), "reads": (len(per) + len(unt)) // ws,
                "never_written_mean_bits": float(np.mean(unt)), "never_written_max_bits": max(unt)}

    traj = [sample(0, 0)]
    t0 = time.perf_counter()
    last = 0
    for c in range(1, cycles + 1):
        k, kw = int(rng.integers(0, pool)), int(rng.integers(0, pool))
        ram.read_prepare_write(addrs[k], keys, download=False)
        ram.write(words[kw], addrs[k], keys)
        for i in range(ws):
            model[i + ws * idxs[k]] = pkg.expected_plain(int(vals[kw][i]), k_pt, written=True)
        last = k
        if c % sample_every == 0 or c == cycles:
            traj.append(sample(c, last))
            if progress:
                progress(traj[-1], time.perf_counter() - t0)
    elapsed = time.perf_counter() - t0
    # the noise variance of a word that is never written grows linearly with the number of cycles (every cycle adds
    # independent external-product and key-switch noise to every row): fit 2^(2 noise) = a + b * cycle on its mean noise,
    # extrapolate to the bound.  (A word that IS written is as noisy as its last write made it: flat.)
    cyc = np.array([t["cycle"] for t in traj], dtype=np.float64)
    var = np.exp2(2.0 * np.array([t["never_written_mean_bits"] for t in traj]))
    b, a = np.polyfit(cyc, var, 1) if len(traj) > 2 else (0.0, float(var[0]))
    out = {
           "what": "read_prepare_write + write cycles on one RAM, random pooled addresses and words; noise of Ram::read results "
                   "(examples/fhe-ram.rs:230-236 metric) at recently written addresses (mean/max_noise_bits) and at addresses that are "
                   "never written (never_written_*: the noise that accumulates); every sample decrypted and checked against a plaintext model",
           "max_addr": max_addr, "word_size": ws, "params": params, "k_glwe_pt": k_pt, "k_glwe_ct": k_ct, "cycles": cycles,
           "sample_every": sample_every, "address_pool": pool, "never_written_probes": probes, "noise_bound_bits": bound,
           "worst_noise_bits": max(t["max_noise_bits"] for t in traj), "first": traj[0], "last": traj[-1],
           "seconds": elapsed, "cycles_per_s": cycles / elapsed,
           "fit_variance": {"a": float(a), "b_per_cycle": float(b),
                            "mean_noise_bits_at_40e6_cycles": (0.5 * float(np.log2(a + b * 40e6)) if a + b * 40e6 > 0 else None),
                            "cycles_until_mean_noise_reaches_bound": (float((np.exp2(2 * bound) - a) / b) if b > 0 else None),
                            "note": "README.md:36 claims at least ~40 million read/write before a refresh"},
           "trajectory": traj, "device": ram.device_info()}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cycles", type=int, default=100000)
    ap.add_argument("--sample-every", type=int, default=1000)
    ap.add_argument("--log-max-addr", type=int, default=18)
    ap.add_argument("--params", choices=["source", "readme"], default="source")
    ap.add_argument("--out", default=None)
    args = ap.parse_args()

    def progress(t, el):
        print(f"cycle {t['cycle']}: written words {t['mean_noise_bits']:.2f} bits, never written {t['never_written_mean_bits']:.2f} "
              f"(max {t['never_written_max_bits']:.2f}) ({el:.0f} s)", flush=True)
    out = run(args.cycles, args.sample_every, args.log_max_addr, args.params, progress=progress)
    s = json.dumps(out, indent=1)
    if args.out:
        open(args.out, "w").write(s)
    print(json.dumps({k: v for k, v in out.items() if k != "trajectory"}))


if __name__ == "__main__":
    main()
