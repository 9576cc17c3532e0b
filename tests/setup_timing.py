#!/usr/bin/env python3
"""Timing of the setup side: Ram / Address / EvaluationKeys encrypt_sk on the MI355X
(host-side sampling, device arithmetic) beside the oracle's CPU setup, one JSON line.  Lives under tests/
because the CPU leg and the seeded sampler are the test-only oracle.

    python tests/setup_timing.py [log2(max_addr)]        (needs an MI355X)
"""
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


class TimedSource:
    """wraps a source and accumulates the time spent sampling (host work that is not the device's)"""

    def __init__(self, src):
        self.src, self.t = src, 0.0

    def uniform_limbs(self, count):
        t = time.perf_counter()
        out = self.src.uniform_limbs(count)
        self.t += time.perf_counter() - t
        return out

    def gaussian(self, count, scale=1.0):
        t = time.perf_counter()
        out = self.src.gaussian(count, scale)
        self.t += time.perf_counter() - t
        return out


def main(log_max_addr=18):
    pkg = load_package()
    max_addr = 1 << log_max_addr
    o = po.Oracle(po.OParams(max_addr=max_addr))
    sk = o.secret_gen(0)
    data = np.random.default_rng(5).integers(0, 256, size=max_addr * 4, dtype=np.uint8)
    ram = pkg.Ram.new_from_ram_params(4, [3, 3, 3, 3], max_addr)
    dsk = pkg.GLWESecret(ram, sk)
    out = {"workload": f"setup side, 2^{log_max_addr} x 4 B", "rows": o.p.rows * 4, "device": ram.device_info()}

    def leg(name, dev, cpu):
        xa, xe = TimedSource(o.source(1)), TimedSource(o.source(2))
        dev(xa, xe)                       # warm (first launch of a kernel shape)
        xa, xe = TimedSource(o.source(1)), TimedSource(o.source(2))
        t = time.perf_counter()
        dev(xa, xe)
        ram.sync()
        total = time.perf_counter() - t
        ram.profile_reset()
        t = time.perf_counter()
        cpu()
        c = time.perf_counter() - t
        out[name] = {"device_total_s": total, "of_which_host_sampling_s": xa.t + xe.t, "device_arith_and_staging_s": total - xa.t - xe.t,
                     "oracle_cpu_1_core_s": c, "speedup_total": c / total, "speedup_excl_sampling": c / max(total - xa.t - xe.t, 1e-9)}

    leg("ram_encrypt_sk", lambda xa, xe: ram.encrypt_sk(data, dsk, xa, xe), lambda: o.ram_encrypt(data, sk, 1, 2))
    leg("address_encrypt_sk", lambda xa, xe: pkg.Address.encrypt_sk(ram, 12345, dsk, xa, xe), lambda: o.address_encrypt(12345, sk, 1, 2))
    leg("evaluation_keys_encrypt_sk", lambda xa, xe: pkg.EvaluationKeysPrepared.encrypt_sk(ram, dsk, xa, xe), lambda: o.evk_gen(sk, 1, 2))
    # kernel time alone: encrypt the RAM once more with per-launch events
    ram.profile_enable(True)
    ram.profile_reset()
    ram.encrypt_sk(data, dsk, o.source(1), o.source(2))
    pr = ram.profile_get("encrypt")
    launches, blocks, ms = pr["launches"], pr["blocks"], pr["ms"]
    ram.profile_enable(False)
    out["k_encrypt_sk"] = {"launches": launches, "glwe": blocks, "ms": ms, "us_per_glwe": ms * 1e3 / max(blocks, 1),
                           "algorithmic_bytes_per_glwe": 3 * 2 * 4096 * 4 * 2,
                           "GBps": blocks * 3 * 2 * 4096 * 4 * 2 / (ms * 1e-3) / 1e9 if ms else None}
    print(json.dumps(out))


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 18)
