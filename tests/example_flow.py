#!/usr/bin/env python3
"""Counterpart of the reference's only driver, /root/reference/examples/fhe-ram.rs:34-177 (SURVEY.md
§8(f) rows N1 + N2), on the MI355X: key generation, RAM and address encryption, timed `read`,
`read_prepare_write`, `write`, and the reference's own assertions after each (decrypted coefficient 0 ==
cast_u8_to_signed(data), noise < -(k_pt+1)) — every step on the device.  The only host-side ingredient is
the sampler (`source_xa` / `source_xe`; Poulpy's `Source` in the reference, here the test-only oracle's
seeded one, which is why this lives under tests/).

    python tests/example_flow.py [log2(max_addr)]        (needs an MI355X)
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import pyoracle as po  # noqa: E402
from _pkg import load_package  # noqa: E402


def main(log_max_addr=14):
    pkg = load_package()
    print("Starting!")                                                     # examples/fhe-ram.rs:35
    max_addr = 1 << log_max_addr
    o = po.Oracle(po.OParams(max_addr=max_addr))                           # sampler + secret generation only
    params = pkg.Parameters(max_addr=max_addr)                             # Parameters::new()            :46
    ws, k_pt = params.word_size(), params.k_glwe_pt()
    ram = pkg.Ram(params)                                                  # Ram::new()                   :76
    sk = pkg.GLWESecret(ram, o.secret_gen(0))                              # fill_ternary_prob(0.5), prepare  :49-59
    t = time.perf_counter()
    keys = pkg.EvaluationKeysPrepared.encrypt_sk(ram, sk, o.source(1), o.source(2))   # EvaluationKeys::encrypt_sk + prepare  :52-63
    print(f"KEYGEN Elapsed time: {(time.perf_counter() - t) * 1e3:.3f} ms")
    rng = np.random.default_rng(5)                                         # Source::new([5u8; 32])       :66
    data = rng.integers(0, 256, size=max_addr * ws, dtype=np.uint8)        # :72-73
    t = time.perf_counter()
    ram.encrypt_sk(data, sk, o.source(3), o.source(4))                     # ram.encrypt_sk               :79
    print(f"RAM ENCRYPT Elapsed time: {(time.perf_counter() - t) * 1e3:.3f} ms")
    idx = int(rng.integers(0, max_addr))                                   # :85
    addr = pkg.Address.encrypt_sk(ram, idx, sk, o.source(5), o.source(6))  # addr.encrypt_sk              :88-95

    def check(ct, data):                                                   # :104-115
        wants = [pkg.cast_u8_to_signed(int(data[i + ws * idx]), k_pt) for i in range(ws)]
        for (value, noise), want in zip(ram.decrypt_coeff(sk, ct, wants), wants):
            assert value == want, (value, want)
            print(f"noise: {noise}")
            assert noise < -(k_pt + 1.0), f"{noise} >= {k_pt + 1.0}"

    for _ in range(3):                                                     # first calls wake the clocks
        ram.read(addr, keys)
    t = time.perf_counter()
    ct = ram.read(addr, keys)                                              # :98-101
    print(f"READ Elapsed time: {(time.perf_counter() - t) * 1e3:.3f} ms")
    check(ct, data)
    t = time.perf_counter()
    ct = ram.read_prepare_write(addr, keys)                                # :118-124
    print(f"READ_PREPARE_WRITE Elapsed time: {(time.perf_counter() - t) * 1e3:.3f} ms")
    check(ct, data)
    value = rng.integers(0, 256, size=ws, dtype=np.uint8)                  # :141-142
    ct_w = ram.encrypt_word(sk, value, o.source(7), o.source(8))           # encrypt_glwe                 :145-148,179-210
    t = time.perf_counter()
    ram.write(ct_w, addr, keys)                                            # :151-154
    ram.sync()
    print(f"WRITE Elapsed time: {(time.perf_counter() - t) * 1e3:.3f} ms")
    for i in range(ws):                                                    # :157-159
        data[i + ws * idx] = value[i]
    check(ram.read(addr, keys), data)                                      # :162-176
    print("ok:", ram.device_info())


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 14)
