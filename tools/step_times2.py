"""Diagnostic: what is it that warms up over the first ~10 steps of a fresh process?  15 steps on context A, then the
same measurement on a freshly created context B (own buffers, keys, address), then A again after 200 ms of idling."""
import sys, os, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from _pkg import load_package
pkg = load_package()
N = 4096
rng = np.random.default_rng(0)
synth = lambda shape: rng.integers(-(1 << 16), 1 << 16, size=shape, dtype=np.int64)
def world():
    ram = pkg.Ram.new_from_ram_params(4, [3, 3, 3, 3], 1 << 18)
    p = ram.params
    keys = pkg.EvaluationKeysPrepared(pkg.galois_elements(12), list(synth((12, 3 * 4 * 2 * N))), synth(4 * 5 * 2 * N), synth(4 * 5 * 2 * N))
    addr = pkg.Address(p, list(synth((p.base2d().as_1d().size(), p.ggsw_len()))))
    ram.load_encrypted(synth((4, ram.local_rows(), p.glwe_len())))
    ram.stage_words(synth((4, p.glwe_len())))
    return ram, keys, addr
def steps(tag, w, n):
    ram, keys, addr = w
    out = []
    for i in range(n):
        t0 = time.perf_counter()
        for fn in (lambda: ram.read(addr, keys, download=False), lambda: ram.read_prepare_write(addr, keys, download=False), lambda: ram.write(None, addr, keys)):
            fn()
        ram.sync()
        out.append((time.perf_counter() - t0) * 1e3)
    print(tag, " ".join("%.2f" % t for t in out), flush=True)
A = world()
steps("A fresh      ", A, 15)
B = world()
steps("B after A    ", B, 15)
time.sleep(0.2)
steps("A after idle ", A, 15)
steps("B again      ", B, 8)
