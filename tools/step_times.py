import sys, os, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from _pkg import load_package
pkg = load_package()
N = 4096
rng = np.random.default_rng(0)
synth = lambda shape: rng.integers(-(1 << 16), 1 << 16, size=shape, dtype=np.int64)
ram = pkg.Ram.new_from_ram_params(4, [3, 3, 3, 3], 1 << 18)
p = ram.params
keys = pkg.EvaluationKeysPrepared(pkg.galois_elements(12), list(synth((12, 3 * 4 * 2 * N))), synth(4 * 5 * 2 * N), synth(4 * 5 * 2 * N))
addr = pkg.Address(p, list(synth((p.base2d().as_1d().size(), p.ggsw_len()))))
ram.load_encrypted(synth((4, ram.local_rows(), p.glwe_len())))
ram.stage_words(synth((4, p.glwe_len())))
if len(sys.argv) > 1:      # N ms of unrelated GPU work first (the external-product microbenchmark): is the ramp the device's clocks?
    t0 = time.perf_counter()
    while (time.perf_counter() - t0) * 1e3 < float(sys.argv[1]):
        ram.bench_external_product(256, 8)
    print("busy for %.0f ms before the first step" % ((time.perf_counter() - t0) * 1e3), flush=True)
for i in range(12):
    ts = []
    t0 = time.perf_counter()
    for fn in (lambda: ram.read(addr, keys, download=False), lambda: ram.read_prepare_write(addr, keys, download=False), lambda: ram.write(None, addr, keys)):
        ram.timer_begin(); fn(); ts.append(ram.timer_end())
    print(i, "wall %.3f ms" % ((time.perf_counter() - t0) * 1e3), " ".join("%.3f" % t for t in ts), flush=True)
