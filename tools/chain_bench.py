"""Per-step time of the two chains that carry the path's work (fheram_bench_chain): n trace steps / n external products on
`batch` ciphertexts, back to back, nothing else on the GPU.  usage: chain_bench.py [batch] [iters]
FHERAM_LIB / FHERAM_CHAIN_Y select the build / the hand-over form."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from _pkg import load_package

pkg = load_package()
batch = int(sys.argv[1]) if len(sys.argv) > 1 else 256
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 40
ram = pkg.Ram.new_from_ram_params(4, [3, 3, 3, 3], 1 << 12)
for kind, name, n in ((0, "trace chain", 12), (0, "trace chain", 6), (1, "product chain", 4)):
    ram.bench_chain(kind, batch, n, 10)
    best = min(ram.bench_chain(kind, batch, n, iters) for _ in range(3))
    print(f"{name:14s} n={n:2d} batch={batch}: {best / iters * 1e3:8.1f} us per launch, {best / iters / n * 1e3:6.2f} us per step")
