"""Per-launch cost model of the chain kernels: time of an n-step launch for several n (batch 256), fitted as n*s + c.
usage: chain_n.py [iters]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from _pkg import load_package

pkg = load_package()
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 60
ram = pkg.Ram.new_from_ram_params(4, [3, 3, 3, 3], 1 << 12)
for kind, name, ns in ((0, "trace chain", (2, 3, 4, 6, 8, 12)), (1, "product chain", (2, 3, 4))):
    pts = []
    for n in ns:
        ram.bench_chain(kind, 256, n, 20)
        t = min(ram.bench_chain(kind, 256, n, iters) for _ in range(3)) / iters * 1e3
        pts.append((n, t))
        print(f"{name:14s} n={n:2d}: {t:8.1f} us per launch")
    (n0, t0), (n1, t1) = pts[0], pts[-1]
    s = (t1 - t0) / (n1 - n0)
    print(f"  -> per step {s:.2f} us, per launch {t0 - n0 * s:.1f} us")
