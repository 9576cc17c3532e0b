"""A/B of the chain kernels inside ONE process (same box, same clocks): per-step time of the fused trace chain and the
product chain at batch 256 for several FHERAM_CHAIN_Y settings of the loaded build, taken round-robin.
usage: chain_ab.py [rounds] [iters]      (FHERAM_LIB selects the build; one build per process)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from _pkg import load_package

pkg = load_package()
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 5
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 60
forms = [f for f in os.environ.get("AB_FORMS", "3,2,1").split(",")]   # "3": what ships (ks_trace_l + ep_step_r); "2": the closed form handed over through global memory; "1": round 3.s
rams = {}
for f in forms:
    os.environ["FHERAM_CHAIN_Y"] = f
    os.environ["FHERAM_EP_R"] = "1" if f == "3" else "0"
    rams[f] = pkg.Ram.new_from_ram_params(4, [3, 3, 3, 3], 1 << 12)
cases = ((0, "trace chain", 12), (0, "trace chain", 6), (1, "product chain", 4))
res = {(f, c): [] for f in forms for c in cases}
for f in forms:
    for c in cases:
        rams[f].bench_chain(c[0], 256, c[2], 20)
for r in range(rounds):
    for c in cases:
        for f in forms:
            res[(f, c)].append(rams[f].bench_chain(c[0], 256, c[2], iters) / iters / c[2] * 1e3)
for c in cases:
    for f in forms:
        v = sorted(res[(f, c)])
        print(f"{c[1]:14s} n={c[2]:2d} CHAIN_Y={f}: min {v[0]:6.2f}  median {v[len(v) // 2]:6.2f}  max {v[-1]:6.2f} us per step")
