"""Diagnostic: cycle stamps inside k_trace_tail, third step, members 0 and 23 of group 0 (needs the -DFK_STAMP build,
FHERAM_LIB=...: make -C fhe-ram_amd/csrc VARIANT=stamp HIPFLAGS+=-DFK_STAMP variant)."""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from _pkg import load_package
pkg = load_package()
N = 4096
rng = np.random.default_rng(0)
synth = lambda shape: rng.integers(-(1 << 16), 1 << 16, size=shape, dtype=np.int64)
ram = pkg.Ram.new_from_ram_params(4, [3, 3, 3, 3], 1 << 14)
keys = pkg.EvaluationKeysPrepared(pkg.galois_elements(12), list(synth((12, 3 * 4 * 2 * N))), synth(4 * 5 * 2 * N), synth(4 * 5 * 2 * N))
L = pkg.library()
L.fheram_debug_stamps.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), C.c_int]
a = synth((4, ram.params.glwe_len()))
names = ["step start", "inputs arrived (L2 loads), staged, barrier", "forward done", "mac + inverse done", "partials stored",
         "hand-off 1 passed", "normalisation phase done", "hand-off 2 passed"]
for rep in range(3):
    ram.glwe_trace(keys, 0, 12, a)
    st = (C.c_uint64 * 192)()
    L.fheram_debug_stamps(ram._h, st, 192)
    s = np.array([int(x) for x in st], dtype=np.int64).reshape(8, 24)
    t0 = s[0].min()
    print(f"== rep {rep}: third step of k_trace_tail, group 0, 10 ns ticks (s_memrealtime) since the first member started the step; min / median / max over the 24 members")
    for i, n in enumerate(names):
        v = s[i] - t0
        print(f"  {n:46s} {v.min():7d} {int(np.median(v)):7d} {v.max():7d}")
    print("  per member: arrival at hand-off 1:", " ".join(str(int(x)) for x in (s[4] - t0)))
    print("  per member: passed hand-off 1:    ", " ".join(str(int(x)) for x in (s[5] - t0)))
