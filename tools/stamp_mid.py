"""Diagnostic: real-time-clock stamps inside k_chain_mid (ciphertext 0, third step, every member).  Needs the -DFK_STAMP build:
    make -C fhe-ram_amd/csrc VARIANT=stamp HIPFLAGS+=-DFK_STAMP variant;  FHERAM_LIB=fhe-ram_amd/libfheram_stamp.so python tools/stamp_mid.py [batch]"""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from _pkg import load_package
pkg = load_package()
N = 4096
batch = int(sys.argv[1]) if len(sys.argv) > 1 else 16
rng = np.random.default_rng(0)
synth = lambda shape: rng.integers(-(1 << 16), 1 << 16, size=shape, dtype=np.int64)
ram = pkg.Ram.new_from_ram_params(4, [3, 3, 3, 3], 1 << 14)
keys = pkg.EvaluationKeysPrepared(pkg.galois_elements(12), list(synth((12, 3 * 4 * 2 * N))), synth(4 * 5 * 2 * N), synth(4 * 5 * 2 * N))
L = pkg.library()
L.fheram_debug_stamps.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), C.c_int]
a = synth((batch, ram.params.glwe_len()))
members = 12 if batch <= 16 else (8 if batch <= 32 else 4)
names = ["step start", "inputs arrived, digits staged + gathered", "forward transform(s) done", "MAC + inverse transforms done, partials stored",
         "hand-off A passed", "normalisation phase done", "hand-off B passed"]
for rep in range(3):
    ram.glwe_trace(keys, 0, 12, a)
    st = (C.c_uint64 * 192)()
    L.fheram_debug_stamps(ram._h, st, 192)
    s = np.array([int(x) for x in st], dtype=np.int64).reshape(8, 24)[:7, :members]
    t0 = s[0].min()
    print(f"== rep {rep}: third step of k_chain_mid, batch {batch} ({members} members), ciphertext 0, 10 ns ticks since the first member started the step; min / median / max over the members")
    for i, n in enumerate(names):
        v = s[i] - t0
        print(f"  {n:52s} {v.min():7d} {int(np.median(v)):7d} {v.max():7d}")
    print("  per member: arrival at hand-off A:", " ".join(str(int(x)) for x in (s[3] - t0)))
print(ram.mid_stats())
