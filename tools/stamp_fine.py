"""Diagnostic: cycle stamps inside one k_keyswitch_fine workgroup (needs the -DFK_STAMP build, FHERAM_LIB=...)."""
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
for rep in range(3):
    ram.glwe_trace(keys, 0, 6, a)
    st = (C.c_uint64 * 192)()
    L.fheram_debug_stamps(ram._h, st, 192)
    s = [int(x) for x in st]
    names = ["start", "loads issued, limbs staged", "twiddles committed (barrier, vmcnt 0)", "gathered", "forward done", "mac done", "inverse done", "stored"]
    for base, label in ((128, "z=0 (no body)"), (144, "z=3 (adds body)")):
        print(f"== rep {rep} {label}")
        for i, n in enumerate(names):
            print(f"  {n:42s} {s[base+i]-s[base]:8d}  (+{s[base+i]-s[base+max(i-1,0)]})")
