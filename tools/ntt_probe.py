"""Diagnostic: cycle stamps inside one inverse + one forward transform (needs the -DFK_STAMP build)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from _pkg import load_package
pkg = load_package()
ram = pkg.Ram.new_from_ram_params(4, [3, 3, 3, 3], 1 << 12)
L = pkg.library()
L.fheram_debug_stamps.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), C.c_int]
L.fheram_debug_ntt_probe.argtypes = [C.c_void_p, C.c_int]
names = ["reduce+barrier", "inv pass3", "exch2 (wave local)", "inv pass2", "exch1 (wave local)", "inv pass1", "exch0 (barrier)", "inv pass0",
         "final reduce", "fwd pass0", "exch0 (2 barriers)", "fwd pass1", "exch1", "fwd pass2", "exch2", "fwd pass3"]
for blocks in (1, 256, 512):
    L.fheram_debug_ntt_probe(ram._h, blocks)
    st = (C.c_uint64 * 64)()
    L.fheram_debug_stamps(ram._h, st, 64)
    s = [int(x) for x in st]
    print(f"== {blocks} workgroups")
    for i, n in enumerate(names):
        print(f"  {n:22s} {s[i+1]-s[i]:6d} cyc")
    print(f"  inverse total {s[9]-s[0]}  forward total {s[16]-s[9]}")
