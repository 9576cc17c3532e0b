"""Diagnostic: per-phase cycle stamps of one key-switch workgroup (needs the -DFK_STAMP build)."""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from _pkg import load_package
pkg = load_package()
N = 4096
rng = np.random.default_rng(0)
synth = lambda shape: rng.integers(-(1 << 16), 1 << 16, size=shape, dtype=np.int64)
ram = pkg.Ram.new_from_ram_params(4, [3, 3, 3, 3], 1 << 18)
keys = pkg.EvaluationKeysPrepared(pkg.galois_elements(12), list(synth((12, 3 * 4 * 2 * N))), synth(4 * 5 * 2 * N), synth(4 * 5 * 2 * N))
L = pkg.library()
L.fheram_debug_stamps.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), C.c_int]
def probe(batch, label):
    a = synth((batch, ram.params.glwe_len()))
    for _ in range(3):
        ram.glwe_trace(keys, 0, 6, a)      # six trace steps = one k_keyswitch_chain launch; the stamps are those of the last step
    st = (C.c_uint64 * 192)()
    L.fheram_debug_stamps(ram._h, st, 192)
    s = [int(x) for x in st]
    t0 = s[0]
    print(f"== {label}: batch {batch}")
    names = {0: "start", 1: "twiddles issued", 2: "x loaded (+rsh), staged, gathered", 3: "forward NTT x3 done"}
    for i in (0, 1, 2, 3):
        print(f"  {names[i]:36s} {s[i]-t0:8d} ticks")
    for c in (0, 1):
        o = 24 * c
        print(f"  column {c}: loop top {s[6+o]-t0}, post-step limbs loaded / body staged {s[4+o]-t0} (+{s[4+o]-s[6+o]})")
        for q in range(4):
            b = 8 + 4 * q + o
            nxt = s[b + 4] if q < 3 else s[5 + o]
            print(f"    limb {3-q}: start {s[b]-t0:7d}  mac {s[b+1]-s[b]:6d}  inv-ntt {s[b+2]-s[b+1]:6d}  body add {s[b+3]-s[b+2]:6d}  emit(+fetch wait) {nxt-s[b+3]:6d}")
        print(f"    column done {s[5+o]-t0}")
os.environ.setdefault("FHERAM_NCO", "0")
probe(256, "full (NCO=2, 256 workgroups)")
