"""Diagnostic: per-phase stamps (s_memtime, wave 0 of workgroup (0,0,0)) of the fused packer combine k_keyswitch<KS_PAIR,3,4,3,1,0>
— the kernel of the pair levels with 128 and 64 outputs of a 2^18 read — on ONE pair (FHERAM_LIMB_SPLIT=0 FHERAM_NCO=1 force that
decomposition for any batch).  Needs the -DFK_STAMP build (FHERAM_LIB=...)."""
import ctypes as C, os, sys
os.environ["FHERAM_LIMB_SPLIT"] = "0"
os.environ["FHERAM_NCO"] = "1"
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from _pkg import load_package
pkg = load_package()
N = 4096
rng = np.random.default_rng(0)
synth = lambda shape: rng.integers(-(1 << 16), 1 << 16, size=shape, dtype=np.int64)
ram = pkg.Ram.new_from_ram_params(4, [3, 3, 3, 3], 1 << 18)
keys = pkg.EvaluationKeysPrepared(pkg.galois_elements(12), list(synth((12, 3 * 4 * 2 * N))), synth(4 * 5 * 2 * N), synth(4 * 5 * 2 * N))
L = pkg.library()
L.fheram_debug_stamps.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), C.c_int]
a = synth((2, ram.params.glwe_len()))
for rep in range(3):
    ram.glwe_pack(keys, a)
    st = (C.c_uint64 * 192)()
    L.fheram_debug_stamps(ram._h, st, 192)
    s = [int(x) for x in st]
    t0 = s[0]
    print(f"== rep {rep}: k_keyswitch<KS_PAIR,3,4,3,NCO=1> column 0, shader-clock ticks")
    for i, n in ((1, "twiddles issued"), (2, "x loaded (+pre-step), staged, gathered"), (3, "forward NTT x3 done")):
        print(f"  {n:40s} {s[i]-t0:8d}")
    print(f"  column: loop top {s[6]-t0}, post-step limbs loaded / body staged {s[4]-t0} (+{s[4]-s[6]})")
    for q in range(4):
        b = 8 + 4 * q
        nxt = s[b + 4] if q < 3 else s[5]
        print(f"    limb {3-q}: start {s[b]-t0:7d}  mac {s[b+1]-s[b]:6d}  inv-ntt {s[b+2]-s[b+1]:6d}  fetch + body add {s[b+3]-s[b+2]:6d}  emit {nxt-s[b+3]:6d}")
    print(f"    column done {s[5]-t0}")
