"""Diagnostic: s_memtime stamps of wave 0, workgroup (0,0,0) inside one inner step (Y in, Y out) of the fused trace chain in
the closed-form variant (ks_trace_z) at batch 256.  Needs the -DFK_STAMP build (FHERAM_LIB=...)."""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from _pkg import load_package
pkg = load_package()
ram = pkg.Ram.new_from_ram_params(4, [3, 3, 3, 3], 1 << 18)
L = pkg.library()
L.fheram_debug_stamps.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), C.c_int]
for rep in range(3):
    ram.bench_chain(0, 256, 12, 5)
    st = (C.c_uint64 * 192)()
    L.fheram_debug_stamps(ram._h, st, 192)
    s = np.array([int(x) for x in st], dtype=np.int64)
    t0 = s[0]
    print(f"== rep {rep}: inner step of k_keyswitch_chain<3,4,3,3> (ks_trace_l; FHERAM_CHAIN_Y=2: ks_trace_z), batch 256, wave 0 of workgroup 0; shader-clock ticks since the step began")
    print(f"  own coefficients taken from LDS / registers   {s[1]-t0:8d}")
    print(f"  gathers through phi_g, digits                 {s[2]-t0:8d}")
    print(f"  three forward transforms                      {s[3]-t0:8d}")
    prev = s[3]
    for ci in range(2):
        for q in (0, 2):
            b = 8 + (ci * 4 + q) * 4
            print(f"  column {1-ci} limbs {3-q},{2-q}: (operand wait +) 2x3 MACs {s[b]-prev:6d}  two inverse transforms {s[b+1]-s[b]:6d}  fold + next operands requested {s[b+3]-s[b+1]:6d}   (at {s[b+3]-t0:8d})")
            prev = s[b + 3]
    print(f"  window, outputs stored, step done             {s[5]-t0:8d}")
