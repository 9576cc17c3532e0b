"""Diagnostic: s_memtime stamps of wave 0, workgroup (0,0,0) inside one inner step (Y in, Y out) of the fused trace chain at
batch 256.  Needs the -DFK_STAMP build (FHERAM_LIB=...).  s_memtime ticks at 100 MHz."""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
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
    tick = 1.0   # raw ticks (the shader clock: divide by ~2.2 for ns)
    print(f"== rep {rep}: inner step of k_keyswitch_chain<3,4,3,true>, batch 256, wave 0 of workgroup 0; ns since the step began")
    print(f"  Y of both columns loaded, mask column staged  {(s[1]-t0)*tick:8.0f}")
    print(f"  barrier, gather through phi_g, digits         {(s[2]-t0)*tick:8.0f}")
    print(f"  three forward transforms                      {(s[3]-t0)*tick:8.0f}")
    print(f"  barrier, body column parked                   {(s[4]-t0)*tick:8.0f}")
    for ci in range(2):
        for q in range(4):
            b = 8 + (ci * 4 + q) * 4
            prev = s[4] if (ci == 0 and q == 0) else s[b - 1]
            print(f"  column {1-ci} limb {3-q}: (operand wait +) MAC {(s[b]-prev)*tick:6.0f}  inverse transform {(s[b+1]-s[b])*tick:6.0f}  fetch + body add {(s[b+2]-s[b+1])*tick:6.0f}  emit {(s[b+3]-s[b+2])*tick:6.0f}   (at {(s[b+3]-t0)*tick:8.0f})")
    print(f"  step done                                     {(s[5]-t0)*tick:8.0f}")

# the same for an inner product of the product chain (kind 1): stamps 0..3 prologue, then per (column, limb): MAC, inverse, post-step
for rep in range(2):
    ram.bench_chain(1, 256, 4, 5)
    st = (C.c_uint64 * 192)()
    L.fheram_debug_stamps(ram._h, st, 192)
    s = np.array([int(x) for x in st], dtype=np.int64)
    t0 = s[0]
    print(f"== rep {rep}: third product of k_ext_product_chain<3,4>, batch 256, wave 0 of workgroup 0; shader-clock ticks since the step began")
    print(f"  column 0 loaded, converted                    {s[1]-t0:8d}")
    print(f"  three forward transforms (column 0)           {s[2]-t0:8d}")
    print(f"  three forward transforms (column 1)           {s[3]-t0:8d}")
    for c in range(2):
        for q in range(4):
            b = 8 + (c * 4 + q) * 4
            prev = s[3] if (c == 0 and q == 0) else s[b - 2]
            print(f"  column {c} limb {3-q}: (operand waits +) 6 MACs {s[b]-prev:6d}  inverse transform {s[b+1]-s[b]:6d}  fetch + normalisation step {s[b+2]-s[b+1]:6d}   (at {s[b+2]-t0:8d})")
    print(f"  step done                                     {s[5]-t0:8d}")
