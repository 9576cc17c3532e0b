"""Prediction of the multi-GPU curve from ONE GPU: for a RAM of 2^TOTAL entries row-sharded over G GPUs, the
ROOT shard's context is created alone on this GPU (`Ram(shard=0, n_shards=G)` — every shard runs the same kernels on its own
GPU, so its share is what it would be there) and timed with HIP events on its stream:

  T_part(G)   read_partial / read_prepare_write partial: coordinate 0's products + the shard's packing levels     (every shard)
  T_fin(G)    read_finish: the top log2 G packing levels, coordinate 1's products, the trace                        (root only)
  T_write(G)  write_begin + write_root + write_shard on the root (the other shards do write_begin + write_shard and
              wait for the root's ct_lo in between: the root is the critical path)

and the two exchanges are priced as latencies (all-gather of G x ws GLWEs of 98 304 B, broadcast of ws GLWEs: device-to-device
copies of < 4 MB over xGMI; XCHG_US each, default 25 us — the one number this model cannot measure here).

  read(G)  = T_part + xchg + T_fin        rpw(G) likewise        write(G) = T_write + xchg
  RAM ops/s = 2 / (read + rpw + write)

Strong scaling: TOTAL fixed (2^21: BASELINE.json configs[4]).  Weak scaling: 2^18 per GPU (TOTAL = G * 2^18), efficiency T(1)/T(G).
usage: python tools/scaling_model.py [--xchg-us 25] > profiles/r06_scaling_model.txt
`tools/scale_check.sh` prints measured / predicted against the committed file on a multi-GPU node."""
import argparse
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from _pkg import load_package   # noqa: E402
import bench                    # noqa: E402  (make_inputs: the synthetic operands of the bench)

ap = argparse.ArgumentParser()
ap.add_argument("--xchg-us", type=float, default=25.0)
ap.add_argument("--iters", type=int, default=12)
ap.add_argument("--json", default=None)
args = ap.parse_args()
pkg = load_package()
WS = 4


def measure(total_log, G):
    p = pkg.Parameters(max_addr=1 << total_log, word_size=WS)
    ram = pkg.Ram(p, device=0, shard=0, n_shards=G) if G > 1 else pkg.Ram.new_from_ram_params(WS, [3, 3, 3, 3], 1 << total_log)
    n_digits = p.base2d().as_1d().size()
    inp = bench.make_inputs(p, WS, 4, n_digits, ram.local_rows())
    keys = pkg.EvaluationKeysPrepared(pkg.galois_elements(12), list(inp["atk"]), inp["atk_inv"], inp["tsk"])
    addr = pkg.Address(p, list(inp["addr"]))
    ram.load_encrypted(inp["rows"])
    ram.stage_words(inp["words"])
    glen = p.glwe_len()

    def timed(fn):
        ram.timer_begin()
        fn()
        return ram.timer_end()

    res = {}
    if G == 1:
        ops = (lambda: ram.read(addr, keys, download=False), lambda: ram.read_prepare_write(addr, keys, download=False), lambda: ram.write(None, addr, keys))
        for _ in range(4):
            for f in ops:
                f()
        ts = np.array([[timed(f) for f in ops] for _ in range(args.iters)])
        r, q, w = np.median(ts, axis=0)
        res = {"part_read": r, "fin_read": 0.0, "part_rpw": q, "fin_rpw": 0.0, "write": w}
    else:
        part = ram.device_malloc(WS * glen * 4)
        gath = ram.device_malloc(G * WS * glen * 4)
        ctlo = ram.device_malloc(WS * glen * 4)
        # the gathered partials: G copies of this shard's (valid normalised limbs: timing only)
        import ctypes as C
        hip = C.CDLL("libamdhip64.so")
        hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]

        def fill_gather():
            for g in range(G):
                hip.hipMemcpy(C.c_void_p(gath + g * WS * glen * 4), C.c_void_p(part), WS * glen * 4, 3)

        def step():
            t = {}
            t["part_read"] = timed(lambda: ram.read_partial(addr, keys, False, out=(part, True)))
            ram.sync(); fill_gather()
            t["fin_read"] = timed(lambda: ram.read_finish(addr, keys, (gath, True), False, download=False))
            t["part_rpw"] = timed(lambda: ram.read_partial(addr, keys, True, out=(part, True)))
            ram.sync(); fill_gather()
            t["fin_rpw"] = timed(lambda: ram.read_finish(addr, keys, (gath, True), True, download=False))

            def wr():
                ram.write_begin(addr, keys)
                ram.write_root(None, addr, keys, out=(ctlo, True))
                ram.write_shard(addr, keys, (ctlo, True))
            t["write"] = timed(wr)
            return t
        for _ in range(3):
            step()
        ts = [step() for _ in range(args.iters)]
        res = {k: float(np.median([t[k] for t in ts])) for k in ts[0]}
        for b in (part, gath, ctlo):
            ram.device_free(b)
    del ram
    return res


x = args.xchg_us * 1e-3
rows = []
print(f"# one MI355X; ms; exchange priced at {args.xchg_us:.0f} us each (not measurable on one GPU); WORDSIZE 4, source constants")
print("# mode    total   G   T_part(read) T_fin(read) T_part(rpw) T_fin(rpw) T_write   read    rpw     write   step    RAM ops/s  speed-up / efficiency")
base = {}
for mode, totals in (("strong", [(21, g) for g in (1, 2, 4, 8)]), ("strong", [(18, g) for g in (1, 2, 4, 8)]), ("weak", [(18 + k, 1 << k) for k in range(4)])):
    for total, G in totals:
        m = measure(total, G)
        xx = 0.0 if G == 1 else x
        read = m["part_read"] + xx + m["fin_read"]
        rpw = m["part_rpw"] + xx + m["fin_rpw"]
        write = m["write"] + xx
        step = read + rpw + write
        key = (mode, totals[0][0])
        if G == 1:
            base[key] = step
        eff = base[key] / step
        rows.append({"mode": mode, "total_log_max_addr": total, "gpus": G, **m, "read_ms": read, "rpw_ms": rpw, "write_ms": write, "ms_per_step": step,
                     "ram_ops_s": 2e3 / step, ("speedup" if mode == "strong" else "weak_efficiency"): eff})
        print(f"{mode:7s}  2^{total:<4d} {G:2d}   {m['part_read']:9.3f}   {m['fin_read']:9.3f}   {m['part_rpw']:9.3f}  {m['fin_rpw']:9.3f}  {m['write']:7.3f}  "
              f"{read:6.3f}  {rpw:6.3f}  {write:6.3f}  {step:6.3f}  {2e3 / step:9.1f}  {eff:6.2f}{'x' if mode == 'strong' else ' (T(1)/T(G))'}", flush=True)
if args.json:
    json.dump({"xchg_us": args.xchg_us, "rows": rows}, open(args.json, "w"), indent=1)
