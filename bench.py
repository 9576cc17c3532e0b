#!/usr/bin/env python3
"""Benchmark of the FHE-RAM hot path on MI355X.

Workload (BASELINE.json configs[2]+[3]): Ram::read, Ram::read_prepare_write and Ram::write at
MAX_ADDR = 2^18, WORDSIZE = 4, default cryptographic parameters (N = 4096, base2k = 17).
One "step" = one read op + one write op on one RAM = `read` followed by
`read_prepare_write` + `write` (the reference times exactly these three calls,
/root/reference/examples/fhe-ram.rs:98-154).  `value` = RAM operations per second (2 per step),
whole job; read / rpw / write rates are reported beside it.

Inputs are synthetic (uniformly random normalised limbs for the RAM rows, the address digits, the
evaluation keys and the written words): the work of every kernel is data independent, and no
setup code is timed.  Everything is resident in HBM before the timed region starts.

N > 1 (`python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N`): one process
per GPU, each rank owns an independent 2^18 RAM (RAM instances are the independent units of this
path: an op needs `&mut Ram`); no data-path collective; weak scaling.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

N = 4096
GLWE_I64 = 3 * 2 * N * 8            # 196 608 B   
GGSW_I64 = 3 * 2 * 4 * 2 * N * 8    # 1 572 864 B
ATK_I64 = 3 * 4 * 2 * N * 8         # 786 432 B
EVK5_I64 = 4 * 5 * 2 * N * 8        # 1 310 720 B
HBM_PEAK_GBS = 8000.0               # MI355X_MICROARCH.md: 8 TB/s spec
FP64_VALU_PEAK_TINSTR = 39.3        # 78.6 TFLOP/s vector FP64 = 39.3 T FMA-class instr/s


def synth(rng, shape):
    return rng.integers(-(1 << 16), 1 << 16, size=shape, dtype=np.int64)


def op_counts(max_addr, ws, base2d):
    """EP / KS launches-worth of work per op, from the control flow."""
    rows = -(-max_addr // N)
    d = [len(b.d) for b in base2d.v]
    k = max(0, (rows - 1).bit_length())
    if len(d) == 1:
        ep_read = ws * d[0]
        ks_pack = 0
    else:
        ep_read = ws * (rows * d[0] + d[1])
        ks_pack = ws * (rows * (12 - k) + (rows - 1))
    ks_read = ks_pack + ws * 12
    ep_write = ep_read
    ks_write = ws * 12 + (ws * rows * 24 if len(d) == 2 else 0)
    return {"rows": rows, "ep_read": ep_read, "ks_read": ks_read, "ep_write": ep_write, "ks_write": ks_write,
            "ggsw_inv": sum(d) * 6}


def algorithmic_bytes(max_addr, ws, n_digits):
    """SURVEY.md §8(d) formulas (int64-limb ABI layout, every object counted once)."""
    rows = -(-max_addr // N)
    ram = ws * rows * GLWE_I64
    addr = n_digits * GGSW_I64
    read = ram + addr + 12 * ATK_I64 + ws * GLWE_I64
    rpw = 2 * ram + addr + 12 * ATK_I64 + ws * GLWE_I64
    write = 2 * ram + addr + 12 * ATK_I64 + 2 * EVK5_I64 + ws * GLWE_I64
    return read, rpw, write


def pmc_traffic_per_launch(kernel_prefix):
    """HBM bytes per launch of a kernel class from the committed PMC profile (None if absent)."""
    path = os.path.join(ROOT, "profiles", "r01_pmc_hbm_traffic.json")
    if not os.path.exists(path):
        return None
    tot, n = 0.0, 0
    for k, v in json.load(open(path)).items():
        if kernel_prefix in k and isinstance(v, dict) and "FETCH_SIZE_per_launch" in v and "WRITE_SIZE_per_launch" in v:
            tot += v["launches"] * (2.0 * v["FETCH_SIZE_per_launch"] + v["WRITE_SIZE_per_launch"]) * 1024.0
            if "_norm" not in k:      # the normalisation pass of a limb-parallel op belongs to the same launch of the class
                n += v["launches"]
    return tot / n if n else None


def cpu_baseline(max_addr, seed):
    """Times the oracle (CPU restatement, kind 'port') on a bounded sample of the same workload:
    ONE of the WORDSIZE sub-RAMs of the 2^18 RAM (sub-RAMs are processed one after the other by
    the reference, ram.rs:187-190), one read + one read_prepare_write + one write, single thread."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import pyoracle as po
    o = po.Oracle(po.OParams(max_addr=max_addr, word_size=1))
    p = o.p
    rng = np.random.default_rng(seed)
    evk = {"gal_els": np.array([int(po.lib().fo_galois_element(12, i)) for i in range(12)], dtype=np.int64),
           "atk_glwe": synth(rng, (12, p.atk_trace_len)), "atk_ggsw_inv": synth(rng, p.evk_inv_len),
           "tsk": synth(rng, p.evk_inv_len)}
    keys = o.keys_prepare(evk)
    addr = o.address_new(synth(rng, (o.n_digits, p.ggsw_len)))
    ram = o.ram_new()
    ram.load(synth(rng, (1, p.rows, p.glwe_len)))
    w = synth(rng, (1, p.glwe_len))
    t0 = time.perf_counter()
    ram.read(addr, keys)
    t1 = time.perf_counter()
    ram.read_prepare_write(addr, keys)
    t2 = time.perf_counter()
    ram.write(w, addr, keys)
    t3 = time.perf_counter()
    return t1 - t0, t2 - t1, t3 - t2


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--log-max-addr", type=int, default=18)
    ap.add_argument("--word-size", type=int, default=4)
    ap.add_argument("--mode", choices=["replicas", "sharded"], default="replicas",
                    help="N > 1: 'replicas' = one independent 2^log_max_addr RAM per GPU (default, weak scaling in RAM "
                         "ops/s); 'sharded' = ONE RAM of N * 2^log_max_addr entries, rows sharded over the GPUs "
                         "(BASELINE.json configs[4]); one RCCL all-gather per read, one broadcast per write")
    ap.add_argument("--dist-backend", default="nccl", help="nccl (= RCCL, default) or gloo (rehearsal on a 1-GPU box)")
    ap.add_argument("--all-ranks-device0", action="store_true", help="rehearsal: every rank uses GPU 0")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timing", action="store_true", help="do not bracket kernels with HIP events")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world != 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    dist = None
    if world > 1:
        import torch
        import torch.distributed as dist
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if args.all_ranks_device0:
            local_rank = 0
        torch.cuda.set_device(local_rank)
        kw = {"device_id": torch.device("cuda", local_rank)} if args.dist_backend == "nccl" else {}
        dist.init_process_group(args.dist_backend, rank=rank, world_size=world, **kw)   # nccl == RCCL on ROCm

    from _pkg import load_package
    pkg = load_package()
    max_addr, ws = 1 << args.log_max_addr, args.word_size
    sharded = args.mode == "sharded" and world > 1
    if sharded:
        max_addr *= world                      # weak scaling of ONE RAM: 2^log_max_addr entries per GPU
        ram = pkg.Ram(pkg.Parameters(max_addr=max_addr, word_size=ws), device=local_rank, shard=rank, n_shards=world)
    else:
        ram = pkg.Ram.new_from_ram_params(ws, [3, 3, 3, 3], max_addr, device=local_rank)
    p = ram.params
    rng = np.random.default_rng(1234 + rank)
    n_digits = p.base2d().as_1d().size()
    keys = pkg.EvaluationKeysPrepared(pkg.galois_elements(12), list(synth(rng, (12, 3 * 4 * 2 * N))),
                                      synth(rng, 4 * 5 * 2 * N), synth(rng, 4 * 5 * 2 * N))
    addr = pkg.Address(p, list(synth(rng, (n_digits, p.ggsw_len()))))
    ram.load_encrypted(synth(rng, (ws, ram.local_rows(), p.glwe_len())))
    ram.stage_words(synth(rng, (ws, p.glwe_len())))
    if sharded:
        from fheram_amd.sharded import ShardedRam, TorchComm
        sram = ShardedRam(ram, TorchComm(device_buffers=args.dist_backend == "nccl"), download=False)
        ops = (lambda: sram.read(addr, keys), lambda: sram.read_prepare_write(addr, keys), lambda: sram.write(None, addr, keys))
    else:
        ops = (lambda: ram.read(addr, keys, download=False),
               lambda: ram.read_prepare_write(addr, keys, download=False),
               lambda: ram.write(None, addr, keys))

    def step(timed):
        ts = []
        for fn in ops:
            ram.timer_begin()
            fn()
            ts.append(ram.timer_end())
        return ts

    def barrier():
        ram.sync()
        if dist is not None:
            import torch
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        step(False)
    barrier()
    t0 = time.perf_counter()
    per_op = [step(True) for _ in range(args.steps)]
    barrier()
    elapsed = time.perf_counter() - t0
    # Kernel-class durations for the roofline: the same K steps once more, now with every launch
    # bracketed by HIP events on its stream.  Those events break back-to-back submission and add
    # ~20 % to a step, so they are kept OUT of the timed region above.
    instr_elapsed = None
    if not args.no_kernel_timing:
        ram.profile_reset()
        ram.profile_enable(True)
        t1 = time.perf_counter()
        for _ in range(args.steps):
            step(True)
        barrier()
        instr_elapsed = time.perf_counter() - t1
        ram.profile_enable(False)
    if dist is not None:
        import torch
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if args.dist_backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    if rank != 0:
        if dist is not None:
            dist.destroy_process_group()
        return

    per_op = np.array(per_op)              # [steps][3] ms, HIP events on the context's stream
    read_ms, rpw_ms, write_ms = per_op.mean(axis=0)
    ms_per_step = elapsed * 1e3 / args.steps
    n_rams = 1 if sharded else world
    ops_per_s = n_rams * 2 * args.steps / elapsed
    a_read, a_rpw, a_write = algorithmic_bytes(max_addr, ws, n_digits)
    cnt = op_counts(max_addr, ws, p.base2d())

    out = {
        "metric": f"encrypted RAM read ops/s + write ops/s at 2^{args.log_max_addr} entries; achieved HBM GB/s vs peak",
        "value": ops_per_s, "unit": "RAM ops/s (1 read + 1 write[=rpw+write] per step)",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
        "higher_is_better": True, "scaling": "weak", "mode": args.mode, "vs_baseline": None, "dtype": "f64", "arithmetic": "exact integers mod 2^48+57345 carried in FP64 (error-free products); int32 limbs in HBM",
        "data": "synthetic",
        "config": {"workload": f"Ram::read + Ram::read_prepare_write + Ram::write, MAX_ADDR=2^{args.log_max_addr}, "
                               f"WORDSIZE={ws}, N=4096, base2k=17, rank=1 (BASELINE.json configs[2]+[3])",
                   "rams": n_rams, "rows_per_subram": cnt["rows"],
                   "parallelism": (f"1 RAM of 2^{args.log_max_addr}*{world} entries, rows sharded over {world} GPUs (RCCL all-gather/broadcast)"
                                   if sharded else f"{world} independent RAM(s), one per GPU")},
        "read_ops_s": n_rams * 1e3 / read_ms, "write_ops_s": n_rams * 1e3 / (rpw_ms + write_ms),
        "read_ms": read_ms, "read_prepare_write_ms": rpw_ms, "write_ms": write_ms,
        "algorithmic_GBs_per_op": {"read": a_read / read_ms / 1e6, "read_prepare_write": a_rpw / rpw_ms / 1e6,
                                   "write": a_write / write_ms / 1e6},
        "reference_published": {"read_ms": 450, "write_ms": 1200, "hw": "i9-12900K single thread (README.md:36)",
                                "speedup_read": 450.0 / read_ms, "speedup_write": 1200.0 / write_ms},
        "device": ram.device_info(),
    }

    if not args.no_kernel_timing:
        ks = ram.profile_get("keyswitch")
        ep = ram.profile_get("ext_product")
        pr = ram.profile_get("prepare")
        el = ram.profile_get("elementwise")
        # dominant kernel: the fused key-switch (trace step / packer combine).  Algorithmic bytes per
        # block = input GLWE + output GLWE at the ABI's int64 width; the key (786 432 B) once per launch.
        if ks["launches"]:
            avg_ms = ks["ms"] / ks["launches"]
            blocks = ks["blocks"] / ks["launches"]
            bytes_per_launch = blocks * 2 * GLWE_I64 + ATK_I64
            achieved = bytes_per_launch / avg_ms / 1e6
            out["roofline"] = {"kernel": "k_keyswitch (glwe_automorphism family: trace step / packer combine)",
                               "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                               "frac": achieved / HBM_PEAK_GBS, "traffic": pmc_traffic_per_launch("k_keyswitch"),
                               "traffic_source": "profiles/r01_pmc_hbm_traffic.json (separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE "
                                                 "passes of this command; bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024, launch-weighted mean)",
                               "avg_launch_ms": avg_ms, "avg_blocks_per_launch": blocks, "launches": ks["launches"],
                               "algorithmic_bytes_per_launch": bytes_per_launch}
            # companion VALU roofline: 11 transforms x 24576 butterflies x 8 FP64 ops + 24 x 4096 MACs x 7
            fp64_per_block = 11 * 24576 * 8 + 24 * 4096 * 7
            out["roofline_valu"] = {"bound": "valu_fp64", "achieved": ks["blocks"] * fp64_per_block / (ks["ms"] * 1e-3) / 1e12,
                                    "peak": FP64_VALU_PEAK_TINSTR, "unit": "T FP64 instr/s",
                                    "frac": ks["blocks"] * fp64_per_block / (ks["ms"] * 1e-3) / 1e12 / FP64_VALU_PEAK_TINSTR,
                                    "measured_issue_peak": 36.0,
                                    "measured_issue_peak_source": "profiles/r01_valu_rate.txt (tools/valu_rate.hip: mulmod chains sustain 1.8 ns per "
                                                                  "wave-instruction per SIMD at the 2.15 GHz clock of FP64 load)"}
        out["kernel_classes"] = {"keyswitch": ks, "ext_product": ep, "prepare": pr, "elementwise": el}
        out["kernel_timing_pass"] = {"what": "separate pass of the same K steps with per-launch HIP events on the launch stream "
                                             "(not part of the timed region: the events add this much to a step)",
                                     "ms_per_step_instrumented": instr_elapsed * 1e3 / args.steps}

    if not args.no_cpu_baseline and world == 1:   # reported baseline: rank 0 at N = 1 only
        r, q, w = cpu_baseline(max_addr, 99)
        # the sample is 1 of `ws` sub-RAMs: scale by ws (prepare_inv is shared, <1 % of a write)
        cpu_step_s = ws * (r + q + w)
        out["cpu_baseline"] = {"value": 2.0 / cpu_step_s, "unit": "RAM ops/s", "cores": 1, "kind": "port",
                               "sample": f"oracle (C++ exact-integer restatement), 1 of {ws} sub-RAMs of the 2^{args.log_max_addr} RAM: "
                                         f"read {r:.2f}s + read_prepare_write {q:.2f}s + write {w:.2f}s, scaled x{ws}",
                               "read_ms": ws * r * 1e3, "read_prepare_write_ms": ws * q * 1e3, "write_ms": ws * w * 1e3,
                               "host_cpu": open("/proc/cpuinfo").read().split("model name")[1].split("\n")[0].strip(": \t") if os.path.exists("/proc/cpuinfo") else "?"}
    print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
