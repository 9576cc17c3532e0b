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
per GPU and ONE RAM whose rows are sharded over the ranks (BASELINE.json configs[4]):
one RCCL all-gather per read, one broadcast per write.  Default = weak scaling, 2^18 entries per GPU
(N = 8 is exactly configs[4]: MAX_ADDR = 2^21 over 8 GPUs); `value` then counts an op on the
N*2^18-entry RAM as N ops of the metric's size, so that value(N) / (N * value(1)) = T(1) / T(N) is the
weak-scaling efficiency.  `--total-log-max-addr 21` fixes the RAM instead (strong scaling, value = raw
RAM ops/s).  `--mode replicas` = N independent 2^18 RAMs, no collective (opt-in).

`--workload ep` = BASELINE.json configs[1]: one GLWE x GGSW external product (latency) and a
batch-256 launch (throughput).
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
# FP64 VALU instructions of the shipped arithmetic (csrc/fft_dev.hpp: negacyclic FFT64 on N/2 = 2048 complex points): a forward
# butterfly is 6 FMA-class instructions, an inverse one 8, the rounding of an inverse transform's output 1 per coefficient, a
# complex multiply-accumulate against a prepared operand 4
FP64_FWD = 11 * 1024 * 6
FP64_INV = 11 * 1024 * 8 + 4096
FP64_MAC = 2048 * 4


def fp64_per_keyswitch(s_evk):
    """trace / packer key-switch with an s_evk-limb key: 3 forward, 2 s_evk inverse transforms, 3 * 2 s_evk polynomial MACs"""
    return 3 * FP64_FWD + 2 * s_evk * FP64_INV + 3 * 2 * s_evk * FP64_MAC


FP64_PER_KS = fp64_per_keyswitch(4)
FP64_PER_EP = 6 * FP64_FWD + 8 * FP64_INV + 48 * FP64_MAC     # external product: 6 forward, 8 inverse transforms, 48 polynomial MACs
# ALGORITHMIC flops (what any FFT-based implementation of the same products needs, not this implementation's instruction stream):
# 5 n log2 n per transform on n = 2048 complex points, 8 per complex multiply-accumulate
ALG_FLOP_TRANSFORM = 5 * 2048 * 11
ALG_FLOP_MAC = 8 * 2048
ALG_FLOP_PER_EP = 14 * ALG_FLOP_TRANSFORM + 48 * ALG_FLOP_MAC


def alg_flop_per_keyswitch(s_evk):
    return (3 + 2 * s_evk) * ALG_FLOP_TRANSFORM + 6 * s_evk * ALG_FLOP_MAC


# round 4's arithmetic (48-bit prime in FP64: 8 instructions per modular butterfly on 4096 points, 7 per MAC), for comparison only
R04_FP64_PER_KS = 11 * 24576 * 8 + 24 * 4096 * 7
R04_FP64_PER_EP = 14 * 24576 * 8 + 48 * 4096 * 7
# SURVEY.md 8(d)'s own unit: "modmuls" (a butterfly of an N-point transform = 1, N/2 log2 N = 24 576 per transform; a pointwise MAC
# = 4096 per polynomial pair): EP = 14 transforms + 48 MACs, trace / pack key-switch = (3 + 2 s) transforms + 6 s MACs
MODMUL_PER_EP = 14 * 24576 + 48 * 4096


def modmul_per_keyswitch(s_evk):
    return (3 + 2 * s_evk) * 24576 + 6 * s_evk * 4096


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


def algorithmic_bytes(max_addr, ws, n_digits, atk_i64=ATK_I64):
    """SURVEY.md §8(d) formulas (int64-limb ABI layout, every object counted once)."""
    rows = -(-max_addr // N)
    ram = ws * rows * GLWE_I64
    addr = n_digits * GGSW_I64
    read = ram + addr + 12 * atk_i64 + ws * GLWE_I64
    rpw = 2 * ram + addr + 12 * atk_i64 + ws * GLWE_I64
    write = 2 * ram + addr + 12 * atk_i64 + 2 * EVK5_I64 + ws * GLWE_I64
    return read, rpw, write


def _first_existing(*names):
    for n in names:
        if os.path.exists(os.path.join(ROOT, n)):
            return n
    return names[0]


PMC_PROFILE = _first_existing("profiles/r06_pmc_hbm_traffic.json", "profiles/r05_pmc_hbm_traffic.json")
SQ_PROFILE = _first_existing("profiles/r06_pmc_sq_summary.txt", "profiles/r05_pmc_sq_summary.txt")


def pmc_profile():
    path = os.path.join(ROOT, PMC_PROFILE)
    return json.load(open(path)) if os.path.exists(path) else None


def pmc_traffic(kernel_prefix):
    """HBM-side bytes per launch (read + written) of the kernel(s) whose name starts with `kernel_prefix` (a string or a tuple of them:
    the two register-budget variants of a chain kernel are one launch class), from the committed PMC profile of this command
    (tools/pmc_hbm.py: separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, FETCH_SIZE calibrated); launches of the same kernel on
    a token grid (the predicated fallback behind the trace tail) are left out; launch-weighted mean over the rest; None if absent"""
    prof = pmc_profile()
    if not prof:
        return None
    prefixes = (kernel_prefix,) if isinstance(kernel_prefix, str) else tuple(kernel_prefix)
    tot = n = 0
    for name, e in prof.get("kernels", {}).items():
        if name.startswith(prefixes) and "read_bytes_per_launch" in e and "write_bytes_per_launch" in e:
            b = e["read_bytes_per_launch"] + e["write_bytes_per_launch"]
            if b < (1 << 20):
                continue
            tot += b * e["launches"]
            n += e["launches"]
    return tot / n if n else None


def pmc_valu_busy(kernel_label):
    """VALU busy fraction of a kernel from the committed SQ-counter summary (None if absent)"""
    path = os.path.join(ROOT, SQ_PROFILE)
    if not os.path.exists(path):
        return None
    import re
    name = kernel_label.split("<")[0]
    for line in open(path):
        if ("fk::" + name + "<") in line:
            m = re.search(r"ACTIVE_INST_VALU=[0-9.e+]+\(([0-9.]+)%\)", line)
            if m:
                return 2 * float(m.group(1)) / 100.0
    return None


def make_inputs(p, ws, s_evk, n_digits, rows_local, seed_keys=1234, seed_rows=4321):
    """The synthetic inputs of a run: evaluation keys, address digits, RAM rows, the words of the write.  ONE function for the
    GPU legs and the oracle leg (identical inputs, examples/fhe-ram.rs:98-115 checks the result it has just timed)."""
    rng = np.random.default_rng(seed_keys)
    inp = {"atk": synth(rng, (12, 3 * s_evk * 2 * N)), "atk_inv": synth(rng, 4 * 5 * 2 * N), "tsk": synth(rng, 4 * 5 * 2 * N),
           "addr": synth(rng, (n_digits, p.ggsw_len()))}
    rng = np.random.default_rng(seed_rows)
    inp["rows"] = synth(rng, (ws, rows_local, p.glwe_len()))
    inp["words"] = synth(rng, (ws, p.glwe_len()))
    return inp


def oracle_native():
    """the CPU leg is built for the host it runs on.  liboracle.so is built portable (-march=x86-64-v2: it
    travels from the build container to the GPU box); this builds the same sources -march=native on THIS machine and
    makes pyoracle load that.  Returns the flags in use."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import subprocess
    if os.environ.get("FO_LIB"):
        return "FO_LIB=" + os.environ["FO_LIB"]
    try:
        subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "_native/liboracle_native.so"], check=True, timeout=300,
                       stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        os.environ["FO_LIB"] = os.path.join(ROOT, "oracle", "_native", "liboracle_native.so")
        return "-O3 -march=native (built on this host)"
    except Exception as e:   # no compiler on the box: the portable build
        return f"-O3 -march=x86-64-v2 (native build failed: {type(e).__name__})"


def cpu_baseline(max_addr, ws, inp, threads, crypto=None, sub_ram=None):
    """Times the oracle (CPU restatement, kind 'port') on the SAME inputs as the GPU legs: one read + one
    read_prepare_write + one write.  threads == 1: sequential, as the reference (ram.rs:187-190); sub_ram = i: only
    sub-RAM i (beyond 2^18; scaled by the caller).  threads > 1: the oracle's OpenMP variant (sub-RAMs and the rows of
    the per-row loops in parallel).  Returns the three times and the three outputs (result of read, of
    read_prepare_write, rows after write)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import pyoracle as po
    o = po.Oracle(po.OParams(max_addr=max_addr, word_size=ws if sub_ram is None else 1, **(crypto or {}))).set_threads(threads)
    evk = {"gal_els": np.array([int(po.lib().fo_galois_element(12, i)) for i in range(12)], dtype=np.int64),
           "atk_glwe": inp["atk"], "atk_ggsw_inv": inp["atk_inv"], "tsk": inp["tsk"]}
    keys = o.keys_prepare(evk)
    addr = o.address_new(inp["addr"])
    ram = o.ram_new()
    sel = slice(None) if sub_ram is None else slice(sub_ram, sub_ram + 1)
    ram.load(np.ascontiguousarray(inp["rows"][sel]))
    w = np.ascontiguousarray(inp["words"][sel])
    t0 = time.perf_counter()
    r = ram.read(addr, keys)
    t1 = time.perf_counter()
    q = ram.read_prepare_write(addr, keys)
    t2 = time.perf_counter()
    ram.write(w, addr, keys)
    t3 = time.perf_counter()
    return (t1 - t0, t2 - t1, t3 - t2), {"read": r, "rpw": q, "rows_after_write": ram.store()}


def readme_leg(pkg, ws, log_max_addr, device, steps, warmup):
    """The README parameter block (README.md:17-27: K_PT = 9, K_EVK = 85, 5-limb trace keys) on a context of its own: the block
    the reference's published 450 / 1200 ms were taken with (README.md:36).  Short: per-op HIP-event times over `steps` steps."""
    crypto = {"k_glwe_pt": 9, "k_evk_trace": 85}
    ram = pkg.Ram.new_from_ram_params(ws, [3, 3, 3, 3], 1 << log_max_addr, device=device, **crypto)
    p = ram.params
    inp = make_inputs(p, ws, 5, p.base2d().as_1d().size(), ram.local_rows(), 1234, 4321)
    keys = pkg.EvaluationKeysPrepared(pkg.galois_elements(12), list(inp["atk"]), inp["atk_inv"], inp["tsk"])
    addr = pkg.Address(p, list(inp["addr"]))
    ram.load_encrypted(inp["rows"])
    ram.stage_words(inp["words"])
    ops = (lambda: ram.read(addr, keys, download=False), lambda: ram.read_prepare_write(addr, keys, download=False), lambda: ram.write(None, addr, keys))
    for _ in range(warmup):
        for fn in ops:
            fn()
            ram.sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        for fn in ops:
            fn()
            ram.sync()
    elapsed = time.perf_counter() - t0
    ts = []
    for _ in range(steps):
        row = []
        for fn in ops:
            ram.timer_begin()
            fn()
            row.append(ram.timer_end())
        ts.append(row)
    r, q, w = np.array(ts).mean(axis=0)
    return {"read_ms": float(r), "read_prepare_write_ms": float(q), "write_ms": float(w), "ms_per_step": elapsed * 1e3 / steps,
            "ram_ops_s": 2 * steps / elapsed, "steps": steps, "roundoff_max": ram.roundoff_max()}


def sha(a):
    import hashlib
    return hashlib.sha256(np.ascontiguousarray(a, dtype=np.int64).tobytes()).hexdigest()


def host_cpu():
    try:
        return open("/proc/cpuinfo").read().split("model name")[1].split("\n")[0].strip(": \t")
    except Exception:
        return "?"


def host_cores():
    """threads of the all-core CPU baseline: every core this process may use (scheduler affinity, then the cgroup's CPU
    quota if it is smaller); FHERAM_CPU_THREADS overrides"""
    if os.environ.get("FHERAM_CPU_THREADS"):
        return int(os.environ["FHERAM_CPU_THREADS"])
    try:
        n = len(os.sched_getaffinity(0))
    except Exception:
        n = os.cpu_count() or 1
    try:   # cgroup v2 quota ("max 100000" = none)
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = max(1, min(n, int(int(q) / int(per))))
    except Exception:
        pass
    return n


def spawn_ranks(args):
    """`python bench.py --gpus N` (N > 1) without torch.distributed.run around it: run that launcher as a child process
    (one rank per GPU over RCCL), pass its output through and return its exit code."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    r = subprocess.run(cmd, env=env)
    if r.returncode != 0:
        print(f"bench.py: the {args.gpus}-rank run failed (exit code {r.returncode}); no 1-GPU fallback", file=sys.stderr)
    return r.returncode


def bench_ep(pkg, args):
    """BASELINE.json configs[1]: single GLWE x GGSW external product at N = 4096."""
    ram = pkg.Ram.new_from_ram_params(4, [3, 3, 3, 3], 1 << 12)
    cus = ram.device_info()["compute_units"]
    for _ in range(args.warmup):
        ram.bench_external_product(1, 8)
    lat = [ram.bench_external_product(1, 64) / 64 for _ in range(args.steps)]
    thr = [ram.bench_external_product(cus, 32) / 32 for _ in range(args.steps)]
    big = [ram.bench_external_product(8 * cus, 8) / 8 for _ in range(args.steps)]
    lat_ms, thr_ms, big_ms = float(np.median(lat)), float(np.median(thr)), float(np.median(big))
    ep_bytes = 2 * GLWE_I64 + GGSW_I64            # a + G + res, int64-limb layout
    fp64_per_ep = FP64_PER_EP
    out = {"metric": "GLWE x GGSW external products per second at N=4096 (BASELINE.json configs[1])",
           "value": cus / (thr_ms * 1e-3), "unit": "external products/s (batch = #CUs per launch)",
           "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "ms_per_step": thr_ms, "higher_is_better": True,
           "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
           "config": {"workload": "glwe_external_product, N=4096, base2k=17, rank=1, GLWE size 3, GGSW size 4 / dnum 3"},
           "latency_us_single_product": lat_ms * 1e3,
           "batch_launch_us": {str(cus): thr_ms * 1e3, str(8 * cus): big_ms * 1e3},
           "amortised_us_per_product": {str(cus): thr_ms * 1e3 / cus, str(8 * cus): big_ms * 1e3 / (8 * cus)},
           "roofline": {"kernel": "k_ext_product<3,4,2,0> (one workgroup per product; FFT64 arithmetic: 1 552 384 FP64 instructions per product)", "bound": "valu_fp64",
                        "achieved": 2 * cus * fp64_per_ep / (thr_ms * 1e-3) / 1e12, "peak": 2 * FP64_VALU_PEAK_TINSTR, "unit": "TFLOP/s",
                        "frac": cus * fp64_per_ep / (thr_ms * 1e-3) / 1e12 / FP64_VALU_PEAK_TINSTR, "traffic": None,
                        "note": "every FP64 VALU instruction priced as one FMA slot (2 FLOP)"},
           "roofline_hbm": {"bound": "hbm", "achieved": (cus * GLWE_I64 + GGSW_I64) / thr_ms / 1e6, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                            "frac": (cus * GLWE_I64 + GGSW_I64) / thr_ms / 1e6 / HBM_PEAK_GBS,
                            "what": "device layout: int32 limbs in and out (= GLWE_I64 bytes per product), f64 GGSW once",
                            "abi_layout_int64": {"achieved": (cus * 2 * GLWE_I64 + GGSW_I64) / thr_ms / 1e6,
                                                 "frac": (cus * 2 * GLWE_I64 + GGSW_I64) / thr_ms / 1e6 / HBM_PEAK_GBS,
                                                 "algorithmic_bytes_single_product": ep_bytes}},
           "device": ram.device_info()}
    if not args.no_cpu_baseline:
        flags = oracle_native()                    # the CPU leg is built for the host it runs on, as the main baseline's is
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import pyoracle as po
        o = po.Oracle(po.OParams(max_addr=1 << 12))
        rng = np.random.default_rng(5)
        a, g = synth(rng, o.p.glwe_len), synth(rng, o.p.ggsw_len)
        o.glwe_external_product(a, g)
        t0 = time.perf_counter()
        n_cpu = 20
        for _ in range(n_cpu):
            o.glwe_external_product(a, g)          # includes the GGSW prepare (48 forward transforms) every call
        dt = (time.perf_counter() - t0) / n_cpu
        out["cpu_baseline"] = {"value": 1.0 / dt, "unit": "external products/s", "cores": 1, "kind": "port",
                               "sample": f"oracle glwe_external_product incl. GGSW prepare, {n_cpu} calls, {dt * 1e3:.2f} ms each",
                               "host_cpu": host_cpu(), "build": flags}
    print(json.dumps(out))


def bench_group(args):
    """ONE process, N GPUs: the row-sharded RAM behind the native group handle (fheram_group_*, include/fheram.h)."""
    from _pkg import load_package
    pkg = load_package()
    n, ws = args.gpus, args.word_size
    crypto = {"k_glwe_pt": 9, "k_evk_trace": 85} if args.params == "readme" else {}
    s_evk = 5 if crypto else 4
    strong = args.total_log_max_addr is not None
    max_addr = (1 << args.total_log_max_addr) if strong else (1 << args.log_max_addr) * n
    params = pkg.Parameters(max_addr=max_addr, word_size=ws, **crypto)
    grp = pkg.GroupRam(params, [0] * n if args.all_ranks_device0 else list(range(n)))
    rng = np.random.default_rng(1234)
    n_digits = params.base2d().as_1d().size()
    keys = pkg.EvaluationKeysPrepared(pkg.galois_elements(12), list(synth(rng, (12, 3 * s_evk * 2 * N))),
                                      synth(rng, 4 * 5 * 2 * N), synth(rng, 4 * 5 * 2 * N))
    addr = pkg.Address(params, list(synth(rng, (n_digits, params.ggsw_len()))))
    grp.load_encrypted(synth(np.random.default_rng(4321), (ws, params.rows(), params.glwe_len())))
    grp.stage_words(synth(rng, (ws, params.glwe_len())))
    ops = (lambda: grp.read(addr, keys, download=False), lambda: grp.read_prepare_write(addr, keys, download=False),
           lambda: grp.write(None, addr, keys))

    def step():
        ts = []
        for fn in ops:
            a = time.perf_counter()
            fn()                               # group ops are synchronous: complete on every shard when they return
            ts.append((time.perf_counter() - a) * 1e3)
        return ts
    for _ in range(args.warmup):
        step()
    t0 = time.perf_counter()
    per_op = np.array([step() for _ in range(args.steps)])
    elapsed = time.perf_counter() - t0
    read_ms, rpw_ms, write_ms = per_op.mean(axis=0)
    weight = 1 if strong else n
    log_entries = int(np.log2(max_addr))
    print(json.dumps({
        "metric": f"encrypted RAM read ops/s + write ops/s at 2^{args.log_max_addr} entries; achieved HBM GB/s vs peak",
        "value": 2 * args.steps / elapsed,
        "unit": "RAM ops/s (1 read + 1 write[=rpw+write] per step)" + (
            f" of the ONE 2^{log_entries}-entry RAM sharded over the {n} GPUs (weak scaling: 2^{args.log_max_addr} entries per GPU; "
            f"value(N) / value(1) = T(1) / T(N) is the weak-scaling efficiency)" if weight > 1 else ""),
        "n_gpus": n, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed * 1e3 / args.steps, "higher_is_better": True,
        "scaling": "strong" if strong else "weak", "mode": "group", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": f"Ram::read + Ram::read_prepare_write + Ram::write, MAX_ADDR=2^{log_entries}, WORDSIZE={ws}, N=4096, base2k=17, rank=1"
                               " (BASELINE.json configs[2]+[3]; configs[4] sharding)",
                   "parallelism": f"1 process, {n} GPUs: 1 RAM of 2^{log_entries} entries, rows r = g (mod {n}) on GPU g, one host thread per GPU; "
                                  "peer-to-peer copies: partial packs to the root per read, ct_lo to every shard per write (fheram_group_*)"},
        "ram_ops_s_raw": 2 * args.steps / elapsed,
        "ops_weighted_by_ram_size": None if weight == 1 else 2 * args.steps / elapsed * weight,
        "read_ms": read_ms, "read_prepare_write_ms": rpw_ms, "write_ms": write_ms,
        "timing": "host wall clock per synchronous group call"}))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", choices=["ram", "ep"], default="ram",
                    help="ram = Ram::read / read_prepare_write / write (BASELINE.json configs[2..4], default); "
                         "ep = single external-product microbenchmark (configs[1])")
    ap.add_argument("--log-max-addr", type=int, default=18, help="log2 of the RAM entries PER GPU (weak scaling)")
    ap.add_argument("--total-log-max-addr", type=int, default=None,
                    help="log2 of the entries of the ONE RAM sharded over all GPUs, fixed as N grows (strong scaling; "
                         "BASELINE.json configs[4] = 21)")
    ap.add_argument("--word-size", type=int, default=4)
    ap.add_argument("--params", choices=["source", "readme"], default="source",
                    help="cryptographic parameter block: 'source' = the constants of src/parameters.rs:11-18 (default, BASELINE.json's "
                         "'default params'); 'readme' = the block of README.md:17-27 the published 450 / 1200 ms were taken with "
                         "(K_PT = 9, K_EVK = 85: 5-limb trace keys)")
    ap.add_argument("--mode", choices=["auto", "sharded", "replicas", "group"], default="auto",
                    help="N > 1: 'sharded' (default) = ONE RAM, rows sharded over the GPUs, one process per GPU, one RCCL all-gather "
                         "per read and one broadcast per write; 'replicas' = one independent RAM per GPU, no collective; 'group' = the "
                         "same sharded RAM driven by ONE process through the native fheram_group_* C ABI (one host thread per GPU, "
                         "peer-to-peer copies instead of a collective)")
    ap.add_argument("--dist-backend", default="nccl", help="nccl (= RCCL, default) or gloo (rehearsal on a 1-GPU box)")
    ap.add_argument("--all-ranks-device0", action="store_true", help="rehearsal: every rank uses GPU 0")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timing", action="store_true", help="do not bracket kernels with HIP events")
    ap.add_argument("--no-boundary", action="store_true", help="skip the pass that includes the host hand-over")
    ap.add_argument("--no-readme-leg", action="store_true", help="skip the short pass on the README parameter block (reference_published)")
    args = ap.parse_args()

    if args.mode == "group":
        return bench_group(args)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ and "RANK" not in os.environ:
        # Started without a launcher: start the N ranks ourselves, as fresh child processes, BEFORE this process has
        # touched the GPU (nothing above imports torch or loads the HIP library), relay rank 0's JSON line and exit with
        # the launcher's code.  Never a silent 1-GPU run.
        raise SystemExit(spawn_ranks(args))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: refusing to measure another job than the one asked for")
    mode = args.mode if args.mode != "auto" else ("sharded" if world > 1 else "single")
    # the sharded path can be rehearsed with ONE rank (RCCL + device buffers + event hand-over) under torch.distributed.run
    use_dist = world > 1 or (mode == "sharded" and "RANK" in os.environ)
    if mode == "sharded" and not use_dist:
        mode = "single"
    dist = None
    if use_dist:
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
    if args.workload == "ep":
        if rank == 0:
            bench_ep(pkg, args)
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()
        return

    ws = args.word_size
    crypto = {"k_glwe_pt": 9, "k_evk_trace": 85} if args.params == "readme" else {}
    s_evk = 5 if args.params == "readme" else 4          # limbs of a trace / packing key
    atk_i64 = 3 * s_evk * 2 * N * 8
    fp64_per_ks = fp64_per_keyswitch(s_evk)
    sharded = mode == "sharded"
    strong = sharded and args.total_log_max_addr is not None
    if strong:
        max_addr = 1 << args.total_log_max_addr
    elif sharded:
        max_addr = (1 << args.log_max_addr) * world      # weak scaling of ONE RAM: 2^log_max_addr entries per GPU
    else:
        max_addr = 1 << args.log_max_addr
    if sharded:
        ram = pkg.Ram(pkg.Parameters(max_addr=max_addr, word_size=ws, **crypto), device=local_rank, shard=rank, n_shards=world)
    else:
        ram = pkg.Ram.new_from_ram_params(ws, [3, 3, 3, 3], max_addr, device=local_rank, **crypto)
    p = ram.params
    n_digits = p.base2d().as_1d().size()
    # sharded: every rank derives the same keys and address, and its own rows; replicas: a RAM of its own per rank
    inp = make_inputs(p, ws, s_evk, n_digits, ram.local_rows(), 1234 + (0 if (sharded or mode == "single") else rank), 4321 + rank)
    keys = pkg.EvaluationKeysPrepared(pkg.galois_elements(12), list(inp["atk"]), inp["atk_inv"], inp["tsk"])
    addr = pkg.Address(p, list(inp["addr"]))
    ram.load_encrypted(inp["rows"])
    words = inp["words"]
    ram.stage_words(words)
    if sharded:
        from fheram_amd.sharded import ShardedRam, TorchComm
        sram = ShardedRam(ram, TorchComm(device_buffers=args.dist_backend == "nccl"), download=False)
        ops = (lambda: sram.read(addr, keys), lambda: sram.read_prepare_write(addr, keys), lambda: sram.write(None, addr, keys))
    else:
        ops = (lambda: ram.read(addr, keys, download=False),
               lambda: ram.read_prepare_write(addr, keys, download=False),
               lambda: ram.write(None, addr, keys))

    def step():
        ts = []
        for fn in ops:
            ram.timer_begin()
            fn()
            ts.append(ram.timer_end())     # HIP events on the context's stream; the host waits for the op here
        return ts

    def barrier():
        ram.sync()
        if dist is not None:
            import torch
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    # the timed region: the three calls of a step, the host waiting for each op before it makes the next call (one round trip
    # per op, as the reference's synchronous calls have) — and nothing else: the per-op split comes from a pass of its own
    t0 = time.perf_counter()
    for _ in range(args.steps):
        for fn in ops:
            fn()
            ram.sync()
    barrier()
    elapsed = time.perf_counter() - t0
    tp0 = time.perf_counter()
    per_op = [step() for _ in range(args.steps)]       # the same K steps with one HIP-event pair per op on the context's stream
    barrier()
    per_op_elapsed = time.perf_counter() - tp0
    tail = ram.tail_stats()       # single-launch trace chains so far (warm-up + timed steps) and how many fell back
    mid = ram.mid_stats() if hasattr(ram, "mid_stats") else {"launches": 0, "fallbacks": 0}
    # For information only (never `value`): the same K steps enqueued back to back — with a NULL result pointer the ABI's
    # ops return as soon as they are enqueued — and ONE synchronisation at the end, i.e. without the host's round trip
    # between ops that the synchronous calls of the reference interface imply.
    pipelined = None
    if not sharded:
        barrier()
        tp = time.perf_counter()
        for _ in range(args.steps):
            for fn in ops:
                fn()
        barrier()
        pipelined = time.perf_counter() - tp
    # Kernel-class durations for the roofline: the same K steps once more, now with every launch
    # bracketed by HIP events on its stream.  Those events break back-to-back submission and add
    # ~20 % to a step, so they are kept OUT of the timed region above.
    instr_elapsed = None
    light = None
    if not args.no_kernel_timing:
        # first the dominant kernel alone: ONE event pair around each chain launch (8 records per step instead of ~200),
        # everything else enqueued as in the timed region — its step time is reported beside the timed one
        ram.profile_reset()
        ram.profile_enable(2)
        t1 = time.perf_counter()
        for _ in range(args.steps):
            step()
        barrier()
        light = {"elapsed": time.perf_counter() - t1, "chain": ram.profile_get("keyswitch_chain_launch")}
        ram.profile_enable(False)
        ram.profile_reset()
        ram.profile_enable(True)
        t1 = time.perf_counter()
        for _ in range(args.steps):
            step()
        barrier()
        instr_elapsed = time.perf_counter() - t1
        ram.profile_enable(False)
    # The same ops through the boundary's HOST buffers (int64 result out, int64 words in): host wall clock
    # per call.  Never part of `value`.
    boundary = None
    if not args.no_boundary and not sharded:
        bt = []
        res_buf = np.zeros((ws, p.glwe_len()), dtype=np.int64)      # the caller's result buffer, reused (as a host loop would)
        for _ in range(max(5, min(args.steps, 20))):
            ts = []
            for fn in (lambda: ram.read(addr, keys, out=res_buf), lambda: ram.read_prepare_write(addr, keys, out=res_buf),
                       lambda: ram.write(words, addr, keys)):
                a = time.perf_counter()
                fn()
                ram.sync()
                ts.append((time.perf_counter() - a) * 1e3)
            bt.append(ts)
        boundary = np.median(np.array(bt), axis=0)
        ram.stage_words(words)
    if dist is not None:
        import torch
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if args.dist_backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # Weak scaling of ONE sharded RAM: the reference point is the same op on an unsharded RAM of the per-GPU size, measured HERE,
    # on rank 0's GPU, in this run (the other ranks wait): weak_scaling_efficiency = T(1) / T(N).  `value` stays the raw ops/s
    # of the RAM that was actually operated (whose size grows with N).
    weak_ref = None
    if sharded and not strong and world > 1:
        if rank == 0:
            r1 = pkg.Ram.new_from_ram_params(ws, [3, 3, 3, 3], 1 << args.log_max_addr, device=local_rank, **crypto)
            p1 = r1.params
            i1 = make_inputs(p1, ws, s_evk, p1.base2d().as_1d().size(), r1.local_rows(), 1234, 4321)
            k1 = pkg.EvaluationKeysPrepared(pkg.galois_elements(12), list(i1["atk"]), i1["atk_inv"], i1["tsk"])
            a1 = pkg.Address(p1, list(i1["addr"]))
            r1.load_encrypted(i1["rows"])
            r1.stage_words(i1["words"])
            ops1 = (lambda: r1.read(a1, k1, download=False), lambda: r1.read_prepare_write(a1, k1, download=False), lambda: r1.write(None, a1, k1))
            for _ in range(max(3, args.warmup)):
                for fn in ops1:
                    fn()
                    r1.sync()
            t1 = time.perf_counter()
            n1 = max(5, min(args.steps, 20))
            for _ in range(n1):
                for fn in ops1:
                    fn()
                    r1.sync()
            weak_ref = (time.perf_counter() - t1) * 1e3 / n1
            del r1
        if dist is not None:
            dist.barrier()

    if rank != 0:
        if dist is not None:
            dist.destroy_process_group()
        return

    per_op = np.array(per_op)              # [steps][3] ms, HIP events on the context's stream
    read_ms, rpw_ms, write_ms = per_op.mean(axis=0)
    ms_per_step = elapsed * 1e3 / args.steps
    n_rams = 1 if mode != "replicas" else world
    raw_ops_per_s = n_rams * 2 * args.steps / elapsed
    # weak scaling of ONE sharded RAM: an op on N*2^k entries counts as N ops of the 2^k-entry metric
    weight = world if (sharded and not strong) else 1
    log_entries = int(np.log2(max_addr)) if max_addr & (max_addr - 1) == 0 else float(np.log2(max_addr))
    a_read, a_rpw, a_write = algorithmic_bytes(max_addr, ws, n_digits, atk_i64)
    cnt = op_counts(max_addr, ws, p.base2d())
    if mode == "replicas":
        par = f"{world} independent RAMs of 2^{args.log_max_addr} entries, one per GPU, no data-path collective"
    elif sharded:
        par = (f"1 RAM of 2^{log_entries} entries, rows r = g (mod {world}) on GPU g; one RCCL all-gather per read, one "
               f"broadcast per write ({'strong' if strong else 'weak'} scaling)")
    else:
        par = "1 RAM on 1 GPU"

    out = {
        "metric": f"encrypted RAM read ops/s + write ops/s at 2^{args.log_max_addr} entries; achieved HBM GB/s vs peak",
        "value": raw_ops_per_s,
        "value_incl_boundary": None,     # filled below: the same steps with the result handed to the host (what the reference's calls return)
        "unit": "RAM ops/s (1 read + 1 write[=rpw+write] per step)" + (
            f" of the ONE 2^{log_entries}-entry RAM sharded over the {world} GPUs (weak scaling: 2^{args.log_max_addr} entries per GPU; "
            f"value(N) / value(1) = T(1) / T(N) is the weak-scaling efficiency)" if weight > 1 else ""),
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
        "higher_is_better": True, "scaling": "strong" if strong else "weak", "mode": mode, "vs_baseline": None, "dtype": "f64",
        "arithmetic": "negacyclic FFT64 (the reference backend's own arithmetic: complex FP64 transform of N/2 points, results rounded to the exact integers); int32 limbs in HBM",
        "data": "synthetic",
        "config": {"workload": f"Ram::read + Ram::read_prepare_write + Ram::write, MAX_ADDR=2^{log_entries}, "
                               f"WORDSIZE={ws}, N=4096, base2k=17, rank=1, "
                               + ("README.md:17-27 parameter block (K_PT=9, K_EVK=85: 5-limb trace keys)" if crypto else "source constants (parameters.rs:11-18)")
                               + " (BASELINE.json configs[2]+[3]"
                               + ("; configs[4] sharding" if sharded else "") + ")",
                   "rams": n_rams, "rows_per_subram": cnt["rows"], "parallelism": par},
        "ram_ops_s_raw": raw_ops_per_s,
        "ops_weighted_by_ram_size": (None if weight == 1 else
                                     {"value": raw_ops_per_s * weight,
                                      "what": f"information only: an op on the {world}*2^{args.log_max_addr}-entry RAM counted as {world} ops of the 2^{args.log_max_addr}-entry metric"}),
        "weak_scaling_efficiency": (None if weak_ref is None else
                                    {"value": weak_ref / ms_per_step, "t1_ms_per_step": weak_ref, "tN_ms_per_step": ms_per_step,
                                     "what": f"T(1) / T(N): the same step on an unsharded 2^{args.log_max_addr}-entry RAM, measured in this run on rank 0's GPU, "
                                             f"over the step on the 2^{log_entries}-entry RAM sharded over {world} GPUs"}),
        "read_ops_s": n_rams * 1e3 / read_ms, "write_ops_s": n_rams * 1e3 / (rpw_ms + write_ms),
        "read_ms": read_ms, "read_prepare_write_ms": rpw_ms, "write_ms": write_ms,
        "per_op_split": {"what": "read_ms / read_prepare_write_ms / write_ms: a pass of the same K steps with one HIP-event pair per op "
                                 "on the context's stream (kept out of the timed region)", "ms_per_step": per_op_elapsed * 1e3 / args.steps},
        "ops_enqueued_back_to_back": (None if pipelined is None else
                                      {"ram_ops_s": n_rams * 2 * args.steps / pipelined, "ms_per_step": pipelined * 1e3 / args.steps,
                                       "note": "information only: K steps enqueued without waiting for each op (NULL result pointer), one "
                                               "synchronisation at the end; `value` keeps one host round trip per op, as the reference's calls have"}),
        "algorithmic_GBs_per_op": {"read": a_read / read_ms / 1e6, "read_prepare_write": a_rpw / rpw_ms / 1e6,
                                   "write": a_write / write_ms / 1e6},
        "reference_published": None,     # filled below: the published numbers against THIS run's times on the published block
        "trace_tail": dict(tail, note="trace chains at the end of a read run as one launch with in-kernel hand-offs; `fallbacks` of them "
                                      "gave up (CUs not available side by side) and were redone by the fused launch behind them"),
        "device": ram.device_info(),
    }
    # The reference's only published numbers (README.md:36: 450 ms read / 1200 ms write, i9-12900K single thread, 2^18 entries) were
    # taken with the README parameter block (5-limb trace keys), not with the source constants `value` is measured on: the ratios
    # are quoted for the block they belong to — this run's own times when it IS that block, else a short leg on that block.
    pub = {"read_ms": 450, "write_ms": 1200, "hw": "i9-12900K single thread (README.md:36), 2^18 entries, README.md:17-27 parameter block"}
    leg = None
    if crypto:
        leg = {"read_ms": float(read_ms), "read_prepare_write_ms": float(rpw_ms), "write_ms": float(write_ms), "ms_per_step": ms_per_step,
               "ram_ops_s": raw_ops_per_s, "steps": args.steps}
    elif mode == "single" and not args.no_readme_leg and args.log_max_addr == 18:
        leg = readme_leg(pkg, ws, args.log_max_addr, local_rank, max(5, min(args.steps, 20)), 5)
    if leg is not None:
        pub.update({"measured_on_the_same_block": leg,
                    "speedup_read": 450.0 / leg["read_ms"], "speedup_write": 1200.0 / leg["write_ms"],
                    "speedup_rpw_plus_write": (450.0 + 1200.0) / (leg["read_prepare_write_ms"] + leg["write_ms"]),
                    "note": "like for like: the README block on this GPU" + ("" if crypto else " (a short leg of this run on a context of its own; `value` stays the source constants)")
                            + "; call for call as the example times them (examples/fhe-ram.rs:98-154); Ram::write here resumes from what "
                              "read_prepare_write kept, so speedup_rpw_plus_write (a read_prepare_write priced as a published read) is the fairer write figure"})
    else:
        pub["note"] = "not measured in this run (the published numbers belong to the README block at 2^18: run --params readme)"
    out["reference_published"] = pub
    out["roundoff_max"] = {"value": ram.roundoff_max(), "limit": 0.375,
                           "what": "largest |x - rint(x)| any rounding of an inverse transform has seen on the timed context (fheram_roundoff_max): the FFT64 "
                                   "arithmetic is exact while it stays below 1/2; above 3/8 every call returns FHERAM_ERR_PRECISION"}
    out["mid_chain"] = dict(mid, note="dependent chains on 9..64 ciphertexts (MAX_ADDR 2^14..2^16) as one launch with in-kernel hand-offs (k_chain_mid); fallbacks = ciphertexts redone by the launch behind")
    if tail["fallbacks"] > 0 or mid["fallbacks"] > 0:
        out["trace_tail_degraded"] = True     # silent degradation made visible: some single-launch chains were redone by their fallback
        print(f"bench.py: WARNING: {tail['fallbacks']} of {tail['launches']} single-launch trace chains fell back", file=sys.stderr)
    if boundary is not None:
        out["read_ms_incl_boundary"], out["rpw_ms_incl_boundary"], out["write_ms_incl_boundary"] = [float(x) for x in boundary]
        bstep = float(sum(boundary))
        out["value_incl_boundary"] = 2e3 / bstep     # (key created near the top of the line)
        out["ms_per_step_incl_boundary"] = bstep
        out["boundary_gap_frac"] = bstep / ms_per_step - 1.0
        out["boundary_note"] = ("what the reference's calls return (ram.rs:176,200: host ciphertexts): host wall clock per call through the ABI's "
                                "int64 HOST buffers — result written out (ws GLWEs) on the two reads, words taken in on write; median; "
                                "`value` stays the device-resident figure the bench contract prescribes, value_incl_boundary is the same steps "
                                "with the hand-over inside")

    if not args.no_kernel_timing:
        names = ("keyswitch", "keyswitch_fused", "keyswitch_chain_launch", "keyswitch_tail_launch", "keyswitch_mid_launch", "ext_product",
                 "ext_product_fused", "ext_product_mid_launch", "prepare", "elementwise", "read_chain_launch", "write_chain_launch")
        classes = {k: ram.profile_get(k) for k in names}
        forms = ram.forms() if hasattr(ram, "forms") else {}
        d0 = len(p.base2d().v[0].d)
        L0r = max(0, 12 - max(0, (-(-max_addr // N) - 1).bit_length()))
        # Every launch class that a step's GPU time splits into (disjoint: the nested event pairs are subtracted), with the FP64
        # work its launches carry (per ciphertext: products x FP64_PER_EP + key-switches x fp64_per_ks) and the kernel it is.
        def entry(kernel, what, ms, launches, fp64, pmc=None, steps_per_launch=None, alg=None):
            if not launches or ms <= 0:
                return None
            t = fp64 / (ms * 1e-3) / 1e12
            e = {"kernel": kernel, "what": what, "launches": launches, "avg_launch_ms": ms / launches, "gpu_ms": ms,
                 "achieved_T_fp64_instr_s": t, "achieved": 2 * t, "peak": 2 * FP64_VALU_PEAK_TINSTR, "unit": "TFLOP/s", "bound": "valu_fp64",
                 "frac": t / FP64_VALU_PEAK_TINSTR,
                 "frac_algorithmic": None if alg is None else alg / (ms * 1e-3) / 1e12 / (2 * FP64_VALU_PEAK_TINSTR),
                 "traffic": pmc_traffic(pmc) if pmc else None}
            if steps_per_launch:
                e["steps_per_launch"] = steps_per_launch
            return e
        rc, wc, pc = classes["read_chain_launch"], classes["write_chain_launch"], classes["keyswitch_chain_launch"]
        tl, md, mde = classes["keyswitch_tail_launch"], classes["keyswitch_mid_launch"], classes["ext_product_mid_launch"]
        ksc, epc = classes["keyswitch"], classes["ext_product"]
        pure_ms = light["chain"]["ms"] if (light is not None and light["chain"]["blocks"] == pc["blocks"] and pc["launches"]) else pc["ms"]
        rest_ks_ms = ksc["ms"] - pc["ms"] - tl["ms"] - md["ms"]
        rest_ks_blocks = ksc["blocks"] - pc["blocks"] - tl["blocks"] - md["blocks"]
        rest_ep_ms = epc["ms"] - mde["ms"]
        rest_ep_blocks = epc["blocks"] - mde["blocks"]
        sk = s_evk
        alg_ks = alg_flop_per_keyswitch(s_evk)
        table = [
            entry(f"k_read_chain<{sk},4> / k_read_chain_w<{sk},4>", f"a row's {d0} products of coordinate 0 + its {L0r} alone packer levels as ONE launch, one workgroup per row (ram.rs:429-435,502-514)",
                  rc["ms"], rc["launches"], rc["blocks"] * (d0 * FP64_PER_EP + L0r * fp64_per_ks), (f"fk::k_read_chain<{sk}, 4>", f"fk::k_read_chain_w<{sk}, 4>"), d0 + L0r,
                  alg=rc["blocks"] * (d0 * ALG_FLOP_PER_EP + L0r * alg_ks)),
            entry(f"k_write_chain<{sk},4>", f"write_mid_step's 12 trace steps of ct_lo X^-row + normalize(ct_hi - trace(ct_hi) + .) + write_last_step's {d0} products as ONE launch (ram.rs:612-646)",
                  wc["ms"], wc["launches"], wc["blocks"] * (d0 * FP64_PER_EP + 12 * fp64_per_ks), f"fk::k_write_chain<{sk}, 4>", d0 + 12,
                  alg=wc["blocks"] * (d0 * ALG_FLOP_PER_EP + 12 * alg_ks)),
            entry(f"k_keyswitch_chain<3,{sk},3,{forms.get('chain_y', 3)}>", "pure trace chains, one workgroup per ciphertext (at 2^18: trace(ct_hi) steps 6..11 on the write's side stream, ram.rs:616)",
                  pure_ms, pc["launches"], pc["blocks"] * fp64_per_ks, (f"fk::k_keyswitch_chain<3, {sk}, 3, {forms.get('chain_y', 3)}>", f"fk::k_keyswitch_chain_w<3, {sk}, 3, {forms.get('chain_y', 3)}>"),
                  (pc["blocks"] / pc["launches"] / max(1.0, classes["keyswitch_fused"]["blocks"] / max(1, classes["keyswitch_fused"]["launches"]))) if pc["launches"] else None,
                  alg=pc["blocks"] * alg_ks),
            entry(f"k_trace_tail<3,{sk},3>", "GLWE::trace on the word_size results at the end of a read: 12 dependent steps as one launch with in-kernel hand-offs (ram.rs:457,540)",
                  tl["ms"], tl["launches"], tl["blocks"] * fp64_per_ks, f"fk::k_trace_tail<3, {sk}, 3>", alg=tl["blocks"] * alg_ks),
            entry("k_chain_mid<...>", "dependent chains on 9..64 ciphertexts as one launch with in-kernel hand-offs (MAX_ADDR 2^13..2^16)",
                  md["ms"] + mde["ms"], md["launches"] + mde["launches"], md["blocks"] * fp64_per_ks + mde["blocks"] * FP64_PER_EP,
                  alg=md["blocks"] * alg_ks + mde["blocks"] * ALG_FLOP_PER_EP),
            entry("k_pair_z / k_keyswitch / k_keyswitch_fine + _norm", "the pair levels of the packing tree and the other key-switch launches of the dependent end of an op (4..128 ciphertexts per launch)",
                  rest_ks_ms, ksc["launches"] - pc["launches"] - tl["launches"] - md["launches"], rest_ks_blocks * fp64_per_ks, alg=rest_ks_blocks * alg_ks),
            entry("k_ext_product_fine + _norm / k_ext_product", "coordinate 1's products on word_size ciphertexts",
                  rest_ep_ms, epc["launches"] - mde["launches"], rest_ep_blocks * FP64_PER_EP, alg=rest_ep_blocks * ALG_FLOP_PER_EP),
            entry("k_prepare", "CoordinatePrepared::prepare: forward transforms of the address digits", classes["prepare"]["ms"], classes["prepare"]["launches"],
                  classes["prepare"]["blocks"] * FP64_FWD),
            entry("elementwise (k_sub_add_norm, k_rotate, k_copy)", "write_first_step, rotations, copies", classes["elementwise"]["ms"], classes["elementwise"]["launches"], 0),
        ]
        table = [e for e in table if e]
        gpu_ms = sum(e["gpu_ms"] for e in table)
        for e in table:
            e["share_of_gpu_time"] = e["gpu_ms"] / gpu_ms
        table.sort(key=lambda e: -e["gpu_ms"])
        note = ("frac prices the implementation's OWN FP64 instruction stream (6 / 8 instructions per complex butterfly of the FFT64 transforms, 4 per "
                "complex MAC, 1 per rounding) against the 39.3 T instr/s issue peak: utilisation of the FP64 pipe, not a bound on the "
                "algorithm; frac_algorithmic prices the same launch in ALGORITHMIC flops (5 n log2 n per transform on n = 2048 complex points, 8 per "
                "complex multiply-accumulate) against the 78.6 TFLOP/s vector FP64 peak.  What shares the step with the FP64 "
                "pipe: the swap rounds and address arithmetic of the transforms (VALU, not FP64), two 32 KB LDS exchanges per transform, and "
                "the prepared operands streamed from the XCD's L2 under the inverse transforms (32 KB per polynomial MAC; at 2^18 the product runs at "
                "~60 % of the L2's share per CU): see `pipes`.")
        if table:
            dom = dict(table[0])
            ratio = None
            if dom["kernel"].startswith("k_read_chain"):   # (both register budgets: one launch class)
                ratio = (d0 * R04_FP64_PER_EP + L0r * R04_FP64_PER_KS) / (d0 * FP64_PER_EP + L0r * FP64_PER_KS) if s_evk == 4 else None
                n_ep, n_ks = d0, L0r
            elif dom["kernel"].startswith("k_write_chain"):
                ratio = (d0 * R04_FP64_PER_EP + 12 * R04_FP64_PER_KS) / (d0 * FP64_PER_EP + 12 * FP64_PER_KS) if s_evk == 4 else None
                n_ep, n_ks = d0, 12
            else:
                ratio = R04_FP64_PER_KS / FP64_PER_KS if s_evk == 4 else None
                n_ep, n_ks = 0, 1
            dom["note"] = note
            # (comparison aid for the r04 profiles only — NOT a roofline figure, hence outside the `roofline` block)
            out["history"] = {"frac_at_round4_instruction_count": None if ratio is None else dom["frac"] * ratio,
                              "what": "the dominant launch of this run priced with ROUND 4's instruction count (modular transforms: 2.47x the FP64 instructions per key-switch, "
                                      "2.66x per product): only for comparing with profiles/r04_*"}
            dom["fp64_instr_per_keyswitch"] = fp64_per_ks
            dom["fp64_instr_per_product"] = FP64_PER_EP
            dom["traffic_source"] = PMC_PROFILE + " (HBM-side bytes per launch of this kernel: separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command, FETCH_SIZE scaled by the factor calibrated on a known-bytes stream, tools/fetch_calib.hip; FETCH_SIZE counts Infinity-Cache hits)"
            dom["timing_source"] = "HIP events around each launch of the class on its launch stream, in a pass of the same K steps (not the timed region)"
            # the other pipes a chain step keeps busy, per ciphertext, from the kernel's structure (csrc/fft_dev.hpp, kernels.hpp)
            n_tr = n_ep * 14 + n_ks * (3 + 2 * s_evk)
            n_mac = n_ep * 48 + n_ks * 6 * s_evk
            t_launch = dom["avg_launch_ms"] * 1e-3
            lds_bytes = n_tr * 2 * 2 * 32768 + n_tr * 13 * 16 * 512        # two exchanges (write + read) + 13 twiddle reads of 16 B per thread
            opnd_bytes = n_mac * 32768
            dom["pipes"] = {"what": "per workgroup (= per CU: one ciphertext each) and launch, from the kernel's structure; rates against the CU's own peaks at the clock the stamps show (2.19 GHz)",
                            "valu_busy_frac_pmc": pmc_valu_busy(dom["kernel"]),
                            "valu_busy_source": SQ_PROFILE + ": SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES of this kernel x 2 waves per SIMD (a separate rocprofv3 --pmc pass of this command): the share of time the vector ALU issues ANY instruction (FP64, swap rounds, address and digit arithmetic)",
                            "transforms": n_tr, "polynomial_macs": n_mac,
                            "fp64_busy_frac": dom["frac"],
                            "lds_bytes": lds_bytes, "lds_GBs_per_cu": lds_bytes / t_launch / 1e9, "lds_peak_GBs_per_cu": "~85 B/clk stores, 256 B/clk loads (MI355X_MICROARCH.md LDS table): 186 / 560",
                            "operand_bytes_from_l2": opnd_bytes, "operand_GBs_per_cu": opnd_bytes / t_launch / 1e9, "operand_peak_GBs_per_cu": 64 * 2.19,
                            "operand_frac": opnd_bytes / t_launch / 1e9 / (64 * 2.19),
                            "operand_l2_share_GBs_per_cu": 70.0,
                            "operand_frac_of_l2_share": opnd_bytes / t_launch / 1e9 / 70.0,
                            "operand_note": "every workgroup of a launch streams the SAME key / GGSW from its XCD's L2: with all 32 CUs of an XCD streaming shared rows the L2 gives each "
                                            "66-73 GB/s (MI355X_MICROARCH.md, 'Indexed rows'), not the 64 B/clk of the CU's own load path.  The stream runs under the inverse transforms "
                                            "(fft_inv1_hooked / fft_inv2_hooked); what it still costs is measured by building with -DFK_NO_OPERANDS / -DFK_HALF_OPERANDS "
                                            "(profiles/r05_experiments.txt): 4.1 us of a 23.1 us trace step, 6.3 us of a 34.1 us product; with half the bytes 1.6 and 1.7"}
            out["roofline"] = dom
            out["roofline_by_kernel"] = {"what": "every launch class of the step (disjoint), largest GPU time first; share_of_gpu_time is of the summed kernel time of the instrumented pass",
                                         "kernels": [{k: v for k, v in e.items()} for e in table if e["share_of_gpu_time"] >= 0.05],
                                         "below_5_percent": [{"kernel": e["kernel"], "share_of_gpu_time": e["share_of_gpu_time"], "avg_launch_ms": e["avg_launch_ms"], "frac": e["frac"]} for e in table if e["share_of_gpu_time"] < 0.05]}
            # the pure trace step (what rounds 2-4 priced): one step of a pure trace chain over the batch
            if pc["launches"]:
                kf = classes["keyswitch_fused"]
                blocks = kf["blocks"] / kf["launches"] if kf["launches"] else 0
                spl = pc["blocks"] / pc["launches"] / blocks if blocks else 1.0
                step_ms = pure_ms / pc["launches"] / spl
                bytes_abi = blocks * 2 * GLWE_I64 + atk_i64              # in + out GLWE (int64 limbs) + key once
                bytes_dev = atk_i64 + blocks * 2 * (GLWE_I64 // 2) / max(1.0, spl)
                out["trace_step"] = {"what": "one fused trace step (ks_trace_l) over the batch: HIP-event time of a pure trace-chain launch / its steps",
                                     "us": step_ms * 1e3, "ciphertexts": blocks, "steps_per_launch": spl,
                                     "frac": blocks * fp64_per_ks / (step_ms * 1e-3) / 1e12 / FP64_VALU_PEAK_TINSTR,
                                     "round4_us": 33.3}
                out["roofline_hbm"] = {"kernel": "the same trace step", "bound": "hbm", "achieved": bytes_dev / step_ms / 1e6, "peak": HBM_PEAK_GBS,
                                       "unit": "GB/s", "frac": bytes_dev / step_ms / 1e6 / HBM_PEAK_GBS,
                                       "device_layout_bytes_per_launch": bytes_dev,
                                       "abi_layout_int64": {"algorithmic_bytes_per_launch": bytes_abi, "achieved": bytes_abi / step_ms / 1e6,
                                                            "frac": bytes_abi / step_ms / 1e6 / HBM_PEAK_GBS},
                                       "note": "not the binding roof: everything a 2^18 op touches sits in the 256 MiB Infinity Cache"}
        if rc["launches"] or wc["launches"]:
            out["fused_row_chains"] = {e["kernel"].split("<")[0]: {k: e[k] for k in ("launches", "avg_launch_ms", "frac", "steps_per_launch") if k in e}
                                       for e in table if e["kernel"].startswith(("k_read_chain", "k_write_chain"))}
        # whole step (read + read_prepare_write + write) against both roofs, and in SURVEY.md 8(d)'s own unit
        ep_ref = 2 * cnt["ep_read"] + cnt["ep_write"]
        ks_ref = 2 * cnt["ks_read"] + cnt["ks_write"]
        L0 = max(0, 12 - max(0, (cnt["rows"] - 1).bit_length())) if cnt["rows"] > 1 else 0
        ks_exec = ks_ref - (ws * 12 + ws * cnt["rows"] * L0)     # the write resumes from what read_prepare_write kept (csrc/path.hpp)
        fp_step = ep_ref * FP64_PER_EP + ks_exec * fp64_per_ks
        mm_ks = modmul_per_keyswitch(s_evk)
        mm = {"read": cnt["ep_read"] * MODMUL_PER_EP + cnt["ks_read"] * mm_ks,
              "read_prepare_write": cnt["ep_read"] * MODMUL_PER_EP + cnt["ks_read"] * mm_ks,
              "write": cnt["ep_write"] * MODMUL_PER_EP + cnt["ks_write"] * mm_ks}
        out["modmul_per_s"] = {"what": "SURVEY.md 8(d)'s unit: the reference's operation count per op (a butterfly of an N-point transform = 1 modmul, 24 576 per transform; a pointwise MAC = 4096 per polynomial pair; EP = 540 672, trace / pack key-switch = 368 640 with 4-limb keys) / the op's time.  Counts follow the REFERENCE's control flow (a write counts all 24 traces per row although this path resumes from what read_prepare_write kept); GGSW prepares and inversions (< 2 %) not counted.  The >= 100x target needed 0.28 T/s",
                               "unit": "T modmul/s",
                               "read": mm["read"] / (read_ms * 1e-3) / 1e12, "read_prepare_write": mm["read_prepare_write"] / (rpw_ms * 1e-3) / 1e12,
                               "write": mm["write"] / (write_ms * 1e-3) / 1e12,
                               "step": (mm["read"] + mm["read_prepare_write"] + mm["write"]) / (ms_per_step * 1e-3) / 1e12,
                               "modmuls_per_op": mm}
        out["roofline_whole_op"] = {"what": "one step = read + read_prepare_write + write, device-resident, against the same two roofs",
                                    "hbm": {"algorithmic_bytes_per_step": a_read + a_rpw + a_write,
                                            "achieved_GBs": (a_read + a_rpw + a_write) / ms_per_step / 1e6,
                                            "frac": (a_read + a_rpw + a_write) / ms_per_step / 1e6 / HBM_PEAK_GBS},
                                    "valu_fp64": {"external_products": ep_ref, "key_switches_reference": ks_ref, "key_switches_executed": ks_exec,
                                                  "fp64_instr_per_step": fp_step,
                                                  "achieved_T_fp64_instr_s": fp_step / (ms_per_step * 1e-3) / 1e12,
                                                  "frac": fp_step / (ms_per_step * 1e-3) / 1e12 / FP64_VALU_PEAK_TINSTR,
                                                  "frac_algorithmic": (ep_ref * ALG_FLOP_PER_EP + ks_exec * alg_ks) / (ms_per_step * 1e-3) / 1e12 / (2 * FP64_VALU_PEAK_TINSTR),
                                                  "note": "GGSW prepares and inversions (< 2 % of the work) not counted"}}
        out["kernel_classes"] = classes
        out["kernel_timing_pass"] = {"what": "separate pass of the same K steps with per-launch HIP events on the launch stream "
                                             "(not part of the timed region: the events add this much to a step)",
                                     "ms_per_step_instrumented": instr_elapsed * 1e3 / args.steps}
        out["forms"] = forms

    parity_failed = False
    if not args.no_cpu_baseline and world == 1 and mode == "single":   # reported baseline: rank 0 at N = 1 only
        # bounded sample (~10 s of CPU work at 2^18): the whole RAM, one step; beyond 2^18 one of the `ws` sub-RAMs, scaled by
        # ws (the reference processes them one after the other, ram.rs:187-190; prepare_inv is shared, <1 % of a write)
        flags = oracle_native()
        whole = max_addr * ws <= (1 << 20)
        k = 1 if whole else ws
        (r, q, w), o1 = cpu_baseline(max_addr, ws, inp, 1, crypto, None if whole else 0)
        out["cpu_baseline"] = {"value": 2.0 / (k * (r + q + w)), "unit": "RAM ops/s", "cores": 1, "kind": "port",
                               "sample": "oracle (C++ exact-integer restatement), single thread, "
                                         + (f"the whole 2^{args.log_max_addr} x {ws}-byte RAM, one step: " if whole else
                                            f"1 of {ws} sub-RAMs of the 2^{args.log_max_addr} RAM, scaled x{ws}: ")
                                         + f"read {r:.2f}s + read_prepare_write {q:.2f}s + write {w:.2f}s; the inputs of the GPU legs",
                               "read_ms": k * r * 1e3, "read_prepare_write_ms": k * q * 1e3, "write_ms": k * w * 1e3,
                               "host_cpu": host_cpu(), "build": flags}
        cores = host_cores()
        (r, q, w), oa = cpu_baseline(max_addr, ws, inp, cores, crypto)
        out["cpu_baseline_allcores"] = {"value": 2.0 / (r + q + w), "unit": "RAM ops/s", "cores": cores, "kind": "port",
                                        "sample": f"oracle, OpenMP over the {ws} sub-RAMs and over the rows (per-row loops; packing level by level with "
                                                  f"the leaves of a level in parallel), whole 2^{args.log_max_addr} RAM, one step; the inputs of the GPU legs",
                                        "read_ms": r * 1e3, "read_prepare_write_ms": q * 1e3, "write_ms": w * 1e3, "host_cpu": host_cpu(),
                                        "build": flags}
        # Parity inside the run (examples/fhe-ram.rs:98-115 checks the result it has just timed): the context that was timed, its rows
        # loaded again, runs the same step once more — untimed, results downloaded — and every output must equal the oracle's on
        # the same inputs, bit for bit (SHA-256 of the int64 limbs).
        ram.load_encrypted(inp["rows"])
        g = {"read": ram.read(addr, keys), "rpw": ram.read_prepare_write(addr, keys)}
        ram.write(words, addr, keys)
        g["rows_after_write"] = ram.store_encrypted()
        ram.stage_words(words)
        par = {kk: bool(np.array_equal(g[kk], oa[kk])) for kk in ("read", "rpw", "rows_after_write")}
        sel = slice(None) if whole else slice(0, 1)
        par["single_thread_leg_agrees"] = all(bool(np.array_equal(np.asarray(g[kk])[sel], o1[kk])) for kk in ("read", "rpw", "rows_after_write"))
        par["sha256"] = {kk: {"hip": sha(g[kk]), "oracle": sha(oa[kk])} for kk in ("read", "rpw", "rows_after_write")}
        par["what"] = ("one more untimed step on the timed context (rows reloaded), results downloaded through the C ABI, against the oracle's "
                       "all-core leg on the same keys, address, rows and words; bit-exact int64 limbs")
        out["parity_in_run"] = par
        parity_failed = not all(par[kk] for kk in ("read", "rpw", "rows_after_write", "single_thread_leg_agrees"))
    print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()
    if parity_failed:
        print("bench.py: PARITY FAILURE: the HIP path and the oracle differ on the run's own inputs (parity_in_run)", file=sys.stderr)
        raise SystemExit(3)


if __name__ == "__main__":
    main()
