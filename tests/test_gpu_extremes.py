"""Adversarial operands THROUGH the chain kernels ("what's weak" 2): a 2^18 x 4 B RAM whose rows, address digits,
evaluation keys and written words are all limbs at the ends of the normalised range — every limb -2^16, every limb 2^16 - 1,
alternating by coefficient, one coefficient flipped, random signs — through Ram::read / read_prepare_write / write (ram.rs:172-294)
against the oracle, which is exact integer arithmetic and does not care what the limbs are.

The end-to-end parity tests and the committed digests only ever feed pseudo-random limbs to k_read_chain / k_write_chain /
ep_step_r / the streamed ks_trace_l, where 75 % of the arithmetic runs; the coherent patterns below concentrate the spectrum
and maximise the sums (6 * 4096 * 2^32), i.e. they are where the FP64 round-off of the FFT64 arithmetic is largest
(tests/test_gpu_fft.py pins it at the transform level: 0.125).  Bit-exactness here = the rounded results are still the exact
integers in the kernels' own summation order (one product per hook of the hooked transforms); the round-off monitor
(fheram_roundoff_max, every coefficient) must agree that they were.  The inputs are not valid ciphertexts: nothing decrypts,
everything compares."""
import numpy as np
import pytest

from _pkg import load_package

pytestmark = pytest.mark.gpu
N = 4096
LO, HI = -(1 << 16), (1 << 16) - 1


def _fill(kind, shape, rng):
    n = int(np.prod(shape))
    if kind == "lo":
        a = np.full(n, LO, dtype=np.int64)
    elif kind == "hi":
        a = np.full(n, HI, dtype=np.int64)
    elif kind == "alternating":
        a = np.where(np.arange(n) % 2 == 0, LO, HI).astype(np.int64)
    elif kind == "flipped":
        a = np.full(n, LO, dtype=np.int64)
        a[::N] = HI                      # coefficient 0 of every polynomial
    elif kind == "signs":
        a = rng.choice(np.array([LO, HI], dtype=np.int64), n)
    else:
        raise ValueError(kind)
    return a.reshape(shape)


def _oracle_flow(po, max_addr, ws, inp, threads):
    o = po.Oracle(po.OParams(max_addr=max_addr, word_size=ws)).set_threads(threads)
    evk = {"gal_els": np.array([int(po.lib().fo_galois_element(12, i)) for i in range(12)], dtype=np.int64),
           "atk_glwe": inp["atk"], "atk_ggsw_inv": inp["atk_inv"], "tsk": inp["tsk"]}
    keys = o.keys_prepare(evk)
    addr = o.address_new(inp["addr"])
    ram = o.ram_new()
    ram.load(np.ascontiguousarray(inp["rows"]))
    out = {"read": ram.read(addr, keys), "rpw": ram.read_prepare_write(addr, keys)}
    ram.write(np.ascontiguousarray(inp["words"]), addr, keys)
    out["rows_after_write"] = ram.store()
    out["read_back"] = ram.read(addr, keys)
    assert o.max_big() < 1 << 47          # the bound the rounding contract is stated for
    return out


def _threads():
    import os
    try:
        return max(1, min(16, len(os.sched_getaffinity(0))))
    except Exception:
        return 4


# (kind of the rows, of the address digits, of the keys, of the written words)
PATTERNS = [
    ("lo", "lo", "lo", "lo"),
    ("hi", "hi", "hi", "hi"),
    ("alternating", "lo", "hi", "alternating"),
    ("flipped", "lo", "lo", "flipped"),
    ("signs", "signs", "signs", "signs"),
    ("lo", "hi", "alternating", "hi"),
]


@pytest.mark.parametrize("log_max_addr", [18, 14])
@pytest.mark.parametrize("pat", PATTERNS, ids=lambda p: "-".join(p))
def test_extreme_limbs_through_the_chain_kernels(po, pat, log_max_addr):
    """2^18: k_read_chain / k_write_chain / the tail; 2^14 (the source default): k_chain_mid.  Default configuration and
    the unfused / limb-form chains (chain_y = 0: ks_run's summation order) must both give the oracle's integers."""
    pkg = load_package()
    max_addr, ws = 1 << log_max_addr, 4
    import zlib
    rng = np.random.default_rng(zlib.crc32("-".join(pat).encode()))
    p = pkg.Parameters(max_addr=max_addr, decomp_n=[3, 3, 3, 3], word_size=ws)
    n_digits = p.base2d().as_1d().size()
    rows_r, addr_r, keys_r, words_r = pat
    inp = {"atk": _fill(keys_r, (12, 3 * 4 * 2 * N), rng), "atk_inv": _fill(keys_r, 4 * 5 * 2 * N, rng), "tsk": _fill(keys_r, 4 * 5 * 2 * N, rng),
           "addr": _fill(addr_r, (n_digits, p.ggsw_len()), rng), "rows": _fill(rows_r, (ws, max_addr // N, p.glwe_len()), rng),
           "words": _fill(words_r, (ws, p.glwe_len()), rng)}
    want = _oracle_flow(po, max_addr, ws, inp, _threads())
    worst = 0.0
    for cfg in ({"monitor": 2}, {"monitor": 2, "chain_y": 0}, {"monitor": 2, "fuse": 0}):
        ram = pkg.Ram(p, config=cfg)
        keys = pkg.EvaluationKeysPrepared(pkg.galois_elements(12), list(inp["atk"]), inp["atk_inv"], inp["tsk"])
        addr = pkg.Address(p, list(inp["addr"]))
        ram.load_encrypted(inp["rows"])
        got = {"read": ram.read(addr, keys), "rpw": ram.read_prepare_write(addr, keys)}
        ram.write(inp["words"], addr, keys)
        got["rows_after_write"] = ram.store_encrypted()
        got["read_back"] = ram.read(addr, keys)
        for k in ("read", "rpw", "rows_after_write", "read_back"):
            bad = np.count_nonzero(np.asarray(got[k]) != np.asarray(want[k]))
            assert bad == 0, (pat, cfg, k, bad)
        ro = ram.roundoff_max()              # raises PRECISION above 1/4
        assert 0.0 < ro <= 0.25, (pat, cfg, ro)
        worst = max(worst, ro)
        t = ram.tail_stats()
        assert t["fallbacks"] == 0
        del ram
    print(f"extremes {pat} 2^{log_max_addr}: max round-off through the chains {worst:.4g}")
