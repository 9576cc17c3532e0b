"""Host-side mirror of the reference's public interface for the RAM path, over the C ABI.

Names, argument meaning and error behaviour follow /root/reference/src:
  Parameters              parameters.rs:147-287
  EvaluationKeysPrepared  keys.rs:27-71
  Address                 address.rs:21-119
  Ram                     ram.rs:25-294  (read :172, read_prepare_write :196, write :226)
The reference panics on misuse (assert!); here every failing call raises FheRamError carrying
the C-ABI status and the reference's message.

All ciphertext data crosses this boundary as numpy int64 arrays in Poulpy's host layouts
(include/fheram.h).  There is NO CPU implementation behind these classes: if the HIP library is
missing or no GPU is present, construction fails loudly.
"""
import ctypes as C
import os
from typing import Dict, List, Optional, Sequence

import numpy as np

from .base import Base2D, get_base_2d

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

I64P = C.POINTER(C.c_int64)

STATUS = {0: "OK", 1: "INVALID_ARG", 2: "STATE", 3: "UNINITIALIZED", 4: "KEYS", 5: "UNSUPPORTED", 6: "RANGE", 7: "DEVICE", 8: "PRECISION"}


class FheRamError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"fheram status {code} ({STATUS.get(code, '?')}): {msg}")
        self.code = code
        self.msg = msg


class _CParams(C.Structure):
    _fields_ = [("log_n", C.c_uint32), ("base2k", C.c_uint32), ("rank", C.c_uint32), ("k_glwe_pt", C.c_uint32),
                ("k_glwe_ct", C.c_uint32), ("k_ggsw_addr", C.c_uint32), ("k_evk_trace", C.c_uint32),
                ("k_evk_ggsw_inv", C.c_uint32), ("word_size", C.c_uint32), ("n_decomp", C.c_uint32),
                ("decomp_n", C.c_uint8 * 16), ("max_addr", C.c_uint64)]


_CONFIG_FIELDS = ("limb_split", "fine_split", "memo", "pre_inv", "tail", "tail_test", "mid", "mid_test", "chain", "chain_y", "pair_z",
                  "fuse", "graph", "safe", "nco", "tail_ep", "monitor", "reserved")


class _CConfig(C.Structure):
    """fheram_config (include/fheram.h): the execution switches of a context"""
    _fields_ = [(f, C.c_int32) for f in _CONFIG_FIELDS]


def library_path() -> str:
    # FHERAM_LIB selects an alternative build of the same HIP library (kernel-tuning experiments)
    return os.environ.get("FHERAM_LIB") or os.path.join(_HERE, "libfheram.so")


# every symbol include/fheram.h declares: (name, restype, argtypes)
_SYMBOLS = [
    ("fheram_params_default", C.c_int, [C.POINTER(_CParams)]),
    ("fheram_ctx_create", C.c_int, [C.POINTER(_CParams), C.c_int, C.POINTER(C.c_void_p)]),
    ("fheram_ctx_destroy", None, [C.c_void_p]),
    ("fheram_last_error", C.c_char_p, [C.c_void_p]),
    ("fheram_glwe_len", C.c_size_t, [C.c_void_p]),
    ("fheram_ggsw_len", C.c_size_t, [C.c_void_p]),
    ("fheram_atk_len", C.c_size_t, [C.c_void_p]),
    ("fheram_evk_inv_len", C.c_size_t, [C.c_void_p]),
    ("fheram_rows", C.c_size_t, [C.c_void_p]),
    ("fheram_n_digits", C.c_int, [C.c_void_p]),
    ("fheram_n_coordinates", C.c_int, [C.c_void_p]),
    ("fheram_keys_load", C.c_int, [C.c_void_p, I64P, C.c_int, C.POINTER(I64P), I64P, C.c_int64, I64P]),
    ("fheram_ram_upload", C.c_int, [C.c_void_p, I64P]),
    ("fheram_ram_download", C.c_int, [C.c_void_p, I64P]),
    ("fheram_ram_tree_download", C.c_int, [C.c_void_p, C.c_int, I64P]),
    ("fheram_ram_state", C.c_int, [C.c_void_p]),
    ("fheram_address_create", C.c_int, [C.c_void_p, C.POINTER(I64P), C.c_int, C.POINTER(C.c_void_p)]),
    ("fheram_address_destroy", None, [C.c_void_p]),
    ("fheram_read", C.c_int, [C.c_void_p, C.c_void_p, I64P]),
    ("fheram_read_prepare_write", C.c_int, [C.c_void_p, C.c_void_p, I64P]),
    ("fheram_write", C.c_int, [C.c_void_p, I64P, C.c_int, C.c_void_p]),
    ("fheram_word_stage", C.c_int, [C.c_void_p, I64P, C.c_int]),
    ("fheram_result_download", C.c_int, [C.c_void_p, I64P]),
    ("fheram_result_map", C.c_int, [C.c_void_p, C.POINTER(I64P)]),
    ("fheram_sync", C.c_int, [C.c_void_p]),
    ("fheram_roundoff_max", C.c_int, [C.c_void_p, C.POINTER(C.c_double)]),
    ("fheram_roundoff_reset", C.c_int, [C.c_void_p]),
    ("fheram_ctx_create_sharded", C.c_int, [C.POINTER(_CParams), C.c_int, C.c_int, C.c_int, C.POINTER(C.c_void_p)]),
    ("fheram_config_default", None, [C.POINTER(_CConfig)]),
    ("fheram_ctx_create_cfg", C.c_int, [C.POINTER(_CParams), C.c_int, C.c_int, C.c_int, C.POINTER(_CConfig), C.POINTER(C.c_void_p)]),
    ("fheram_ctx_config", C.c_int, [C.c_void_p, C.POINTER(_CConfig)]),
    ("fheram_shard_info", C.c_int, [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_size_t)]),
    ("fheram_stream_signal", C.c_int, [C.c_void_p, C.c_void_p]),
    ("fheram_stream_wait", C.c_int, [C.c_void_p, C.c_void_p]),
    ("fheram_write_begin", C.c_int, [C.c_void_p, C.c_void_p]),
    ("fheram_device_malloc", C.c_int, [C.c_void_p, C.c_size_t, C.POINTER(C.c_void_p)]),
    ("fheram_device_free", C.c_int, [C.c_void_p, C.c_void_p]),
    ("fheram_read_partial", C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int]),
    ("fheram_read_finish", C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, I64P]),
    ("fheram_write_root", C.c_int, [C.c_void_p, I64P, C.c_int, C.c_void_p, C.c_void_p, C.c_int]),
    ("fheram_write_shard", C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]),
    ("fheram_glwe_external_product", C.c_int, [C.c_void_p, I64P, C.c_int, I64P, I64P]),
    ("fheram_glwe_automorphism", C.c_int, [C.c_void_p, C.c_int, C.c_int64, I64P, C.c_int, I64P]),
    ("fheram_glwe_trace", C.c_int, [C.c_void_p, C.c_int, C.c_int, I64P, C.c_int, I64P]),
    ("fheram_glwe_pack", C.c_int, [C.c_void_p, I64P, C.c_int, I64P]),
    ("fheram_ggsw_automorphism_inv", C.c_int, [C.c_void_p, I64P, I64P]),
    ("fheram_secret_create", C.c_int, [C.c_void_p, I64P, C.POINTER(C.c_void_p)]),
    ("fheram_secret_destroy", None, [C.c_void_p]),
    ("fheram_glwe_encrypt_sk", C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, I64P, C.c_int, C.c_int, I64P, I64P, I64P]),
    ("fheram_glwe_decrypt", C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, I64P, I64P]),
    ("fheram_ram_encrypt_sk", C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(C.c_uint8), C.c_size_t, I64P, I64P]),
    ("fheram_address_encrypt_sk", C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32, I64P, I64P, C.POINTER(C.c_void_p)]),
    ("fheram_address_download", C.c_int, [C.c_void_p, C.c_void_p, I64P]),
    ("fheram_keys_encrypt_sk", C.c_int, [C.c_void_p, C.c_void_p, I64P, I64P, I64P]),
    ("fheram_fheuint_ggsw_len", C.c_size_t, [C.c_void_p]),
    ("fheram_fheuint_create", C.c_int, [C.c_void_p, I64P, C.c_int, C.POINTER(C.c_void_p)]),
    ("fheram_fheuint_encrypt_sk", C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32, C.c_int, I64P, I64P, C.POINTER(C.c_void_p)]),
    ("fheram_fheuint_download", C.c_int, [C.c_void_p, C.c_void_p, I64P]),
    ("fheram_fheuint_destroy", None, [C.c_void_p]),
    ("fheram_address_set_from_fheuint", C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_void_p)]),
    ("fheram_timer_begin", C.c_int, [C.c_void_p]),
    ("fheram_timer_end", C.c_int, [C.c_void_p, C.POINTER(C.c_float)]),
    ("fheram_profile_enable", C.c_int, [C.c_void_p, C.c_int]),
    ("fheram_profile_get", C.c_int, [C.c_void_p, C.c_char_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_double)]),
    ("fheram_tail_stats", C.c_int, [C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    ("fheram_mid_stats", C.c_int, [C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    ("fheram_mid_state", C.c_int, [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_uint64)]),
    ("fheram_profile_reset", C.c_int, [C.c_void_p]),
    ("fheram_bench_external_product", C.c_int, [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_float)]),
    ("fheram_bench_chain", C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_float)]),
    ("fheram_device_info", C.c_int, [C.c_void_p, C.c_char_p, C.c_size_t, C.POINTER(C.c_int)]),
    ("fheram_group_create", C.c_int, [C.POINTER(_CParams), C.POINTER(C.c_int), C.c_int, C.POINTER(C.c_void_p)]),
    ("fheram_group_destroy", None, [C.c_void_p]),
    ("fheram_group_last_error", C.c_char_p, [C.c_void_p]),
    ("fheram_group_size", C.c_int, [C.c_void_p]),
    ("fheram_group_ctx", C.c_void_p, [C.c_void_p, C.c_int]),
    ("fheram_group_keys_load", C.c_int, [C.c_void_p, I64P, C.c_int, C.POINTER(I64P), I64P, C.c_int64, I64P]),
    ("fheram_group_ram_upload", C.c_int, [C.c_void_p, I64P]),
    ("fheram_group_ram_download", C.c_int, [C.c_void_p, I64P]),
    ("fheram_group_ram_tree_download", C.c_int, [C.c_void_p, C.c_int, I64P]),
    ("fheram_group_ram_state", C.c_int, [C.c_void_p]),
    ("fheram_group_address_create", C.c_int, [C.c_void_p, C.POINTER(I64P), C.c_int, C.POINTER(C.c_void_p)]),
    ("fheram_group_address_destroy", None, [C.c_void_p]),
    ("fheram_group_read", C.c_int, [C.c_void_p, C.c_void_p, I64P]),
    ("fheram_group_read_prepare_write", C.c_int, [C.c_void_p, C.c_void_p, I64P]),
    ("fheram_group_write", C.c_int, [C.c_void_p, I64P, C.c_int, C.c_void_p]),
    ("fheram_group_word_stage", C.c_int, [C.c_void_p, I64P, C.c_int]),
    ("fheram_group_result_download", C.c_int, [C.c_void_p, I64P]),
    ("fheram_group_peer_info", C.c_int, [C.c_void_p, C.POINTER(C.c_int), C.c_int]),
    ("fheram_group_poisoned", C.c_int, [C.c_void_p]),
    ("fheram_group_roundoff_max", C.c_int, [C.c_void_p, C.POINTER(C.c_double)]),
    ("fheram_selftest_convolve", C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_double), C.c_int]),
    ("fheram_selftest_convolve_rounded", C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_double), C.c_double]),
]


def exported_symbols() -> List[str]:
    return [s[0] for s in _SYMBOLS]


def library():
    """Load the HIP C-ABI library.  Fails loudly when it has not been built."""
    global _LIB
    if _LIB is None:
        path = library_path()
        if not os.path.exists(path):
            raise FheRamError(7, f"HIP extension missing: {path} not built (run __graft_entry__.build()); "
                                 "there is no CPU fallback")
        L = C.CDLL(path)
        lenient = bool(os.environ.get("FHERAM_LIB"))   # an experimental build named by FHERAM_LIB (A/B runs) may predate newer entry points
        for name, res, args in _SYMBOLS:
            if lenient and not hasattr(L, name):
                continue
            f = getattr(L, name)  # AttributeError if the .so does not export a declared symbol
            f.restype = res
            f.argtypes = args
        _LIB = L
    return _LIB


def _p(a: np.ndarray):
    assert a.dtype == np.int64 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(I64P)


def _i64(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.int64)


def galois_elements(log_n: int = 12) -> List[int]:
    """GLWE::trace_galois_elements (keys.rs:39,158): -1, then 5^(2^(i-1)) mod 2N."""
    two_n = 2 << log_n
    return [-1] + [pow(5, 1 << (i - 1), two_n) for i in range(1, log_n)]


class Parameters:
    """parameters.rs:147-176.  Defaults are the source constants (parameters.rs:11-21).  The layout factories of the
    reference are generic in every precision (parameters.rs:53-104); the kernels accept any precision inside the limb
    counts of its two published blocks (source constants and README.md:17-27)."""

    def __init__(self, max_addr: int = 1 << 14, decomp_n: Sequence[int] = (3, 3, 3, 3), word_size: int = 4,
                 k_glwe_pt: int = 3, k_glwe_ct: int = 51, k_ggsw_addr: int = 68, k_evk_trace: int = 68, k_evk_ggsw_inv: int = 85):
        self.log_n, self._base2k, self._rank = 12, 17, 1
        self._k_glwe_pt, self._k_glwe_ct, self._k_ggsw_addr = int(k_glwe_pt), int(k_glwe_ct), int(k_ggsw_addr)
        self._k_evk_trace, self._k_evk_ggsw_inv = int(k_evk_trace), int(k_evk_ggsw_inv)
        assert sum(decomp_n) == self.log_n  # parameters.rs:168
        self._max_addr, self._decomp_n, self._word_size = int(max_addr), [int(x) for x in decomp_n], int(word_size)

    @classmethod
    def new(cls):  # parameters.rs:167
        return cls()

    @classmethod
    def readme(cls, max_addr: int = 1 << 18, word_size: int = 4):
        """The parameter block of README.md:17-34, with which the published 450 ms / 1200 ms were taken:
        K_PT = 9, K_CT = 3*17, K_ADDR = 4*17, K_EVK = 5*17 for EVERY evaluation key (5-limb trace keys), MAX_ADDR = 2^18."""
        return cls(max_addr=max_addr, word_size=word_size, k_glwe_pt=9, k_evk_trace=85, k_evk_ggsw_inv=85)

    def n(self):
        return 1 << self.log_n

    def max_addr(self):  # parameters.rs:237
        return self._max_addr

    def basek(self):
        return self._base2k

    def rank(self):
        return self._rank

    def k_glwe_ct(self):
        return self._k_glwe_ct

    def k_glwe_pt(self):
        return self._k_glwe_pt

    def k_ggsw_addr(self):
        return self._k_ggsw_addr

    def k_evk_trace(self):
        return self._k_evk_trace

    def k_evk_ggsw_inv(self):
        return self._k_evk_ggsw_inv

    def word_size(self):
        return self._word_size

    def decomp_n(self):
        return list(self._decomp_n)

    def dnum_ct(self):  # parameters.rs:273-275
        return -(-self._k_glwe_ct // self._base2k)

    def dnum_ggsw(self):  # parameters.rs:277-279
        return -(-self._k_ggsw_addr // self._base2k)

    def base2d(self) -> Base2D:  # parameters.rs:285-287
        return get_base_2d(self._max_addr, self._decomp_n)

    def rows(self):
        return -(-self._max_addr // self.n())

    # element counts of the host layouts
    def glwe_len(self):
        return self.dnum_ct() * 2 * self.n()

    def ggsw_len(self):
        return self.dnum_ct() * 2 * self.dnum_ggsw() * 2 * self.n()

    def _c(self) -> _CParams:
        cp = _CParams()
        library().fheram_params_default(C.byref(cp))
        cp.k_glwe_pt, cp.k_glwe_ct, cp.k_ggsw_addr = self._k_glwe_pt, self._k_glwe_ct, self._k_ggsw_addr
        cp.k_evk_trace, cp.k_evk_ggsw_inv = self._k_evk_trace, self._k_evk_ggsw_inv
        cp.max_addr = self._max_addr
        cp.word_size = self._word_size
        cp.n_decomp = len(self._decomp_n)
        for i, d in enumerate(self._decomp_n):
            cp.decomp_n[i] = d
        return cp


class EvaluationKeysPrepared:
    """keys.rs:27-71.  Holds the std-form keys; `prepare` happens on the device when a Ram first
    uses them (fheram_keys_load)."""

    def __init__(self, gal_els: Sequence[int], atk_glwe, atk_ggsw_inv, tsk_ggsw_inv, atk_ggsw_inv_p: int = -1):
        self.gal_els = _i64(gal_els)
        self.atk_glwe = [_i64(k).ravel() for k in atk_glwe]
        self.atk_ggsw_inv = _i64(atk_ggsw_inv).ravel()
        self.tsk_ggsw_inv = _i64(tsk_ggsw_inv).ravel()
        self.atk_ggsw_inv_p = int(atk_ggsw_inv_p)

    @classmethod
    def from_dict(cls, evk: Dict):
        return cls(evk["gal_els"], list(evk["atk_glwe"]), evk["atk_ggsw_inv"], evk["tsk"])

    @classmethod
    def encrypt_sk(cls, ram: "Ram", sk: "GLWESecret", source_xa, source_xe, keep_std: bool = False):
        """EvaluationKeys::encrypt_sk (keys.rs:135-180) + prepare (keys.rs:57-71) on `ram`'s device.
        The sources stay on the host: source_xa.uniform_limbs(count), source_xe.gaussian(count, scale)
        are drawn in the reference's order (12 trace keys, tensor key, p = -1 key).  With keep_std the
        std forms come back too, so the keys can be loaded into other contexts."""
        p = ram.params
        n, b2k = p.n(), p.basek()
        s4, s5 = -(-p.k_evk_trace() // b2k), -(-p.k_evk_ggsw_inv() // b2k)
        n4, n5 = p.log_n * p.dnum_ct(), 2 * p.dnum_ggsw()
        mask = np.concatenate([source_xa.uniform_limbs(n4 * s4 * n), source_xa.uniform_limbs(n5 * s5 * n)])
        noise = np.concatenate([source_xe.gaussian(n4 * n, noise_scale(p.k_evk_trace(), b2k)),
                                source_xe.gaussian(n5 * n, noise_scale(p.k_evk_ggsw_inv(), b2k))])
        L = library()
        atk_len, inv_len = L.fheram_atk_len(ram._h), L.fheram_evk_inv_len(ram._h)
        std = np.zeros(p.log_n * atk_len + 2 * inv_len, dtype=np.int64) if keep_std else None
        ram._chk(L.fheram_keys_encrypt_sk(ram._h, sk._h, _p(mask), _p(noise), _p(std) if keep_std else None))
        gal = galois_elements(p.log_n)
        if keep_std:
            o = p.log_n * atk_len
            keys = cls(gal, [std[i * atk_len:(i + 1) * atk_len] for i in range(p.log_n)], std[o + inv_len:o + 2 * inv_len], std[o:o + inv_len])
        else:
            keys = cls.__new__(cls)
            keys.gal_els, keys.atk_glwe, keys.atk_ggsw_inv, keys.tsk_ggsw_inv, keys.atk_ggsw_inv_p = _i64(gal), None, None, None, -1
        ram._keys = keys   # already prepared on this context
        return keys


class Address:
    """address.rs:21-24: one Coordinate (list of GGSW digits) per Base1D of the plan."""

    def __init__(self, params: Parameters, ggsw_digits):
        self.base2d = params.base2d()
        n_digits = self.base2d.as_1d().size()
        self._digits = [_i64(g).ravel() for g in ggsw_digits]
        if len(self._digits) != n_digits:
            raise FheRamError(1, f"address needs {n_digits} GGSW digits, got {len(self._digits)}")
        self._handles = {}  # ctx handle -> device address

    @property
    def digits(self):
        """std-form GGSW digits on the host; an address encrypted on the device downloads them on first use"""
        if self._digits is None:
            h = next(iter(self._handles.values()))
            n_digits = self.base2d.as_1d().size()
            out = np.zeros((n_digits, h.ram.params.ggsw_len()), dtype=np.int64)
            h.ram._chk(library().fheram_address_download(h.ram._h, h.h, _p(out)))
            self._digits = [out[i] for i in range(n_digits)]
        return self._digits

    @classmethod
    def encrypt_sk(cls, ram: "Ram", value: int, sk: "GLWESecret", source_xa, source_xe):
        """Address::encrypt_sk (address.rs:86-109) on `ram`'s device with host-side sources (see
        EvaluationKeysPrepared.encrypt_sk).  The GGSW digits never touch the host unless `.digits` is read."""
        p = ram.params
        self = cls.__new__(cls)
        self.base2d = p.base2d()
        n_digits = self.base2d.as_1d().size()
        n, b2k = p.n(), p.basek()
        size = -(-p.k_ggsw_addr() // b2k)
        n_glwe = n_digits * p.dnum_ct() * 2
        mask = source_xa.uniform_limbs(n_glwe * size * n)
        noise = source_xe.gaussian(n_glwe * n, noise_scale(p.k_ggsw_addr(), b2k))
        out = C.c_void_p()
        ram._chk(library().fheram_address_encrypt_sk(ram._h, sk._h, int(value), _p(mask), _p(noise), C.byref(out)))
        self._digits = None
        self._handles = {id(ram): _AddrHandle(out.value, ram)}
        return self

    @classmethod
    def set_from_fheuint(cls, ram: "Ram", fheuint: "FheUintPrepared", sign: bool = True):
        """Address::set_from_fheuint (conversion.rs:68-82): every digit derived from the encrypted integer on the
        device.  sign=True: digits of X^{+digit} (what the reference's test decrypts to); sign=False: X^{-digit}, the
        convention of Address::encrypt_sk (address.rs:102-108), i.e. an address Ram.read accepts."""
        self = cls.__new__(cls)
        self.base2d = ram.params.base2d()
        out = C.c_void_p()
        ram._chk(library().fheram_address_set_from_fheuint(ram._h, fheuint._h, int(bool(sign)), C.byref(out)))
        self._digits = None
        self._handles = {id(ram): _AddrHandle(out.value, ram)}
        return self

    @classmethod
    def alloc_from_params(cls, params: Parameters):  # address.rs:58
        glen = params.ggsw_len()
        return cls(params, [np.zeros(glen, dtype=np.int64) for _ in range(params.base2d().as_1d().size())])

    def n2(self):  # address.rs:113
        return len(self.base2d.v)

    def at(self, i):  # address.rs:117
        s = sum(b.size() for b in self.base2d.v[:i])
        return self.digits[s:s + self.base2d.v[i].size()]

    def _group(self, grp: "GroupRam"):
        h = self._handles.get(id(grp))
        if h is None:
            arr = (I64P * len(self.digits))(*[_p(d) for d in self.digits])
            out = C.c_void_p()
            grp._chk(library().fheram_group_address_create(grp._h, arr, len(self.digits), C.byref(out)))
            h = _GroupAddrHandle(out.value, grp)
            self._handles[id(grp)] = h
        return h.h

    def _device(self, ram: "Ram"):
        h = self._handles.get(id(ram))
        if h is None:
            L = library()
            arr = (I64P * len(self.digits))(*[_p(d) for d in self.digits])
            out = C.c_void_p()
            ram._chk(L.fheram_address_create(ram._h, arr, len(self.digits), C.byref(out)))
            h = _AddrHandle(out.value, ram)
            self._handles[id(ram)] = h
        return h.h


class _GroupAddrHandle:
    def __init__(self, h, grp):
        self.h = h
        self.ram = grp

    def __del__(self):
        if self.h and _LIB is not None:
            _LIB.fheram_group_address_destroy(self.h)
            self.h = None


class FheUintPrepared:
    """poulpy-schemes' FheUintPrepared<u32> as the RAM path needs it (conversion.rs:68-82): one GGSW per bit of an
    encrypted integer, LSB first, on a Ram's device.  Layout: k = k_evk_ggsw_inv, 5 limbs, dnum 4 (include/fheram.h)."""

    def __init__(self, ram: "Ram", handle, n_bits):
        self.ram, self._h, self.n_bits = ram, handle, n_bits

    @classmethod
    def from_host(cls, ram: "Ram", bits):
        bits = _i64(bits)
        out = C.c_void_p()
        ram._chk(library().fheram_fheuint_create(ram._h, _p(bits), bits.shape[0], C.byref(out)))
        return cls(ram, out.value, bits.shape[0])

    @classmethod
    def encrypt_sk(cls, ram: "Ram", value: int, sk: "GLWESecret", source_xa, source_xe, n_bits: int = 32):
        """FheUintPrepared::encrypt_sk (conversion.rs:160-168) with host-side sources, as the other encrypt_sk mirrors."""
        p = ram.params
        n, b2k, k = p.n(), p.basek(), p.k_evk_ggsw_inv()
        size = -(-k // b2k)
        n_glwe = n_bits * p.dnum_ggsw() * 2
        mask = source_xa.uniform_limbs(n_glwe * size * n)
        noise = source_xe.gaussian(n_glwe * n, noise_scale(k, b2k))
        out = C.c_void_p()
        ram._chk(library().fheram_fheuint_encrypt_sk(ram._h, sk._h, int(value), n_bits, _p(mask), _p(noise), C.byref(out)))
        return cls(ram, out.value, n_bits)

    def download(self) -> np.ndarray:
        out = np.zeros((self.n_bits, library().fheram_fheuint_ggsw_len(self.ram._h)), dtype=np.int64)
        self.ram._chk(library().fheram_fheuint_download(self.ram._h, self._h, _p(out)))
        return out

    def __del__(self):
        if getattr(self, "_h", None) and _LIB is not None:
            _LIB.fheram_fheuint_destroy(self._h)
            self._h = None


class _AddrHandle:
    def __init__(self, h, ram):
        self.h = h
        self.ram = ram  # keeps the owning context alive for as long as the device address exists

    def __del__(self):
        if self.h and _LIB is not None:
            _LIB.fheram_address_destroy(self.h)
            self.h = None


def cast_u8_to_signed(value: int, bit_length: int) -> int:
    """examples/fhe-ram.rs:25-32"""
    shift = 8 - bit_length
    v = (value << shift) & 0xFF
    v = v - 256 if v >= 128 else v
    return v >> shift


def expected_plain(value: int, k_pt: int, written: bool = False) -> int:
    """cast_u8_to_signed generalised to any plaintext precision (README.md:20 has K_PT = 9, beyond the 8 bits the example's
    helper asserts): Ram::encrypt_sk encodes `(x as i8) as i64` (ram.rs:361-363), encrypt_glwe `value as i64`
    (examples/fhe-ram.rs:196); the torus keeps either mod 2^k_pt, centred."""
    v = int(value) if written else (int(value) - 256 if int(value) >= 128 else int(value))
    m = 1 << k_pt
    v %= m
    return v - m if v >= m // 2 else v


def encode_coeff(value: int, k: int, base2k: int = 17):
    """encode_coeff_i64(value, k, idx): value * 2^-k on ceil(k/base2k) normalised limbs"""
    size = -(-k // base2k)
    x = value << (size * base2k - k)
    limbs = [0] * size
    for j in range(size - 1, -1, -1):
        d = ((x + (1 << (base2k - 1))) % (1 << base2k)) - (1 << (base2k - 1))
        limbs[j] = d
        x = (x - d) >> base2k
    return limbs


def noise_scale(k: int, base2k: int = 17) -> float:
    """Noise of a ciphertext at precision k sits on limb ceil(k/base2k)-1, scaled by 2^((limb+1)*base2k - k)
    (Poulpy add_normal): the factor the host sampler multiplies sigma by."""
    return float(1 << (-(-k // base2k) * base2k - k))


class GLWESecret:
    """GLWESecret + GLWESecretPrepared (examples/fhe-ram.rs:49-59) on a Ram's device: coefficients in {-1,0,1}."""

    def __init__(self, ram: "Ram", sk):
        self.ram = ram
        self._h = None
        out = C.c_void_p()
        ram._chk(library().fheram_secret_create(ram._h, _p(_i64(sk)), C.byref(out)))
        self._h = out.value

    def __del__(self):
        if getattr(self, "_h", None) and _LIB is not None:
            _LIB.fheram_secret_destroy(self._h)
            self._h = None


class Ram:
    """ram.rs:25-29.  Owns the device-resident sub-RAMs, tree, packer scratch and prepared keys."""

    def __init__(self, params: Optional[Parameters] = None, device: int = 0, shard: int = 0, n_shards: int = 1, config: Optional[dict] = None):
        """shard / n_shards: row sharding across GPUs; this context then owns rows
        r = shard (mod n_shards) of every sub-RAM and only the *_partial / *_root / *_shard ops apply.
        config: execution switches that differ from fheram_config_default() (fheram_config's field names)."""
        self.params = params or Parameters.new()
        self.shard, self.n_shards = int(shard), int(n_shards)
        self._h = None
        L = library()
        out = C.c_void_p()
        cp = self.params._c()
        if config:
            cfg = _CConfig()
            L.fheram_config_default(C.byref(cfg))
            for k, v in config.items():
                if k not in _CONFIG_FIELDS:
                    raise FheRamError(1, f"unknown execution switch {k!r}")
                setattr(cfg, k, int(v))
            rc = L.fheram_ctx_create_cfg(C.byref(cp), device, self.shard, self.n_shards, C.byref(cfg), C.byref(out))
        else:
            rc = L.fheram_ctx_create_sharded(C.byref(cp), device, self.shard, self.n_shards, C.byref(out))
        if rc != 0:
            raise FheRamError(rc, L.fheram_last_error(None).decode())
        self._h = out.value
        self._keys = None

    def forms(self) -> dict:
        """the execution switches in effect on this context (fheram_ctx_config)"""
        cfg = _CConfig()
        self._chk(library().fheram_ctx_config(self._h, C.byref(cfg)))
        return {f: int(getattr(cfg, f)) for f in _CONFIG_FIELDS if f != "reserved"}

    @classmethod
    def new(cls, device: int = 0):  # ram.rs:59
        return cls(Parameters.new(), device)

    @classmethod
    def new_from_ram_params(cls, word_size: int, decomp_n: Sequence[int], max_addr: int, device: int = 0, **crypto):  # ram.rs:72
        """crypto: k_glwe_pt / k_evk_trace / ... (CryptographicParameters, parameters.rs:23-32), default = source constants"""
        return cls(Parameters(max_addr=max_addr, decomp_n=decomp_n, word_size=word_size, **crypto), device)

    def __del__(self):
        if getattr(self, "_h", None) and _LIB is not None:
            _LIB.fheram_ctx_destroy(self._h)
            self._h = None

    def _chk(self, rc):
        if rc != 0:
            raise FheRamError(rc, library().fheram_last_error(self._h).decode())

    def _use_keys(self, keys: EvaluationKeysPrepared):
        if self._keys is keys:
            return
        if keys.atk_glwe is None:
            raise FheRamError(5, "these keys were generated on another context's device (EvaluationKeysPrepared.encrypt_sk "
                                 "without keep_std=True): their std forms are not on the host")
        L = library()
        atk_len, inv_len = L.fheram_atk_len(self._h), L.fheram_evk_inv_len(self._h)
        if any(k.size != atk_len for k in keys.atk_glwe) or keys.atk_ggsw_inv.size != inv_len or keys.tsk_ggsw_inv.size != inv_len:
            raise FheRamError(1, f"evaluation-key layout does not match the context's (trace keys of {atk_len} limbs-elements, "
                                 f"inverse / tensor keys of {inv_len}: evk_glwe_infos / evk_ggsw_infos, parameters.rs:71-93)")
        arr = (I64P * len(keys.atk_glwe))(*[_p(k) for k in keys.atk_glwe])
        self._chk(L.fheram_keys_load(self._h, _p(keys.gal_els), len(keys.gal_els), arr, _p(keys.atk_ggsw_inv),
                                     keys.atk_ggsw_inv_p, _p(keys.tsk_ggsw_inv)))
        self._keys = keys

    def _out(self):
        return np.zeros((self.params.word_size(), self.params.glwe_len()), dtype=np.int64)

    # -- data hand-over (Ram::encrypt_sk output, ram.rs:129-167)
    def local_rows(self) -> int:
        return self.params.rows() // self.n_shards

    def load_encrypted(self, rows: np.ndarray):
        """rows: [word_size][rows][GLWE]; a sharded context takes its own rows (rows[:, shard::n_shards])."""
        p = self.params
        rows = _i64(rows)
        if rows.size != p.word_size() * self.local_rows() * p.glwe_len():
            raise FheRamError(1, f"invalid data: expected {p.word_size()}x{self.local_rows()} GLWE rows (ram.rs:144-155)")
        self._chk(library().fheram_ram_upload(self._h, _p(rows)))

    def encrypt_sk(self, data, sk: GLWESecret, source_xa, source_xe):
        """Ram::encrypt_sk (ram.rs:129-167) on the device.  data: max_addr*word_size bytes.  The sources are
        host objects (uniform_limbs(count) / gaussian(count, scale)); draws are made for every row of the
        RAM in the reference's order and a sharded context keeps those of its own rows."""
        p = self.params
        data = np.ascontiguousarray(data, dtype=np.uint8).ravel()
        n, b2k = p.n(), p.basek()
        size = -(-p.k_glwe_ct() // b2k)
        ws, rows = p.word_size(), p.rows()
        mask = source_xa.uniform_limbs(ws * rows * size * n).reshape(ws, rows, size * n)
        noise = source_xe.gaussian(ws * rows * n, noise_scale(p.k_glwe_ct(), b2k)).reshape(ws, rows, n)
        if self.n_shards > 1:
            mask = np.ascontiguousarray(mask[:, self.shard::self.n_shards])
            noise = np.ascontiguousarray(noise[:, self.shard::self.n_shards])
        self._chk(library().fheram_ram_encrypt_sk(self._h, sk._h, data.ctypes.data_as(C.POINTER(C.c_uint8)), data.size,
                                                  _p(mask), _p(noise)))

    def glwe_encrypt_sk(self, sk: GLWESecret, n_glwe: int, size: int, k: int, pt, pt_col: int, source_xa, source_xe):
        """GLWE::encrypt_sk on n_glwe ciphertexts (examples/fhe-ram.rs:179-210); pt [n_glwe][pt_size][N] or None."""
        n = self.params.n()
        mask = source_xa.uniform_limbs(n_glwe * size * n)
        noise = source_xe.gaussian(n_glwe * n, noise_scale(k, self.params.basek()))
        out = np.zeros((n_glwe, size * 2 * n), dtype=np.int64)
        pt_size = 0
        if pt is not None:
            pt = _i64(pt).reshape(n_glwe, -1, n)
            pt_size = pt.shape[1]
        self._chk(library().fheram_glwe_encrypt_sk(self._h, sk._h, n_glwe, size, k, _p(pt) if pt is not None else None,
                                                   pt_size, pt_col, _p(mask), _p(noise), _p(out)))
        return out

    def glwe_decrypt(self, sk: GLWESecret, cts, size: int = 3):
        """GLWE::decrypt (examples/fhe-ram.rs:217-222): normalised plaintext limbs [n_glwe][size][N]."""
        n = self.params.n()
        cts = _i64(cts).reshape(-1, size * 2 * n)
        pt = np.zeros((cts.shape[0], size, n), dtype=np.int64)
        self._chk(library().fheram_glwe_decrypt(self._h, sk._h, cts.shape[0], size, _p(cts), _p(pt)))
        return pt

    def encrypt_word(self, sk: GLWESecret, values, source_xa, source_xe):
        """encrypt_glwe of the example (examples/fhe-ram.rs:179-210): one GLWE per byte, value on coefficient 0
        at precision k_pt."""
        p = self.params
        values = [int(v) for v in values]
        size_pt = -(-p.k_glwe_pt() // p.basek())
        pt = np.zeros((len(values), size_pt, p.n()), dtype=np.int64)
        for i, v in enumerate(values):
            pt[i, :, 0] = encode_coeff(v, p.k_glwe_pt(), p.basek())
        return self.glwe_encrypt_sk(sk, len(values), -(-p.k_glwe_ct() // p.basek()), p.k_glwe_ct(), pt, 0, source_xa, source_xe)

    def decrypt_coeff(self, sk: GLWESecret, cts, wants, coeff: int = 0):
        """decrypt_glwe + noise metric of the example (examples/fhe-ram.rs:212-237) for each ct:
        returns [(value, log2 noise)]."""
        p = self.params
        k, b2k = p.k_glwe_ct(), p.basek()
        size = -(-k // b2k)
        pt = self.glwe_decrypt(sk, cts, size)
        out = []
        for limbs, want in zip(pt[:, :, coeff], wants):
            res = 0
            rem = b2k - (k % b2k)
            for j in range(size):                       # decode_coeff_i64(k, coeff)
                x = int(limbs[j])
                res = (res << (b2k - rem)) + (x >> rem) if (j == size - 1 and rem != b2k) else (res << b2k) + x
            log_scale = k - p.k_glwe_pt()                # :228
            diff = res - (int(want) << log_scale)        # :230
            noise = float(np.log2(abs(diff))) - k if diff != 0 else float("-inf")   # :231
            mag = int(np.floor(abs(res) / (1 << log_scale) + 0.5))   # round half away from zero  :232-233
            out.append((mag if res >= 0 else -mag, noise))
        return out

    def store_encrypted(self) -> np.ndarray:
        p = self.params
        rows = np.zeros((p.word_size(), self.local_rows(), p.glwe_len()), dtype=np.int64)
        self._chk(library().fheram_ram_download(self._h, _p(rows)))
        return rows

    def tree(self, level: int = 0) -> np.ndarray:
        out = self._out()
        self._chk(library().fheram_ram_tree_download(self._h, level, _p(out)))
        return out

    @property
    def state(self) -> bool:
        return bool(library().fheram_ram_state(self._h))

    # -- the path
    def _check_out(self, out):
        """a caller-owned result buffer is written by the library (word_size * GLWE int64): refuse anything it could overrun"""
        p = self.params
        if not (isinstance(out, np.ndarray) and out.dtype == np.int64 and out.flags["C_CONTIGUOUS"] and out.flags["WRITEABLE"]
                and out.size == p.word_size() * p.glwe_len()):
            raise FheRamError(1, f"out must be a writeable C-contiguous int64 array of {p.word_size()} x {p.glwe_len()} elements")

    def read(self, address: Address, keys: EvaluationKeysPrepared, download: bool = True, out=None):  # ram.rs:172
        """out: an int64 array [word_size][GLWE] to receive the result (a host that reads in a loop reuses one)"""
        self._use_keys(keys)
        if download and out is None:
            out = self._out()
        elif download:
            self._check_out(out)
        self._chk(library().fheram_read(self._h, address._device(self), _p(out) if download else None))
        return out if download else None

    def read_prepare_write(self, address: Address, keys: EvaluationKeysPrepared, download: bool = True, out=None):  # ram.rs:196
        self._use_keys(keys)
        if download and out is None:
            out = self._out()
        elif download:
            self._check_out(out)
        self._chk(library().fheram_read_prepare_write(self._h, address._device(self), _p(out) if download else None))
        return out if download else None

    def write(self, w, address: Address, keys: EvaluationKeysPrepared):  # ram.rs:226
        self._use_keys(keys)
        if w is None:
            self._chk(library().fheram_write(self._h, None, self.params.word_size(), address._device(self)))
            return
        w = _i64(w)
        n_w = w.shape[0] if w.ndim > 1 else w.size // self.params.glwe_len()
        self._chk(library().fheram_write(self._h, _p(w), n_w, address._device(self)))

    # -- row-sharded path (every buffer: host int64 ndarray, or (device_ptr, True) for an int32 device buffer)
    @staticmethod
    def _buf(b):
        if isinstance(b, tuple):
            return C.c_void_p(int(b[0])), 1
        return b.ctypes.data_as(C.c_void_p), 0

    def read_partial(self, address: Address, keys: EvaluationKeysPrepared, prepare_write: bool = False, out=None):
        self._use_keys(keys)
        if out is None:
            out = self._out()
        ptr, dev = self._buf(out)
        self._chk(library().fheram_read_partial(self._h, address._device(self), int(prepare_write), ptr, dev))
        return out

    def read_finish(self, address: Address, keys: EvaluationKeysPrepared, partials, prepare_write: bool = False, download: bool = True):
        self._use_keys(keys)
        if not isinstance(partials, tuple):
            partials = _i64(partials)
        ptr, dev = self._buf(partials)
        out = self._out() if download else None
        self._chk(library().fheram_read_finish(self._h, address._device(self), int(prepare_write), ptr, dev, _p(out) if download else None))
        return out

    def write_root(self, w, address: Address, keys: EvaluationKeysPrepared, out=None):
        self._use_keys(keys)
        if out is None:
            out = self._out()
        ptr, dev = self._buf(out)
        wp = None if w is None else _p(_i64(w))
        self._chk(library().fheram_write_root(self._h, wp, self.params.word_size(), address._device(self), ptr, dev))
        return out

    def write_begin(self, address: Address, keys: EvaluationKeysPrepared):
        """start the part of a write that does not need ct_lo (overlaps the root's work and the broadcast)"""
        self._use_keys(keys)
        self._chk(library().fheram_write_begin(self._h, address._device(self)))

    def stream_signal(self, hip_stream: int):
        """work enqueued later on `hip_stream` (a hipStream_t as int, 0 = default stream) waits for this context"""
        self._chk(library().fheram_stream_signal(self._h, C.c_void_p(int(hip_stream))))

    def stream_wait(self, hip_stream: int):
        """work this context enqueues later waits for everything enqueued on `hip_stream` so far"""
        self._chk(library().fheram_stream_wait(self._h, C.c_void_p(int(hip_stream))))

    def device_malloc(self, n_bytes: int) -> int:
        """device memory on this context's GPU for exchange buffers (pass (ptr, True) to the sharded ops)"""
        out = C.c_void_p()
        self._chk(library().fheram_device_malloc(self._h, n_bytes, C.byref(out)))
        return int(out.value)

    def device_free(self, ptr: int):
        self._chk(library().fheram_device_free(self._h, C.c_void_p(int(ptr))))

    def write_shard(self, address: Address, keys: EvaluationKeysPrepared, ct_lo):
        self._use_keys(keys)
        if not isinstance(ct_lo, tuple):
            ct_lo = _i64(ct_lo)
        ptr, dev = self._buf(ct_lo)
        self._chk(library().fheram_write_shard(self._h, address._device(self), ptr, dev))

    def stage_words(self, w):
        w = _i64(w)
        self._chk(library().fheram_word_stage(self._h, _p(w), w.shape[0]))

    def result(self):
        out = self._out()
        self._chk(library().fheram_result_download(self._h, _p(out)))
        return out

    def result_view(self) -> np.ndarray:
        """the result of the last read in place (the context's pinned host buffer, fheram_result_map): valid until the
        next operation on this Ram — copy it if it has to live longer"""
        ptr = I64P()
        self._chk(library().fheram_result_map(self._h, C.byref(ptr)))
        p = self.params
        return np.ctypeslib.as_array(ptr, shape=(p.word_size(), p.glwe_len()))

    def sync(self):
        self._chk(library().fheram_sync(self._h))

    # -- the exactness contract of the FFT64 arithmetic, checked (fheram_roundoff_max)
    def roundoff_max(self, check: bool = True) -> float:
        """largest |x - rint(x)| any monitored rounding of an inverse transform has seen on this context since its creation / the
        last reset; raises FheRamError(8, PRECISION) above 3/8 unless check is False"""
        m = C.c_double()
        rc = library().fheram_roundoff_max(self._h, C.byref(m))
        if rc != 0 and (check or rc != 8):
            self._chk(rc)
        return float(m.value)

    def roundoff_reset(self):
        self._chk(library().fheram_roundoff_reset(self._h))

    # -- Poulpy-level ops (parity tests / micro-benchmarks)
    def glwe_external_product(self, a, ggsw):
        a = _i64(a).reshape(-1, self.params.glwe_len())
        res = np.zeros_like(a)
        self._chk(library().fheram_glwe_external_product(self._h, _p(a), a.shape[0], _p(_i64(ggsw).ravel()), _p(res)))
        return res

    def glwe_automorphism(self, keys, gal_el, mode, a):
        self._use_keys(keys)
        a = _i64(a).reshape(-1, self.params.glwe_len())
        res = np.zeros_like(a)
        self._chk(library().fheram_glwe_automorphism(self._h, mode, gal_el, _p(a), a.shape[0], _p(res)))
        return res

    def glwe_trace(self, keys, start, end, a):
        self._use_keys(keys)
        a = _i64(a).reshape(-1, self.params.glwe_len())
        res = np.zeros_like(a)
        self._chk(library().fheram_glwe_trace(self._h, start, end, _p(a), a.shape[0], _p(res)))
        return res

    def glwe_pack(self, keys, cts):
        self._use_keys(keys)
        cts = _i64(cts).reshape(-1, self.params.glwe_len())
        out = np.zeros(self.params.glwe_len(), dtype=np.int64)
        self._chk(library().fheram_glwe_pack(self._h, _p(cts), cts.shape[0], _p(out)))
        return out

    def ggsw_automorphism_inv(self, keys, ggsw):
        self._use_keys(keys)
        g = _i64(ggsw).ravel()
        out = np.zeros_like(g)
        self._chk(library().fheram_ggsw_automorphism_inv(self._h, _p(g), _p(out)))
        return out

    # -- measurement hooks
    def timer_begin(self):
        self._chk(library().fheram_timer_begin(self._h))

    def timer_end(self) -> float:
        ms = C.c_float()
        self._chk(library().fheram_timer_end(self._h, C.byref(ms)))
        return float(ms.value)

    def profile_enable(self, on=True):
        self._chk(library().fheram_profile_enable(self._h, int(on)))

    def profile_reset(self):
        self._chk(library().fheram_profile_reset(self._h))

    def profile_get(self, cls: str):
        a, b, ms = C.c_uint64(), C.c_uint64(), C.c_double()
        self._chk(library().fheram_profile_get(self._h, cls.encode(), C.byref(a), C.byref(b), C.byref(ms)))
        return {"launches": int(a.value), "blocks": int(b.value), "ms": float(ms.value)}

    def tail_stats(self):
        """single-launch trace chains since the context was created, and how many of them fell back (fheram_tail_stats)"""
        a, b = C.c_uint64(), C.c_uint64()
        self._chk(library().fheram_tail_stats(self._h, C.byref(a), C.byref(b)))
        return {"launches": int(a.value), "fallbacks": int(b.value)}

    def mid_stats(self):
        """single-launch chains on 9..64 ciphertexts since the context was created, and how many ciphertexts were redone by the launch behind (fheram_mid_stats)"""
        a, b = C.c_uint64(), C.c_uint64()
        self._chk(library().fheram_mid_stats(self._h, C.byref(a), C.byref(b)))
        return {"launches": int(a.value), "fallbacks": int(b.value)}

    def mid_state(self):
        """setting of the single-launch mid chains in effect (0: switched off by the path itself) and how often it has been"""
        en, n = C.c_int(), C.c_uint64()
        self._chk(library().fheram_mid_state(self._h, C.byref(en), C.byref(n)))
        return {"enabled": int(en.value), "times_disabled": int(n.value)}

    def bench_external_product(self, batch: int, iters: int) -> float:
        """ms for a dependent chain of `iters` launches of `batch` GLWE x GGSW products (BASELINE.json configs[1])"""
        ms = C.c_float()
        self._chk(library().fheram_bench_external_product(self._h, batch, iters, C.byref(ms)))
        return float(ms.value)

    # -- self-test of the FP64 FFT arithmetic (tests/test_gpu_fft.py)
    def selftest_convolve(self, a, g, singles: bool = False):
        """raw (unrounded) sums  out[0] = sum_r a_r * g_r,  out[1] = sum_r a_r * g_{r ^ 1}  of negacyclic products of int32 polynomials"""
        a = np.ascontiguousarray(a, dtype=np.int32).reshape(-1, self.params.n())
        g = np.ascontiguousarray(g, dtype=np.int32).reshape(-1, self.params.n())
        assert a.shape == g.shape
        out = np.zeros((2, self.params.n()), dtype=np.float64)
        ip = lambda v: v.ctypes.data_as(C.POINTER(C.c_int32))   # noqa: E731
        self._chk(library().fheram_selftest_convolve(self._h, a.shape[0], ip(a), ip(g), out.ctypes.data_as(C.POINTER(C.c_double)), int(singles)))
        return out

    def selftest_convolve_rounded(self, a, g, operand_scale: float = 1.0):
        """the same sums ROUNDED as the path rounds them (through the round-off monitor); operand_scale = 0.5 drives the monitor over its limit"""
        a = np.ascontiguousarray(a, dtype=np.int32).reshape(-1, self.params.n())
        g = np.ascontiguousarray(g, dtype=np.int32).reshape(-1, self.params.n())
        assert a.shape == g.shape
        out = np.zeros((2, self.params.n()), dtype=np.float64)
        ip = lambda v: v.ctypes.data_as(C.POINTER(C.c_int32))   # noqa: E731
        self._chk(library().fheram_selftest_convolve_rounded(self._h, a.shape[0], ip(a), ip(g), out.ctypes.data_as(C.POINTER(C.c_double)), float(operand_scale)))
        return out

    def bench_chain(self, kind: int, batch: int, n: int, iters: int) -> float:
        """ms for `iters` back-to-back runs of a dependent chain on `batch` ciphertexts: kind 0 = n trace steps, 1 = n external products"""
        ms = C.c_float()
        self._chk(library().fheram_bench_chain(self._h, kind, batch, n, iters, C.byref(ms)))
        return float(ms.value)

    def device_info(self):
        buf = C.create_string_buffer(256)
        cus = C.c_int()
        self._chk(library().fheram_device_info(self._h, buf, 256, C.byref(cus)))
        return {"name": buf.value.decode(), "compute_units": int(cus.value)}


class GroupRam:
    """Ram (ram.rs:25-29) over several GPUs of one node behind ONE handle: the native row-sharded path (fheram_group_*,
    include/fheram.h) for a single-process host.  Same calls as Ram: read / read_prepare_write / write, one per op
    (ram.rs:172-176,196-200,226-231).  `devices`: HIP device indices, a power of two of them (the same index may repeat:
    rehearsal on one GPU)."""

    def __init__(self, params: Optional[Parameters], devices: Sequence[int]):
        self.params = params or Parameters.new()
        self.devices = [int(d) for d in devices]
        self._h = None
        L = library()
        out = C.c_void_p()
        cp = self.params._c()
        dv = (C.c_int * len(self.devices))(*self.devices)
        rc = L.fheram_group_create(C.byref(cp), dv, len(self.devices), C.byref(out))
        if rc != 0:
            raise FheRamError(rc, L.fheram_group_last_error(None).decode())
        self._h = out.value
        self._keys = None

    def __del__(self):
        if getattr(self, "_h", None) and _LIB is not None:
            _LIB.fheram_group_destroy(self._h)
            self._h = None

    def _chk(self, rc):
        if rc != 0:
            raise FheRamError(rc, library().fheram_group_last_error(self._h).decode())

    def _use_keys(self, keys: EvaluationKeysPrepared):
        if self._keys is keys:
            return
        L = library()
        c0 = L.fheram_group_ctx(self._h, 0)
        atk_len, inv_len = L.fheram_atk_len(c0), L.fheram_evk_inv_len(c0)
        if keys.atk_glwe is None or any(k.size != atk_len for k in keys.atk_glwe) or keys.atk_ggsw_inv.size != inv_len or keys.tsk_ggsw_inv.size != inv_len:
            raise FheRamError(1, "evaluation-key layout does not match the group's (parameters.rs:71-93)")
        arr = (I64P * len(keys.atk_glwe))(*[_p(k) for k in keys.atk_glwe])
        self._chk(L.fheram_group_keys_load(self._h, _p(keys.gal_els), len(keys.gal_els), arr, _p(keys.atk_ggsw_inv),
                                           keys.atk_ggsw_inv_p, _p(keys.tsk_ggsw_inv)))
        self._keys = keys

    def _out(self):
        return np.zeros((self.params.word_size(), self.params.glwe_len()), dtype=np.int64)

    def load_encrypted(self, rows: np.ndarray):
        """rows: the whole RAM, [word_size][rows][GLWE] (Ram::encrypt_sk output, ram.rs:129-167)"""
        p = self.params
        rows = _i64(rows)
        if rows.size != p.word_size() * p.rows() * p.glwe_len():
            raise FheRamError(1, f"invalid data: expected {p.word_size()}x{p.rows()} GLWE rows (ram.rs:144-155)")
        self._chk(library().fheram_group_ram_upload(self._h, _p(rows)))

    def store_encrypted(self) -> np.ndarray:
        p = self.params
        rows = np.zeros((p.word_size(), p.rows(), p.glwe_len()), dtype=np.int64)
        self._chk(library().fheram_group_ram_download(self._h, _p(rows)))
        return rows

    def tree(self, level: int = 0) -> np.ndarray:
        out = self._out()
        self._chk(library().fheram_group_ram_tree_download(self._h, level, _p(out)))
        return out

    @property
    def state(self) -> bool:
        return bool(library().fheram_group_ram_state(self._h))

    def read(self, address: Address, keys: EvaluationKeysPrepared, download: bool = True):  # ram.rs:172
        self._use_keys(keys)
        out = self._out() if download else None
        self._chk(library().fheram_group_read(self._h, address._group(self), _p(out) if download else None))
        return out

    def read_prepare_write(self, address: Address, keys: EvaluationKeysPrepared, download: bool = True):  # ram.rs:196
        self._use_keys(keys)
        out = self._out() if download else None
        self._chk(library().fheram_group_read_prepare_write(self._h, address._group(self), _p(out) if download else None))
        return out

    def write(self, w, address: Address, keys: EvaluationKeysPrepared):  # ram.rs:226
        self._use_keys(keys)
        ws = self.params.word_size()
        if w is None:
            self._chk(library().fheram_group_write(self._h, None, ws, address._group(self)))
            return
        w = _i64(w)
        n_w = w.shape[0] if w.ndim > 1 else w.size // self.params.glwe_len()
        self._chk(library().fheram_group_write(self._h, _p(w), n_w, address._group(self)))

    def stage_words(self, w):
        w = _i64(w)
        self._chk(library().fheram_group_word_stage(self._h, _p(w), w.shape[0]))

    def result(self):
        out = self._out()
        self._chk(library().fheram_group_result_download(self._h, _p(out)))
        return out

    def peer_info(self) -> List[bool]:
        """per shard: the exchange copies between its device and the root's go peer to peer (the self-test copies of the
        constructor went through either way)"""
        n = len(self.devices)
        d = (C.c_int * n)()
        self._chk(library().fheram_group_peer_info(self._h, d, n))
        return [bool(x) for x in d]

    @property
    def poisoned(self) -> bool:
        """an op failed part-way: every further op is refused until load_encrypted has replaced the rows"""
        return bool(library().fheram_group_poisoned(self._h))

    def roundoff_max(self) -> float:
        """the largest round-off over the shards' monitors (fheram_group_roundoff_max); raises PRECISION above 3/8"""
        m = C.c_double()
        self._chk(library().fheram_group_roundoff_max(self._h, C.byref(m)))
        return float(m.value)
