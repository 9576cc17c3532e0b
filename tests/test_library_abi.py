"""The C-ABI shared library loads and exports every symbol include/fheram.h declares
(no compute calls: this runs on the CPU-only build box)."""
import ctypes as C
import os
import re

import pytest

from _pkg import load_package

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "fheram.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(fheram_[a-z0-9_]+)\s*\(", src)))


def test_header_symbols_are_all_bound_and_exported():
    pkg = load_package()
    import __graft_entry__ as ge
    if not os.path.exists(pkg.library_path()):
        ge.build()
    declared = _declared()
    assert len(declared) >= 30
    bound = sorted(pkg.api.exported_symbols())
    assert bound == declared, (set(declared) ^ set(bound))
    lib = pkg.library()  # raises if a declared symbol is missing from the .so
    for name in declared:
        assert hasattr(lib, name)


def test_params_default_matches_reference_constants():  # parameters.rs:11-21, :296-323
    pkg = load_package()
    import ctypes as C
    cp = pkg.api._CParams()
    assert pkg.library().fheram_params_default(C.byref(cp)) == 0
    assert (cp.log_n, cp.base2k, cp.rank) == (12, 17, 1)
    assert (cp.k_glwe_pt, cp.k_glwe_ct, cp.k_ggsw_addr, cp.k_evk_trace, cp.k_evk_ggsw_inv) == (3, 51, 68, 68, 85)
    assert cp.word_size == 4 and cp.max_addr == 1 << 14
    assert list(cp.decomp_n[:cp.n_decomp]) == [3, 3, 3, 3]
    p = pkg.Parameters.new()
    assert p.dnum_ct() == -(-51 // 17) == 3 and p.dnum_ggsw() == 4          # parameters.rs:321-322
    assert p.basek() == 17 and p.k_glwe_ct() == 51 and p.k_glwe_pt() == 3 and p.rank() == 1
    assert p.word_size() == 4 and p.max_addr() == 1 << 14 and p.n() == 4096
    assert sum(p.decomp_n()) == 12


def test_unsupported_and_invalid_parameters_are_rejected():
    pkg = load_package()
    import ctypes as C
    L = pkg.library()
    cp = pkg.api._CParams()
    L.fheram_params_default(C.byref(cp))
    out = C.c_void_p()
    cp.log_n = 11
    assert L.fheram_ctx_create(C.byref(cp), 0, C.byref(out)) == 5  # UNSUPPORTED
    assert b"LOG_N=12" in L.fheram_last_error(None)
    L.fheram_params_default(C.byref(cp))
    cp.decomp_n[0] = 4
    assert L.fheram_ctx_create(C.byref(cp), 0, C.byref(out)) == 1  # DECOMP_N must sum to LOG_N


def test_no_gpu_means_loud_failure_not_a_cpu_fallback():
    """On a box without a GPU the product path must refuse to run."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    pkg = load_package()
    with pytest.raises(pkg.FheRamError) as e:
        pkg.Ram.new()
    assert e.value.code == 7 and "no CPU path" in e.value.msg


def test_product_package_never_imports_the_oracle():
    pkg_dir = os.path.join(ROOT, "fhe-ram_amd")
    for dirpath, _, files in os.walk(pkg_dir):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h", ".cpp")):
                txt = open(os.path.join(dirpath, f)).read()
                assert "pyoracle" not in txt and "liboracle" not in txt, f
                assert not re.search(r"#include\s*[<\"][^>\"]*oracle", txt), f
                assert not re.search(r"^\s*(from|import)\s+\S*oracle", txt, flags=re.M), f


def test_cpp_host_mirror_builds_and_links(tmp_path):
    """fhe-ram_amd/host/fheram.hpp (the C++ mirror of the reference API) compiles against the C ABI
    and, on a box without a GPU, fails loudly with the C ABI's DEVICE status."""
    import shutil
    import subprocess
    pkg = load_package()
    if shutil.which("g++") is None:
        pytest.skip("no g++")
    host = os.path.join(ROOT, "fhe-ram_amd", "host")
    exe = str(tmp_path / "host_check")
    subprocess.check_call(["g++", "-std=c++17", "-Wall", "-o", exe, os.path.join(host, "host_check.cpp"),
                           "-L" + os.path.dirname(pkg.library_path()), "-lfheram",
                           "-Wl,-rpath," + os.path.dirname(pkg.library_path())])
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr


def test_config_defaults_and_environment_overrides(monkeypatch):
    """fheram_config_default (include/fheram.h): the library defaults, then the FHERAM_* environment variables of the same names —
    how the GPU suite forces every decomposition.  Pure host code: runs without a GPU."""
    pkg = load_package()
    L = pkg.library()
    from fheram_amd.api import _CConfig, _CONFIG_FIELDS
    for k in list(os.environ):
        if k.startswith("FHERAM_") and k != "FHERAM_LIB":
            monkeypatch.delenv(k)
    cfg = _CConfig()
    L.fheram_config_default(C.byref(cfg))
    got = {f: getattr(cfg, f) for f in _CONFIG_FIELDS}
    assert got == {"limb_split": 1, "fine_split": 1, "memo": 1, "pre_inv": 1, "tail": 1, "tail_test": 0, "mid": 2, "mid_test": 0, "chain": 1,
                   "chain_y": 3, "pair_z": 1, "fuse": 1, "graph": 0, "safe": 0, "nco": 0, "tail_ep": 1, "monitor": 1, "reserved": 0}
    for name, val, field, want in (("FHERAM_SAFE", "1", "safe", 1), ("FHERAM_CHAIN_Y", "0", "chain_y", 0), ("FHERAM_TAIL", "2", "tail_test", 1),
                                   ("FHERAM_TAIL", "0", "tail", 0), ("FHERAM_MID", "2", "mid_test", 1), ("FHERAM_MID", "1", "mid", 1),
                                   ("FHERAM_NCO", "2", "nco", 2), ("FHERAM_PRE_INV", "2", "pre_inv", 2), ("FHERAM_GRAPH", "1", "graph", 1),
                                   ("FHERAM_FUSE", "0", "fuse", 0), ("FHERAM_MEMO", "0", "memo", 0), ("FHERAM_LIMB_SPLIT", "0", "limb_split", 0),
                                   ("FHERAM_MONITOR", "0", "monitor", 0), ("FHERAM_MONITOR", "2", "monitor", 2), ("FHERAM_TAIL_EP", "0", "tail_ep", 0)):
        monkeypatch.setenv(name, val)
        L.fheram_config_default(C.byref(cfg))
        assert getattr(cfg, field) == want, (name, val)
        monkeypatch.delenv(name)
