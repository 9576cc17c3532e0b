"""Known-answer tests of the digit plan, restated from the reference's own unit tests
(/root/reference/src/base.rs:110-439) and run against BOTH the oracle (C++) and the product's
host-side mirror (fhe-ram_amd/base.py)."""
import pytest

from _pkg import load_package


def impls(po):
    pkg = load_package()
    return [
        ("oracle", po.Base1D, po.Base2D, po.get_base_2d),
        ("host", pkg.Base1D, pkg.Base2D, lambda v, b: [x.d for x in pkg.get_base_2d(v, b).v]),
    ]


@pytest.fixture(params=[0, 1], ids=["oracle", "host"])
def impl(request, po):
    return impls(po)[request.param]


def test_base1d_max_calculation(impl):  # base.rs:115-128
    _, B1, _, _ = impl
    assert B1([4, 4, 4]).max() == 1 << 12
    assert B1([8, 8]).max() == 1 << 16
    assert B1([12]).max() == 1 << 12
    assert B1([1, 1, 1, 1]).max() == 1 << 4


def test_base1d_decomp_recomp_roundtrip(impl):  # base.rs:131-158
    _, B1, _, _ = impl
    base = B1([4, 4, 4])
    for value in [0, 1, 15, 255, 1000, 4095]:
        d = base.decomp(value)
        assert base.recomp(d) == value
        assert len(d) == 3 and all(e < 16 for e in d)


def test_base1d_decomp_correctness(impl):  # base.rs:161-182
    _, B1, _, _ = impl
    base = B1([4, 4, 4])
    assert base.decomp(0b0000_0000_1111) == [15, 0, 0]
    assert base.recomp([15, 0, 0]) == 15
    assert base.decomp(0b1010_1100_1111) == [15, 12, 10]
    assert base.recomp([15, 12, 10]) == 0b1010_1100_1111


def test_base1d_gap_calculation(impl):  # base.rs:185-200
    _, B1, _, _ = impl
    assert B1([4, 4, 4]).gap(12) == 1
    assert B1([6, 6]).gap(12) == 1
    assert B1([3, 3, 3, 3]).gap(12) == 1


def test_base2d_creation_and_conversion(impl):  # base.rs:203-212
    _, _, B2, _ = impl
    assert B2([[4, 4], [4, 4]]).as_1d().d == [4, 4, 4, 4]


def test_base2d_max_calculation(impl):  # base.rs:215-230
    _, _, B2, _ = impl
    assert B2([[4, 4], [4, 4]]).max() == 1 << 16
    assert B2([[6], [6]]).max() == 1 << 12


def test_base2d_decomp_recomp_roundtrip(impl):  # base.rs:233-262
    _, _, B2, _ = impl
    b = B2([[4, 4], [4, 4]])
    for value in [0, 1, 255, 1000, 65535]:
        d = b.decomp(value)
        assert b.recomp(d) == value
        assert len(d) == 4 and all(e < 16 for e in d)


def test_get_base_2d_functionality(impl):  # base.rs:265-301
    _, _, B2, g = impl
    r = g(1000, [4, 4, 4])
    assert len(r) == 1 and len(r[0]) > 0
    r2 = g(1000, [4, 4])
    assert len(r2) >= 1
    b = B2(r2)
    d = b.decomp(1000)
    assert b.recomp(d) == 1000
    assert len(d) >= 1 and all(e < 16 for e in d)


def test_base1d_edge_cases(impl):  # base.rs:304-318
    _, B1, _, _ = impl
    e = B1([])
    assert e.max() == 1 and e.decomp(0) == [] and e.recomp([]) == 0
    s = B1([1])
    assert s.max() == 2 and s.decomp(0) == [0] and s.decomp(1) == [1]
    assert s.recomp([0]) == 0 and s.recomp([1]) == 1


def test_base1d_comprehensive_roundtrip(impl):  # base.rs:321-336
    _, B1, _, _ = impl
    base = B1([4, 4, 4])
    assert base.max() == 1 << 12
    for v in [0, 1, 2, 3, 4, 5, 10, 15, 16, 17, 31, 32, 63, 64, 127, 128, 255, 256, 511, 512, 1023, 1024, 2047, 2048, 4095]:
        assert base.recomp(base.decomp(v)) == v


def test_base2d_comprehensive_roundtrip(impl):  # base.rs:339-355
    _, _, B2, _ = impl
    b = B2([[6, 6], [4, 4]])
    for v in [0, 1, 15, 16, 31, 32, 63, 64, 127, 128, 255, 256, 511, 512, 1023, 1024, 2047, 2048, 4095, 4096, 8191,
              8192, 16383, 16384, 32767, 32768, 65535]:
        assert b.recomp(b.decomp(v)) == v


def test_base1d_different_sizes(impl):  # base.rs:358-382
    _, B1, _, _ = impl
    for d in ([1, 1, 1, 1], [2, 2, 2], [3, 3, 3], [4, 4, 4], [8, 8]):
        base = B1(d)
        mx = base.max() - 1
        for v in [0, 1, mx // 4, mx // 2, mx]:
            assert base.recomp(base.decomp(v)) == v


def test_base2d_edge_cases(impl):  # base.rs:385-400
    _, _, B2, _ = impl
    e = B2([])
    assert e.max() == 1 and e.decomp(0) == [] and e.recomp([]) == 0
    s = B2([[4, 4]])
    assert s.max() == 1 << 8 and s.as_1d().d == [4, 4]


def test_base_decomposition_boundary_tests(impl):  # base.rs:403-438
    _, B1, _, _ = impl
    base = B1([4, 4, 4])
    for v in [0, 1, (1 << 4) - 1, 1 << 4, (1 << 8) - 1, 1 << 8, (1 << 12) - 1]:
        d = base.decomp(v)
        assert base.recomp(d) == v and len(d) == 3 and all(e < 16 for e in d)


def test_survey_table_plans(impl):  # SURVEY.md §8 table, derived from base.rs:84-108
    _, _, _, g = impl
    assert g(1 << 12, [3, 3, 3, 3]) == [[3, 3, 3, 3]]
    assert g(1 << 14, [3, 3, 3, 3]) == [[3, 3, 3, 3], [2]]
    assert g(1 << 18, [3, 3, 3, 3]) == [[3, 3, 3, 3], [3, 3]]
    assert g(1 << 21, [3, 3, 3, 3]) == [[3, 3, 3, 3], [3, 3, 3]]


def test_reverse_bits_msb(po):  # lib.rs:23-26
    pkg = load_package()
    for f in (lambda x, n: int(po.lib().fo_reverse_bits_msb(x, n)), pkg.reverse_bits_msb):
        assert f(1, 12) == 2048 and f(2, 12) == 1024 and f(3, 12) == 3072 and f(0, 12) == 0
        assert sorted(f(i, 6) for i in range(64)) == list(range(64))
