"""Digit plans of the FHE-RAM address (host-side control data).

Mirrors /root/reference/src/base.rs:3-108 (`Base1D`, `Base2D`, `get_base_2d`) and
src/lib.rs:23-26 (`reverse_bits_msb`): same names, same argument meaning.  Pure integers; the
reference's own known-answer tests (base.rs:110-439) are restated in tests/test_base.py.
"""
from typing import List, Sequence


class Base1D:
    """Bit-widths of the digits of one coordinate (base.rs:3)."""

    def __init__(self, d: Sequence[int]):
        self.d: List[int] = [int(x) for x in d]

    def size(self) -> int:  # base.rs:6
        return len(self.d)

    def max(self) -> int:  # base.rs:10-14
        m = 1
        for b in self.d:
            m <<= b
        return m

    def gap(self, log_n: int) -> int:  # base.rs:17-21
        g = log_n
        for b in self.d:
            g >>= b
        return 1 << g

    def decomp(self, value: int) -> List[int]:  # base.rs:24-33
        out, s = [], 0
        for b in self.d:
            out.append((value >> s) & ((1 << b) - 1))
            s += b
        return out

    def recomp(self, decomp: Sequence[int]) -> int:  # base.rs:36-45
        v, s = 0, 0
        for i, b in enumerate(self.d):
            v |= int(decomp[i]) << s
            s += b
        return v

    def __eq__(self, o):
        return isinstance(o, Base1D) and self.d == o.d

    def __repr__(self):
        return f"Base1D({self.d})"


class Base2D:
    """One `Base1D` per coordinate (base.rs:49)."""

    def __init__(self, v: Sequence):
        self.v: List[Base1D] = [x if isinstance(x, Base1D) else Base1D(x) for x in v]

    def max_len(self) -> int:  # base.rs:52-58
        return max((b.size() for b in self.v), default=0)

    def as_1d(self) -> Base1D:  # base.rs:64-71
        return Base1D([x for b in self.v for x in b.d])

    def max(self) -> int:  # base.rs:60
        return self.as_1d().max()

    def decomp(self, value: int) -> List[int]:
        return self.as_1d().decomp(value)

    def recomp(self, decomp: Sequence[int]) -> int:
        return self.as_1d().recomp(decomp)

    def __eq__(self, o):
        return isinstance(o, Base2D) and self.v == o.v

    def __repr__(self):
        return f"Base2D({[b.d for b in self.v]})"


def get_base_2d(value: int, base: Sequence[int]) -> Base2D:
    """Split log2(value) address bits into coordinates of at most sum(base) bits (base.rs:84-108)."""
    out = []
    bits = (int(value) - 1).bit_length()  # 32 - (value-1).leading_zeros()
    while bits != 0:
        v = []
        for b in base:
            if b <= bits:
                v.append(b)
                bits -= b
            else:
                if bits != 0:
                    v.append(bits)
                    bits = 0
                break
        out.append(Base1D(v))
    return Base2D(out)


def reverse_bits_msb(x: int, n: int) -> int:
    """lib.rs:23-26"""
    r = 0
    for _ in range(n):
        r = (r << 1) | (x & 1)
        x >>= 1
    return r
