"""CPython-2.7 behaviours that leak into smCounter's output columns.

The reference is Python 2.7 code (README.md:6).  Three interpreter behaviours reach the
`.all.txt` columns and differ under Python 3:

* `round()` rounds the exact binary value half away from zero (py3: half to even),
  used for every 2-/4-decimal column (smCounter.py:576-593);
* `str(float)` prints 12 significant digits (`'%.12g'`, plus a forced `.0`);
* `sorted(finalDict.items(), ...)` (smCounter.py:534) is stable, so alleles with bit-equal PI keep
  the iteration order of a py2 `dict` with `str` keys - the slot order of the open-addressing
  table under the (unrandomised) py2 string hash.

Everything here is plain host-side Python; nothing is on the GPU path.
"""
from __future__ import annotations

import decimal

_M64 = (1 << 64) - 1


def py2_round(x: float, ndigits: int = 0) -> float:
    """CPython 2.7 `round(x, ndigits)`: correctly rounded, ties away from zero."""
    if x != x or x in (float("inf"), float("-inf")):
        return x
    q = decimal.Decimal(1).scaleb(-ndigits)
    d = decimal.Decimal(x).quantize(q, rounding=decimal.ROUND_HALF_UP)
    r = float(d)
    if r == 0.0 and (x < 0 or str(x).startswith("-")):
        return -0.0
    return r


def py2_str_float(x: float) -> str:
    """CPython 2.7 `str(float)`: '%.12g' with a '.0' appended to integer-looking output."""
    if x != x:
        return "nan"
    if x in (float("inf"), float("-inf")):
        return "inf" if x > 0 else "-inf"
    s = "%.12g" % x
    if "." not in s and "e" not in s and "n" not in s:
        s += ".0"
    return s


def py2_str(v) -> str:
    """`str(v)` as the reference's `'\\t'.join(str(x) ...)` would print it (smCounter.py:599)."""
    if isinstance(v, float):
        return py2_str_float(v)
    return str(v)


def py2_str_hash(s: str) -> int:
    """Unrandomised CPython 2.7 string hash (64-bit build), as a signed 64-bit value."""
    if not s:
        return 0
    b = s.encode("latin-1")
    x = (b[0] << 7) & _M64
    for c in b:
        x = ((1000003 * x) & _M64) ^ c
    x ^= len(b)
    if x >= 1 << 63:
        x -= 1 << 64
    if x == -1:
        x = -2
    return x


class Py2Dict(object):
    """Key-order model of a CPython 2.7 dict (insert-only), enough to reproduce `keys()` order.

    Follows Objects/dictobject.c of 2.7: 8-slot small table, probe `i = 5*i + 1 + perturb`,
    `perturb >>= 5`, resize when `fill*3 >= size*2` to the first power of two above
    `4*used` (`2*used` above 50000 entries).
    """

    def __init__(self):
        self.size = 8
        self.slots = [None] * 8
        self.used = 0

    def _place(self, slots, size, key):
        h = py2_str_hash(key)
        mask = size - 1
        i = h & mask
        perturb = h & _M64
        while slots[i] is not None:
            if slots[i] == key:
                return False
            i = (5 * i + 1 + perturb) & mask
            perturb >>= 5
        slots[i] = key
        return True

    def insert(self, key):
        if not self._place(self.slots, self.size, key):
            return
        self.used += 1
        if self.used * 3 >= self.size * 2:
            want = (2 if self.used > 50000 else 4) * self.used
            newsize = 8
            while newsize <= want:
                newsize <<= 1
            old = [k for k in self.slots if k is not None]
            self.size = newsize
            self.slots = [None] * newsize
            for k in old:
                self._place(self.slots, newsize, k)

    def keys(self):
        return [k for k in self.slots if k is not None]


def py2_dict_order(keys_in_insertion_order):
    d = Py2Dict()
    for k in keys_in_insertion_order:
        d.insert(k)
    return d.keys()
