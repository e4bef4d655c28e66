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


def _py2_round_exact(x: float, ndigits: int) -> float:
    q = decimal.Decimal(1).scaleb(-ndigits)
    d = decimal.Decimal(x).quantize(q, rounding=decimal.ROUND_HALF_UP)
    r = float(d)
    if r == 0.0 and (x < 0 or str(x).startswith("-")):
        return -0.0
    return r


def py2_round(x: float, ndigits: int = 0) -> float:
    """CPython 2.7 `round(x, ndigits)`: correctly rounded, ties away from zero.

    This interpreter's `round` is correctly rounded too and differs only on exact ties (half to even).  A double
    is a dyadic rational, so x * 10^n can be exactly half-way between integers only when x * 2^(n+1) is an odd
    integer (x = odd / 2^(n+1)); that scaling is exact in floating point, so the test is: everything else takes
    the built-in."""
    if x != x or x in (float("inf"), float("-inf")):
        return x
    if 0 <= ndigits <= 20:
        t = x * (1 << (ndigits + 1))
        if t != int(t) or not (int(t) & 1):
            return round(x, ndigits)
    return _py2_round_exact(x, ndigits)


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


# ------------------------------------------------------------------------------------------------
# random.seed(str) + random.sample(list, k) of CPython 2.7 (smCounter.py:496-498)
# ------------------------------------------------------------------------------------------------
class Py2Random(object):
    """`random.seed(a)`, `random.random()`, `random.sample(list, k)` as CPython 2.7 computes them.

    * Modules/_randommodule.c (2.7) `random_seed`: an int/long seeds with its absolute value; any other
      object with `(unsigned long) hash(obj)` - for a `str` the unrandomised 64-bit string hash - split into
      32-bit words, low word first, fed to MT19937 `init_by_array`.  CPython 3 seeds an `int` the same way,
      so `random.Random(n)` of this interpreter reproduces the state; `random()` (genrand_res53) is unchanged.
    * Lib/random.py (2.7) `sample`: `setsize = 21 (+ 4 ** ceil(log(3k, 4)) if k > 5)`; a list of
      `n <= setsize` items is sampled by pool swaps `j = int(random() * (n - i))`, a longer one by rejection
      `j = int(random() * n)` against a set of chosen indices (py3 draws with `_randbelow` instead).
    """

    def __init__(self, seed):
        import random as _random
        if isinstance(seed, int):
            n = abs(seed)
        elif isinstance(seed, str):
            n = py2_str_hash(seed) & _M64
        else:
            raise TypeError("Py2Random: int or str seed")
        self._r = _random.Random(n)

    def random(self) -> float:
        return self._r.random()

    def sample(self, population, k: int):
        import math
        population = list(population)
        n = len(population)
        if not 0 <= k <= n:
            raise ValueError("sample larger than population")
        result = [None] * k
        setsize = 21
        if k > 5:
            setsize += 4 ** math.ceil(math.log(k * 3, 4))
        if n <= setsize:
            pool = list(population)
            for i in range(k):
                j = int(self.random() * (n - i))
                result[i] = pool[j]
                pool[j] = pool[n - i - 1]
        else:
            selected = set()
            for i in range(k):
                j = int(self.random() * n)
                while j in selected:
                    j = int(self.random() * n)
                selected.add(j)
                result[i] = population[j]
        return result


def py2_downsample_barcodes(pos: str, barcodes_in_insertion_order, ds: int):
    """`random.seed(pos); bcKeys = random.sample(bcDict.keys(), ds)` (smCounter.py:496-498): `pos` is the
    locus position AS A STRING (vc() receives it as text, :274/:680), the population is the py2 dict key order
    of the barcodes, inserted in order of their first included read (:467-468).  -> kept barcodes, in the
    order the reference would iterate them."""
    keys = py2_dict_order(barcodes_in_insertion_order)
    if len(keys) <= ds:
        return keys
    return Py2Random(str(pos)).sample(keys, ds)
