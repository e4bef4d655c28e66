/*
 * py2_pin.c - an INDEPENDENT restatement of the three CPython 2.7 behaviours smCounter's down-sampling step leans on
 * (smCounter.py:496-498: `random.seed(pos); bcKeys = random.sample(bcDict.keys(), ds)`), written from the published
 * interpreter sources (CPython 2.7: Objects/stringobject.c string_hash, Objects/dictobject.c lookdict_string /
 * insertdict / dictresize, Modules/_randommodule.c random_seed / init_by_array / genrand_res53, Lib/random.py
 * Random.sample) and NOT from smcounter_amd/py2compat.py.
 *
 * TEST INFRASTRUCTURE ONLY (like everything under oracle/): the product's emulation (py2compat.Py2Dict / Py2Random /
 * py2_downsample_barcodes) is checked against this second implementation on random barcode sets, so that a bug in the
 * dict-resize or sample() emulation - which both the product and the harnessed reference would share - does not go
 * unseen.  No Python 2 exists in the build container, so neither is pinned against the real
 * interpreter beyond the known values tests/test_host_logic.py holds.
 *
 * Build: gcc -O2 -shared -fPIC -o libpy2_pin.so py2_pin.c  (oracle/Makefile)
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* ---- str.__hash__ (unrandomised, 64-bit long): x = ord(s[0]) << 7; x = (1000003 * x) ^ ord(c) for every c; x ^= len;
 * -1 is reserved and becomes -2; the empty string hashes to 0. */
int64_t py2pin_str_hash(const char* s, int64_t len) {
    if (len == 0) return 0;
    const unsigned char* p = (const unsigned char*)s;
    uint64_t x = (uint64_t)p[0] << 7;
    for (int64_t i = 0; i < len; ++i) x = (1000003ull * x) ^ (uint64_t)p[i];
    x ^= (uint64_t)len;
    if (x == (uint64_t)-1) x = (uint64_t)-2;
    return (int64_t)x;
}

/* ---- dict with str keys: open addressing, table of 8 slots at first; probe i = (5 i + perturb + 1) & mask with perturb
 * starting at the hash (as an unsigned value) and shifted right by 5 after every probe; after an insertion that brings
 * the fill to >= 2/3 of the table it is rebuilt with the smallest power of two > 4 * used (2 * used above 50000 keys),
 * re-inserting the old table's entries in slot order.  keys() walks the slots in index order.  (No deletions: bcDict
 * never loses a key, smCounter.py:467-479 deletes inside bcDict[BC].) */
typedef struct { int64_t hash; int32_t key; } slot_t;           /* key: index into the caller's array, -1 = empty */

static void place(slot_t* tab, uint64_t mask, int64_t hash, int32_t key) {
    uint64_t i = (uint64_t)hash & mask;
    uint64_t perturb = (uint64_t)hash;
    while (tab[i & mask].key >= 0) {
        i = (i << 2) + i + perturb + 1;
        perturb >>= 5;
    }
    tab[i & mask].hash = hash;
    tab[i & mask].key = key;
}

/* out[0 .. n): the indices of the n DISTINCT keys in the order dict.keys() returns them after inserting keys[0], keys[1],
 * ... in that order.  Returns 0, or -1 on allocation failure. */
int py2pin_dict_order(const char* const* keys, const int64_t* lens, int32_t n, int32_t* out) {
    uint64_t size = 8;
    slot_t* tab = (slot_t*)malloc(size * sizeof(slot_t));
    if (!tab) return -1;
    for (uint64_t i = 0; i < size; ++i) tab[i].key = -1;
    int64_t used = 0;
    for (int32_t k = 0; k < n; ++k) {
        place(tab, size - 1, py2pin_str_hash(keys[k], lens[k]), k);
        ++used;
        if (used * 3 >= (int64_t)size * 2) {
            const int64_t minused = (used > 50000 ? 2 : 4) * used;
            uint64_t nsize = 8;
            while ((int64_t)nsize <= minused) nsize <<= 1;
            slot_t* nt = (slot_t*)malloc(nsize * sizeof(slot_t));
            if (!nt) { free(tab); return -1; }
            for (uint64_t i = 0; i < nsize; ++i) nt[i].key = -1;
            for (uint64_t i = 0; i < size; ++i)
                if (tab[i].key >= 0) place(nt, nsize - 1, tab[i].hash, tab[i].key);
            free(tab);
            tab = nt; size = nsize;
        }
    }
    int32_t m = 0;
    for (uint64_t i = 0; i < size; ++i)
        if (tab[i].key >= 0) out[m++] = tab[i].key;
    free(tab);
    return m == n ? 0 : -2;
}

/* ---- MT19937 as _randommodule.c drives it */
typedef struct { uint32_t mt[624]; int idx; } mt_t;

static void init_genrand(mt_t* g, uint32_t s) {
    g->mt[0] = s;
    for (int i = 1; i < 624; ++i) g->mt[i] = 1812433253u * (g->mt[i - 1] ^ (g->mt[i - 1] >> 30)) + (uint32_t)i;
    g->idx = 624;
}
static void init_by_array(mt_t* g, const uint32_t* key, int klen) {
    init_genrand(g, 19650218u);
    int i = 1, j = 0;
    for (int k = 624 > klen ? 624 : klen; k; --k) {
        g->mt[i] = (g->mt[i] ^ ((g->mt[i - 1] ^ (g->mt[i - 1] >> 30)) * 1664525u)) + key[j] + (uint32_t)j;
        if (++i >= 624) { g->mt[0] = g->mt[623]; i = 1; }
        if (++j >= klen) j = 0;
    }
    for (int k = 623; k; --k) {
        g->mt[i] = (g->mt[i] ^ ((g->mt[i - 1] ^ (g->mt[i - 1] >> 30)) * 1566083941u)) - (uint32_t)i;
        if (++i >= 624) { g->mt[0] = g->mt[623]; i = 1; }
    }
    g->mt[0] = 0x80000000u;
}
static uint32_t genrand_int32(mt_t* g) {
    if (g->idx >= 624) {
        for (int k = 0; k < 624; ++k) {
            const uint32_t y = (g->mt[k] & 0x80000000u) | (g->mt[(k + 1) % 624] & 0x7fffffffu);
            g->mt[k] = g->mt[(k + 397) % 624] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
        }
        g->idx = 0;
    }
    uint32_t y = g->mt[g->idx++];
    y ^= y >> 11;
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= y >> 18;
    return y;
}
static double genrand_res53(mt_t* g) {
    const uint32_t a = genrand_int32(g) >> 5, b = genrand_int32(g) >> 6;
    return (a * 67108864.0 + b) * (1.0 / 9007199254740992.0);
}
/* random.seed(a_str): a non-integer argument is hashed, the hash taken as an UNSIGNED long and cut into 32-bit words,
 * least significant first (a zero hash gives the single word 0) */
static void seed_with_str(mt_t* g, const char* s, int64_t len) {
    const uint64_t h = (uint64_t)py2pin_str_hash(s, len);
    uint32_t key[2] = {(uint32_t)h, (uint32_t)(h >> 32)};
    init_by_array(g, key, key[1] ? 2 : 1);
}

/* first `count` values of random.random() after random.seed(<str>) (for the pins) */
void py2pin_random_after_seed(const char* s, int64_t len, int32_t count, double* out) {
    mt_t g;
    seed_with_str(&g, s, len);
    for (int32_t i = 0; i < count; ++i) out[i] = genrand_res53(&g);
}

/* ---- random.sample(population_list, k): n <= setsize -> a pool with swap-removal, drawing int(random() * (n - i));
 * else index rejection with int(random() * n).  setsize = 21, + 4 ** ceil(log(3 k, 4)) when k > 5 (floats, as written
 * in Lib/random.py).  out[0 .. k): positions in the population, in the order sample() returns them. */
static int sample_positions(mt_t* g, int32_t n, int32_t k, int32_t* out) {
    double setsize = 21.0;
    if (k > 5) setsize += pow(4.0, ceil(log((double)k * 3.0) / log(4.0)));
    if ((double)n <= setsize) {
        int32_t* pool = (int32_t*)malloc((size_t)(n > 0 ? n : 1) * sizeof(int32_t));
        if (!pool) return -1;
        for (int32_t i = 0; i < n; ++i) pool[i] = i;
        for (int32_t i = 0; i < k; ++i) {
            const int32_t j = (int32_t)(genrand_res53(g) * (double)(n - i));
            out[i] = pool[j];
            pool[j] = pool[n - i - 1];
        }
        free(pool);
    } else {
        unsigned char* sel = (unsigned char*)calloc((size_t)n, 1);
        if (!sel) return -1;
        for (int32_t i = 0; i < k; ++i) {
            int32_t j = (int32_t)(genrand_res53(g) * (double)n);
            while (sel[j]) j = (int32_t)(genrand_res53(g) * (double)n);
            sel[j] = 1;
            out[i] = j;
        }
        free(sel);
    }
    return 0;
}

/* smCounter.py:496-498 for one locus: barcodes = the keys of bcDict in INSERTION order (n distinct strings), pos = the
 * position string that seeds the generator, ds = sample size (< n).  out[0 .. ds): indices into `barcodes` of the kept
 * keys, in the order random.sample returns them.  0 on success. */
int py2pin_downsample(const char* pos, int64_t pos_len, const char* const* barcodes, const int64_t* lens, int32_t n,
                      int32_t ds, int32_t* out) {
    if (ds < 0 || ds > n) return -3;
    int32_t* order = (int32_t*)malloc((size_t)(n > 0 ? n : 1) * sizeof(int32_t));
    int32_t* posn = (int32_t*)malloc((size_t)(ds > 0 ? ds : 1) * sizeof(int32_t));
    if (!order || !posn) { free(order); free(posn); return -1; }
    int rc = py2pin_dict_order(barcodes, lens, n, order);            /* bcDict.keys() */
    if (rc == 0) {
        mt_t g;
        seed_with_str(&g, pos, pos_len);                              /* random.seed(pos) */
        rc = sample_positions(&g, n, ds, posn);                      /* random.sample(keys, ds) */
        if (rc == 0)
            for (int32_t i = 0; i < ds; ++i) out[i] = order[posn[i]];
    }
    free(order); free(posn);
    return rc;
}
