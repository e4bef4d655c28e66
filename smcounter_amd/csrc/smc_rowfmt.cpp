// Native printer of the numeric columns of the 45-field row (DP ... PI_C, smCounter.py:575-597), byte for byte what
// CPython 2.7 prints for `'\t'.join(str(x) ...)` (:599): integers as decimal, `round(x, n)` as the correctly rounded
// decimal of the exact binary value with ties away from zero, `str(float)` as '%.12g' + a forced '.0'.
// Host-side companion of rows.py (which keeps what needs strings: CHROM..TYPE and FILTER); plain C++, no GPU.
#include "smcounter_host.h"

#include <climits>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <thread>
#include <vector>

namespace {

inline char* put_u(char* p, uint64_t v) {
    char tmp[24];
    int n = 0;
    do { tmp[n++] = (char)('0' + v % 10); v /= 10; } while (v);
    while (n) *p++ = tmp[--n];
    return p;
}

inline char* put_i(char* p, int64_t v) {
    if (v < 0) { *p++ = '-'; return put_u(p, (uint64_t)(-v)); }
    return put_u(p, (uint64_t)v);
}

// str(round(x, nd)) of CPython 2.7 for nd in {2, 4}; returns nullptr when the value is outside the range this
// printer proves itself exact for (|x| >= 1e8: the caller falls back to the Python formatter).
// |x| * 10^nd is taken as an exact two-term sum p + err (fma); p < 2^40, so p - floor(p) - 0.5 is exact and a
// multiple of ulp(p), while |err| <= ulp(p) / 2: the sign of the exact remainder is decided without rounding.
// The rounded value k / 10^nd has at most 12 significant digits, so '%.12g' of its nearest double prints exactly
// those digits (trailing zeros dropped, '.0' for integers).
inline char* put_round(char* p, double x, int nd) {
    if (x != x) { memcpy(p, "nan", 3); return p + 3; }
    const bool neg = std::signbit(x);
    const double ax = std::fabs(x);
    if (std::isinf(ax)) {
        if (neg) *p++ = '-';
        memcpy(p, "inf", 3);
        return p + 3;
    }
    if (!(ax < 1e8)) return nullptr;
    const double S = nd == 2 ? 100.0 : 10000.0;
    const uint64_t Si = nd == 2 ? 100u : 10000u;
    const double pr = ax * S;
    const double err = std::fma(ax, S, -pr);
    const double fl = std::floor(pr);
    const double d = (pr - fl) - 0.5;
    uint64_t k = (uint64_t)fl;
    if (d > 0.0 || (d == 0.0 && err >= 0.0)) ++k;      // (d == 0, err == 0: the exact tie goes away from zero)
    if (neg) *p++ = '-';
    p = put_u(p, k / Si);
    *p++ = '.';
    uint64_t fp = k % Si;
    char dig[4];
    for (int i = nd - 1; i >= 0; --i) { dig[i] = (char)('0' + fp % 10); fp /= 10; }
    int last = nd - 1;
    while (last > 0 && dig[last] == '0') --last;
    for (int i = 0; i <= last; ++i) *p++ = dig[i];
    return p;
}

// columns DP..PI_C of one row (no trailing separator); nullptr = not printable here (the caller's Python prints the row).
// *pred (optional) = int(float(<the PI column as printed>)), what the post-filter and the writers test (smCounter.py:757,838)
inline char* put_tail(char* p, const smc_row& R, int c, int32_t* pred) {
    bool ok = c >= 0 && c <= 1 && (R.status & 0xff) == 0 && !(R.status & SMC_ST_BAD_INPUT) && R.cvg > 0 && R.used_mt > 0;
    if (!ok) return nullptr;
    const smc_cand& C = R.cand[c];
    const double cvg = (double)R.cvg, used = (double)R.used_mt;
    auto I = [&](int64_t v) { p = put_i(p, v); *p++ = '\t'; };
    auto F = [&](double v, int nd) {
        if (!ok) return;
        char* q = put_round(p, v, nd);
        if (!q) { ok = false; return; }
        p = q; *p++ = '\t';
    };
    I(R.cvg); I(R.all_frag); I(R.all_mt); I(R.used_frag); I(R.used_mt);
    {
        char* const pi0 = p;
        F(C.pi, 2);
        if (ok && pred) {
            // integer part of the printed value, truncated toward zero ('nan' / 'inf' cannot be converted by int(): left to Python)
            const char* q = pi0;
            const bool neg = *q == '-';
            if (neg) ++q;
            if (*q < '0' || *q > '9') return nullptr;
            int64_t v = 0;
            while (*q >= '0' && *q <= '9') { v = v * 10 + (*q - '0'); ++q; }
            *pred = (int32_t)(neg ? -v : v);
        }
    }
    I(C.vdp); F(1.0 * C.vdp / cvg, 4); I(C.vmt); F(1.0 * C.vmt / used, 4); I(C.vsm);
    for (int a = 0; a < 4; ++a) I(R.dp[a]);
    for (int a = 0; a < 4; ++a) F(1.0 * R.dp[a] / cvg, 4);
    I(R.mt3); I(R.mt5); I(R.mt7); I(R.mt10);
    for (int a = 0; a < 4; ++a) I(R.umt[a]);
    for (int a = 0; a < 4; ++a) F(1.0 * R.umt[a] / used, 4);
    for (int a = 0; a < 4; ++a) I(R.vsm[a]);
    for (int a = 0; a < 4; ++a) F(R.pi[a], 2);
    if (!ok) return nullptr;
    return p - 1;                                      // drop the last TAB
}

}  // namespace

extern "C" {

/* Upper bound of one printed tail (39 fields) + separator. */
int smc_rowfmt_stride(void) { return 640; }

/* For each of the n rows: the columns DP..PI_C (fields 6-44 of the 45, smCounter.py:575-597) TAB-joined, using
 * candidate chosen[i] (0 or 1; the bi-allelic decision :567-573 is the caller's) for PI/VDP/VAF/VMT/VMF/VSM; rows end
 * with '\n'.  chosen[i] < 0, a row whose status is not 0, zero denominators or an out-of-range float leave an
 * EMPTY line (the caller prints those rows itself).  out must hold n * smc_rowfmt_stride() bytes.  Returns the
 * number of bytes written. */
int64_t smc_format_tails(const smc_row* rows, const int8_t* chosen, int64_t n, char* out) {
    char* p = out;
    for (int64_t i = 0; i < n; ++i) {
        char* q = put_tail(p, rows[i], chosen ? chosen[i] : 0, nullptr);
        p = q ? q : p;
        *p++ = '\n';
    }
    return (int64_t)(p - out);
}

int smc_rowfmt_line_stride(int max_chrom_len) { return 640 + 64 + (max_chrom_len > 0 ? max_chrom_len : 0); }

int64_t smc_format_lines(const smc_row* rows, const int8_t* chosen, int64_t n, const char* chroms, const int32_t* chrom_off,
                         const int32_t* chrom_id, const int64_t* pos, const uint8_t* ref, const uint8_t* alt, int max_chrom_len,
                         int nthreads, char* out, int32_t* pred) {
    const int64_t stride = smc_rowfmt_line_stride(max_chrom_len);
    int T = nthreads < 1 ? 1 : nthreads > 16 ? 16 : nthreads;
    if (n < 4096) T = 1;
    std::vector<int64_t> len((size_t)T, 0);
    auto work = [&](int t) {
        const int64_t lo = n * t / T, hi = n * (t + 1) / T;
        char* const base = out + lo * stride;        // (a chunk's lines are written densely from its own start)
        char* p = base;
        for (int64_t i = lo; i < hi; ++i) {
            char* const row0 = p;
            const bool full = alt && alt[i] && ref && ref[i];
            if (full) {
                const int32_t c = chrom_id[i];
                const int32_t cl = chrom_off[c + 1] - chrom_off[c];
                memcpy(p, chroms + chrom_off[c], (size_t)cl); p += cl; *p++ = '\t';
                p = put_i(p, pos[i]); *p++ = '\t';
                *p++ = (char)ref[i]; *p++ = '\t'; *p++ = (char)alt[i]; *p++ = '\t';
                memcpy(p, "SNP\t", 4); p += 4;
            }
            int32_t pr = INT32_MIN;
            char* q = put_tail(p, rows[i], chosen ? chosen[i] : 0, &pr);
            if (!q) { p = row0; pr = INT32_MIN; }
            else { p = q; if (full) { *p++ = '\t'; *p++ = ';'; } }
            if (pred) pred[i] = pr;
            *p++ = '\n';
        }
        len[(size_t)t] = p - base;
    };
    if (T == 1) work(0);
    else {
        std::vector<std::thread> th;
        for (int t = 1; t < T; ++t) th.emplace_back(work, t);
        work(0);
        for (auto& x : th) x.join();
    }
    int64_t total = len[0];
    for (int t = 1; t < T; ++t) {
        memmove(out + total, out + (n * t / T) * stride, (size_t)len[(size_t)t]);
        total += len[(size_t)t];
    }
    return total;
}

}  // extern "C"
