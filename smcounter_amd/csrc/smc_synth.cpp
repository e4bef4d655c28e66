// smc_synth.cpp - fast host-side generator of the synthetic pileup workloads (SURVEY.md 8d /
// BASELINE.json configs C2-C5), multithreaded, writing the HBM structure-of-arrays layout directly.
//
// Same workload definition as smcounter_amd/synth.py::generate (which stays the readable reference
// and the source of the golden fixtures): per locus `n_umi` barcodes x `rpb` reads, fragments with
// a fraction p_overlap of mate pairs, random interleave, ids relabelled by first appearance, ref
// base "ACGT"[pos % 4], per-read error / in-deletion / insertion-start / deletion-start events, the
// fixed small distributions for quality, MAPQ, mismatches, lengths, soft clips and positions, and the
// per-read feature arithmetic of smCounter.py:352-356 / :432-452.  Each locus draws from its own
// counter-seeded stream (seed, locus index), so output does not depend on the thread count.
//
// Used by bench.py and the large-size tests to build inputs; not on the calling path.
#include <math.h>
#include <stdint.h>
#include <string.h>

#include <algorithm>
#include <thread>
#include <vector>

#include "smcounter_hip.h"

namespace {

struct Rng {   // xoshiro256** seeded by splitmix64
    uint64_t s[4];
    static uint64_t sm(uint64_t& x) {
        uint64_t z = (x += 0x9E3779B97F4A7C15ull);
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        return z ^ (z >> 31);
    }
    Rng(uint64_t seed, uint64_t stream) {
        uint64_t x = seed * 0xD1342543DE82EF95ull + stream;
        for (auto& v : s) v = sm(x);
    }
    static uint64_t rotl(uint64_t x, int k) { return (x << k) | (x >> (64 - k)); }
    uint64_t next() {
        const uint64_t r = rotl(s[1] * 5, 7) * 9, t = s[1] << 17;
        s[2] ^= s[0]; s[3] ^= s[1]; s[1] ^= s[2]; s[0] ^= s[3]; s[2] ^= t; s[3] = rotl(s[3], 45);
        return r;
    }
    double uni() { return (next() >> 11) * (1.0 / 9007199254740992.0); }
    uint32_t below(uint32_t n) { return (uint32_t)(((next() >> 32) * (uint64_t)n) >> 32); }
};

// uniform [0,1) from (key, a, b): the variant decisions of smc_synth_alignments
static inline double hash_uni(uint64_t key, uint64_t a, uint64_t b) {
    uint64_t x = key ^ (a * 0x9E3779B97F4A7C15ull) ^ (b * 0xC2B2AE3D27D4EB4Full);
    x = Rng::sm(x);
    return (x >> 11) * (1.0 / 9007199254740992.0);
}

struct Rd {
    uint32_t umi, frag;
    uint8_t allele, bq, mq, flags;   // flags: device flag byte
    uint16_t dbc, dpr;
};

}  // namespace

extern "C" {

typedef struct smc_synth_cfg {
    int64_t n_loci_total;
    int32_t n_umi, rpb;
    uint64_t seed;
    int64_t start_pos;
    double p_overlap, p_err, p_gap, p_ins, p_delstart, p_n, alt_locus_frac, alt_af;
    double mismatch_thr;
    int32_t min_bq, min_mq, primer_dist, pad_;   // the run's parameters: the read class (frag bits 27-31) depends on them
} smc_synth_cfg;

// slots needed for loci [lo,hi): every locus padded to a multiple of 4 reads
int64_t smc_synth_slots(const smc_synth_cfg* c, int64_t lo, int64_t hi) {
    const int64_t r = (int64_t)c->n_umi * c->rpb;
    return (hi - lo) * ((r + 3) / 4 * 4);
}

// extra[l]: bit0 insertion allele present, bit1 deletion-start allele present, bit2 insertion got
// the lower id (6).  Strings are rebuilt on the host from pos (smcounter_amd/synth.py).
// umi_start: (hi-lo) * (n_umi + 1) entries
int smc_synth_generate(const smc_synth_cfg* c, int64_t lo, int64_t hi, uint32_t* meta, uint32_t* umi, uint32_t* frag,
                       uint32_t* dist, uint32_t* umi_start, smc_locus* loci, uint8_t* extra, int nthreads) {
    const int U = c->n_umi, B = c->rpb, R = U * B;
    const int64_t stride = (R + 3) / 4 * 4;
    const int f0 = std::max(1, (int)llround(B / (1.0 + c->p_overlap)));
    static const uint8_t ref_by_mod[4] = {0, 3, 2, 1};   // "ACGT"[p%4] -> allele id (A0 T1 G2 C3)
    static const uint8_t trans[4] = {2, 3, 0, 1};        // A<->G, T<->C
    static const uint8_t bqv[5] = {12, 25, 30, 37, 40};
    static const double bqc[5] = {.03, .10, .30, .70, 1.0};
    auto work = [&](int64_t a, int64_t b) {
        std::vector<Rd> rd(R), out(R);
        std::vector<int> perm(R), ufirst(U), urank(U), forig(R), kp(U);
        std::vector<int> ffirst, frank;
        for (int64_t l = a; l < b; ++l) {
            Rng g(c->seed, (uint64_t)l);
            const int64_t pos = c->start_pos + l;
            const int ref = ref_by_mod[pos & 3];
            const bool alt_locus = c->alt_locus_frac > 0 && g.uni() < c->alt_locus_frac;
            int n_frag_total = 0;
            int idx = 0;
            for (int u = 0; u < U; ++u) {
                int k = 0;
                for (int t = 0; t < f0; ++t) k += g.uni() < c->p_overlap;
                k = std::min(k, B / 2);
                kp[u] = k;
                n_frag_total += B - k;
                int tru = ref;
                if (alt_locus && g.uni() < c->alt_af) tru = trans[ref];
                for (int j = 0; j < B; ++j, ++idx) {
                    Rd& r = rd[idx];
                    const bool in_pair = j < 2 * k;
                    forig[idx] = in_pair ? j / 2 : j - k;
                    const bool r2 = in_pair ? (j & 1) : (g.uni() < 0.5);
                    const bool rev = r2 ^ (g.uni() < 0.1);
                    int al = tru;
                    if (g.uni() < c->p_err) al = (al + 1 + (int)g.below(3)) & 3;
                    const bool carries_alt = al != ref;
                    const double ev = g.uni();
                    int kind = SMC_KIND_BASE;
                    bool is_n = false;
                    if (ev < c->p_gap) { kind = SMC_KIND_GAP; al = 5; }
                    else if (ev < c->p_gap + c->p_ins) { kind = SMC_KIND_INS; al = 254; }
                    else if (ev < c->p_gap + c->p_ins + c->p_delstart) { kind = SMC_KIND_DELSTART; al = 253; }
                    else if (ev >= 1.0 - c->p_n) { is_n = true; al = 4; }
                    const double qx = g.uni();
                    int qi = 0;
                    while (qx >= bqc[qi]) ++qi;
                    const int mq = g.uni() < 0.02 ? 20 : 60;
                    const double mx = g.uni();
                    int mism = mx < .8 ? 0 : (mx < .95 ? 1 : 2);
                    if (carries_alt && kind == SMC_KIND_BASE && !is_n) mism += 1;
                    if (g.uni() < 0.01) mism = 8;
                    const int qlen = 100 + (int)g.below(51);
                    const int lsp = g.uni() < 0.05 ? 1 + (int)g.below(10) : 0;
                    const int qalen = qlen - lsp;
                    const int qpos = lsp + (int)(g.uni() * qalen);
                    // features (smCounter.py:352-356, :432-452)
                    const double mm100 = 100.0 * mism / qlen;
                    const bool mm_ok = mm100 <= c->mismatch_thr;
                    const int rel = qpos - lsp, far = qalen - rel;
                    int dbc = 0, dpr = 0;
                    if (kind == SMC_KIND_BASE) {
                        dbc = r2 ? (rev ? rel : far) : (rev ? far : rel);
                        dpr = r2 ? (rev ? far : rel) : 0;
                    }
                    r.umi = (uint32_t)u;
                    r.frag = 0;
                    r.allele = (uint8_t)al;
                    r.bq = bqv[qi];
                    r.mq = (uint8_t)mq;
                    r.flags = (uint8_t)((r2 ? SMC_FL_R2 : 0) | (rev ? SMC_FL_REV : 0) | (mm_ok ? SMC_FL_MMOK : 0) |
                                        (kind << SMC_KIND_SHIFT));
                    r.dbc = (uint16_t)std::min(dbc, 65535);
                    r.dpr = (uint16_t)std::min(dpr, 65535);
                }
            }
            // random interleave (Fisher-Yates)
            for (int i = 0; i < R; ++i) perm[i] = i;
            for (int i = R - 1; i > 0; --i) std::swap(perm[i], perm[g.below((uint32_t)i + 1)]);
            // relabel by first appearance
            std::fill(ufirst.begin(), ufirst.end(), -1);
            ffirst.assign(R, -1);   // indexed u*B + original fragment
            int nu = 0;
            std::vector<int> fcount(U, 0);
            int first_ins = R + 1, first_dst = R + 1;
            for (int i = 0; i < R; ++i) {
                const int src = perm[i];
                Rd r = rd[src];
                const int u = (int)r.umi;
                if (ufirst[u] < 0) ufirst[u] = nu++;
                int& ff = ffirst[u * B + forig[src]];
                if (ff < 0) ff = fcount[u]++;
                r.umi = (uint32_t)ufirst[u];
                r.frag = (uint32_t)ff;
                if (r.allele == 254 && first_ins > R) first_ins = i;
                if (r.allele == 253 && first_dst > R) first_dst = i;
                out[i] = r;
            }
            // fragment slots are locus-level, grouped by (new) barcode id
            std::vector<int> ubase(U + 1, 0);
            for (int u = 0; u < U; ++u) ubase[ufirst[u] + 1] = fcount[u];
            for (int u = 0; u < U; ++u) ubase[u + 1] += ubase[u];
            for (int i = 0; i < R; ++i) out[i].frag += (uint32_t)ubase[out[i].umi];
            const int ins_id = first_ins < first_dst ? 6 : 7;
            const int dst_id = first_dst < first_ins ? 6 : (first_ins <= R ? 7 : 6);
            // batch layout: reads sorted barcode-major (barcode, fragment slot, pileup order)
            {
                std::vector<int> idx(R);
                for (int i = 0; i < R; ++i) idx[i] = i;
                std::stable_sort(idx.begin(), idx.end(), [&](int x, int y) { return out[x].frag < out[y].frag; });
                std::vector<Rd> tmp(R);
                for (int i = 0; i < R; ++i) tmp[i] = out[idx[i]];
                out.swap(tmp);
                uint32_t* us = umi_start + (l - lo) * (int64_t)(U + 1);
                for (int u = 0; u <= U; ++u) us[u] = 0;
                for (int i = 0; i < R; ++i) us[out[i].umi + 1]++;
                for (int u = 0; u < U; ++u) us[u + 1] += us[u];
            }
            const int64_t off = (l - lo) * stride;
            for (int i = 0; i < R; ++i) {
                Rd& r = out[i];
                int al = r.allele;
                if (al == 254) al = ins_id; else if (al == 253) al = dst_id;
                const uint32_t bq_out = ((r.flags >> SMC_KIND_SHIFT) & 3) == SMC_KIND_GAP ? (uint32_t)c->min_bq : (uint32_t)r.bq;   // :418
                meta[off + i] = (uint32_t)al | (bq_out << 8) | ((uint32_t)r.flags << 16) | ((uint32_t)r.mq << 24);
                umi[off + i] = r.umi;
                {
                    const int kind = (r.flags >> SMC_KIND_SHIFT) & 3, bq_ok = r.bq >= c->min_bq;
                    const int inc = (bq_ok || kind == SMC_KIND_GAP) && r.mq >= c->min_mq && (r.flags & SMC_FL_MMOK);
                    frag[off + i] = r.frag | smc_read_class(kind, r.flags & SMC_FL_REV, r.flags & SMC_FL_R2, inc, bq_ok, r.dbc <= 20,
                                                            r.dpr <= c->primer_dist) << SMC_FRAG_CLASS_SHIFT;
                }
                dist[off + i] = (uint32_t)r.dbc | ((uint32_t)r.dpr << 16);
            }
            for (int64_t i = R; i < stride; ++i) meta[off + i] = umi[off + i] = frag[off + i] = dist[off + i] = 0;
            smc_locus& L = loci[l - lo];
            L.read_off4 = (uint32_t)(off / 4);
            L.umi_off = (uint32_t)((l - lo) * (int64_t)(U + 1));
            L.n_reads = R;
            L.n_umi = U;
            L.n_frag = n_frag_total;
            L.ref_allele = (uint8_t)ref;
            const int n_extra = (first_ins <= R) + (first_dst <= R);
            L.n_alleles = (uint8_t)(6 + n_extra);
            L.flags = (uint16_t)(smc_param_fingerprint(c->min_bq, c->min_mq, c->mismatch_thr, c->primer_dist) << SMC_LF_FP_SHIFT);
            L.snp_mask = 0x1F;
            extra[l - lo] = (uint8_t)((first_ins <= R ? 1 : 0) | (first_dst <= R ? 2 : 0) | (first_ins < first_dst ? 4 : 0));
        }
    };
    nthreads = std::max(1, nthreads);
    std::vector<std::thread> th;
    const int64_t n = hi - lo, per = (n + nthreads - 1) / nthreads;
    for (int t = 0; t < nthreads; ++t) {
        const int64_t a = lo + t * per, b = std::min(hi, a + per);
        if (a < b) th.emplace_back(work, a, b);
    }
    for (auto& t : th) t.join();
    return 0;
}

}  // extern "C"


extern "C" {

// ---------------------------------------------------------------------------------------------------------------
// Synthetic ALIGNMENTS of stated depth (round 3): the input of the device plane builder (smc_build_planes) in the form the
// BAM decoder hands it over (smc_bam_alignments: smc_dev_aln records, CIGAR / base / quality pools, per-locus windows), for
// bench.py's `from_alignments` leg - the hot path timed from where the reference's hot loop starts (smCounter.py:316), not
// from planes that are already sorted and classified.
// Shape: molecules (barcodes) start every L / n_umi positions (L = mean read length 125), so a locus is covered by ~ n_umi
// barcodes; a molecule has `rpb` reads - a fraction p_overlap of its fragments as mate pairs (R1 forward + R2 reverse), the
// rest single - all starting within 4 positions of the molecule's start, 100-150 bases long: depth ~ n_umi x rpb.  Reference
// base at 1-based position p is "ACGT"[p % 4]; per-base sequencing error p_err; qualities / MAPQ / mismatch counts from the
// distributions of the pileup generator above; 5 % of the alignments begin with a soft clip, p_ins_aln carry a one-base
// insertion, p_del_aln a two-base deletion; alt_locus_frac / alt_af put variants under the reads (see the struct).  Alignments come out in coordinate order with run-wide barcode / fragment ids
// numbered by first appearance, exactly as the decoder numbers them.  Deterministic for a given (seed, shape), whatever
// the thread count.
typedef struct smc_synth_acfg {
    int64_t n_loci, start0;      // loci [start0, start0 + n_loci), 0-based
    int32_t n_umi, rpb;
    uint64_t seed;
    double p_overlap, p_err, p_ins_aln, p_del_aln, p_clip, mismatch_thr;
    // variants (round 5; the pileup generator's alt_locus_frac / alt_af carried over to alignments): a fraction alt_locus_frac of
    // the reference positions are variant sites; at such a site every MOLECULE (barcode) carries the transition (A<->G, C<->T)
    // with probability alt_af - all its reads then show it there (sequencing errors on top), and each adds one to the read's
    // mismatch count (the NM a mapper would report).  Both decisions are hashes of (seed, position[, molecule]): the same
    // whatever the thread count or the run's extent.
    double alt_locus_frac, alt_af;
} smc_synth_acfg;
typedef void (*smc_aln_alloc)(void* ctx, int64_t n_aln, int64_t n_cig, int64_t n_seq, int64_t n_loci, void** out);

int64_t smc_synth_alignments(const smc_synth_acfg* c, smc_aln_alloc alloc, void* alloc_ctx, int64_t* n_slots, int32_t* n_bc_out,
                             int32_t* n_pair_out, int nthreads) {
    struct A { int32_t pos, len; uint32_t mol; uint16_t frag; uint8_t flag /* 1 R1, 2 R2, 4 reverse */, kind /* 0 plain, 1 clip, 2 ins, 3 del */; uint8_t clip, mapq; uint16_t mism; uint64_t seed; };
    const double L = 125.0;
    const int64_t lead = 160;                                    // molecules start this far before the first locus
    const int64_t n_mol = (int64_t)((double)(c->n_loci + lead) * c->n_umi / L) + 1;
    const int B = c->rpb;
    const int f0 = std::max(1, (int)llround(B / (1.0 + c->p_overlap)));
    nthreads = std::max(1, nthreads);
    std::vector<std::vector<A>> per_thread((size_t)nthreads);
    {
        std::vector<std::thread> th;
        const int64_t per = (n_mol + nthreads - 1) / nthreads;
        for (int t = 0; t < nthreads; ++t)
            th.emplace_back([&, t]() {
                std::vector<A>& out = per_thread[(size_t)t];
                for (int64_t m = t * per; m < std::min(n_mol, (t + 1) * per); ++m) {
                    Rng g(c->seed ^ 0xA11C0DEull, (uint64_t)m);
                    const int64_t s = c->start0 - lead + (int64_t)((double)m * L / c->n_umi);
                    int k = 0;
                    for (int t2 = 0; t2 < f0; ++t2) k += g.uni() < c->p_overlap;
                    k = std::min(k, B / 2);
                    for (int j = 0; j < B; ++j) {
                        A a;
                        const bool in_pair = j < 2 * k;
                        a.frag = (uint16_t)(in_pair ? j / 2 : j - k);
                        const bool r2 = in_pair ? (j & 1) : (g.uni() < 0.5);
                        const bool rev = r2 ^ (g.uni() < 0.1);
                        a.flag = (uint8_t)((r2 ? 2 : 1) | (rev ? 4 : 0));
                        a.pos = (int32_t)(s + (int64_t)g.below(4));
                        a.len = 100 + (int)g.below(51);
                        const double ev = g.uni();
                        a.kind = ev < c->p_clip ? 1 : ev < c->p_clip + c->p_ins_aln ? 2 : ev < c->p_clip + c->p_ins_aln + c->p_del_aln ? 3 : 0;
                        a.clip = (uint8_t)(1 + g.below(10));
                        a.mapq = g.uni() < 0.02 ? 20 : 60;
                        const double mx = g.uni();
                        a.mism = (uint16_t)(g.uni() < 0.01 ? 8 : (mx < .8 ? 0 : (mx < .95 ? 1 : 2)));
                        a.mol = (uint32_t)m;
                        a.seed = g.next();
                        if (a.pos < 0) continue;
                        out.push_back(a);
                    }
                }
            });
        for (auto& t : th) t.join();
    }
    // coordinate order, stable in (molecule, read): a counting sort by start position
    int64_t n_aln = 0;
    for (auto& v : per_thread) n_aln += (int64_t)v.size();
    const int64_t p_lo = c->start0 - lead, span = c->n_loci + lead + 8;
    std::vector<int64_t> first((size_t)span + 1, 0);
    for (auto& v : per_thread) for (const A& a : v) if (a.pos - p_lo < span) ++first[(size_t)(a.pos - p_lo) + 1];
    for (int64_t i = 0; i < span; ++i) first[(size_t)i + 1] += first[(size_t)i];
    n_aln = first[(size_t)span];
    std::vector<A> al((size_t)n_aln);
    {
        std::vector<int64_t> cur(first.begin(), first.end() - 1);
        for (auto& v : per_thread) { for (const A& a : v) if (a.pos - p_lo < span) al[(size_t)cur[(size_t)(a.pos - p_lo)]++] = a; std::vector<A>().swap(v); }
    }
    // run-wide ids by first appearance (the decoder's numbering), CIGAR / base offsets, reference span
    std::vector<int32_t> bc_of((size_t)n_mol, -1);
    std::vector<int32_t> pair_first((size_t)n_mol * (size_t)B, -1);
    std::vector<uint32_t> bc_gid((size_t)n_aln), pair_gid((size_t)n_aln), offc((size_t)n_aln + 1, 0), offs((size_t)n_aln + 1, 0);
    std::vector<int32_t> endp((size_t)n_aln);
    int32_t n_bc = 0, n_pair = 0;
    uint32_t pool_end = 0;
    for (int64_t i = 0; i < n_aln; ++i) {
        const A& a = al[(size_t)i];
        if (bc_of[a.mol] < 0) bc_of[a.mol] = n_bc++;
        int32_t& pf = pair_first[(size_t)a.mol * (size_t)B + a.frag];
        if (pf < 0) pf = n_pair++;
        bc_gid[(size_t)i] = (uint32_t)bc_of[a.mol]; pair_gid[(size_t)i] = (uint32_t)pf;
        offc[(size_t)i + 1] = offc[(size_t)i] + (a.kind == 0 ? 1u : a.kind == 1 ? 2u : 3u);
        {   // the pool, laid out as the decoder lays it out (smc_bam_alignments): a tile's 64 positions of an alignment = one 128-byte line
            const uint32_t want = (uint32_t)(a.pos - (a.kind == 1 ? a.clip : 0) - c->start0) & 63u;
            offs[(size_t)i] = pool_end + ((want - pool_end) & 63u);
            pool_end = offs[(size_t)i] + (uint32_t)a.len;
        }
        // reference span: kind 1: len - clip; kind 2: len - 1 (one inserted base); kind 3: len + 2 (two deleted)
        endp[(size_t)i] = a.pos + (a.kind == 1 ? a.len - a.clip : a.kind == 2 ? a.len - 1 : a.kind == 3 ? a.len + 2 : a.len);
    }
    // depth per locus, slots
    std::vector<int64_t> cov((size_t)c->n_loci + 1, 0);
    const int64_t s0 = c->start0, e0 = c->start0 + c->n_loci;
    for (int64_t i = 0; i < n_aln; ++i) {
        const int64_t lo = std::max<int64_t>(al[(size_t)i].pos, s0), hi = std::min<int64_t>(endp[(size_t)i], e0);
        if (lo < hi) { ++cov[(size_t)(lo - s0)]; --cov[(size_t)(hi - s0)]; }
    }
    void* bufs[4] = {nullptr, nullptr, nullptr, nullptr};
    offs[(size_t)n_aln] = pool_end;
    alloc(alloc_ctx, n_aln, (int64_t)offc[(size_t)n_aln], (int64_t)pool_end, c->n_loci, bufs);
    smc_dev_aln* pa = (smc_dev_aln*)bufs[0]; uint32_t* pc = (uint32_t*)bufs[1];
    uint8_t* ps = (uint8_t*)bufs[2]; smc_dev_locus* pl = (smc_dev_locus*)bufs[3];   // ps: (letter, quality) byte pairs, as the decoder writes them
    if (!pa || !pc || !ps || !pl) return -9;
    int64_t slots = 0, total = 0, run = 0;
    {
        size_t w0 = 0, w1 = 0;
        for (int64_t l = 0; l < c->n_loci; ++l) {
            run += cov[(size_t)l];
            total += run;
            const int64_t p0 = s0 + l;
            while (w0 < (size_t)n_aln && endp[w0] <= p0) ++w0;
            while (w1 < (size_t)n_aln && al[w1].pos <= p0) ++w1;
            pl[l].w0 = (uint32_t)w0; pl[l].w1 = (uint32_t)std::max(w0, w1);
            pl[l].slot_off = (uint32_t)slots; pl[l].n = (uint32_t)run;
            slots += (run + 3) / 4 * 4;
        }
    }
    if (slots >= (1ll << 32)) return -10;
    // variant sites: one byte per 1-based reference position the run's reads can touch
    const bool alt_on = c->alt_locus_frac > 0 && c->alt_af > 0;
    const int64_t alt_base = p_lo - 16;                                    // (1-based position alt_base + k <-> altmap[k])
    std::vector<uint8_t> altmap;
    if (alt_on) {
        altmap.resize((size_t)(span + 16 + 512));
        for (size_t k = 0; k < altmap.size(); ++k)
            altmap[k] = hash_uni(c->seed ^ 0x5A17E5ull, (uint64_t)(alt_base + (int64_t)k), 0) < c->alt_locus_frac;
    }
    // records and pools, in parallel by alignment
    {
        static const char REF[5] = "ACGT";
        static const uint8_t bqv[5] = {12, 25, 30, 37, 40};
        // quality by one byte of a random word: cumulative thresholds of {.03, .07, .2, .4, .3} over 256
        uint8_t qlut[256];
        for (int v = 0; v < 256; ++v) { const double x = (v + 0.5) / 256.0; qlut[v] = bqv[x < .03 ? 0 : x < .10 ? 1 : x < .30 ? 2 : x < .70 ? 3 : 4]; }
        std::vector<std::thread> th;
        const int64_t per = (n_aln + nthreads - 1) / nthreads;
        for (int t = 0; t < nthreads; ++t)
            th.emplace_back([&, t]() {
                for (int64_t i = t * per; i < std::min(n_aln, (t + 1) * per); ++i) {
                    const A& a = al[(size_t)i];
                    Rng g(a.seed, (uint64_t)i);
                    smc_dev_aln& d = pa[i];
                    d.pos = a.pos; d.end = endp[(size_t)i];
                    d.cig_off = offc[(size_t)i]; d.seq_off = offs[(size_t)i];
                    d.n_cig = (uint16_t)(offc[(size_t)i + 1] - offc[(size_t)i]);
                    d.mapq = a.mapq;
                    d.left_sp = (uint16_t)(a.kind == 1 ? a.clip : 0);
                    d.qalen = (uint16_t)(a.len - (a.kind == 1 ? a.clip : 0));
                    d.l_seq = (uint16_t)a.len; d.pad = 0;
                    d.bc_gid = bc_gid[(size_t)i]; d.pair_gid = pair_gid[(size_t)i];
                    uint32_t* cg = pc + offc[(size_t)i];
                    const int half = a.len / 2;
                    if (a.kind == 0) cg[0] = (uint32_t)a.len << 4;
                    else if (a.kind == 1) { cg[0] = (uint32_t)a.clip << 4 | 4u; cg[1] = (uint32_t)(a.len - a.clip) << 4; }
                    else if (a.kind == 2) { cg[0] = (uint32_t)half << 4; cg[1] = 1u << 4 | 1u; cg[2] = (uint32_t)(a.len - half - 1) << 4; }
                    else { cg[0] = (uint32_t)half << 4; cg[1] = 2u << 4 | 2u; cg[2] = (uint32_t)(a.len - half) << 4; }
                    // bases: the reference under every query position (clips and the inserted base: a random letter), errors rare
                    uint8_t* sq = ps + 2 * (size_t)offs[(size_t)i];
                    uint8_t* ql = sq + 1;
                    for (size_t g = i ? (size_t)offs[(size_t)i - 1] + (size_t)al[(size_t)i - 1].len : 0; g < (size_t)offs[(size_t)i]; ++g) { ps[2 * g] = 'A'; ps[2 * g + 1] = 0; }   // (the gap before it)
                    int64_t rp = a.pos + 1;                              // 1-based reference position of the next match
                    int n_alt = 0;                                       // variant sites at which this read's molecule carries the alt
                    for (int q = 0; q < a.len; ++q) {
                        bool off_ref = (a.kind == 1 && q < a.clip) || (a.kind == 2 && q == half);
                        if (a.kind == 3 && q == half) rp += 2;           // the two deleted bases
                        uint8_t b = (uint8_t)(off_ref ? REF[g.below(4)] : REF[rp & 3]);
                        if (alt_on && !off_ref) {
                            const int64_t k = rp - alt_base;
                            if (k >= 0 && k < (int64_t)altmap.size() && altmap[(size_t)k] &&
                                hash_uni(c->seed ^ 0xA17A11E1Eull, (uint64_t)rp, (uint64_t)a.mol + 1) < c->alt_af) {
                                b = (uint8_t)REF[(rp & 3) ^ 2];          // the transition: A<->G, C<->T
                                ++n_alt;
                            }
                        }
                        sq[2 * q] = b;
                        if (!off_ref) ++rp;
                    }
                    const double mm100 = 100.0 * (double)(a.mism + n_alt) / (double)a.len;          // smCounter.py:352-356
                    d.oflag = (uint8_t)((a.flag & 7) | (mm100 <= c->mismatch_thr ? SMC_DA_MMOK : 0));
                    // sequencing errors by geometric skipping
                    if (c->p_err > 0) {
                        const double lg = log1p(-c->p_err);
                        for (int q = (int)(log(1.0 - g.uni()) / lg); q < a.len; q += 1 + (int)(log(1.0 - g.uni()) / lg))
                            sq[2 * q] = (uint8_t)REF[((sq[2 * q] == 'A' ? 0 : sq[2 * q] == 'C' ? 1 : sq[2 * q] == 'G' ? 2 : 3) + 1 + (int)g.below(3)) & 3];
                    }
                    for (int q = 0; q < a.len; q += 8) {
                        uint64_t w = g.next();
                        for (int b = 0; b < 8 && q + b < a.len; ++b, w >>= 8) ql[2 * (q + b)] = qlut[w & 255];
                    }
                }
            });
        for (auto& t : th) t.join();
    }
    *n_slots = slots; *n_bc_out = n_bc; *n_pair_out = n_pair;
    return total;
}

}  // extern "C"
