// smcounter_hip.hip - MI355X (gfx950) kernels and C ABI for smCounter's per-locus hot path.
//
// One workgroup per locus.  The whole of vc() up to (not including) string formatting runs on
// the device (reference: /root/reference/smCounter.py):
//   scan     per-read inclusion test + per-allele tallies            smCounter.py:368-460
//   group    barcode -> fragment table in LDS, mate merge             :462-479
//   score    calProb per barcode, PI / consensus accumulation         :26-98, :506-532
//   rank     top-2 alleles, candidate(s), bi-allelic pre-condition    :534-555
//   filter   filterVariants minus the two FASTA-dependent flags       :182-269  (second kernel)
//
// Data layout (see include/smcounter_hip.h, smcounter_amd/features.py): four uint32 planes, 16 B
// per pileup read, reads of a locus contiguous and 16-byte aligned; barcode ids dense per locus,
// fragment ids dense per barcode, so the on-chip tables are directly indexed (no hashing).
//
// On-chip tables per locus (dynamic LDS, or a global scratch slab for loci that do not fit):
//   umi_base[nU+1]  first fragment slot of each barcode (slot of its first read, from umi_start)
//   frag word[nF]   one 32-bit word per fragment: (allele, quality) of its first and of its second
//                   included read, in pileup order - which is memory order, a fragment's reads being
//                   adjacent in the barcode-major batch: a read is "second" when the read just before it
//                   has the same slot and is included (smCounter.py:468-479 depends on that order).
//                   Written with one LDS atomicOr per included read; slots with >= 3 reads are flagged
//                   and replayed sequentially.  The merge then overwrites it with the fragment's state.
//   worklist[nU], umi_flag[nU], chunk masks (2 x u64 per 64 slots): see the U phase.
// This is integer/branchy, HBM-streaming work: no MFMA.
#include <hip/hip_runtime.h>

#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <string>
#include <vector>

#include "smcounter_hip.h"

#define WAVE 64
#define N_ID 4
#define GAP_ID 5

// ------------------------------------------------------------------------------------------
// device side
// ------------------------------------------------------------------------------------------
struct KParams {
    int min_bq, min_mq, mt_drop, primer_dist, ds;
    double smt;
};

// header of the per-block LDS image; the tables follow it
struct Hdr {
    uint32_t misc[32];
    uint32_t scan_tmp[32];
};
enum {
    M_NINC = 0,   // included reads
    M_SPARE0,
    M_ERR,
    M_NBC,        // barcodes with an included read
    M_ALLMT,
    M_TOTFRAG,
    M_MT3, M_MT5, M_MT7, M_MT10,
    M_USEDFRAG,
    M_TOUCH_LO, M_TOUCH_HI,
    M_CVG,
    M_NEEDFIX,
    M_NCOMPLEX,   // barcodes deferred to the general calProb path
    M_NKEPT,      // bcDict keys the host's down-sampling kept (SMC_LF_SAMPLED loci)
};

#ifdef SMC_STAMPS
// diagnostic build only (scripts/stamps.py): per-phase cycle totals of k_call_loci, thread 0 of each
// block; deltas are kept in registers and flushed with one burst of atomics at the very end.
__device__ unsigned long long g_stamps[16];
#define STAMP_INIT() unsigned long long t_prev_ = clock64(), t_d_[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}
#define STAMP(k)                                                             \
    do {                                                                     \
        const unsigned long long t_ = clock64();                             \
        t_d_[k] += t_ - t_prev_;                                             \
        t_prev_ = t_;                                                        \
    } while (0)
#define STAMP_FLUSH()                                                        \
    do {                                                                     \
        if (threadIdx.x == 0)                                                \
            for (int k_ = 0; k_ < 12; ++k_) atomicAdd(&g_stamps[k_], t_d_[k_]); \
    } while (0)
#else
#define STAMP(k) do { } while (0)
#define STAMP_INIT() do { } while (0)
#define STAMP_FLUSH() do { } while (0)
#endif

#define NT_K1 11  // tallies kept per allele in LDS: the SMC_T_* of the header, without the pad

__device__ __forceinline__ uint32_t lds_hdr_bytes(int a_cap) {
    // Hdr + tal[a_cap][SMC_NT] + pifx[a_cap] (u64) + mtc[a_cap] + strong[a_cap] + lut[LUT_N] doubles
    // (the row is staged over the LUT, which is dead by then)
    return (uint32_t)(sizeof(Hdr) + a_cap * SMC_NT * 4 + a_cap * 8 + a_cap * 4 + a_cap * 4 + 128 * 8 + 32 * 8);
}

__device__ __forceinline__ double wave_reduce_mul(double v, int width) {
    for (int m = 1; m < width; m <<= 1) v *= __shfl_xor(v, m);
    return v;
}
__device__ __forceinline__ int wave_reduce_add(int v, int width) {
    for (int m = 1; m < width; m <<= 1) v += __shfl_xor(v, m);
    return v;
}
__device__ __forceinline__ uint32_t wave_reduce_or(uint32_t v, int width) {
    for (int m = 1; m < width; m <<= 1) v |= __shfl_xor((int)v, m);
    return v;
}
__device__ __forceinline__ long long wave_reduce_add64(long long v, int width) {
    for (int m = 1; m < width; m <<= 1) v += __shfl_xor(v, m);
    return v;
}

// ---- reductions through DPP (one VALU instruction per step, no LDS round trip, unlike __shfl = ds_bpermute).
// Callers keep whole 8-lane groups (grp8_*) or the whole wavefront (wave_*) active.
#define DPP_XOR1 0xB1          // quad_perm:[1,0,3,2]
#define DPP_XOR2 0x4E          // quad_perm:[2,3,0,1]
#define DPP_HALF_MIRROR 0x141  // lane i <- lane 7-i of its 8
#define DPP_MIRROR 0x140       // lane i <- lane 15-i of its row
template <int CTRL>
__device__ __forceinline__ int dpp_i32(int v) { return __builtin_amdgcn_update_dpp(v, v, CTRL, 0xF, 0xF, false); }
template <int CTRL>
__device__ __forceinline__ double dpp_f64(double v) {
    return __hiloint2double(dpp_i32<CTRL>(__double2hiint(v)), dpp_i32<CTRL>(__double2loint(v)));
}
template <int CTRL>
__device__ __forceinline__ long long dpp_i64(long long v) {
    return (long long)(((unsigned long long)(uint32_t)dpp_i32<CTRL>((int)(v >> 32)) << 32) | (uint32_t)dpp_i32<CTRL>((int)v));
}
__device__ __forceinline__ int grp8_add(int v) { v += dpp_i32<DPP_XOR1>(v); v += dpp_i32<DPP_XOR2>(v); v += dpp_i32<DPP_HALF_MIRROR>(v); return v; }
__device__ __forceinline__ uint32_t grp8_or(uint32_t v) {
    v |= (uint32_t)dpp_i32<DPP_XOR1>((int)v); v |= (uint32_t)dpp_i32<DPP_XOR2>((int)v); v |= (uint32_t)dpp_i32<DPP_HALF_MIRROR>((int)v);
    return v;
}
__device__ __forceinline__ double grp8_mul(double v) { v *= dpp_f64<DPP_XOR1>(v); v *= dpp_f64<DPP_XOR2>(v); v *= dpp_f64<DPP_HALF_MIRROR>(v); return v; }
// whole-wavefront totals, returned wave-uniform (row totals by DPP, the four rows added on the scalar unit)
__device__ __forceinline__ int wave_add(int v) {
    v = grp8_add(v); v += dpp_i32<DPP_MIRROR>(v);
    return __builtin_amdgcn_readlane(v, 0) + __builtin_amdgcn_readlane(v, 16) + __builtin_amdgcn_readlane(v, 32) + __builtin_amdgcn_readlane(v, 48);
}
__device__ __forceinline__ uint32_t wave_or(uint32_t v) {
    v = grp8_or(v); v |= (uint32_t)dpp_i32<DPP_MIRROR>((int)v);
    return (uint32_t)(__builtin_amdgcn_readlane((int)v, 0) | __builtin_amdgcn_readlane((int)v, 16) | __builtin_amdgcn_readlane((int)v, 32) | __builtin_amdgcn_readlane((int)v, 48));
}
__device__ __forceinline__ long long wave_add64(long long v) {
    v += dpp_i64<DPP_XOR1>(v); v += dpp_i64<DPP_XOR2>(v); v += dpp_i64<DPP_HALF_MIRROR>(v); v += dpp_i64<DPP_MIRROR>(v);
    long long t = 0;
#pragma unroll
    for (int r = 0; r < 64; r += 16)
        t += (long long)(((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)(v >> 32), r) << 32) | (uint32_t)__builtin_amdgcn_readlane((int)v, r));
    return t;
}

#define KEY_ALLELE(k) (((k) >> 8) & 63u)
// fragment state word (after resolve), kept in fmin[]: present | allele << 8 | probability index.
// The index is the merged quality of a 'Paired' fragment (error prob 10^(-q/10)) or PIDX_UNPAIRED
// for a single read (prob forced to 0.1, smCounter.py:67-68); qualities are <= 126 by the batch contract
// (a BAM holds 0..93) and clamped to that.
// read-class table entry (two words): nine 5-bit tally increments in SMC_T_* order (six in .x, three in .y),
// .y bit 31 = the read enters bcDict (incCond).  (In-deletion reads need nothing special here: the batch carries
// minBQ as their quality, smCounter.py:418.)
#define CLS_INC 0x80000000u
#define ST_PRESENT 0x80000000u
#define ST_PAIRED 0x40000000u
// raw fragment word (P1): first read in bits 0-13 (allele << 8 | quality), second in 14-27
#define FW_HAS1 0x10000000u
#define FW_HAS2 0x20000000u
#define FW_OVERFLOW 0x40000000u   // three or more reads share the slot: exact replay
#define ST_MARK 0x08000000u       // (after R) slot waits for the replay
#define ST_HAD 0x20000000u     // slot had included reads (its barcode is a key of bcDict) but the fragment was deleted
#define PIDX_UNPAIRED 127u
__device__ __forceinline__ uint32_t make_state(int allele, int bq, bool paired) {
    return ST_PRESENT | (paired ? ST_PAIRED : 0u) | ((uint32_t)allele << 8) | (uint32_t)bq;
}

struct ReadRec {
    int allele, bq_eff, kind;
    bool inc, r2, rev, lowq;
    int dbc, dpr;
};

__device__ __forceinline__ ReadRec decode_read(uint32_t m, uint32_t d, const KParams& P) {
    ReadRec r;
    r.allele = m & 0xff;
    int bq = (m >> 8) & 0xff, fl = (m >> 16) & 0xff, mq = m >> 24;
    r.kind = (fl >> SMC_KIND_SHIFT) & 3;
    r.r2 = fl & SMC_FL_R2;
    r.rev = fl & SMC_FL_REV;
    r.lowq = (r.kind == SMC_KIND_BASE) && bq < P.min_bq;          // smCounter.py:428
    r.bq_eff = (r.kind == SMC_KIND_GAP) ? P.min_bq : bq;          // :418
    r.inc = r.bq_eff >= P.min_bq && mq >= P.min_mq && (fl & SMC_FL_MMOK);  // :378
    r.dbc = d & 0xffff;
    r.dpr = d >> 16;
    return r;
}

// 10^x and log10 in double.  The LUT of 10^(-q/10) is computed on the host with the same libm the CPU
// restatement uses; the PCR-error terms and the final -log10 use the device library (a few ulp from
// glibc's pow/log10: invisible at the 1e-6 tolerance on PI, and symmetric across alleles so exact
// ties between alleles stay exact).
__device__ __forceinline__ double pcr_of(int cnt, double denom) { return exp10(-6.0 * ((cnt + 0.5) / denom)); }

#define LUT_N 128
// Predicates are kept as 64-bit lane masks: a ballot of a single compare is one v_cmp into an SGPR
// pair, combinations are scalar ANDs, "counter += predicate" is one v_addc_co_u32 with the mask as
// carry-in, and a branch on a mask is s_and_saveexec (inverse ballot) - no per-lane 0/1 integers.
typedef unsigned long long lmask;
#define BAL(x) __builtin_amdgcn_ballot_w64(x)
#define LANES(m) __builtin_amdgcn_inverse_ballot_w64(m)
#define ADDM(acc, m) asm volatile("v_addc_co_u32 %0, vcc, 0, %0, %1" : "+v"(acc) : "s"((lmask)(m)) : "vcc")   // qualities with an LDS-resident error probability; rarer ones read the global table


// calProb for a barcode whose fragments all show ONE allele (the common case) depends only on the fragment
// count nf: prodP[allele] == rightP bit for bit (same factors, same order), so rightP cancels in the posterior
// (smCounter.py:83-96) and   post(allele) = (pne + pcr(0)) / (pne + pcr(0) + 3 pcr(nf)),  post(pad) = pcr(nf) / (same)
// with pcr(c) = 10^(-6 (c + .5) / (nf + 2))  (:79-81, |uniqBase| = 4).  The two -log10(1 - post) values are
// tabulated once per context for nf < SMC_SIMPLE_N by the device code below; larger barcodes take the general path.
#define SMC_SIMPLE_N 4096
__global__ void k_simple_table(double* __restrict__ out, int n) {
    const int nf = blockIdx.x * blockDim.x + threadIdx.x;
    if (nf >= n) return;
    const double pne = 1.0 - 3e-5;
    double pred0 = 0.0, predpad = 0.0;
    if (nf > 0) {
        const double denom = nf + 2.0;
        const double pcr_self = pcr_of(nf, denom), pcr_zero = pcr_of(0, denom);
        const double tmp0 = pne + pcr_zero, padOut = pcr_self;
        double sumP = tmp0;
        sumP += padOut; sumP += padOut; sumP += padOut;
        const double post0 = tmp0 / sumP, postp = padOut / sumP;
        const double x0 = 1.0 - post0;
        pred0 = x0 > 0.0 ? -log10(x0) : 16.0;                         // :508-510
        // -log10(1 - t) for the padded keys: t is tiny, the series is exact to < 1e-19 below 1e-6
        if (postp < 1e-6) predpad = postp * (1.0 + postp * (0.5 + postp * (1.0 / 3.0))) * 0.43429448190325182765;
        else { const double xp = 1.0 - postp; predpad = xp > 0.0 ? -log10(xp) : 16.0; }
    }
    out[2 * nf] = pred0;
    out[2 * nf + 1] = predpad;
}

// ---- fixed-point PI sums: value * 2^shift as 64-bit integers (order-independent adds).  Conversions are spelled
// out (5 / 3 instructions) instead of the generic 64-bit casts (~ 50 each).
__device__ __forceinline__ long long to_fx(double pred, double fxscale) {
    const double t = pred * fxscale + 0.5;                   // in [0, 2^53): pred <= 16, shift <= 48
    const double hi_d = floor(t * 0x1p-32);
    const uint32_t hi = (uint32_t)hi_d, lo = (uint32_t)fma(hi_d, -0x1p32, t);
    return (long long)(((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ double from_fx(unsigned long long v, double inv /* 2^-shift */) {
    return fma((double)(uint32_t)(v >> 32), 0x1p32, (double)(uint32_t)v) * inv;   // one rounding, like the cast
}

// ---- E stage shared by both locus kernels: ranking and candidates (smCounter.py:534-555), one thread.
// R points at a zeroed row staged in LDS; misc[] holds the M_* counters.
// Called either by one thread (lane 0, step 1) or by every lane of one wavefront (lane, step 64): the scalar logic
// is then computed redundantly by all lanes (same instructions, same values) and the tally copies are spread over
// the lanes.
__device__ __forceinline__ void finish_row(smc_row* R, const smc_locus& L, int li, int n, int nF, int used, bool downsampled,
                                           double fxscale, const uint32_t* misc, const uint32_t* tal,
                                           const unsigned long long* pifx, const uint32_t* mtc, const uint32_t* strong,
                                           uint32_t* flt_list, int lane, int step) {
    const int nA = L.n_alleles;
    const unsigned long long touched = ((unsigned long long)misc[M_TOUCH_HI] << 32) | misc[M_TOUCH_LO];
    const int nkeys = __popcll(touched);
    const double inv = 1.0 / fxscale;                               // exact: fxscale is a power of two
    auto PI = [&](int a) { return from_fx(pifx[a], inv); };
    // py2 dict slot order of the fixed keys A,T,G,C,N,DEL: 8-slot table (<= 5 keys) / 32-slot table (py2compat.py)
    auto rank = [&](int a) {
        return a < 6 ? (int)(((nkeys <= 5 ? 0x040702060500ull : 0x140F02061500ull) >> (8 * a)) & 0xffull) : 64 + a;
    };
    // top two by (PI desc, rank asc): one pass (a strict total order, so this equals two argmax passes)
    int best = -1, second = -1;
    double vb = 0.0, vs = 0.0;
    int rb = 0, rs = 0;
    for (int a = 0; a < nA; ++a) {
        if (!((touched >> a) & 1ull)) continue;
        const double v = PI(a);
        const int r = rank(a);
        if (best < 0 || v > vb || (v == vb && r < rb)) { second = best; vs = vb; rs = rb; best = a; vb = v; rb = r; }
        else if (second < 0 || v > vs || (v == vs && r < rs)) { second = a; vs = v; rs = r; }
    }
    R->status = downsampled ? SMC_ST_DOWNSAMPLED : SMC_ST_OK;
    R->n_touched = nkeys;
    R->cvg = n;
    R->all_frag = nF;
    R->all_mt = misc[M_ALLMT];
    R->used_frag = misc[M_USEDFRAG];
    R->used_mt = used;
    R->mt3 = misc[M_MT3]; R->mt5 = misc[M_MT5]; R->mt7 = misc[M_MT7]; R->mt10 = misc[M_MT10];
    R->max_allele = best; R->second_allele = second;
    R->touched_mask = touched;
    for (int k = lane; k < 4; k += step) {
        R->dp[k] = tal[k * SMC_NT + SMC_T_CNT];
        R->umt[k] = mtc[k];
        R->vsm[k] = strong[k];
        R->pi[k] = PI(k);
    }
    const int ref = L.ref_allele;
    if (ref < nA) for (int k = lane; k < SMC_NT; k += step) R->ref_tal[k] = tal[ref * SMC_NT + k];
    auto fill = [&](smc_cand& C, int a, double pia) {
        C.allele = a;
        C.p_sb = C.p_r1 = C.p_r2 = C.p_pr = NAN;
        if (a < 0) return;
        C.pi = pia;
        C.vdp = tal[a * SMC_NT + SMC_T_CNT];
        C.vmt = mtc[a];
        C.vsm = strong[a];
        for (int k = lane; k < SMC_NT; k += step) C.tal[k] = tal[a * SMC_NT + k];
    };
    auto filterable = [&](int a) { return ((L.snp_mask >> a) & 1ull) || a != GAP_ID; };  // SNP or INDEL
    const int alt = best == ref ? second : best;                                   // :541
    const double pi_alt = best == ref ? vs : vb;
    fill(R->cand[0], alt, pi_alt);
    const bool flt0 = alt >= 0 && pi_alt >= 5 && filterable(alt);                 // :549
    bool flt1 = false;
    if (flt0) R->cand[0].flt_applied = 1;
    if (best >= 0 && second >= 0 && best != ref && second != ref && 1.0 * mtc[best] / used >= 0.45 &&
        1.0 * mtc[second] / used >= 0.45) {                                        // :553-555
        R->biallelic = 1;
        fill(R->cand[1], second, vs);
        flt1 = vs >= 5 && filterable(second);                                      // :563
        if (flt1) R->cand[1].flt_applied = 1;
    } else {
        fill(R->cand[1], -1, 0.0);
    }
    // loci whose candidate(s) go through filterVariants are queued for k_filter_loci
    if ((flt0 || flt1) && lane == 0) flt_list[1 + atomicAdd(&flt_list[0], 1u)] = (uint32_t)li;
}

// ------------------------------------------------------------------------------------------
// kernel 1: scan + group + score + rank
// ------------------------------------------------------------------------------------------
#ifndef SMC_ABLATE
#define SMC_ABLATE 0   // diagnostic builds: return after phase N (timing only, rows are garbage)
#endif
// 6 waves per SIMD (<= 80 VGPRs; LDS allows 14 workgroups of 2 waves per CU on the C3 shape): measured best of 4..8
#ifndef SMC_WAVES_PER_EU
#define SMC_WAVES_PER_EU 6
#endif
#ifndef SMC_WALK_UNROLL
#define SMC_WALK_UNROLL 2
#endif
template <int BLOCK, bool GLOBAL_TABLES>
__global__ __launch_bounds__(BLOCK) __attribute__((amdgpu_waves_per_eu(SMC_WAVES_PER_EU, 8))) void k_call_loci(
    KParams P, const smc_locus* __restrict__ loci, const int* __restrict__ order, int a_cap,
    const uint32_t* __restrict__ g_meta, const uint32_t* __restrict__ g_umi_start, const uint32_t* __restrict__ g_frag,
    const uint32_t* __restrict__ g_dist, const double* __restrict__ g_lut, const double* __restrict__ g_simple,
    smc_row* __restrict__ rows, uint8_t* __restrict__ scratch, const int64_t* __restrict__ scratch_off,
    uint32_t* __restrict__ flt_list, const uint8_t* __restrict__ redo_flag) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    // `loci` is this bin's descriptor array in launch order; order[] maps back to the batch index
    const int li = order[blockIdx.x];
    // when the sorted-stream kernel ran first, only the loci it handed over are processed here
    if (redo_flag != nullptr && !redo_flag[li]) return;
    const smc_locus L = loci[blockIdx.x];
    const int n = L.n_reads, nU = L.n_umi, nF = L.n_frag, nA = L.n_alleles;
    const uint4* meta4 = (const uint4*)(g_meta + 4ll * L.read_off4);
    const uint4* frag4 = (const uint4*)(g_frag + 4ll * L.read_off4);

    // ---- carve LDS
    Hdr* H = (Hdr*)smem;
    uint32_t* tal = (uint32_t*)(smem + sizeof(Hdr));                 // [a_cap][SMC_NT]
    unsigned long long* pifx = (unsigned long long*)(tal + a_cap * SMC_NT);
    uint32_t* mtc = (uint32_t*)(pifx + a_cap);
    uint32_t* strong = mtc + a_cap;
    double* lut = (double*)(strong + a_cap);                         // [LUT_N]
    smc_row* rowst = (smc_row*)lut;                                   // row stage: over the LUT once it is dead
    uint2* cls_lut = (uint2*)(lut + LUT_N);                           // [32] read class -> tally increments / flags
    static_assert(sizeof(smc_row) <= LUT_N * sizeof(double), "row stage must fit in the LUT");
    unsigned char* tab = GLOBAL_TABLES ? (scratch + scratch_off[blockIdx.x]) : (smem + lds_hdr_bytes(a_cap));
    uint32_t* umi_base = (uint32_t*)tab;                              // [nU+1] first slot of each barcode
    uint32_t* fmin = umi_base + (nU + 1);                             // [nF] fragment word: raw (P1) then state (R)
    uint32_t* worklist = fmin + nF;                                   // [nU] barcodes queued for the general calProb path
    uint32_t* bcinfo = worklist + nU;                                 // [nU] fragment count of a one-allele barcode still to score
    unsigned char* umi_flag = (unsigned char*)(bcinfo + nU);          // [nU]
    // per 64 fragment slots, after the merge: which slots hold a fragment, which of those show the reference allele
    unsigned long long* cmask = (unsigned long long*)(tab + ((4u * (uint32_t)(nU + 1) + 4u * (uint32_t)nF + 9u * (uint32_t)nU + 7u) & ~7u));

    STAMP_INIT();
    // first step's reads are requested before the LDS image is initialised (HBM latency overlaps it)
    const int n4 = (n + 3) >> 2;
    uint4 m4, f4;
    // + the frag words of the two reads before the lane's quad: a read's rank inside its fragment
    // comes from its predecessors, which sit right before it in barcode-major order
    uint2 pf2 = make_uint2(0xFFFFFFFFu, 0xFFFFFFFFu);
    const uint2* frag2 = (const uint2*)frag4;
    if (tid < n4) {
        m4 = meta4[tid]; f4 = frag4[tid];
        // (only the first lane of a wavefront fetches them; the others take them from their neighbour lane)
        if (lane == 0 && tid > 0) pf2 = frag2[2 * tid - 1];
    }
    // ---- S0: init
    uint32_t b0_early = 0xFFFFFFFFu;
    {
        uint32_t* z = (uint32_t*)smem;
        const int nz = (int)((sizeof(Hdr) + a_cap * 64) / 4);
        for (int i = tid; i < nz; i += BLOCK) z[i] = 0;
        for (int i = tid; i < nF; i += BLOCK) fmin[i] = 0u;
        // quality -> error-probability table (read by the calProb phase; 'unpaired' -> 0.1, smCounter.py:65-68)
        for (int i = tid; i < LUT_N; i += BLOCK) lut[i] = i == (int)PIDX_UNPAIRED ? 0.1 : g_lut[i];
        if (tid < 32) cls_lut[tid] = ((const uint2*)(g_lut + 256))[tid];   // the class table sits behind the quality table
        // first read of every barcode (+ closing entry); S2 turns it into the first fragment slot
        const uint32_t* ustart = g_umi_start + L.umi_off;
        for (int i = tid; i <= nU; i += BLOCK) umi_base[i] = ustart[i] & ~SMC_USTART_DROPPED;
        if (L.flags & SMC_LF_SAMPLED)                                  // the host's down-sampling marks, until U
            for (int i = tid; i < nU; i += BLOCK) umi_flag[i] = (unsigned char)(ustart[i] >> 31);
        // one barcode per thread at most: its first slot (= slot of its first read) is fetched now and rides in
        // a register through the scan - the scan streams the same lines right after, so they are fetched from
        // HBM once (a gather after the scan finds them evicted: + 10 % traffic)
        if (nU <= BLOCK && tid < nU) {
            const uint32_t r0 = ustart[tid] & ~SMC_USTART_DROPPED;
            b0_early = r0 < (uint32_t)n ? ((g_frag + 4ll * L.read_off4)[r0] & SMC_FRAG_SLOT_MASK) : 0xFFFFFFFFu;
        }
    }
    __syncthreads();
    STAMP(0);
    if (SMC_ABLATE == 1) return;

    // ---- P1: ONE pass over the reads, 4 reads per lane per step (16-byte loads of each plane):
    // inclusion test and tallies (smCounter.py:368-460), first slot of each barcode, which barcodes
    // enter bcDict (:467-468), and per fragment the smallest / largest key of its included reads.
    // Tallies of the locus's reference allele (nearly every read) are kept in per-lane registers and
    // reduced once; other alleles are aggregated per wave step, only when present.
    {
        uint32_t accv[9];
#pragma unroll
        for (int k = 0; k < 9; ++k) accv[k] = 0;
        // per-lane tallies of the reference allele, nine 5-bit fields in two words (the class table's layout);
        // a step adds at most 4 to a field, so they are spilled into accv[] every 7 steps
        uint32_t acc0 = 0, acc1 = 0;
        int steps = 0;
        auto spill = [&]() {
#pragma unroll
            for (int t = 0; t < 6; ++t) accv[t] += (acc0 >> (5 * t)) & 31u;
#pragma unroll
            for (int t = 6; t < 9; ++t) accv[t] += (acc1 >> (5 * (t - 6))) & 31u;
            acc0 = acc1 = 0u;
        };
        uint32_t n_inc_s = 0;
        lmask err_m = 0, ovf_any = 0;
        const uint32_t refa = L.ref_allele;
        for (int qb = 0; qb < n4; qb += BLOCK) {
            const int q = qb + tid;
            const uint4 cm = m4, cf = f4;
            const bool full = 4 * (qb + BLOCK) <= n;                     // every read of this step exists
            const uint2 cpf = pf2;
            {   // prefetch the next step while this one is processed
                const int qn = q + BLOCK;
                if (qn < n4) {
                    m4 = meta4[qn]; f4 = frag4[qn];
                    if (lane == 0) pf2 = frag2[2 * qn - 1];
                }
            }
            const uint32_t ms[4] = {cm.x, cm.y, cm.z, cm.w};
            const uint32_t fw[4] = {cf.x, cf.y, cf.z, cf.w};
            // What each read adds: looked up by its class (frag word bits 27-31; smcounter_hip.h) - four LDS reads
            // issued together; + the class of the read just before the quad (lane 0's loaded word; the other lanes
            // read a dummy entry)
            uint2 cw[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) cw[k] = cls_lut[fw[k] >> SMC_FRAG_CLASS_SHIFT];
            const uint2 cwp = cls_lut[cpf.y >> SMC_FRAG_CLASS_SHIFT];
            const uint32_t fs[4] = {fw[0] & SMC_FRAG_SLOT_MASK, fw[1] & SMC_FRAG_SLOT_MASK, fw[2] & SMC_FRAG_SLOT_MASK,
                                    fw[3] & SMC_FRAG_SLOT_MASK};
            // slots of the two reads before the quad: the neighbour lane's last two (wavefront shift right by one
            // lane, DPP), lane 0 keeps what it loaded
            const uint32_t fprev1 = (uint32_t)__builtin_amdgcn_update_dpp((int)(cpf.y & SMC_FRAG_SLOT_MASK), (int)fs[3], 0x138, 0xF, 0xF, false);
            const uint32_t fprev2 = (uint32_t)__builtin_amdgcn_update_dpp((int)(cpf.x & SMC_FRAG_SLOT_MASK), (int)fs[2], 0x138, 0xF, 0xF, false);
            lmask m_okk[4], m_inck[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const lmask m_valid = full ? ~0ull : BAL(4 * q + k < n);  // n4 = ceil(n/4): q < n4 follows
                m_okk[k] = m_valid & BAL(fs[k] < (uint32_t)nF) & BAL((ms[k] & 0xffu) < (uint32_t)nA) &
                           BAL((fw[k] >> SMC_FRAG_CLASS_SHIFT) < (uint32_t)SMC_N_READ_CLASS);
                err_m |= m_valid & ~m_okk[k];
                m_inck[k] = m_okk[k] & BAL((int)cw[k].y < 0);             // incCond (:378), evaluated by the host
            }
            // inclusion of the read just before the quad: the neighbour lane's fourth read, or lane 0's loaded one
            lmask m_inc_prev = (m_inck[3] << 1) | (BAL((int)cwp.y < 0) & BAL(q > 0) & 1ull);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const uint32_t mw = ms[k], f = fs[k];
                const uint32_t a = mw & 0xffu;
                const lmask m_ok = m_okk[k], m_inc = m_inck[k];
                const lmask m_ref = m_ok & BAL(a == refa);
                n_inc_s += (uint32_t)__popcll(m_inc);
                acc0 += LANES(m_ref) ? cw[k].x : 0u;
                acc1 += LANES(m_ref) ? cw[k].y : 0u;                      // (the flag bit above the fields just wraps)
                lmask nr = m_ok & ~m_ref;
                if (nr) {
                    // stray reads (sequencing errors, indel alleles): the class's increments, one tally at a time
                    lmask tm[9];
#pragma unroll
                    for (int t = 0; t < 6; ++t) tm[t] = BAL(((cw[k].x >> (5 * t)) & 1u) != 0u);
#pragma unroll
                    for (int t = 6; t < 9; ++t) tm[t] = BAL(((cw[k].y >> (5 * (t - 6))) & 1u) != 0u);
                    if (__popcll(nr) <= 6) {
                        uint32_t* t_ = tal + a * SMC_NT;
#pragma unroll
                        for (int t = 0; t < 9; ++t)
                            if (LANES(nr & tm[t])) atomicAdd(&t_[t], 1u);
                    } else {                                            // many: aggregate per allele
                        while (nr) {
                            const int src = __ffsll((long long)nr) - 1;
                            const uint32_t a0 = (uint32_t)__builtin_amdgcn_readlane((int)a, src);
                            const lmask ma = m_ok & BAL(a == a0);
                            nr &= ~ma;
                            if (lane == 0) {
                                uint32_t* t_ = tal + a0 * SMC_NT;
#pragma unroll
                                for (int t = 0; t < 9; ++t) {
                                    const uint32_t c = (uint32_t)__popcll(ma & tm[t]);
                                    if (c) atomicAdd(&t_[t], c);
                                }
                            }
                        }
                    }
                }
                {
                    // The fragment word: (allele, quality) of its first and of its second read, in pileup order
                    // (= memory order: a fragment's reads are adjacent).  A read is second when the read before it
                    // has the same slot and is included; a slot with three or more reads is flagged for the
                    // exact replay below.
                    const uint32_t fp1 = k == 0 ? fprev1 : fs[k > 0 ? k - 1 : 0];
                    const uint32_t fp2 = k == 0 ? fprev2 : (k == 1 ? fprev1 : fs[k > 1 ? k - 2 : 0]);
                    const lmask m_same1 = BAL(f == fp1);
                    const lmask m_second = m_same1 & m_inc_prev;
                    const lmask m_ovf = m_ok & m_same1 & BAL(f == fp2);
                    ovf_any |= m_ovf;
                    if (LANES(m_inc | m_ovf)) {
                        uint32_t bq_eff = (mw >> 8) & 0xffu;              // (in-deletion reads carry minBQ, :418)
                        bq_eff = bq_eff < PIDX_UNPAIRED ? bq_eff : PIDX_UNPAIRED - 1u;   // contract: quality <= 126
                        uint32_t w = (a << 8) | bq_eff;
                        w = LANES(m_second) ? (w << 14) | FW_HAS2 : w | FW_HAS1;
                        w = LANES(m_inc) ? w : 0u;
                        w |= LANES(m_ovf) ? FW_OVERFLOW : 0u;
                        atomicOr(&fmin[f], w);
                    }
                    m_inc_prev = m_inc;
                }
            }
            if (++steps == 7) { spill(); steps = 0; }
        }
        spill();
#pragma unroll
        for (int k = 0; k < 9; ++k) {
            const uint32_t v = (uint32_t)wave_add((int)accv[k]);
            if (lane == 0 && v && refa < (uint32_t)nA) atomicAdd(&tal[refa * SMC_NT + k], v);
        }
        if (lane == 0 && n_inc_s) atomicAdd(&H->misc[M_NINC], n_inc_s);
        if (err_m && lane == 0) H->misc[M_ERR] = 1;
        if (ovf_any && lane == 0) H->misc[M_NEEDFIX] = 1;
    }
    __syncthreads();
    STAMP(1);
    if (SMC_ABLATE == 2) return;

    // ---- S2: first fragment slot of every barcode = slot of its first read (reads are barcode-major, slots
    // ascending inside a barcode); read ranges and slot ranges must be ordered and cover [0, n) / [0, nF)
    {
        const uint32_t* frag = g_frag + 4ll * L.read_off4;
        uint32_t bad = 0;
        for (int ub = 0; ub < nU; ub += BLOCK) {
            const int u = ub + tid;
            uint32_t b0 = 0;
            if (u < nU) {
                const uint32_t r0 = umi_base[u], r1 = umi_base[u + 1];
                bad |= !(r0 < r1 && r1 <= (uint32_t)n) || (u == 0 && r0 != 0) || (u == nU - 1 && r1 != (uint32_t)n);
                b0 = nU <= BLOCK ? b0_early : (r0 < (uint32_t)n ? (frag[r0] & SMC_FRAG_SLOT_MASK) : 0xFFFFFFFFu);
            }
            __syncthreads();                                            // every r1 of this step is read
            if (u < nU) umi_base[u] = b0;
        }
        if (tid == 0) umi_base[nU] = (uint32_t)nF;
        __syncthreads();
        for (int u = tid; u < nU; u += BLOCK) {
            const uint32_t b0 = umi_base[u], b1 = umi_base[u + 1];
            bad |= b1 <= b0 || b1 > (uint32_t)nF || (u == 0 && b0 != 0);
        }
        if (nU == 0 && n != 0) bad = 1;
        if (__builtin_amdgcn_ballot_w64(bad != 0) && lane == 0) H->misc[M_ERR] = 1;
        if (tid == 0) H->misc[M_ALLMT] = (uint32_t)nU;                  // allMT (:482): every barcode has a read
    }
    __syncthreads();
    STAMP(2);
    smc_row* out = rows + li;
    if (H->misc[M_ERR] || H->misc[M_NINC] == 0 || P.ds <= 0) {          // (ds <= 0: usedMT = min(ds, .) = 0)
        // bad input, or no read enters bcDict: the Zero_Coverage row (smCounter.py:489-494)
        for (int i = tid; i < (int)(sizeof(smc_row) / 4); i += BLOCK) ((uint32_t*)rowst)[i] = 0u;
        __syncthreads();
        if (tid == 0) {
            smc_row* R = rowst;
            R->status = H->misc[M_ERR] ? SMC_ST_BAD_INPUT : SMC_ST_ZERO_COVERAGE;
            R->cvg = n;
            R->all_mt = H->misc[M_ALLMT];
            R->all_frag = nF;
            R->max_allele = R->second_allele = -1;
            for (int k = 0; k < 4; ++k) R->dp[k] = tal[k * SMC_NT + SMC_T_CNT];
            for (int c = 0; c < 2; ++c) {
                R->cand[c].allele = -1;
                R->cand[c].p_sb = R->cand[c].p_r1 = R->cand[c].p_r2 = R->cand[c].p_pr = NAN;
            }
        }
        __syncthreads();
        const uint32_t* src = (const uint32_t*)rowst;
        uint32_t* dst = (uint32_t*)out;
        for (int i = tid; i < (int)(sizeof(smc_row) / 4); i += BLOCK) dst[i] = src[i];
        return;
    }

    // ---- R: mate merge (smCounter.py:468-479) from the fragment words; slots flagged FW_OVERFLOW wait for the
    // exact replay below.
    {
        uint32_t conc_ref = 0, disc_ref = 0;                             // wave-uniform counts
        const uint32_t refa = L.ref_allele;
        // one chunk of 64 slots per wavefront: classify, write the state word, leave the chunk masks
        auto chunk = [&](int s, lmask m_in, uint32_t w) {
            const lmask m_marked = m_in & BAL((w & FW_OVERFLOW) != 0u);
            const lmask m_has = m_in & ~m_marked & BAL((w & FW_HAS1) != 0u);
            const lmask m_pair = m_has & BAL((w & FW_HAS2) != 0u), m_single = m_has & ~m_pair;
            const uint32_t a1 = (w >> 8) & 63u, a2 = (w >> 22) & 63u, q1 = w & 255u, q2 = (w >> 14) & 255u;
            const lmask m_same = BAL(a1 == a2);
            const lmask m_merge = m_pair & (m_same | BAL(a2 == (uint32_t)N_ID));
            const lmask m_conc = m_pair & m_same, m_disc = m_pair & ~m_merge;   // :475-476 / :478-479
            // state: first read's allele; prob = max(prob_new, prob_old) <=> min quality (:473)
            uint32_t st = 0u;
            if (LANES(m_has)) st = ST_HAD;                                // discordant pair: deleted (:477-479)
            if (LANES(m_single)) st = ST_PRESENT | (w & 0x3F00u) | PIDX_UNPAIRED;
            if (LANES(m_merge)) st = ST_PRESENT | ST_PAIRED | (w & 0x3F00u) | (q1 < q2 ? q1 : q2);
            if (LANES(m_marked)) st = ST_MARK;
            if (LANES(m_in)) fmin[s] = st;
            const lmask m_a1ref = BAL(a1 == refa);
            if (lane == 0 && s < nF) {                                   // chunk masks for the calProb phase
                const lmask m_live = m_single | m_merge;
                cmask[2 * (s >> 6)] = m_live;
                cmask[2 * (s >> 6) + 1] = m_live & m_a1ref;
            }
            const lmask c_ref = m_conc & m_a1ref, d_ref = m_disc & BAL(a2 == refa);
            conc_ref += (uint32_t)__popcll(c_ref);
            disc_ref += (uint32_t)__popcll(d_ref);
            const lmask rare = (m_conc & ~c_ref) | (m_disc & ~d_ref);
            if (rare) {
                if (LANES(m_conc & ~c_ref)) atomicAdd(&tal[a1 * SMC_NT + SMC_T_CONCORD], 1u);
                if (LANES(m_disc & ~d_ref)) atomicAdd(&tal[a2 * SMC_NT + SMC_T_DISCORD], 1u);
            }
        };
        // two chunks per step, their LDS reads issued together
        for (int sb = 0; sb < nF; sb += 2 * BLOCK) {
            const int s0 = sb + tid, s1 = s0 + BLOCK;
            const lmask in0 = BAL(s0 < nF), in1 = BAL(s1 < nF);
            uint32_t w0 = 0u, w1 = 0u;
            if (LANES(in0)) w0 = fmin[s0];
            if (LANES(in1)) w1 = fmin[s1];
            chunk(s0, in0, w0);
            if (in1) chunk(s1, in1, w1);
        }
        if (lane == 0 && refa < (uint32_t)nA) {
            if (conc_ref) atomicAdd(&tal[refa * SMC_NT + SMC_T_CONCORD], conc_ref);
            if (disc_ref) atomicAdd(&tal[refa * SMC_NT + SMC_T_DISCORD], disc_ref);
        }
    }
    __syncthreads();
    STAMP(4);
    if (SMC_ABLATE == 3) return;
    if (H->misc[M_NEEDFIX]) {
        // Some read name has three or more alignments on this locus (rare): replay each flagged fragment
        // sequentially in read order, one wavefront per fragment, then rebuild the chunk masks.
        const uint32_t* meta = g_meta + 4ll * L.read_off4;
        const uint32_t* frag = g_frag + 4ll * L.read_off4;
        constexpr int NW = BLOCK / WAVE;
        for (int sb = 0; sb < nF; ++sb) {
            if (fmin[sb] != ST_MARK) continue;                // uniform over the block (LDS value)
            if ((sb % NW) != wid) continue;                   // one wave per marked fragment
            bool present = false, paired = false, had = false;
            int sa = 0, sq = 0;
            for (int base = 0; base < n; base += WAVE) {
                const int i = base + lane;
                bool hit = false;
                ReadRec r;
                r.allele = 0; r.bq_eff = 0;
                if (i < n) {
                    r = decode_read(meta[i], 0, P);
                    hit = r.inc && (frag[i] & SMC_FRAG_SLOT_MASK) == (uint32_t)sb;
                }
                unsigned long long hm = __ballot(hit);
                while (hm) {
                    const int src = __ffsll((long long)hm) - 1;
                    hm &= hm - 1;
                    const int a = __shfl(r.allele, src), q = __shfl(r.bq_eff, src);
                    had = true;
                    if (!present) { present = true; paired = false; sa = a; sq = q; }
                    else if (a == sa || a == N_ID) {
                        sq = q < sq ? q : sq; paired = true;
                        if (a == sa && lane == 0) atomicAdd(&tal[a * SMC_NT + SMC_T_CONCORD], 1u);
                    } else {
                        present = false;
                        if (lane == 0) atomicAdd(&tal[a * SMC_NT + SMC_T_DISCORD], 1u);
                    }
                }
            }
            if (lane == 0)
                fmin[sb] = present ? make_state(sa, paired ? sq : (int)PIDX_UNPAIRED, paired) : (had ? ST_HAD : 0u);
        }
        __syncthreads();
        for (int sb = 0; sb < nF; sb += BLOCK) {                          // chunk masks from the final states
            const int s = sb + tid;
            const uint32_t st = s < nF ? fmin[s] : 0u;
            const lmask m_live = BAL((st & ST_PRESENT) != 0u);
            const lmask m_lref = m_live & BAL(KEY_ALLELE(st) == (uint32_t)L.ref_allele);
            if (lane == 0 && s < nF) { cmask[2 * (s >> 6)] = m_live; cmask[2 * (s >> 6) + 1] = m_lref; }
        }
        __syncthreads();
    }

    STAMP(5);
    // ---- U: per-barcode posterior (calProb, :26-98) and PI / consensus accumulation (:506-532)
    {
        // fixed-point scale of the PI sums: order-independent, hence bit-reproducible
        // (sized for the most barcodes the locus can use, so it does not wait for the bcDict count)
        const int ubound = nU < P.ds ? nU : P.ds;
        int bits = 32 - __clz(ubound);
        int shift = 58 - bits; if (shift > 48) shift = 48;
        const double fxscale = (double)(1ull << shift);
        // lane accumulators of the table-scored barcodes: everything they add goes to the reference allele
        // (fx0, consensus, strong) or equally to its padded keys (fxp)
        long long acc_fx0 = 0, acc_fxp = 0;
        int acc_mt = 0, acc_st = 0;
        int c3 = 0, c5 = 0, c7 = 0, c10 = 0, ufrag = 0;
        uint32_t touch_lo = 0, touch_hi = 0;
        const double pne = 1.0 - 3e-5;                                 // pcr_no_error, :20
        const int refa = L.ref_allele;
        // error probability of a fragment: one LDS read, no branch.  The state's low byte is the merged
        // quality (<= 126 by the batch contract, features.py) or PIDX_UNPAIRED (-> 0.1, :65-68).
        auto prob_of = [&](uint32_t st) -> double { return lut[st & (LUT_N - 1)]; };

        // pass A of one barcode: fragment count, allele set, P(no sequencing error); speculatively also
        // the count and product for the locus's reference allele (the only allele of most barcodes)
        constexpr int Gc = 8;         // lanes per barcode on the general path (grp8_* reductions)
        const int jc = tid % Gc;
        auto walk = [&](int u, int& b0, int& b1, int& nf, int& cnt_ref, unsigned long long& mask, double& rightP,
                        double& prod_ref, double& prod_x) {
            b0 = umi_base[u]; b1 = umi_base[u + 1];
            // pass A: fragment count, allele set, P(no sequencing error); speculatively also the
            // count and product for the locus's reference allele (the only allele of most barcodes)
            nf = 0; cnt_ref = 0;
            unsigned long long mk = 0;
            rightP = 1.0; prod_ref = 1.0; prod_x = 1.0;
            {
                // WU slots per lane per step, independent partial products (the walk is a chain of dependent
                // LDS reads and FP64 multiplies: instruction-level parallelism hides it; two keep the
                // register count low enough for 6 waves per SIMD)
                constexpr int WU = SMC_WALK_UNROLL;
                double rp[WU], pr[WU], px[WU];
#pragma unroll
                for (int t = 0; t < WU; ++t) rp[t] = pr[t] = px[t] = 1.0;
                for (int s0 = b0 + jc; s0 < b1; s0 += WU * Gc) {
                    uint32_t st[WU];
#pragma unroll
                    for (int t = 0; t < WU; ++t) { const int s = s0 + t * Gc; st[t] = s < b1 ? fmin[s] : 0u; }
                    double pv[WU];
#pragma unroll
                    for (int t = 0; t < WU; ++t) pv[t] = prob_of(st[t]);
#pragma unroll
                    for (int t = 0; t < WU; ++t) {
                        const bool present = (st[t] & ST_PRESENT) != 0u;
                        const int a = KEY_ALLELE(st[t]);
                        const bool same = a == refa;
                        const double q1 = 1.0 - pv[t];
                        nf += present;
                        cnt_ref += present && same;
                        mk |= present ? (1ull << a) : 0ull;
                        rp[t] *= present ? q1 : 1.0;
                        pr[t] *= present ? (same ? q1 : pv[t]) : 1.0;
                        px[t] *= present ? (same ? pv[t] : q1) : 1.0;   // P(reads | the other allele), if there is just one
                    }
                }
                rightP = rp[0]; prod_ref = pr[0]; prod_x = px[0];
#pragma unroll
                for (int t = 1; t < WU; ++t) { rightP *= rp[t]; prod_ref *= pr[t]; prod_x *= px[t]; }
            }
            nf = grp8_add(nf);
            cnt_ref = grp8_add(cnt_ref);
            const uint32_t mlo = grp8_or((uint32_t)mk), mhi = grp8_or((uint32_t)(mk >> 32));
            rightP = grp8_mul(rightP);
            prod_ref = grp8_mul(prod_ref);
            prod_x = grp8_mul(prod_x);
            mask = ((unsigned long long)mhi << 32) | mlo;
        };

        // ---- phase 0: one lane per barcode.  Fragment count and "every fragment shows the reference allele"
        // come from the chunk masks the merge left (no walk); such barcodes (nearly all) are scored from the
        // per-count table, the others are queued
        // fragment count / reference-only count / "is a key of bcDict" (:467-468: has an included read, even if
        // every fragment was deleted later) of barcode u
        auto barcode_counts = [&](int u, int& nf, int& cr, unsigned long long& live1, uint32_t& c1) -> bool {
            const uint32_t b0 = umi_base[u], b1 = umi_base[u + 1];
            nf = 0; cr = 0; live1 = 0; c1 = 0;
            for (uint32_t c = b0 >> 6; c <= ((b1 - 1u) >> 6); ++c) {
                const uint32_t lo = b0 > 64u * c ? b0 - 64u * c : 0u, hi = b1 < 64u * c + 64u ? b1 - 64u * c : 64u;
                const unsigned long long range = (hi >= 64u ? ~0ull : ((1ull << hi) - 1ull)) & ~((1ull << lo) - 1ull);
                const unsigned long long lv = cmask[2 * c] & range;
                nf += __popcll(lv);
                cr += __popcll(cmask[2 * c + 1] & range);
                if (lv) { live1 = lv; c1 = c; }
            }
            if (nf) return true;
            for (uint32_t sl = b0; sl < b1; ++sl)
                if (fmin[sl] & ST_HAD) return true;
            return false;
        };
        // More barcodes than the cap can only happen when nU > ds: then a first pass finds bcDict's keys and the
        // down-sampling stand-in (non-parity; the reference random.samples, :496-498) keeps the ds lowest ids.
        // With SMC_LF_SAMPLED the host has run the reference's random.sample (py2compat.py) and marked the keys
        // it dropped; the kept count is checked in E.
        const bool sampled = (L.flags & SMC_LF_SAMPLED) != 0;
        const bool two_pass = nU > P.ds || sampled;
        if (two_pass) {
            uint32_t nb = 0, nkeep = 0;
            for (int u = tid; u < nU; u += BLOCK) {
                int nf, cr; unsigned long long l1; uint32_t c1;
                const bool key = barcode_counts(u, nf, cr, l1, c1);
                const bool keep = key && !(sampled && umi_flag[u]);
                umi_flag[u] = keep;
                nb += key;
                nkeep += keep;
            }
            nb = (uint32_t)wave_add((int)nb);
            nkeep = (uint32_t)wave_add((int)nkeep);
            if (lane == 0 && nb) { atomicAdd(&H->misc[M_NBC], nb); atomicAdd(&H->misc[M_NKEPT], nkeep); }
            __syncthreads();
            if (!sampled && (int)H->misc[M_NBC] > P.ds) {
                if (tid == 0) {
                    int k = 0;
                    for (int u = 0; u < nU; ++u)
                        if (umi_flag[u]) { if (k >= P.ds) umi_flag[u] = 0; ++k; }
                }
                __syncthreads();
            }
        }
        uint32_t nb1 = 0;
        for (int u = tid; u < nU; u += BLOCK) {
            int nf, cr;
            unsigned long long live1;                                  // live slots of the last chunk touched
            uint32_t c1;
            const bool key = barcode_counts(u, nf, cr, live1, c1);
            bcinfo[u] = 0xFFFFFFFFu;                                   // nothing to score in pass B (default)
            if (two_pass ? !umi_flag[u] : !key) continue;              // not a (kept) key of bcDict
            ++nb1;
            ufrag += nf; c3 += nf >= 3; c5 += nf >= 5; c7 += nf >= 7; c10 += nf >= 10;
            if (nf <= P.mt_drop) {                                     // :28-32 -> all four posteriors 0
                touch_lo |= 0xFu;                                      // finalDict gets A,T,G,C (+ -0.0)
                if (nf == 1) {                                         // tie -> single-fragment rule, :521-523
                    const int a = (int)KEY_ALLELE(fmin[64u * c1 + (uint32_t)(__ffsll((long long)live1) - 1)]);
                    atomicAdd(&mtc[a], 1u);
                }
                continue;
            }
            if (!(refa < 64 && cr == nf && nf < SMC_SIMPLE_N)) {
                worklist[atomicAdd(&H->misc[M_NCOMPLEX], 1u)] = (uint32_t)u;
                continue;
            }
            bcinfo[u] = (uint32_t)nf;
        }
        if (!two_pass) {
            nb1 = (uint32_t)wave_add((int)nb1);
            if (lane == 0 && nb1) atomicAdd(&H->misc[M_NBC], nb1);
        }
        __syncthreads();                                               // the queue of the general path is complete
        STAMP(6);
        // From here every wavefront works on its own, no barrier until the end of U: pass B scores the
        // one-allele barcodes of the wave's threads from the table, then the wave takes its share of the queued
        // barcodes - groups are handed out from the LAST thread down, so when the barcodes fill only the first
        // wavefront(s) the two kinds of work run on different wavefronts at the same time.
        bool scored = false;
        for (int u = tid; u < nU; u += BLOCK) {
            const uint32_t info = bcinfo[u];
            if (info == 0xFFFFFFFFu) continue;
            const int nf = (int)info;
            // one existing allele (the reference), three padded keys (:49-54): nk = 4
            const double pred0 = g_simple[2 * nf], predpad = g_simple[2 * nf + 1];
            acc_fx0 += to_fx(pred0, fxscale);
            acc_fxp += to_fx(predpad, fxscale);
            scored = true;
            if (pred0 > predpad) {                                     // unique maximum (:514-519)
                ++acc_mt;
                acc_st += pred0 > P.smt;
            } else if (nf == 1) {                                      // :521-523
                ++acc_mt;
            }
        }
        // flush the lane accumulators of passes A and B (order-independent integer adds); phase 1 adds straight
        // to LDS, so none of these registers stays live through its FP64 code
        if (tid - lane < nU) {                                         // (wave-uniform: the wave owns barcodes)
        {
            // one existing allele (the reference), three padded keys (:49-54): nk = 4
            const unsigned long long padmask = refa < 4 ? (0xFull & ~(1ull << refa)) : 0x7ull;
            const long long s0 = wave_add64(acc_fx0), sp = wave_add64(acc_fxp);
            const int m = wave_add(acc_mt), st = wave_add(acc_st);
            if (BAL(scored)) {
                const unsigned long long uq = (1ull << refa) | padmask;
                touch_lo |= (uint32_t)uq; touch_hi |= (uint32_t)(uq >> 32);
            }
            if (lane == 0 && refa < 64) {
                if (s0) atomicAdd(&pifx[refa], (unsigned long long)s0);
                if (sp)
                    for (int a = 0; a < 4; ++a)
                        if ((padmask >> a) & 1ull) atomicAdd(&pifx[a], (unsigned long long)sp);
                if (m) atomicAdd(&mtc[refa], (uint32_t)m);
                if (st) atomicAdd(&strong[refa], (uint32_t)st);
            }
        }
        c3 = wave_add(c3); c5 = wave_add(c5);
        c7 = wave_add(c7); c10 = wave_add(c10);
        ufrag = wave_add(ufrag);
        touch_lo = wave_or(touch_lo); touch_hi = wave_or(touch_hi);
        if (lane == 0) {
            atomicAdd(&H->misc[M_MT3], (uint32_t)c3); atomicAdd(&H->misc[M_MT5], (uint32_t)c5);
            atomicAdd(&H->misc[M_MT7], (uint32_t)c7); atomicAdd(&H->misc[M_MT10], (uint32_t)c10);
            atomicAdd(&H->misc[M_USEDFRAG], (uint32_t)ufrag);
            atomicOr(&H->misc[M_TOUCH_LO], touch_lo); atomicOr(&H->misc[M_TOUCH_HI], touch_hi);
        }
        }

        // ---- phase 1: the queued barcodes (more than one allele, or not the reference), general path
        // (8 lanes per barcode here: few barcodes, short walks, and idle waves skip the phase)
        const int n_complex = SMC_ABLATE == 5 ? 0 : (int)H->misc[M_NCOMPLEX];   // 5: diagnostic, skips the general path
        const int grp1 = (BLOCK - 1 - tid) / Gc, ngrp1 = BLOCK / Gc;   // groups from the last thread down
        for (int w = grp1; w < n_complex; w += ngrp1) {
            const int u = (int)worklist[w];
            int b0, b1, nf, cnt_ref;
            unsigned long long mask;
            double rightP, prod_ref, prod_x;
            walk(u, b0, b1, nf, cnt_ref, mask, rightP, prod_ref, prod_x);
            const int n_exist = __popcll(mask);
            int npad = n_exist < 4 ? 4 - n_exist : 0;
            unsigned long long padmask = 0;
            for (int a = 0, k = 0; a < 4 && k < npad; ++a)
                if (!((mask >> a) & 1ull)) { padmask |= 1ull << a; ++k; }   // :49-54, atgc order
            const int nk = n_exist + npad;
            const double denom = nf + 0.5 * nk;                        // :80

            if (n_exist <= 4) {
                int ida[4] = {0, 0, 0, 0}, cnta[4] = {0, 0, 0, 0};
                double proda[4] = {1.0, 1.0, 1.0, 1.0};
                {
                    unsigned long long mm = mask;
#pragma unroll
                    for (int k = 0; k < 4; ++k)
                        if (k < n_exist) { ida[k] = __ffsll((long long)mm) - 1; mm &= mm - 1; }
                }
                if (refa < 64 && mask == (1ull << refa)) {
                    cnta[0] = cnt_ref; proda[0] = prod_ref;            // speculation hit: no second pass
                } else if (refa < 64 && n_exist == 2 && ((mask >> refa) & 1ull)) {
                    // the reference allele and one other (the usual queued barcode): the walk's two
                    // speculative products are exactly the two P(reads | allele), no second pass
                    const bool ref_first = ida[0] == refa;
                    cnta[0] = ref_first ? cnt_ref : nf - cnt_ref; cnta[1] = ref_first ? nf - cnt_ref : cnt_ref;
                    proda[0] = ref_first ? prod_ref : prod_x; proda[1] = ref_first ? prod_x : prod_ref;
                } else {
                    // pass B: per existing allele, count and P(reads | allele)  (:62-77)
                    for (int s = b0 + jc; s < b1; s += Gc) {
                        const uint32_t st = fmin[s];
                        if (st & ST_PRESENT) {
                            const int a = KEY_ALLELE(st);
                            const double p = prob_of(st);
#pragma unroll
                            for (int k = 0; k < 4; ++k)
                                if (k < n_exist) {
                                    const bool same = a == ida[k];
                                    cnta[k] += same;
                                    proda[k] *= same ? 1.0 - p : p;
                                }
                        }
                    }
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        cnta[k] = grp8_add(cnta[k]);
                        proda[k] = grp8_mul(proda[k]);
                    }
                }
                // PCR-error terms (:79-81); min over the other keys == value at their max count
                int max1 = -1, max2 = -1, arg1 = -1, arg2 = -1;
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if (k < n_exist) {
                        if (cnta[k] > max1) { max2 = max1; arg2 = arg1; max1 = cnta[k]; arg1 = k; }
                        else if (cnta[k] > max2) { max2 = cnta[k]; arg2 = k; }
                    }
                // The transcendental chains run lane-parallel: lane k (< n_exist) of the barcode's first quad owns
                // existing allele k, lane n_exist owns the zero-count PCR term and then the padded keys; values
                // are exchanged by quad broadcasts (DPP).  Every value is produced by the same instruction
                // sequence whichever lane computes it, so results do not depend on the lane assignment.
                const int myc = jc == 0 ? cnta[0] : jc == 1 ? cnta[1] : jc == 2 ? cnta[2] : jc == 3 ? cnta[3] : 0;
                const double mypcr = pcr_of(myc, denom);               // lanes >= n_exist: count 0
                double pcrv[4];
                pcrv[0] = dpp_f64<0x00>(mypcr); pcrv[1] = dpp_f64<0x55>(mypcr);
                pcrv[2] = dpp_f64<0xAA>(mypcr); pcrv[3] = dpp_f64<0xFF>(mypcr);
                double prodpcr = 1.0;
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if (k < n_exist) prodpcr *= pcrv[k];
                // the "other keys" term needs the PCR value at the largest count among the other keys:
                // that is one of the values just computed (equal counts give bit-equal values), or the
                // zero-count value (lane 1's) when the only other keys are padded ones
                const double pcr0 = n_exist == 1 ? pcrv[1] : 0.0;
                const double padOut = rightP * prodpcr;                // :88-91
                double mytmp = padOut;                                 // lane n_exist: the padded keys
                {
                    const int oi = (jc == arg1) ? arg2 : arg1;
                    double po = pcr0;
#pragma unroll
                    for (int m = 0; m < 4; ++m) po = (m == oi) ? pcrv[m] : po;
                    const double myprod = jc == 0 ? proda[0] : jc == 1 ? proda[1] : jc == 2 ? proda[2] : proda[3];
                    if (jc < n_exist) mytmp = pne * myprod + rightP * po;                 // :86
                }
                double tmpv[4];
                tmpv[0] = dpp_f64<0x00>(mytmp); tmpv[1] = dpp_f64<0x55>(mytmp);
                tmpv[2] = dpp_f64<0xAA>(mytmp); tmpv[3] = dpp_f64<0xFF>(mytmp);
                double sumP = 0.0;
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if (k < n_exist) sumP += tmpv[k];
                for (int k = 0; k < npad; ++k) sumP += padOut;
                // posteriors -> -log10(1-p) (:95-96, :508-510)
                const double mypost = sumP <= 0 ? 0.0 : mytmp / sumP;
                double mypred;
                if (jc >= n_exist && mypost < 1e-6) {
                    // padded keys: t is tiny, the series is exact to < 1e-19 below 1e-6
                    mypred = mypost * (1.0 + mypost * (0.5 + mypost * (1.0 / 3.0))) * 0.43429448190325182765;
                } else {
                    const double x = 1.0 - mypost;
                    mypred = x > 0.0 ? -log10(x) : 16.0;
                }
                double predv[4], mx = -1.0;
                predv[0] = dpp_f64<0x00>(mypred); predv[1] = dpp_f64<0x55>(mypred);
                predv[2] = dpp_f64<0xAA>(mypred); predv[3] = dpp_f64<0xFF>(mypred);
                double predpad = 0.0;
                if (npad) predpad = n_exist == 1 ? predv[1] : n_exist == 2 ? predv[2] : predv[3];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    if (k >= n_exist) predv[k] = 0.0;
                    else if (predv[k] > mx) mx = predv[k];
                }
                if (npad && predpad > mx) mx = predpad;
                // PI sums (:512): each owning lane adds its own value to its own accumulators (the lane
                // accumulators are summed over the wavefront at the end, integer adds: order-free)
                if (jc <= n_exist && jc < 4) {
                    const long long fx = to_fx(mypred, fxscale);
                    if (jc < n_exist) {
                        const int a = jc == 0 ? ida[0] : jc == 1 ? ida[1] : jc == 2 ? ida[2] : ida[3];
                        atomicAdd(&pifx[a], (unsigned long long)fx);
                    } else if (npad) {
                        for (int b = 0; b < 4; ++b)
                            if ((padmask >> b) & 1ull) atomicAdd(&pifx[b], (unsigned long long)fx);
                    }
                }
                if (jc == 0) {
                    int n_max = 0, cons = -1;
#pragma unroll
                    for (int k = 0; k < 4; ++k)
                        if (k < n_exist && predv[k] == mx) { ++n_max; cons = ida[k]; }  // :514
                    if (npad && predpad == mx) {
#pragma unroll
                        for (int a = 0; a < 4; ++a)
                            if ((padmask >> a) & 1ull) { ++n_max; cons = a; }
                    }
                    const unsigned long long uq = mask | padmask;
                    atomicOr(&H->misc[M_TOUCH_LO], (uint32_t)uq); atomicOr(&H->misc[M_TOUCH_HI], (uint32_t)(uq >> 32));
                    if (n_max == 1) {                                                    // :515-519
                        const bool str = mx > P.smt;
                        { atomicAdd(&mtc[cons], 1u); if (str) atomicAdd(&strong[cons], 1u); }
                    } else if (nf == 1) {                                                // :521-523
                        const int a = ida[0];
                        atomicAdd(&mtc[a], 1u);
                    }
                }
            } else {
                // > 4 distinct alleles inside one barcode: same arithmetic, recomputed per allele
                // instead of cached in registers.
                auto scan_allele = [&](int a, int& cnt, double& prod) {
                    cnt = 0; prod = 1.0;
                    for (int s = b0 + jc; s < b1; s += Gc) {
                        const uint32_t st = fmin[s];
                        if (st & ST_PRESENT) {
                            const double p = prob_of(st);
                            const bool same = (int)KEY_ALLELE(st) == a;
                            cnt += same;
                            prod *= same ? 1.0 - p : p;
                        }
                    }
                    cnt = grp8_add(cnt);
                    prod = grp8_mul(prod);
                };
                int max1 = -1, max2 = -1, arg1 = -1;
                for (unsigned long long mm = mask; mm; mm &= mm - 1) {
                    const int a = __ffsll((long long)mm) - 1;
                    int c; double pr;
                    scan_allele(a, c, pr);
                    if (c > max1) { max2 = max1; max1 = c; arg1 = a; }
                    else if (c > max2) max2 = c;
                }
                double sumP = 0.0;
                for (unsigned long long mm = mask; mm; mm &= mm - 1) {
                    const int a = __ffsll((long long)mm) - 1;
                    int c; double pr;
                    scan_allele(a, c, pr);
                    const int other = (a == arg1) ? max2 : max1;
                    sumP += pne * pr + rightP * pcr_of(other, denom);
                }
                double mx = -1.0;
                int n_max = 0, cons = -1;
                for (unsigned long long mm = mask; mm; mm &= mm - 1) {
                    const int a = __ffsll((long long)mm) - 1;
                    int c; double pr;
                    scan_allele(a, c, pr);
                    const int other = (a == arg1) ? max2 : max1;
                    const double t = pne * pr + rightP * pcr_of(other, denom);
                    const double post = sumP <= 0 ? 0.0 : t / sumP;
                    const double x = 1.0 - post;
                    const double pred = x > 0.0 ? -log10(x) : 16.0;
                    if (pred > mx) { mx = pred; n_max = 1; cons = a; }
                    else if (pred == mx) ++n_max;
                    if (jc == 0) {
                        const long long fx = to_fx(pred, fxscale);
                        atomicAdd(&pifx[a], (unsigned long long)fx);
                    }
                }
                if (jc == 0) {
                    atomicOr(&H->misc[M_TOUCH_LO], (uint32_t)mask); atomicOr(&H->misc[M_TOUCH_HI], (uint32_t)(mask >> 32));
                    if (n_max == 1) {
                        const bool str = mx > P.smt;
                        { atomicAdd(&mtc[cons], 1u); if (str) atomicAdd(&strong[cons], 1u); }
                    }
                }
            }
        }
        __syncthreads();
        STAMP(7);
        if (SMC_ABLATE == 4) return;

        // ---- E: ranking and candidates (:534-555), one thread; the row is staged over the (dead) LUT
        for (int i = tid; i < (int)(sizeof(smc_row) / 4); i += BLOCK) ((uint32_t*)rowst)[i] = 0u;
        __syncthreads();
        if (wid == 0) {                                                 // the first wavefront, every lane
            const int n_bc = (int)H->misc[M_NBC];
            finish_row(rowst, L, li, n, nF, n_bc < P.ds ? n_bc : P.ds /* usedMT, :489 */, n_bc > P.ds, fxscale, H->misc, tal,
                       pifx, mtc, strong, flt_list, lane, WAVE);
            if (sampled && (int)H->misc[M_NKEPT] != (n_bc < P.ds ? n_bc : P.ds)) rowst->status = SMC_ST_BAD_INPUT;
        }
        __syncthreads();
        const uint32_t* src = (const uint32_t*)rowst;
        uint32_t* dst = (uint32_t*)out;
        for (int i = tid; i < (int)(sizeof(smc_row) / 4); i += BLOCK) dst[i] = src[i];
        STAMP(9);
        STAMP_FLUSH();
    }
}

// ------------------------------------------------------------------------------------------
// kernel 1s: the sorted-stream locus kernel - one wavefront per locus, no atomics on the read path
// ------------------------------------------------------------------------------------------
// The batch contract sorts a locus's reads barcode-major (barcode, fragment slot, pileup order), so a
// barcode is one contiguous run of reads and the reads of a fragment are adjacent, first-seen mate
// first.  Lane groups of G lanes walk one barcode each, G reads per step: the mate merge
// (smCounter.py:468-479) is a compare with the previous lane, per-fragment terms accumulate in lane
// registers and are reduced across the group once per barcode; per-barcode records go to a 64-entry LDS
// buffer and the calProb arithmetic then runs one lane per barcode.  Barcodes holding an allele other
// than the reference are re-walked for per-allele products.  A fragment with >= 3 reads on the locus
// (re-created fragments need a sequential replay) or a barcode with > 4 alleles sends the whole locus
// to the table-based kernel k_call_loci instead (redo_flag); nothing is written for it here.
struct URec {
    double rightP, prod_ref;
    uint32_t mlo, mhi;
    int nf, cnt_ref;
    uint32_t rb, re;
    uint32_t in_bc, pad;
};
#define UB 64
#define PK_NONE 0xFFFFFFFFu
// frag plane words without their read-class bits (this kernel computes the predicates from the raw fields)
__device__ __forceinline__ uint4 slots_of(uint4 f) {
    return make_uint4(f.x & SMC_FRAG_SLOT_MASK, f.y & SMC_FRAG_SLOT_MASK, f.z & SMC_FRAG_SLOT_MASK, f.w & SMC_FRAG_SLOT_MASK);
}

__device__ __forceinline__ uint32_t lds_sorted_bytes(int a_cap) {
    return (uint32_t)(sizeof(Hdr) + a_cap * 64 + sizeof(smc_row) + 128 * 8 + UB * sizeof(URec) + UB * 4);
}

__global__ __launch_bounds__(WAVE) void k_call_sorted(
    KParams P, const smc_locus* __restrict__ loci, const int* __restrict__ order, int a_cap,
    const uint32_t* __restrict__ g_meta, const uint32_t* __restrict__ g_umi, const uint32_t* __restrict__ g_frag,
    const uint32_t* __restrict__ g_dist, const uint32_t* __restrict__ g_umi_start, const double* __restrict__ g_lut,
    smc_row* __restrict__ rows, uint32_t* __restrict__ flt_list, uint8_t* __restrict__ redo_flag) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x;
    const smc_locus L = loci[blockIdx.x];
    const int li = order[blockIdx.x];
    const int n = L.n_reads, nU = L.n_umi, nF = L.n_frag, nA = L.n_alleles;
    const uint32_t refa = L.ref_allele;
    const uint32_t* meta = g_meta + 4ll * L.read_off4;
    const uint32_t* umi = g_umi + 4ll * L.read_off4;
    const uint32_t* frag = g_frag + 4ll * L.read_off4;
    const uint32_t* dist = g_dist + 4ll * L.read_off4;
    const uint32_t* ustart = g_umi_start + L.umi_off;

    Hdr* H = (Hdr*)smem;
    uint32_t* tal = (uint32_t*)(smem + sizeof(Hdr));
    unsigned long long* pifx = (unsigned long long*)(tal + a_cap * SMC_NT);
    uint32_t* mtc = (uint32_t*)(pifx + a_cap);
    uint32_t* strong = mtc + a_cap;
    smc_row* rowst = (smc_row*)(strong + a_cap);
    double* lut = (double*)(rowst + 1);
    URec* urec = (URec*)(lut + LUT_N);
    uint32_t* clist = (uint32_t*)(urec + UB);
    {
        uint32_t* z = (uint32_t*)smem;
        const int nz = (int)((sizeof(Hdr) + a_cap * 64 + sizeof(smc_row)) / 4);
        for (int i = lane; i < nz; i += WAVE) z[i] = 0;
        for (int i = lane; i < LUT_N; i += WAVE) lut[i] = i == (int)PIDX_UNPAIRED ? 0.1 : g_lut[i];
    }
    __syncthreads();

    // lanes per barcode (a function of the locus only): each lane takes a quad of reads per step and a
    // barcode should take a handful of steps - few barcode starts, whose first loads are not prefetched
    int G = 1;
    while (G < 16 && (long long)n > 24ll * G * nU) G <<= 1;
    const int j = lane & (G - 1), gbase = lane - j, g = lane / G, ngrp = WAVE / G;
    int bits = 32 - __clz(nU);
    int shift = 58 - bits; if (shift > 48) shift = 48;
    const double fxscale = (double)(1ull << shift);
    const double pne = 1.0 - 3e-5;                                     // pcr_no_error, :20

    uint32_t accv[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) accv[k] = 0;
    uint32_t conc_ref = 0, disc_ref = 0;
    lmask err_m = 0, redo_m = 0;
    uint32_t n_frag_seen = 0, n_bc = 0;
    long long pi_acc[4] = {0, 0, 0, 0};
    int mt_acc[4] = {0, 0, 0, 0}, st_acc[4] = {0, 0, 0, 0};
    int c3 = 0, c5 = 0, c7 = 0, c10 = 0, ufrag = 0;
    uint32_t touch_lo = 0, touch_hi = 0;

    // One barcode: every lane of the group takes one aligned quad of 4 reads per step (one 16-byte load
    // per plane, next step prefetched in registers); reads outside [rb, re) belong to a neighbouring barcode
    // and are masked.  fn(present, allele, pidx) is called for every read slot; only the read closing a
    // fragment passes present = true.
    const uint4* meta4 = (const uint4*)meta;
    const uint4* umi4 = (const uint4*)umi;
    const uint4* frag4 = (const uint4*)frag;
    const uint4* dist4 = (const uint4*)dist;
    auto walk = [&](uint32_t u_expect, uint32_t rb, uint32_t re, bool tally, auto&& fn) -> bool {
        const uint32_t q0 = rb >> 2, q1 = (re + 3u) >> 2;
        const uint4 none4 = make_uint4(PK_NONE, PK_NONE, PK_NONE, PK_NONE), zero4 = make_uint4(0, 0, 0, 0);
        uint4 cm = zero4, cf = none4, cd = zero4, cu = zero4, nm, nf4, nd, nu;
        {
            const uint32_t q = q0 + j;
            if (q < q1) { cm = meta4[q]; cf = slots_of(frag4[q]); cd = dist4[q]; cu = umi4[q]; }
        }
        uint32_t carry = PK_NONE;
        bool any_inc = false;
        for (uint32_t qs = q0; qs < q1; qs += G) {
            const uint32_t q = qs + j;
            {
                const uint32_t qn = q + G;
                nm = zero4; nf4 = none4; nd = zero4; nu = zero4;
                if (qn < q1) { nm = meta4[qn]; nf4 = slots_of(frag4[qn]); nd = dist4[qn]; nu = umi4[qn]; }
            }
            const uint32_t ms[4] = {cm.x, cm.y, cm.z, cm.w}, fs[4] = {cf.x, cf.y, cf.z, cf.w};
            const uint32_t ds[4] = {cd.x, cd.y, cd.z, cd.w}, us[4] = {cu.x, cu.y, cu.z, cu.w};
            uint32_t pk[4], slot[4];
            lmask m_okk[4], m_inck[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const uint32_t r = 4u * q + k;
                const uint32_t mw = ms[k], f = fs[k], dw = ds[k];
                const uint32_t a = mw & 0xffu, kind = (mw >> 19) & 3u;
                const lmask m_valid = BAL(r >= rb) & BAL(r < re);
                const lmask m_ok = m_valid & BAL(f < (uint32_t)nF) & BAL(a < (uint32_t)nA) & BAL(us[k] == u_expect);
                err_m |= m_valid & ~m_ok;
                const lmask m_base = BAL(kind == SMC_KIND_BASE), m_gap = BAL(kind == SMC_KIND_GAP);
                const lmask m_qok = BAL((int)((mw >> 8) & 0xffu) >= P.min_bq);
                const lmask m_inc = m_ok & (m_qok | m_gap) & BAL((int)(mw >> 24) >= P.min_mq) & BAL((mw & 0x40000u) != 0u);
                m_okk[k] = m_ok; m_inck[k] = m_inc;
                if (tally) {
                    const lmask m_r2 = BAL((mw & 0x10000u) != 0u), m_rev = BAL((mw & 0x20000u) != 0u);
                    const lmask m_ref = m_ok & BAL(a == refa);
                    const lmask m_ib = m_inc & m_base;
                    const lmask m_r1i = m_ib & ~m_r2, m_r2i = m_ib & m_r2;
                    const lmask m_le20 = BAL((dw & 0xffffu) <= 20u), m_prle = BAL((int)(dw >> 16) <= P.primer_dist);
                    const lmask e_fwd = ~m_gap & ~m_rev, e_rev = ~m_gap & m_rev, e_lowq = m_base & ~m_qok;
                    ADDM(accv[SMC_T_CNT], m_ref);
                    ADDM(accv[SMC_T_FWD], m_ref & e_fwd);
                    ADDM(accv[SMC_T_REV], m_ref & e_rev);
                    ADDM(accv[SMC_T_LOWQ], m_ref & e_lowq);
                    ADDM(accv[SMC_T_R1N], m_ref & m_r1i);
                    ADDM(accv[SMC_T_R1LE], m_ref & m_r1i & m_le20);
                    ADDM(accv[SMC_T_R2N], m_ref & m_r2i);
                    ADDM(accv[SMC_T_R2BCLE], m_ref & m_r2i & m_le20);
                    ADDM(accv[SMC_T_R2PRLE], m_ref & m_r2i & m_prle);
                    const lmask nr = m_ok & ~m_ref;
                    if (nr) {                                           // stray alleles: predicated LDS adds
                        uint32_t* t = tal + a * SMC_NT;
                        if (LANES(nr)) atomicAdd(&t[SMC_T_CNT], 1u);
                        if (LANES(nr & e_fwd)) atomicAdd(&t[SMC_T_FWD], 1u);
                        if (LANES(nr & e_rev)) atomicAdd(&t[SMC_T_REV], 1u);
                        if (LANES(nr & e_lowq)) atomicAdd(&t[SMC_T_LOWQ], 1u);
                        if (LANES(nr & m_r1i)) atomicAdd(&t[SMC_T_R1N], 1u);
                        if (LANES(nr & m_r1i & m_le20)) atomicAdd(&t[SMC_T_R1LE], 1u);
                        if (LANES(nr & m_r2i)) atomicAdd(&t[SMC_T_R2N], 1u);
                        if (LANES(nr & m_r2i & m_le20)) atomicAdd(&t[SMC_T_R2BCLE], 1u);
                        if (LANES(nr & m_r2i & m_prle)) atomicAdd(&t[SMC_T_R2PRLE], 1u);
                    }
                }
                const bool inc = LANES(m_inc);
                uint32_t qv = LANES(m_gap) ? (uint32_t)P.min_bq : ((mw >> 8) & 0xffu);             // :418
                qv = qv < PIDX_UNPAIRED ? qv : PIDX_UNPAIRED - 1u;
                slot[k] = LANES(m_ok) ? f : PK_NONE;
                pk[k] = LANES(m_ok) ? ((f << 14) | (a << 8) | (qv << 1) | (uint32_t)inc) : PK_NONE;
                any_inc |= inc;
            }
            // neighbours across lanes: the previous lane's last read, the next lane's first read
            uint32_t prev0 = (uint32_t)__shfl_up((int)pk[3], 1, G);
            if (j == 0) prev0 = carry;
            uint32_t next3 = (uint32_t)__shfl_down((int)slot[0], 1, G);
            {
                // first read of the next step (lane 0 of the group), if it still belongs to this barcode
                const uint32_t nx = (uint32_t)__shfl((int)nf4.x, gbase);
                const uint32_t look = (4u * (qs + G) < re) ? nx : PK_NONE;
                if (j == G - 1) next3 = look;
            }
            carry = (uint32_t)__shfl((int)pk[3], gbase + G - 1);
            // ---- mate merge by neighbour compare (smCounter.py:468-479)
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const uint32_t prev = k ? pk[k - 1] : prev0;
                const uint32_t next_slot = k < 3 ? slot[k + 1] : next3;
                const uint32_t cur = pk[k], sl = slot[k];
                const uint32_t a = (cur >> 8) & 63u, qv = (cur >> 1) & 127u;
                const bool inc = (cur & 1u) != 0u && cur != PK_NONE;
                const bool has_prev = prev != PK_NONE;
                const uint32_t pslot = prev >> 14, pa = (prev >> 8) & 63u, pq = (prev >> 1) & 127u;
                const bool pinc = has_prev && (prev & 1u);
                const bool is_first = !(has_prev && pslot == sl), is_last = next_slot != sl;
                const lmask m_v = m_okk[k];
                redo_m |= m_v & BAL(!is_first) & BAL(!is_last);          // a run of >= 3 reads
                err_m |= m_v & BAL(has_prev && pslot > sl);              // not sorted
                const bool both = !is_first && pinc && inc;
                const bool same = a == pa || a == (uint32_t)N_ID;
                const bool present = is_first ? inc : (both ? same : (pinc || inc));
                const uint32_t fa = (!is_first && pinc) ? pa : a;
                const uint32_t fq = both ? (pq < qv ? pq : qv) : ((!is_first && pinc) ? pq : qv);
                const bool paired = both && same;
                const lmask m_close = m_v & BAL(is_last);
                if (tally) {
                    ADDM(n_frag_seen, m_close);          // per lane (groups diverge); summed at the end
                    const lmask m_conc = m_close & BAL(both && a == pa), m_disc = m_close & BAL(both && !same);
                    const lmask c_ref = m_conc & BAL(a == refa), d_ref = m_disc & BAL(a == refa);
                    ADDM(conc_ref, c_ref);
                    ADDM(disc_ref, d_ref);
                    const lmask rare = (m_conc & ~c_ref) | (m_disc & ~d_ref);
                    if (rare) {
                        if (LANES(m_conc & ~c_ref)) atomicAdd(&tal[a * SMC_NT + SMC_T_CONCORD], 1u);
                        if (LANES(m_disc & ~d_ref)) atomicAdd(&tal[a * SMC_NT + SMC_T_DISCORD], 1u);
                    }
                }
                fn(LANES(m_close) && present, fa, paired ? fq : PIDX_UNPAIRED);
            }
            cm = nm; cf = nf4; cd = nd; cu = nu;
        }
        return any_inc;
    };

    for (int ub0 = 0; ub0 < nU; ub0 += UB) {
        const int nb = nU - ub0 < UB ? nU - ub0 : UB;
        // ---- walk phase
        for (int t = g; t < nb; t += ngrp) {
            const int u = ub0 + t;
            const uint32_t rb = ustart[u] & ~SMC_USTART_DROPPED, re = ustart[u + 1] & ~SMC_USTART_DROPPED;
            if (!(rb < re && re <= (uint32_t)n) || (u == 0 && rb != 0)) { err_m |= 1; continue; }
            int nf = 0, cnt_ref = 0;
            unsigned long long mk = 0;
            double rp = 1.0, pr = 1.0;
            const bool any_inc = walk((uint32_t)u, rb, re, true, [&](bool present, uint32_t fa, uint32_t pidx) {
                const double p = lut[pidx & (LUT_N - 1)], q1 = 1.0 - p;
                const bool isref = fa == refa;
                nf += present;
                cnt_ref += present && isref;
                mk |= present ? (1ull << fa) : 0ull;
                rp *= present ? q1 : 1.0;
                pr *= present ? (isref ? q1 : p) : 1.0;
            });
            nf = wave_reduce_add(nf, G);
            cnt_ref = wave_reduce_add(cnt_ref, G);
            const uint32_t mlo = wave_reduce_or((uint32_t)mk, G), mhi = wave_reduce_or((uint32_t)(mk >> 32), G);
            const uint32_t inb = wave_reduce_or((uint32_t)any_inc, G);
            rp = wave_reduce_mul(rp, G);
            pr = wave_reduce_mul(pr, G);
            if (j == 0) {
                URec& R = urec[t];
                R.rightP = rp; R.prod_ref = pr; R.mlo = mlo; R.mhi = mhi; R.nf = nf; R.cnt_ref = cnt_ref;
                R.rb = rb; R.re = re; R.in_bc = inb;
            }
        }
        __syncthreads();
        // ---- math phase: one lane per barcode of the batch (calProb :26-98, PI / consensus :506-532)
        bool complex = false;
        {
            URec R;
            R.in_bc = 0; R.nf = 0; R.mlo = R.mhi = 0; R.cnt_ref = 0; R.rightP = R.prod_ref = 1.0;
            if (lane < nb) R = urec[lane];
            const lmask m_bc = BAL(lane < nb && R.in_bc != 0u);
            // down-sampling stand-in (non-parity, :496-498): keep the ds lowest barcode ids of bcDict
            const int rank = (int)n_bc + __builtin_amdgcn_mbcnt_hi((uint32_t)(m_bc >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m_bc, 0u));
            n_bc += (uint32_t)__popcll(m_bc);
            const bool kept = LANES(m_bc) && rank < P.ds;
            if (lane < nb && R.in_bc) urec[lane].in_bc = kept ? 1u : 0u;
            if (kept) {
                const int nf = R.nf;
                const unsigned long long mask = ((unsigned long long)R.mhi << 32) | R.mlo;
                ufrag += nf; c3 += nf >= 3; c5 += nf >= 5; c7 += nf >= 7; c10 += nf >= 10;
                if (nf <= P.mt_drop) {                                 // :28-32 -> all four posteriors 0
                    touch_lo |= 0xFu;
                    if (nf == 1) {                                     // :521-523
                        const int a = __ffsll((long long)mask) - 1;
                        if (a < 4) { for (int k = 0; k < 4; ++k) if (k == a) mt_acc[k]++; } else atomicAdd(&mtc[a], 1u);
                    }
                } else if (!(refa < 64u && mask == (1ull << refa))) {
                    complex = true;
                } else {
                    // one existing allele (the reference), three padded keys (:49-54): nk = 4
                    const unsigned long long padmask = refa < 4 ? (0xFull & ~(1ull << refa)) : 0x7ull;
                    const double denom = nf + 2.0;                     // :80
                    const double pcr_self = pcr_of(nf, denom), pcr_zero = pcr_of(0, denom);
                    const double tmp0 = pne * R.prod_ref + R.rightP * pcr_zero;    // :86
                    const double padOut = R.rightP * pcr_self;         // :88-91
                    double sumP = tmp0;
                    sumP += padOut; sumP += padOut; sumP += padOut;
                    const double post0 = sumP <= 0 ? 0.0 : tmp0 / sumP, postp = sumP <= 0 ? 0.0 : padOut / sumP;
                    const double x0 = 1.0 - post0;
                    const double pred0 = x0 > 0.0 ? -log10(x0) : 16.0; // :508-510
                    double predpad;
                    if (postp < 1e-6) predpad = postp * (1.0 + postp * (0.5 + postp * (1.0 / 3.0))) * 0.43429448190325182765;
                    else { const double xp = 1.0 - postp; predpad = xp > 0.0 ? -log10(xp) : 16.0; }
                    const long long fx0 = to_fx(pred0, fxscale), fxp = to_fx(predpad, fxscale);
                    if (refa >= 4) atomicAdd(&pifx[refa], (unsigned long long)fx0);
#pragma unroll
                    for (int a = 0; a < 4; ++a) {
                        if (a == (int)refa) pi_acc[a] += fx0;
                        else if ((padmask >> a) & 1ull) pi_acc[a] += fxp;
                    }
                    const unsigned long long uq = mask | padmask;
                    touch_lo |= (uint32_t)uq; touch_hi |= (uint32_t)(uq >> 32);
                    if (pred0 > predpad) {                             // unique maximum (:514-519)
                        const bool str = pred0 > P.smt;
                        if (refa < 4) {
#pragma unroll
                            for (int a = 0; a < 4; ++a) if (a == (int)refa) { mt_acc[a]++; st_acc[a] += str; }
                        } else { atomicAdd(&mtc[refa], 1u); if (str) atomicAdd(&strong[refa], 1u); }
                    } else if (nf == 1) {                              // :521-523
                        if (refa < 4) {
#pragma unroll
                            for (int a = 0; a < 4; ++a) if (a == (int)refa) mt_acc[a]++;
                        } else atomicAdd(&mtc[refa], 1u);
                    }
                }
            }
        }
        // ---- barcodes with another allele: queue, then re-walk for per-allele counts and products
        const lmask m_cx = BAL(complex);
        const int n_complex = __popcll(m_cx);
        if (complex) clist[__builtin_amdgcn_mbcnt_hi((uint32_t)(m_cx >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m_cx, 0u))] = (uint32_t)lane;
        __syncthreads();
        for (int w = g; w < n_complex; w += ngrp) {
            const URec R = urec[clist[w]];
            const int nf = R.nf;
            const unsigned long long mask = ((unsigned long long)R.mhi << 32) | R.mlo;
            const int n_exist = __popcll(mask);
            if (n_exist > 4) { redo_m |= 1; continue; }              // rare: the table kernel handles it
            int npad = 4 - n_exist;
            unsigned long long padmask = 0;
            for (int a = 0, k = 0; a < 4 && k < npad; ++a)
                if (!((mask >> a) & 1ull)) { padmask |= 1ull << a; ++k; }   // :49-54, atgc order
            const int nk = n_exist + npad;
            const double denom = nf + 0.5 * nk;                        // :80
            int ida[4] = {0, 0, 0, 0}, cnta[4] = {0, 0, 0, 0};
            double proda[4] = {1.0, 1.0, 1.0, 1.0};
            {
                unsigned long long mm = mask;
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if (k < n_exist) { ida[k] = __ffsll((long long)mm) - 1; mm &= mm - 1; }
            }
            (void)walk((uint32_t)(ub0 + (int)clist[w]), R.rb, R.re, false, [&](bool present, uint32_t fa, uint32_t pidx) {     // :62-77
                const double p = lut[pidx & (LUT_N - 1)];
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if (k < n_exist) {
                        const bool sm = (int)fa == ida[k];
                        cnta[k] += present && sm;
                        proda[k] *= present ? (sm ? 1.0 - p : p) : 1.0;
                    }
            });
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                cnta[k] = wave_reduce_add(cnta[k], G);
                proda[k] = wave_reduce_mul(proda[k], G);
            }
            int max1 = -1, max2 = -1, arg1 = -1, arg2 = -1;
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (k < n_exist) {
                    if (cnta[k] > max1) { max2 = max1; arg2 = arg1; max1 = cnta[k]; arg1 = k; }
                    else if (cnta[k] > max2) { max2 = cnta[k]; arg2 = k; }
                }
            double prodpcr = 1.0, tmpv[4] = {0, 0, 0, 0}, pcrv[4] = {0, 0, 0, 0}, sumP = 0.0;
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (k < n_exist) { pcrv[k] = pcr_of(cnta[k], denom); prodpcr *= pcrv[k]; }
            const double pcr0 = n_exist == 1 ? pcr_of(0, denom) : 0.0;
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (k < n_exist) {
                    const int oi = (k == arg1) ? arg2 : arg1;
                    double po = pcr0;
#pragma unroll
                    for (int m = 0; m < 4; ++m) po = (m == oi) ? pcrv[m] : po;
                    tmpv[k] = pne * proda[k] + R.rightP * po;                       // :86
                    sumP += tmpv[k];
                }
            const double padOut = R.rightP * prodpcr;                  // :88-91
            for (int k = 0; k < npad; ++k) sumP += padOut;
            double predv[4] = {0, 0, 0, 0}, mx = -1.0;
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (k < n_exist) {
                    const double post = sumP <= 0 ? 0.0 : tmpv[k] / sumP;
                    const double x = 1.0 - post;
                    predv[k] = x > 0.0 ? -log10(x) : 16.0;
                    if (predv[k] > mx) mx = predv[k];
                }
            double predpad = 0.0;
            if (npad) {
                const double post = sumP <= 0 ? 0.0 : padOut / sumP;
                if (post < 1e-6) predpad = post * (1.0 + post * (0.5 + post * (1.0 / 3.0))) * 0.43429448190325182765;
                else { const double x = 1.0 - post; predpad = x > 0.0 ? -log10(x) : 16.0; }
                if (predpad > mx) mx = predpad;
            }
            if (j == 0) {
                int n_max = 0, cons = -1;
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if (k < n_exist) {
                        const int a = ida[k];
                        const long long fx = to_fx(predv[k], fxscale);
                        if (a < 4) { for (int m = 0; m < 4; ++m) if (m == a) pi_acc[m] += fx; }
                        else atomicAdd(&pifx[a], (unsigned long long)fx);
                        if (predv[k] == mx) { ++n_max; cons = a; }                 // :514
                    }
                if (npad) {
                    const long long fx = to_fx(predpad, fxscale);
#pragma unroll
                    for (int a = 0; a < 4; ++a)
                        if ((padmask >> a) & 1ull) {
                            pi_acc[a] += fx;
                            if (predpad == mx) { ++n_max; cons = a; }
                        }
                }
                const unsigned long long uq = mask | padmask;
                touch_lo |= (uint32_t)uq; touch_hi |= (uint32_t)(uq >> 32);
                if (n_max == 1) {                                                    // :515-519
                    const bool str = mx > P.smt;
                    if (cons < 4) { for (int m = 0; m < 4; ++m) if (m == cons) { mt_acc[m]++; st_acc[m] += str; } }
                    else { atomicAdd(&mtc[cons], 1u); if (str) atomicAdd(&strong[cons], 1u); }
                } else if (nf == 1) {                                                // :521-523
                    const int a = ida[0];
                    if (a < 4) { for (int m = 0; m < 4; ++m) if (m == a) mt_acc[m]++; } else atomicAdd(&mtc[a], 1u);
                }
            }
        }
        __syncthreads();
    }

    // ---- anything this kernel does not handle exactly goes to the table kernel
    if (BAL(redo_m != 0) || (L.flags & SMC_LF_SAMPLED)) {              // (host-sampled loci: table kernel)
        if (lane == 0) redo_flag[li] = 1;
        return;
    }
    // ---- fold lane accumulators into the LDS image the row is built from
#pragma unroll
    for (int k = 0; k < 9; ++k) {
        const uint32_t v = (uint32_t)wave_reduce_add((int)accv[k], WAVE);
        if (lane == 0 && v && refa < (uint32_t)nA) tal[refa * SMC_NT + k] += v;
    }
    {
        const uint32_t c = (uint32_t)wave_reduce_add((int)conc_ref, WAVE), d = (uint32_t)wave_reduce_add((int)disc_ref, WAVE);
        if (lane == 0 && refa < (uint32_t)nA) { tal[refa * SMC_NT + SMC_T_CONCORD] += c; tal[refa * SMC_NT + SMC_T_DISCORD] += d; }
    }
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        const long long p = wave_reduce_add64(pi_acc[a], WAVE);
        const int m = wave_reduce_add(mt_acc[a], WAVE), s = wave_reduce_add(st_acc[a], WAVE);
        if (lane == 0) { pifx[a] += (unsigned long long)p; mtc[a] += (uint32_t)m; strong[a] += (uint32_t)s; }
    }
    c3 = wave_reduce_add(c3, WAVE); c5 = wave_reduce_add(c5, WAVE);
    c7 = wave_reduce_add(c7, WAVE); c10 = wave_reduce_add(c10, WAVE);
    ufrag = wave_reduce_add(ufrag, WAVE);
    touch_lo = wave_reduce_or(touch_lo, WAVE); touch_hi = wave_reduce_or(touch_hi, WAVE);
    const bool bad = BAL(err_m != 0) != 0 || (uint32_t)wave_reduce_add((int)n_frag_seen, WAVE) != (uint32_t)nF;
    const int used = (int)n_bc < P.ds ? (int)n_bc : P.ds;             // smCounter.py:489
    if (lane == 0) {
        H->misc[M_MT3] = c3; H->misc[M_MT5] = c5; H->misc[M_MT7] = c7; H->misc[M_MT10] = c10;
        H->misc[M_USEDFRAG] = ufrag; H->misc[M_TOUCH_LO] = touch_lo; H->misc[M_TOUCH_HI] = touch_hi;
        H->misc[M_ALLMT] = nU;
        smc_row* R = rowst;
        if (bad || used == 0) {
            R->status = bad ? SMC_ST_BAD_INPUT : SMC_ST_ZERO_COVERAGE;    // :492-494
            R->cvg = n; R->all_mt = nU; R->all_frag = nF;
            R->max_allele = R->second_allele = -1;
            for (int k = 0; k < 4; ++k) R->dp[k] = tal[k * SMC_NT + SMC_T_CNT];
            for (int c = 0; c < 2; ++c) {
                R->cand[c].allele = -1;
                R->cand[c].p_sb = R->cand[c].p_r1 = R->cand[c].p_r2 = R->cand[c].p_pr = NAN;
            }
        } else {
            finish_row(R, L, li, n, nF, used, (int)n_bc > P.ds, fxscale, H->misc, tal, pifx, mtc, strong, flt_list, 0, 1);
        }
    }
    __syncthreads();
    const uint32_t* src = (const uint32_t*)rowst;
    uint32_t* dst = (uint32_t*)(rows + li);
    for (int i = lane; i < (int)(sizeof(smc_row) / 4); i += WAVE) dst[i] = src[i];
}

// ------------------------------------------------------------------------------------------
// kernel 2: filterVariants (smCounter.py:182-269), one wave per locus, only where it applies
// ------------------------------------------------------------------------------------------
// log(n!) for an integer-valued n >= 0: Stirling's series on x = n + 1 >= 9 (truncation < 1e-12), exact constants
// below (one log either way, unlike the general lgamma of the device library, which is several hundred instructions)
__device__ __forceinline__ double d_lfact(double n) {
    const double x = n + 1.0;
    const double r = 1.0 / x, r2 = r * r;
    const double series = r * (8.33333333333333333e-2 + r2 * (-2.77777777777777778e-3 + r2 * (7.93650793650793651e-4 +
                          r2 * (-5.95238095238095238e-4 + r2 * 8.41750841750841751e-4))));
    const double st = (x - 0.5) * log(x) - x + 0.918938533204672742 + series;
    // n = 0..7: log of 1, 1, 2, 6, 24, 120, 720, 5040
    double small = 0.0;
    small = n == 2.0 ? 0.693147180559945309 : small;
    small = n == 3.0 ? 1.79175946922805500 : small;
    small = n == 4.0 ? 3.17805383034794562 : small;
    small = n == 5.0 ? 4.78749174278204599 : small;
    small = n == 6.0 ? 6.57925121201010100 : small;
    small = n == 7.0 ? 8.52516136106541430 : small;
    return n < 8.0 ? small : st;
}

// scipy.stats.fisher_exact(table) two-sided, evaluated by one wavefront: the support is cut into
// 64 contiguous chunks, each lane anchors its chunk with one log-factorial-based pmf and walks it with the
// exact ratio pmf(k+1)/pmf(k) = (n1-k)(n-k) / ((k+1)(n2-n+k+1)).  The nine log-factorials every lane needs
// (margins, the observed table) are evaluated once, one per lane, and broadcast.
__device__ void wave_fisher(long long a, long long b, long long c, long long d, double* orat, double* pval) {
    const int lane = threadIdx.x & 63;
    if (a + b == 0 || c + d == 0 || a + c == 0 || b + d == 0) { *orat = NAN; *pval = 1.0; return; }
    *orat = (c > 0 && b > 0) ? ((double)(a * d)) / ((double)(c * b)) : INFINITY;
    const long long n1 = a + b, n2 = c + d, n = a + c;
    const long long lo = n - n2 > 0 ? n - n2 : 0, hi = n < n1 ? n : n1;
    // uniform terms: lane i evaluates argument i
    const long long args[9] = {n1, n2, n1 + n2, n, n1 + n2 - n, a, n1 - a, n - a, n2 - n + a};
    long long mine = 0;
#pragma unroll
    for (int i = 0; i < 9; ++i) mine = lane == i ? args[i] : mine;
    const double lf_mine = d_lfact((double)mine);
    double u[9];
#pragma unroll
    for (int i = 0; i < 9; ++i)
        u[i] = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(lf_mine), i), __builtin_amdgcn_readlane(__double2loint(lf_mine), i));
    const double lden = u[2] - u[3] - u[4];                              // log C(n1 + n2, n)
    const double lnum0 = u[0] + u[1] - lden;                             // log(n1!) + log(n2!) - log C(n1 + n2, n)
    const double pexact = exp(lnum0 - u[5] - u[6] - u[7] - u[8]);
    const double thr = pexact * (1.0 + 1e-7);
    const long long len = hi - lo + 1, chunk = (len + 63) / 64;
    const long long k0 = lo + chunk * lane, k1 = (k0 + chunk - 1 < hi) ? k0 + chunk - 1 : hi;
    double p = 0.0;
    if (k0 <= hi) {
        double pk = exp(lnum0 - d_lfact((double)k0) - d_lfact((double)(n1 - k0)) - d_lfact((double)(n - k0)) -
                        d_lfact((double)(n2 - n + k0)));
        for (long long k = k0;; ++k) {
            if (pk <= thr) p += pk;
            if (k == k1) break;
            pk *= ((double)(n1 - k) * (double)(n - k)) / ((double)(k + 1) * (double)(n2 - n + k + 1));
        }
    }
    p += dpp_f64<DPP_XOR1>(p); p += dpp_f64<DPP_XOR2>(p); p += dpp_f64<DPP_HALF_MIRROR>(p); p += dpp_f64<DPP_MIRROR>(p);
    p = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(p), 0), __builtin_amdgcn_readlane(__double2loint(p), 0)) +
        __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(p), 16), __builtin_amdgcn_readlane(__double2loint(p), 16)) +
        __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(p), 32), __builtin_amdgcn_readlane(__double2loint(p), 32)) +
        __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(p), 48), __builtin_amdgcn_readlane(__double2loint(p), 48));
    *pval = p < 1.0 ? p : 1.0;
}

__global__ __launch_bounds__(WAVE) void k_filter_loci(KParams P, const smc_locus* __restrict__ loci, smc_row* __restrict__ rows,
                                                      const uint32_t* __restrict__ flt_list) {
    const uint32_t n_work = flt_list[0];
    const int lane = threadIdx.x;
    for (uint32_t w = blockIdx.x; w < n_work; w += gridDim.x) {
    const int li = (int)flt_list[1 + w];
    smc_row* R = rows + li;
    if ((R->status & 0xff) != SMC_ST_OK) continue;
    const smc_locus L = loci[li];
    for (int ci = 0; ci < 2; ++ci) {
        smc_cand* C = &R->cand[ci];
        if (C->allele < 0 || !C->flt_applied) continue;        // wave-uniform
        const int alt = C->allele;
        const bool snp = (L.snp_mask >> alt) & 1ull;
        const int* ta = C->tal;
        const int* tr = R->ref_tal;
        uint32_t f = 0;
        if (R->used_mt < 5) f |= SMC_F_LM;                                        // :187
        if (C->vsm < 2) f |= SMC_F_LSM;                                           // :191
        const int vmf = (1.0 * C->vmt / R->used_mt < 0.99);                       // :198,:202
        const double af_alt = 100.0 * ta[SMC_T_CNT] / R->cvg;                     // :206
        const int pairs = ta[SMC_T_DISCORD] + ta[SMC_T_CONCORD];                  // :207
        double p_sb = NAN, p_r1 = NAN, p_r2 = NAN, p_pr = NAN, orat, p;
        if (pairs >= 1000 && 1.0 * ta[SMC_T_DISCORD] / pairs >= 0.5) {
            f |= SMC_F_DP;                                                        // :208-209
        } else if (af_alt <= 60.0) {
            wave_fisher(tr[SMC_T_REV], tr[SMC_T_FWD], ta[SMC_T_REV], ta[SMC_T_FWD], &orat, &p);   // :211-215
            p_sb = p;
            if (p < 0.00001 && (orat >= 50 || orat <= 1.0 / 50)) f |= SMC_F_SB;
        }
        double bq_alt = 0.0;                                                      // :222-227
        if (snp && ta[SMC_T_LOWQ] > 0) bq_alt = 1.0 * ta[SMC_T_LOWQ] / ta[SMC_T_CNT];
        if (bq_alt > 0.4) f |= SMC_F_LOWQ;
        if (snp) {                                                                // :230-266
            wave_fisher(tr[SMC_T_R1LE], tr[SMC_T_R1N] - tr[SMC_T_R1LE], ta[SMC_T_R1LE], ta[SMC_T_R1N] - ta[SMC_T_R1LE], &orat, &p);
            p_r1 = p;
            if (p < 0.001 && orat < 0.05 && af_alt <= 60.0) f |= SMC_F_R1CP;
            wave_fisher(tr[SMC_T_R2BCLE], tr[SMC_T_R2N] - tr[SMC_T_R2BCLE], ta[SMC_T_R2BCLE], ta[SMC_T_R2N] - ta[SMC_T_R2BCLE], &orat, &p);
            p_r2 = p;
            if (p < 0.001 && orat < 0.05 && af_alt <= 60.0) f |= SMC_F_R2CP;
            const int alt_le = ta[SMC_T_R2PRLE], alt_gt = ta[SMC_T_R2N] - ta[SMC_T_R2PRLE];
            wave_fisher(tr[SMC_T_R2PRLE], tr[SMC_T_R2N] - tr[SMC_T_R2PRLE], alt_le, alt_gt, &orat, &p);
            p_pr = p;
            if (alt_le + alt_gt > 0)
                if (1.0 * alt_le / (alt_le + alt_gt) >= 0.98 || (p < 0.001 && orat < 1.0 / 20)) f |= SMC_F_PRIMERCP;
        }
        if (lane == 0) {
            C->flt = f; C->vmf_lt_099 = vmf;
            C->p_sb = p_sb; C->p_r1 = p_r1; C->p_r2 = p_r2; C->p_pr = p_pr;
        }
    }
    }
}

// ------------------------------------------------------------------------------------------
// host side: C ABI
// ------------------------------------------------------------------------------------------
static thread_local std::string g_err;
static int fail(int code, const std::string& msg) { g_err = msg; return code; }
#define HIPCHK(x)                                                                                   \
    do {                                                                                            \
        hipError_t e_ = (x);                                                                        \
        if (e_ != hipSuccess)                                                                       \
            return fail(SMC_E_HIP, std::string(#x) + ": " + hipGetErrorString(e_));                 \
    } while (0)

struct smc_ctx {
    int device;
    double* lut;   // 10^(-q/10), q = 0..255
    double* simple; // [SMC_SIMPLE_N][2]: -log10(1 - posterior) of a one-allele barcode by fragment count
    int max_lds;   // bytes of LDS a workgroup may use
};

static size_t host_hdr_bytes(int a_cap) {
    return sizeof(Hdr) + (size_t)a_cap * SMC_NT * 4 + (size_t)a_cap * 8 + (size_t)a_cap * 4 + (size_t)a_cap * 4 + 128 * 8 + 32 * 8;
}
static size_t table_bytes(const smc_locus& L) {
    size_t b = 4 * ((size_t)L.n_umi + 1) + 4 * (size_t)L.n_frag + 9 * (size_t)L.n_umi;
    b = (b + 7) & ~(size_t)7;
    b += 16 * (((size_t)L.n_frag + 63) / 64);             // chunk masks (live, live & reference allele)
    return (b + 15) & ~(size_t)15;
}

#ifndef SMC_CLS1_BLOCK
#define SMC_CLS1_BLOCK 128
#endif

struct Bin {
    int cls;              // 0: 64 thr, 1: 256, 2: 512, 3: 1024 (LDS tables), 4: 1024 (global tables)
    int a_cap;
    size_t lds_bytes;
    std::vector<int> order;
    int* d_order = nullptr;
    smc_locus* d_loci = nullptr;   // descriptors in launch order
    int64_t* d_scratch_off = nullptr;
    uint8_t* d_scratch = nullptr;
    size_t scratch_bytes = 0;
};

struct smc_plan {
    smc_ctx* ctx;
    int64_t n_loci;
    smc_locus* d_loci = nullptr;
    uint32_t* d_flt_list = nullptr;   // [0] = count, then locus indices queued for k_filter_loci
    // sorted-stream path: every locus in one launch (heaviest first), hand-over flags for the table kernel
    int use_sorted = 1;
    int* d_all_order = nullptr;
    smc_locus* d_all_loci = nullptr;
    uint8_t* d_redo = nullptr;
    int a_cap_all = 8;
    int64_t total_reads = 0;
    std::vector<Bin> bins;
    // optional HIP-event timing of the dominant k_call_loci launch (the bin with most reads)
    int timing = 0, dom_bin = -1;
    int64_t dom_loci = 0, dom_reads = 0;
    std::vector<hipEvent_t> ev0, ev1;   // ring of event pairs, one per timed run
    int64_t n_timed = 0;
};

template <int BLOCK, bool GT>
static hipError_t launch_bin(const Bin& b, const KParams& kp, const smc_plan* p, const uint32_t* meta, const uint32_t* umi_start,
                             const uint32_t* frag, const uint32_t* dist, smc_row* rows, hipStream_t st) {
    auto kern = k_call_loci<BLOCK, GT>;
    if (b.lds_bytes > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)b.lds_bytes);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)b.order.size()), dim3(BLOCK), b.lds_bytes, st, kp, b.d_loci, b.d_order, b.a_cap,
                       meta, umi_start, frag, dist, p->ctx->lut, p->ctx->simple, rows, b.d_scratch, b.d_scratch_off, p->d_flt_list,
                       p->use_sorted ? p->d_redo : (const uint8_t*)nullptr);
    return hipGetLastError();
}

extern "C" {

int smc_abi_version(void) { return SMC_ABI_VERSION; }
const char* smc_last_error(void) { return g_err.c_str(); }
int smc_row_size(void) { return (int)sizeof(smc_row); }
int smc_locus_size(void) { return (int)sizeof(smc_locus); }

int smc_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

// read class -> what the read adds to its allele's tallies (include/smcounter_hip.h: smc_read_class).  out[2c], out[2c+1]:
// nine 5-bit increments in SMC_T_* order (six in the first word, three in the second), second word bit 31 = incCond.
void smc_class_table(uint32_t* out /* [64] */) {
    uint32_t cls[32][2];
    memset(cls, 0, sizeof cls);
    for (int kind = 0; kind < 4; ++kind)
        for (int bits = 0; bits < 64; ++bits) {
            const int rev = bits & 1, r2 = (bits >> 1) & 1, inc = (bits >> 2) & 1, bq_ok = (bits >> 3) & 1,
                      le20 = (bits >> 4) & 1, prle = (bits >> 5) & 1;
            if (kind == SMC_KIND_BASE && inc && !bq_ok) continue;           // included implies bq >= minBQ
            const uint32_t c_ = smc_read_class(kind, rev, r2, inc, bq_ok, le20, prle);
            uint32_t f[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
            f[SMC_T_CNT] = 1;                                                // :379,401,459
            if (kind != SMC_KIND_GAP) f[rev ? SMC_T_REV : SMC_T_FWD] = 1;    // :386-389,408-411,454-457
            if (kind == SMC_KIND_BASE) {
                if (!bq_ok) f[SMC_T_LOWQ] = 1;                               // :428-429
                if (inc && !r2) { f[SMC_T_R1N] = 1; f[SMC_T_R1LE] = (uint32_t)le20; }                  // :432-440
                if (inc && r2) { f[SMC_T_R2N] = 1; f[SMC_T_R2BCLE] = (uint32_t)le20; f[SMC_T_R2PRLE] = (uint32_t)prle; }   // :441-452
            }
            uint32_t lo = 0, hi = 0;
            for (int t = 0; t < 6; ++t) lo |= f[t] << (5 * t);
            for (int t = 6; t < 9; ++t) hi |= f[t] << (5 * (t - 6));
            hi |= inc ? CLS_INC : 0u;
            cls[c_][0] = lo; cls[c_][1] = hi;
        }
    memcpy(out, cls, sizeof cls);
}

int smc_create(int device, smc_ctx** out) {
    if (!out) return fail(SMC_E_ARG, "smc_create: out is NULL");
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return fail(SMC_E_NOGPU, "no HIP device visible");
    if (device < 0 || device >= n) return fail(SMC_E_ARG, "smc_create: device index out of range");
    HIPCHK(hipSetDevice(device));
    hipDeviceProp_t prop;
    HIPCHK(hipGetDeviceProperties(&prop, device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(SMC_E_NOGPU, std::string("device is ") + prop.gcnArchName + ", this library is built for gfx950 only");
    smc_ctx* c = new smc_ctx();
    c->device = device;
    c->max_lds = 160 * 1024;
    double h[256 + 32];
    for (int q = 0; q < 256; ++q) h[q] = pow(10.0, -q / 10.0);   // smCounter.py:469
    smc_class_table((uint32_t*)&h[256]);
    HIPCHK(hipMalloc(&c->lut, sizeof h));
    HIPCHK(hipMemcpy(c->lut, h, sizeof h, hipMemcpyHostToDevice));
    HIPCHK(hipMalloc(&c->simple, sizeof(double) * 2 * SMC_SIMPLE_N));
    hipLaunchKernelGGL(k_simple_table, dim3(SMC_SIMPLE_N / 256), dim3(256), 0, 0, c->simple, SMC_SIMPLE_N);
    HIPCHK(hipGetLastError());
    HIPCHK(hipDeviceSynchronize());
    *out = c;
    return SMC_OK;
}

void smc_destroy(smc_ctx* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    (void)hipFree(c->lut);
    (void)hipFree(c->simple);
    delete c;
}

void smc_plan_destroy(smc_plan* p) {
    if (!p) return;
    (void)hipSetDevice(p->ctx->device);
    (void)hipFree(p->d_loci);
    (void)hipFree(p->d_flt_list);
    (void)hipFree(p->d_all_order);
    (void)hipFree(p->d_all_loci);
    (void)hipFree(p->d_redo);
    for (auto e : p->ev0) (void)hipEventDestroy(e);
    for (auto e : p->ev1) (void)hipEventDestroy(e);
    for (auto& b : p->bins) {
        (void)hipFree(b.d_order);
        (void)hipFree(b.d_loci);
        (void)hipFree(b.d_scratch_off);
        (void)hipFree(b.d_scratch);
    }
    delete p;
}

int smc_plan_create(smc_ctx* ctx, const smc_locus* loci, int64_t n_loci, smc_plan** out) {
    if (!ctx || !out || (n_loci > 0 && !loci)) return fail(SMC_E_ARG, "smc_plan_create: NULL argument");
    if (n_loci > 0x7fffffff) return fail(SMC_E_ARG, "smc_plan_create: more than 2^31-1 loci in one batch");
    HIPCHK(hipSetDevice(ctx->device));
    smc_plan* p = new smc_plan();
    p->ctx = ctx;
    p->n_loci = n_loci;
    static const size_t cls_cap[4] = {8 * 1024, 24 * 1024, 64 * 1024, 160 * 1024};
    std::vector<Bin> bins(5);
    for (int c = 0; c < 5; ++c) { bins[c].cls = c; bins[c].a_cap = 8; bins[c].lds_bytes = 0; }
    for (int64_t l = 0; l < n_loci; ++l) {
        const smc_locus& L = loci[l];
        if (L.n_alleles > SMC_MAX_ALLELES || L.n_reads < 0 || L.n_umi < 0 || L.n_frag < 0 || L.n_reads >= (1 << 18)) {
            delete p;
            return fail(SMC_E_INPUT, "smc_plan_create: locus " + std::to_string(l) +
                                         " violates the layout contract (alleles<=64, reads<2^18)");
        }
        const int a_cap = (L.n_alleles + 7) & ~7;
        const size_t need = host_hdr_bytes(a_cap < 8 ? 8 : a_cap) + table_bytes(L);
        int c = 4;
        for (int k = 0; k < 4; ++k)
            if (need <= cls_cap[k]) { c = k; break; }
        Bin& b = bins[c];
        b.order.push_back((int)l);
        if (a_cap > b.a_cap) b.a_cap = a_cap;
    }
    HIPCHK(hipMalloc(&p->d_flt_list, sizeof(uint32_t) * ((size_t)n_loci + 4)));
    {
        // SMC_KERNEL=sorted runs the sorted-stream kernel first and the table kernel only on the loci it
        // hands over; the default is the table kernel alone, which is faster on every shape measured
        // ("Explored: sorted-stream kernel")
        const char* env = getenv("SMC_KERNEL");
        p->use_sorted = (env && strcmp(env, "sorted") == 0);
        std::vector<int> all((size_t)n_loci);
        for (int64_t l = 0; l < n_loci; ++l) { all[(size_t)l] = (int)l; p->total_reads += loci[l].n_reads; }
        std::stable_sort(all.begin(), all.end(), [&](int x, int y) { return loci[x].n_reads > loci[y].n_reads; });
        std::vector<smc_locus> perm((size_t)n_loci);
        for (int64_t k = 0; k < n_loci; ++k) {
            perm[(size_t)k] = loci[all[(size_t)k]];
            const int ac = (perm[(size_t)k].n_alleles + 7) & ~7;
            if (ac > p->a_cap_all) p->a_cap_all = ac;
        }
        HIPCHK(hipMalloc(&p->d_all_order, sizeof(int) * (size_t)(n_loci + 1)));
        HIPCHK(hipMalloc(&p->d_all_loci, sizeof(smc_locus) * (size_t)(n_loci + 1)));
        HIPCHK(hipMalloc(&p->d_redo, (size_t)n_loci + 16));
        if (n_loci) {
            HIPCHK(hipMemcpy(p->d_all_order, all.data(), sizeof(int) * (size_t)n_loci, hipMemcpyHostToDevice));
            HIPCHK(hipMemcpy(p->d_all_loci, perm.data(), sizeof(smc_locus) * (size_t)n_loci, hipMemcpyHostToDevice));
        }
    }
    if (n_loci) {
        HIPCHK(hipMalloc(&p->d_loci, sizeof(smc_locus) * (size_t)n_loci));
        HIPCHK(hipMemcpy(p->d_loci, loci, sizeof(smc_locus) * (size_t)n_loci, hipMemcpyHostToDevice));
    }
    for (auto& b : bins) {
        if (b.order.empty()) continue;
        // heaviest loci first: the tail of the launch is then made of light blocks
        std::stable_sort(b.order.begin(), b.order.end(), [&](int x, int y) { return loci[x].n_reads > loci[y].n_reads; });
        size_t mx = 0;
        std::vector<int64_t> soff;
        for (int l : b.order) {
            const size_t t = table_bytes(loci[l]);
            if (b.cls == 4) { soff.push_back((int64_t)b.scratch_bytes); b.scratch_bytes += t; }
            if (t > mx) mx = t;
        }
        b.lds_bytes = host_hdr_bytes(b.a_cap) + (b.cls == 4 ? 0 : mx);
        b.lds_bytes = (b.lds_bytes + 255) & ~(size_t)255;
        HIPCHK(hipMalloc(&b.d_order, sizeof(int) * b.order.size()));
        HIPCHK(hipMemcpy(b.d_order, b.order.data(), sizeof(int) * b.order.size(), hipMemcpyHostToDevice));
        {
            std::vector<smc_locus> perm(b.order.size());
            for (size_t k = 0; k < b.order.size(); ++k) perm[k] = loci[b.order[k]];
            HIPCHK(hipMalloc(&b.d_loci, sizeof(smc_locus) * perm.size()));
            HIPCHK(hipMemcpy(b.d_loci, perm.data(), sizeof(smc_locus) * perm.size(), hipMemcpyHostToDevice));
        }
        if (b.cls == 4) {
            HIPCHK(hipMalloc(&b.d_scratch_off, sizeof(int64_t) * soff.size()));
            HIPCHK(hipMemcpy(b.d_scratch_off, soff.data(), sizeof(int64_t) * soff.size(), hipMemcpyHostToDevice));
            HIPCHK(hipMalloc(&b.d_scratch, b.scratch_bytes));
        }
        int64_t reads = 0;
        for (int l : b.order) reads += loci[l].n_reads;
        if (reads > p->dom_reads) { p->dom_reads = reads; p->dom_loci = (int64_t)b.order.size(); p->dom_bin = (int)p->bins.size(); }
        p->bins.push_back(b);
    }
    *out = p;
    return SMC_OK;
}

int smc_plan_set_timing(smc_plan* p, int slots) {
    if (!p || slots < 0) return fail(SMC_E_ARG, "smc_plan_set_timing: bad argument");
    HIPCHK(hipSetDevice(p->ctx->device));
    for (auto e : p->ev0) (void)hipEventDestroy(e);
    for (auto e : p->ev1) (void)hipEventDestroy(e);
    p->ev0.assign((size_t)slots, nullptr);
    p->ev1.assign((size_t)slots, nullptr);
    for (int k = 0; k < slots; ++k) { HIPCHK(hipEventCreate(&p->ev0[k])); HIPCHK(hipEventCreate(&p->ev1[k])); }
    p->timing = slots;
    p->n_timed = 0;
    return SMC_OK;
}

int smc_plan_kernel_ms(smc_plan* p, float* avg_ms, int32_t* n_samples, int64_t* n_loci, int64_t* n_reads) {
    if (!p || !avg_ms) return fail(SMC_E_ARG, "smc_plan_kernel_ms: NULL argument");
    const int64_t ns = p->n_timed < p->timing ? p->n_timed : p->timing;
    if (ns <= 0) return fail(SMC_E_ARG, "smc_plan_kernel_ms: no timed run (smc_plan_set_timing(plan, slots) then smc_plan_run)");
    double tot = 0;
    for (int64_t k = 0; k < ns; ++k) {
        float ms = 0;
        HIPCHK(hipEventSynchronize(p->ev1[k]));
        HIPCHK(hipEventElapsedTime(&ms, p->ev0[k], p->ev1[k]));
        tot += ms;
    }
    *avg_ms = (float)(tot / ns);
    if (n_samples) *n_samples = (int32_t)ns;
    if (n_loci) *n_loci = p->use_sorted ? p->n_loci : p->dom_loci;
    if (n_reads) *n_reads = p->use_sorted ? p->total_reads : p->dom_reads;
    return SMC_OK;
}


int smc_plan_info(const smc_plan* p, int32_t* n_launches, int64_t* scratch_bytes) {
    if (!p) return fail(SMC_E_ARG, "smc_plan_info: NULL plan");
    if (n_launches) *n_launches = (int32_t)p->bins.size() + (p->n_loci ? 1 : 0);
    if (scratch_bytes) {
        int64_t s = 0;
        for (auto& b : p->bins) s += (int64_t)b.scratch_bytes;
        *scratch_bytes = s;
    }
    return SMC_OK;
}

int smc_plan_run(smc_plan* p, const smc_params* prm, const uint32_t* meta, const uint32_t* umi, const uint32_t* frag,
                 const uint32_t* dist, const uint32_t* umi_start, smc_row* rows, void* stream) {
    if (!p || !prm) return fail(SMC_E_ARG, "smc_plan_run: NULL argument");
    if (p->n_loci == 0) return SMC_OK;
    if (!meta || !umi || !frag || !dist || !umi_start || !rows) return fail(SMC_E_ARG, "smc_plan_run: NULL device pointer");
    HIPCHK(hipSetDevice(p->ctx->device));
    hipStream_t st = (hipStream_t)stream;
    KParams kp{prm->min_bq, prm->min_mq, prm->mt_drop, prm->primer_dist, prm->ds, prm->smt};
    HIPCHK(hipMemsetAsync(p->d_flt_list, 0, 16, st));
    if (p->use_sorted) {
        HIPCHK(hipMemsetAsync(p->d_redo, 0, (size_t)p->n_loci, st));
        const size_t lds = (sizeof(Hdr) + (size_t)p->a_cap_all * 64 + sizeof(smc_row) + 128 * 8 + UB * sizeof(URec) + UB * 4 + 255) & ~(size_t)255;
        const bool timed = p->timing > 0;
        const size_t slot = timed ? (size_t)(p->n_timed % p->timing) : 0;
        if (timed) HIPCHK(hipEventRecord(p->ev0[slot], st));
        hipLaunchKernelGGL(k_call_sorted, dim3((unsigned)p->n_loci), dim3(WAVE), lds, st, kp, p->d_all_loci, p->d_all_order,
                           p->a_cap_all, meta, umi, frag, dist, umi_start, p->ctx->lut, rows, p->d_flt_list, p->d_redo);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return fail(SMC_E_HIP, std::string("k_call_sorted launch: ") + hipGetErrorString(e));
        if (timed) { HIPCHK(hipEventRecord(p->ev1[slot], st)); p->n_timed++; }
    }
    for (size_t bi = 0; bi < p->bins.size(); ++bi) {
        const Bin& b = p->bins[bi];
        const bool timed = !p->use_sorted && p->timing > 0 && (int)bi == p->dom_bin;
        const size_t slot = timed ? (size_t)(p->n_timed % p->timing) : 0;
        if (timed) HIPCHK(hipEventRecord(p->ev0[slot], st));
        hipError_t e = hipSuccess;
        switch (b.cls) {
            case 0: e = launch_bin<64, false>(b, kp, p, meta, umi_start, frag, dist, rows, st); break;
            case 1: e = launch_bin<SMC_CLS1_BLOCK, false>(b, kp, p, meta, umi_start, frag, dist, rows, st); break;
            case 2: e = launch_bin<512, false>(b, kp, p, meta, umi_start, frag, dist, rows, st); break;
            case 3: e = launch_bin<1024, false>(b, kp, p, meta, umi_start, frag, dist, rows, st); break;
            default: e = launch_bin<1024, true>(b, kp, p, meta, umi_start, frag, dist, rows, st); break;
        }
        if (e != hipSuccess) return fail(SMC_E_HIP, std::string("k_call_loci launch: ") + hipGetErrorString(e));
        if (timed) { HIPCHK(hipEventRecord(p->ev1[slot], st)); p->n_timed++; }
    }
    const unsigned fgrid = (unsigned)(p->n_loci < 6144 ? p->n_loci : 6144);   // 78 VGPRs: 24 one-wave workgroups per CU
    hipLaunchKernelGGL(k_filter_loci, dim3(fgrid), dim3(WAVE), 0, st, kp, p->d_loci, rows, p->d_flt_list);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(SMC_E_HIP, std::string("k_filter_loci launch: ") + hipGetErrorString(e));
    return SMC_OK;
}

int smc_call_batch_host(smc_ctx* ctx, const smc_params* prm, const smc_locus* loci, int64_t n_loci, const uint32_t* meta,
                        const uint32_t* umi, const uint32_t* frag, const uint32_t* dist, int64_t n_slots,
                        const uint32_t* umi_start, int64_t n_umi_start, smc_row* rows_out) {
    if (!ctx || !prm || (n_loci && (!loci || !rows_out))) return fail(SMC_E_ARG, "smc_call_batch_host: NULL argument");
    if (n_loci == 0) return SMC_OK;
    HIPCHK(hipSetDevice(ctx->device));
    smc_plan* plan = nullptr;
    int rc = smc_plan_create(ctx, loci, n_loci, &plan);
    if (rc) return rc;
    uint32_t* d[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
    const uint32_t* h[5] = {meta, umi, frag, dist, umi_start};
    const int64_t hn[5] = {n_slots, n_slots, n_slots, n_slots, n_umi_start};
    smc_row* d_rows = nullptr;

    auto cleanup = [&]() {
        if (d[1] == d[0]) d[1] = nullptr;
        if (d[3] == d[0]) d[3] = nullptr;
        for (auto q : d) (void)hipFree(q);
        (void)hipFree(d_rows);
        smc_plan_destroy(plan);
    };
    // the default kernel reads the meta and frag planes and umi_start only (8 of the 16 bytes per read): the umi and
    // dist planes cross PCIe only for the sorted-stream variant
    for (int k = 0; k < 5; ++k) {
        if ((k == 1 || k == 3) && !plan->use_sorted) continue;
        hipError_t e = hipMalloc(&d[k], sizeof(uint32_t) * (size_t)(hn[k] > 0 ? hn[k] : 1));
        if (e == hipSuccess && hn[k] > 0) e = hipMemcpy(d[k], h[k], sizeof(uint32_t) * (size_t)hn[k], hipMemcpyHostToDevice);
        if (e != hipSuccess) { cleanup(); return fail(SMC_E_HIP, std::string("plane upload: ") + hipGetErrorString(e)); }
    }
    if (!plan->use_sorted) { d[1] = d[0]; d[3] = d[0]; }    // (never dereferenced; smc_plan_run wants non-NULL)
    hipError_t e = hipMalloc(&d_rows, sizeof(smc_row) * (size_t)n_loci);
    if (e != hipSuccess) { cleanup(); return fail(SMC_E_HIP, std::string("rows alloc: ") + hipGetErrorString(e)); }
    rc = smc_plan_run(plan, prm, d[0], d[1], d[2], d[3], d[4], d_rows, nullptr);
    if (rc == SMC_OK) {
        e = hipDeviceSynchronize();
        if (e == hipSuccess) e = hipMemcpy(rows_out, d_rows, sizeof(smc_row) * (size_t)n_loci, hipMemcpyDeviceToHost);
        if (e != hipSuccess) rc = fail(SMC_E_HIP, std::string("run/download: ") + hipGetErrorString(e));
    }
    cleanup();
    return rc;
}

#ifdef SMC_STAMPS
int smc_debug_stamps(unsigned long long* out, int reset) {
    HIPCHK(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_stamps), sizeof(unsigned long long) * 16));
    if (reset) {
        unsigned long long z[16] = {0};
        HIPCHK(hipMemcpyToSymbol(HIP_SYMBOL(g_stamps), z, sizeof z));
    }
    return SMC_OK;
}
#endif

int smc_event_create(void** ev) {
    hipEvent_t e;
    HIPCHK(hipEventCreate(&e));
    *ev = (void*)e;
    return SMC_OK;
}
int smc_event_record(void* ev, void* stream) {
    HIPCHK(hipEventRecord((hipEvent_t)ev, (hipStream_t)stream));
    return SMC_OK;
}
int smc_event_elapsed_ms(void* start, void* stop, float* ms) {
    HIPCHK(hipEventSynchronize((hipEvent_t)stop));
    HIPCHK(hipEventElapsedTime(ms, (hipEvent_t)start, (hipEvent_t)stop));
    return SMC_OK;
}
void smc_event_destroy(void* ev) { (void)hipEventDestroy((hipEvent_t)ev); }

}  // extern "C"
