/*
 * smc_oracle.c - CPU restatement of smCounter's vc() hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library; the
 * product (smcounter_amd + libsmcounter_hip.so) never does.
 *
 * It restates, sequentially and in double precision, the algorithm of
 * /root/reference/smCounter.py (vc :274-600, calProb :26-98, filterVariants :182-269) on the same
 * structure-of-arrays batch the HIP kernels consume (include/smcounter_hip.h), and fills the
 * same smc_row records, so a GPU row can be compared field by field.
 *
 * Parity pin: tests/test_oracle_golden.py checks this restatement against tests/golden/, vectors
 * produced by running the reference itself (oracle/ref_harness.py, build container only).
 * Third-party arithmetic restated here because its source is not under /root/reference:
 * scipy.stats.fisher_exact two-sided (scipy 1.15.3 semantics: zero margin -> (nan, 1); sample odds
 * ratio, inf when b*c == 0; p = sum of hypergeometric pmf(k) not above pmf(observed), relative
 * slack 1e-7, capped at 1).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "../include/smcounter_hip.h"

#define N_ID 4
#define GAP_ID 5

typedef struct {
    unsigned char present, paired, base;
    double prob;
    int64_t seq; /* dict insertion stamp: order of bcDict[BC].values() */
} frag_t;

typedef struct {
    int base;
    double prob;
    int paired;
    int64_t seq;
} fobs_t;

static int cmp_fobs(const void* a, const void* b) {
    int64_t x = ((const fobs_t*)a)->seq, y = ((const fobs_t*)b)->seq;
    return (x > y) - (x < y);
}

/* ---- scipy.stats.fisher_exact(table)  (smCounter.py:215,238,248,260) ---- */
static double lchoose(double n, double k) { return lgamma(n + 1.0) - lgamma(k + 1.0) - lgamma(n - k + 1.0); }

static void fisher_exact(int64_t a, int64_t b, int64_t c, int64_t d, double* oddsratio, double* pvalue) {
    if (a + b == 0 || c + d == 0 || a + c == 0 || b + d == 0) {
        *oddsratio = NAN;
        *pvalue = 1.0;
        return;
    }
    *oddsratio = (c > 0 && b > 0) ? ((double)(a * d)) / ((double)(c * b)) : INFINITY;
    int64_t n1 = a + b, n2 = c + d, n = a + c;
    int64_t lo = n - n2 > 0 ? n - n2 : 0, hi = n < n1 ? n : n1;
    double lden = lchoose((double)(n1 + n2), (double)n);
    double pexact = exp(lchoose((double)n1, (double)a) + lchoose((double)n2, (double)(n - a)) - lden);
    double thr = pexact * (1.0 + 1e-7), p = 0.0;
    for (int64_t k = lo; k <= hi; ++k) {
        double pk = exp(lchoose((double)n1, (double)k) + lchoose((double)n2, (double)(n - k)) - lden);
        if (pk <= thr) p += pk;
    }
    *pvalue = p < 1.0 ? p : 1.0;
}

static int allele_type(const smc_locus* L, int a) { /* 0 SNP, 1 SDEL, 2 INDEL  (convertToVcf :103-117) */
    if ((L->snp_mask >> a) & 1) return 0;
    if (a == GAP_ID) return 1;
    return 2;
}

/* filterVariants (smCounter.py:182-269) minus the two flags that need the FASTA */
static void filter_cand(const smc_params* P, const smc_locus* L, smc_row* R, smc_cand* C,
                        int (*tal)[SMC_NT], const int* mtcnt, const int* strong) {
    int alt = C->allele, ref = L->ref_allele;
    static const int zero[SMC_NT] = {0};
    const int* ta = tal[alt];
    const int* tr = ref < SMC_MAX_ALLELES ? tal[ref] : zero;
    int vtype = allele_type(L, alt);
    uint32_t f = 0;
    C->p_sb = C->p_r1 = C->p_r2 = C->p_pr = NAN;
    if (R->used_mt < 5) f |= SMC_F_LM;                                   /* :187 */
    if (strong[alt] < 2) f |= SMC_F_LSM;                                 /* :191 */
    C->vmf_lt_099 = (1.0 * mtcnt[alt] / R->used_mt < 0.99);              /* :198,:202 */
    double af_alt = 100.0 * ta[SMC_T_CNT] / R->cvg;                      /* :206 */
    int pairs = ta[SMC_T_DISCORD] + ta[SMC_T_CONCORD];                   /* :207 */
    double orat, p;
    if (pairs >= 1000 && 1.0 * ta[SMC_T_DISCORD] / pairs >= 0.5) {
        f |= SMC_F_DP;                                                   /* :208-209 */
    } else if (af_alt <= 60.0) {
        fisher_exact(tr[SMC_T_REV], tr[SMC_T_FWD], ta[SMC_T_REV], ta[SMC_T_FWD], &orat, &p); /* :211-215 */
        C->p_sb = p;
        if (p < 0.00001 && (orat >= 50 || orat <= 1.0 / 50)) f |= SMC_F_SB;
    }
    double bq_alt = 0.0;                                                 /* :222-227 */
    if (vtype == 0 && ta[SMC_T_LOWQ] > 0) bq_alt = 1.0 * ta[SMC_T_LOWQ] / ta[SMC_T_CNT];
    if (bq_alt > 0.4) f |= SMC_F_LOWQ;
    if (vtype == 0) {                                                    /* :230-266 */
        fisher_exact(tr[SMC_T_R1LE], tr[SMC_T_R1N] - tr[SMC_T_R1LE], ta[SMC_T_R1LE],
                     ta[SMC_T_R1N] - ta[SMC_T_R1LE], &orat, &p);
        C->p_r1 = p;
        if (p < 0.001 && orat < 0.05 && af_alt <= 60.0) f |= SMC_F_R1CP;
        fisher_exact(tr[SMC_T_R2BCLE], tr[SMC_T_R2N] - tr[SMC_T_R2BCLE], ta[SMC_T_R2BCLE],
                     ta[SMC_T_R2N] - ta[SMC_T_R2BCLE], &orat, &p);
        C->p_r2 = p;
        if (p < 0.001 && orat < 0.05 && af_alt <= 60.0) f |= SMC_F_R2CP;
        int alt_le = ta[SMC_T_R2PRLE], alt_gt = ta[SMC_T_R2N] - ta[SMC_T_R2PRLE];
        fisher_exact(tr[SMC_T_R2PRLE], tr[SMC_T_R2N] - tr[SMC_T_R2PRLE], alt_le, alt_gt, &orat, &p);
        C->p_pr = p;
        if (alt_le + alt_gt > 0)
            if (1.0 * alt_le / (alt_le + alt_gt) >= 0.98 || (p < 0.001 && orat < 1.0 / 20))
                f |= SMC_F_PRIMERCP;
    }
    C->flt = f;
}

static void fill_cand(smc_cand* C, int a, double pi, int (*tal)[SMC_NT], const int* mtcnt, const int* strong) {
    memset(C, 0, sizeof *C);
    C->allele = a;
    C->p_sb = C->p_r1 = C->p_r2 = C->p_pr = NAN;
    if (a < 0) return;
    C->pi = pi;
    C->vdp = tal[a][SMC_T_CNT];
    C->vmt = mtcnt[a];
    C->vsm = strong[a];
    memcpy(C->tal, tal[a], sizeof C->tal);
}

/* tie order of sorted(finalDict.items()) (smCounter.py:534): py2 dict slot order, see
 * smcounter_amd/py2compat.py.  8-slot table while finalDict has <= 5 keys, 32 slots from 6. */
static int tie_rank(int a, int n_keys) {
    static const int r8[6] = {0, 5, 6, 2, 7, 4};      /* A T G C N DEL */
    static const int r32[6] = {0, 21, 6, 2, 15, 20};
    if (a < 6) return n_keys <= 5 ? r8[a] : r32[a];
    return 64 + a;
}

/* `fragile`, when not NULL, receives the number of barcodes with >= 3 fragments whose two largest
 * per-barcode prediction indices are equal to within 1e-9: whether such a barcode has a unique
 * maximum (smCounter.py:514) is decided by floating-point rounding of products whose factor order is
 * the iteration order of a dict (smCounter.py:62) - not pinned by the algorithm, so parity tests
 * skip the MT-count columns of the few loci that contain one. */
static int call_locus(const smc_params* P, const smc_locus* L, const uint32_t* meta, const uint32_t* umi,
                      const uint32_t* frag, const uint32_t* dist, const uint32_t* ustart /* may be NULL */, smc_row* R,
                      int32_t* fragile, double* pi_all /* may be NULL: [SMC_MAX_ALLELES], PI of every allele, NaN = not a key */) {
    if (pi_all) for (int a = 0; a < SMC_MAX_ALLELES; ++a) pi_all[a] = NAN;
    if (fragile) *fragile = 0;
    memset(R, 0, sizeof *R);
    R->max_allele = R->second_allele = -1;
    fill_cand(&R->cand[0], -1, 0, NULL, NULL, NULL);
    fill_cand(&R->cand[1], -1, 0, NULL, NULL, NULL);
    const int n = L->n_reads, nU = L->n_umi;
    int tal[SMC_MAX_ALLELES][SMC_NT];
    int mtcnt[SMC_MAX_ALLELES], strong[SMC_MAX_ALLELES], touched[SMC_MAX_ALLELES];
    double fin[SMC_MAX_ALLELES];
    memset(tal, 0, sizeof tal);
    memset(mtcnt, 0, sizeof mtcnt);
    memset(strong, 0, sizeof strong);
    memset(touched, 0, sizeof touched);
    for (int a = 0; a < SMC_MAX_ALLELES; ++a) fin[a] = 0.0;

    int* nfrag = (int*)calloc((size_t)nU + 1, sizeof(int));
    int* foff = (int*)calloc((size_t)nU + 1, sizeof(int));
    int* in_bc = (int*)calloc((size_t)nU + 1, sizeof(int));
    int* bc_order = (int*)calloc((size_t)nU + 1, sizeof(int));
    int bad = 0;
    /* allBcDict (smCounter.py:463-464): fragments per barcode over ALL reads.  frag[] is the
     * locus-level slot; a barcode's fragments are the slots between its smallest and largest. */
    for (int u = 0; u < nU; ++u) { foff[u] = L->n_frag; nfrag[u] = 0; }
    for (int i = 0; i < n; ++i) {
        uint32_t u = umi[i], f = frag[i] & SMC_FRAG_SLOT_MASK;   /* (bits 27-31: the read class, not used here) */
        if (u >= (uint32_t)nU || f >= (uint32_t)L->n_frag || (meta[i] & 0xff) >= L->n_alleles) { bad = 1; continue; }
        if ((int)f < foff[u]) foff[u] = (int)f;
        if ((int)f + 1 > nfrag[u]) nfrag[u] = (int)f + 1;      /* end of the range, for now */
    }
    int64_t tot = 0;
    for (int u = 0; u < nU; ++u) { nfrag[u] = nfrag[u] > foff[u] ? nfrag[u] - foff[u] : 0; tot += nfrag[u]; }
    if (tot != L->n_frag) bad = 1;
    if (bad) {
        R->status = SMC_ST_BAD_INPUT;
        free(nfrag); free(foff); free(in_bc); free(bc_order);
        return 0;
    }
    frag_t* ft = (frag_t*)calloc((size_t)tot + 1, sizeof(frag_t));
    int n_bc = 0, cvg = 0;
    int64_t stamp = 0;
    for (int i = 0; i < n; ++i) {                                        /* smCounter.py:316-479 */
        uint32_t m = meta[i];
        int a = m & 0xff, bq = (m >> 8) & 0xff, fl = (m >> 16) & 0xff, mq = m >> 24;
        int kind = (fl >> SMC_KIND_SHIFT) & 3, r2 = fl & SMC_FL_R2, rev = fl & SMC_FL_REV;
        int dbc = dist[i] & 0xffff, dpr = dist[i] >> 16;
        cvg++;                                                           /* :368 */
        if (kind == SMC_KIND_BASE && bq < P->min_bq) tal[a][SMC_T_LOWQ]++; /* :428 */
        if (kind == SMC_KIND_GAP) bq = P->min_bq;                        /* :418 */
        int inc = bq >= P->min_bq && mq >= P->min_mq && (fl & SMC_FL_MMOK); /* :378 */
        tal[a][SMC_T_CNT]++;
        if (kind != SMC_KIND_GAP) tal[a][rev ? SMC_T_REV : SMC_T_FWD]++;
        if (kind == SMC_KIND_BASE && inc) {
            if (!r2) {                                                   /* :432-440 */
                tal[a][SMC_T_R1N]++;
                tal[a][SMC_T_R1LE] += dbc <= 20;
            } else {                                                     /* :441-452 */
                tal[a][SMC_T_R2N]++;
                tal[a][SMC_T_R2BCLE] += dbc <= 20;
                tal[a][SMC_T_R2PRLE] += dpr <= P->primer_dist;
            }
        }
        if (inc) {                                                       /* :467-479 */
            uint32_t u = umi[i];
            if (!in_bc[u]) { in_bc[u] = 1; bc_order[n_bc++] = (int)u; }
            frag_t* s = &ft[frag[i] & SMC_FRAG_SLOT_MASK];
            double prob = pow(10.0, -bq / 10.0);
            if (!s->present) {
                s->present = 1; s->paired = 0; s->base = (unsigned char)a; s->prob = prob; s->seq = stamp++;
            } else if (a == s->base || a == N_ID) {
                if (prob > s->prob) s->prob = prob;
                s->paired = 1;
                if (a == s->base) tal[a][SMC_T_CONCORD]++;
            } else {
                s->present = 0;
                tal[a][SMC_T_DISCORD]++;
            }
        }
    }
    R->cvg = cvg;
    R->all_mt = 0;
    for (int u = 0; u < nU; ++u) R->all_mt += nfrag[u] > 0;              /* :482 */
    R->all_frag = (int)tot;                                              /* :483 */
    int used = n_bc < P->ds ? n_bc : P->ds;                              /* :489 */
    R->used_mt = used;
    for (int k = 0; k < 4; ++k) R->dp[k] = tal[k][SMC_T_CNT];
    if (used == 0) {                                                     /* :492-494 */
        R->status = SMC_ST_ZERO_COVERAGE;
        free(nfrag); free(foff); free(in_bc); free(bc_order); free(ft);
        return 0;
    }
    if ((L->flags & SMC_LF_SAMPLED) && ustart) {
        /* random.sample() of the reference (:496-498) was applied by the host (features.py, py2compat.py):
         * keys of bcDict marked in umi_start are dropped; the count kept must be min(#keys, ds) */
        int k = 0;
        for (int u = 0; u < nU; ++u)
            if (in_bc[u] && !(ustart[u] & SMC_USTART_DROPPED)) { if (k < n_bc) bc_order[k] = u; ++k; }
        if (k != used) {
            memset(R, 0, sizeof *R);
            R->max_allele = R->second_allele = -1;
            fill_cand(&R->cand[0], -1, 0, NULL, NULL, NULL);
            fill_cand(&R->cand[1], -1, 0, NULL, NULL, NULL);
            R->status = SMC_ST_BAD_INPUT;
            free(nfrag); free(foff); free(in_bc); free(bc_order); free(ft);
            return 0;
        }
        if (n_bc > P->ds) R->status |= SMC_ST_DOWNSAMPLED;
    } else if (n_bc > P->ds) {
        /* no host sample: deterministic stand-in shared with the device path (non-parity): keep the ds
         * lowest barcode ids */
        R->status |= SMC_ST_DOWNSAMPLED;
        int k = 0;
        for (int u = 0; u < nU && k < used; ++u) if (in_bc[u]) bc_order[k++] = u;
    }
    const double pcr_no_error = 1.0 - 3e-5;                              /* :20 */
    int underflow = 0;
    fobs_t* obs = (fobs_t*)malloc(sizeof(fobs_t) * (size_t)(tot + 1));
    for (int b = 0; b < used; ++b) {                                     /* :506-532 */
        int u = bc_order[b], nf = 0;
        for (int f = 0; f < nfrag[u]; ++f) {
            frag_t* s = &ft[foff[u] + f];
            if (s->present) { obs[nf].base = s->base; obs[nf].prob = s->prob; obs[nf].paired = s->paired; obs[nf].seq = s->seq; nf++; }
        }
        qsort(obs, (size_t)nf, sizeof(fobs_t), cmp_fobs);
        R->used_frag += nf;                                              /* :501 */
        /* ---- calProb (:26-98) ---- */
        int keys[SMC_MAX_ALLELES], nk = 0;
        double pred[SMC_MAX_ALLELES];
        if (nf <= P->mt_drop) {                                          /* :28-32 */
            for (int k = 0; k < 4; ++k) { keys[nk] = k; pred[nk++] = 0.0; }
            for (int k = 0; k < nk; ++k) pred[k] = -log10(1.0 - pred[k]);
        } else {
            int exist[SMC_MAX_ALLELES] = {0}, inuniq[SMC_MAX_ALLELES] = {0}, cnt[SMC_MAX_ALLELES] = {0};
            double prodP[SMC_MAX_ALLELES], pcrP[SMC_MAX_ALLELES], tmp[SMC_MAX_ALLELES];
            int n_exist = 0;
            for (int i = 0; i < nf; ++i) if (!exist[obs[i].base]) { exist[obs[i].base] = 1; n_exist++; }
            int n_uniq = n_exist;
            for (int a = 0; a < SMC_MAX_ALLELES; ++a) inuniq[a] = exist[a];
            for (int k = 0; k < 4 && n_uniq < 4; ++k) if (!inuniq[k]) { inuniq[k] = 1; n_uniq++; } /* :49-54 */
            for (int a = 0; a < SMC_MAX_ALLELES; ++a) if (inuniq[a]) { keys[nk++] = a; prodP[a] = 1.0; }
            double rightP = 1.0, sumP = 0.0;
            for (int i = 0; i < nf; ++i) {                               /* :62-77 */
                double prob = obs[i].paired ? obs[i].prob : 0.1;
                int base = obs[i].base;
                prodP[base] *= 1.0 - prob;
                cnt[base]++;
                for (int k = 0; k < nk; ++k) if (keys[k] != base) prodP[keys[k]] *= prob;
                rightP *= 1.0 - prob;
            }
            if (rightP < 0x1p-1000) underflow = 1;   /* SMC_ST_UNDERFLOW: what follows is denormal rounding in this loop's order */
            for (int k = 0; k < nk; ++k) {                               /* :79-81 */
                double ratio = (cnt[keys[k]] + 0.5) / (nf + 0.5 * nk);
                pcrP[keys[k]] = pow(10.0, -6.0 * ratio);
            }
            for (int k = 0; k < nk; ++k) {                               /* :83-93 */
                int key = keys[k];
                if (exist[key]) {
                    double mn = INFINITY;
                    for (int j = 0; j < nk; ++j) if (keys[j] != key && pcrP[keys[j]] < mn) mn = pcrP[keys[j]];
                    tmp[key] = pcr_no_error * prodP[key] + rightP * mn;
                } else {
                    tmp[key] = rightP;
                    for (int j = 0; j < nk; ++j) if (exist[keys[j]] && keys[j] != key) tmp[key] *= pcrP[keys[j]];
                }
                sumP += tmp[key];
            }
            for (int k = 0; k < nk; ++k) {                               /* :95-96, :508-510 */
                double post = sumP <= 0 ? 0.0 : tmp[keys[k]] / sumP;
                double x = 1.0 - post;
                pred[k] = x > 0.0 ? -log10(x) : 16.0;
            }
        }
        double mx = -INFINITY;
        int n_max = 0, arg = -1;
        for (int k = 0; k < nk; ++k) {                                   /* :511-514 */
            fin[keys[k]] += pred[k];
            touched[keys[k]] = 1;
            if (pred[k] > mx) mx = pred[k];
        }
        for (int k = 0; k < nk; ++k) if (pred[k] == mx) { n_max++; arg = k; }
        if (fragile && nf >= 3) {
            int close = 0;
            for (int k = 0; k < nk; ++k) if (fabs(pred[k] - mx) <= 1e-9 * (fabs(mx) > 1 ? fabs(mx) : 1)) close++;
            if (close >= 2) (*fragile)++;
        }
        if (n_max == 1) {                                                /* :515-519 */
            mtcnt[keys[arg]]++;
            if (pred[arg] > P->smt) strong[keys[arg]]++;
        } else if (nf == 1) {                                            /* :521-523 */
            mtcnt[obs[0].base]++;
        }
        R->mt3 += nf >= 3; R->mt5 += nf >= 5; R->mt7 += nf >= 7; R->mt10 += nf >= 10; /* :525-532 */
    }
    free(obs);

    /* ---- ranking (:534-542) ---- */
    int nkeys = 0;
    for (int a = 0; a < SMC_MAX_ALLELES; ++a) if (touched[a]) { nkeys++; R->touched_mask |= 1ull << a; }
    R->n_touched = nkeys;
    int best = -1, second = -1;
    for (int pass = 0; pass < 2; ++pass) {
        int pick = -1;
        for (int a = 0; a < SMC_MAX_ALLELES; ++a) {
            if (!touched[a] || a == best) continue;
            if (pick < 0 || fin[a] > fin[pick] ||
                (fin[a] == fin[pick] && tie_rank(a, nkeys) < tie_rank(pick, nkeys)))
                pick = a;
        }
        if (pass == 0) best = pick; else second = pick;
    }
    if (underflow) R->status |= SMC_ST_UNDERFLOW;
    R->max_allele = best;
    R->second_allele = second;
    for (int k = 0; k < 4; ++k) { R->umt[k] = mtcnt[k]; R->vsm[k] = strong[k]; R->pi[k] = fin[k]; }
    if (pi_all) for (int a = 0; a < SMC_MAX_ALLELES; ++a) if (touched[a]) pi_all[a] = fin[a];
    if (L->ref_allele < SMC_MAX_ALLELES) memcpy(R->ref_tal, tal[L->ref_allele], sizeof R->ref_tal);
    int ref = L->ref_allele;
    int alt = best == ref ? second : best;                               /* :541-542 */
    double alt_pi = best == ref ? fin[second] : fin[best];
    fill_cand(&R->cand[0], alt, alt_pi, tal, mtcnt, strong);
    if (alt_pi >= 5 && allele_type(L, alt) != 1) {                        /* :549 */
        R->cand[0].flt_applied = 1;
        filter_cand(P, L, R, &R->cand[0], tal, mtcnt, strong);
    }
    double mf1 = 1.0 * mtcnt[best] / used, mf2 = 1.0 * mtcnt[second] / used; /* :553-554 */
    if (best != ref && second != ref && mf1 >= 0.45 && mf2 >= 0.45) {     /* :555 */
        R->biallelic = 1;
        fill_cand(&R->cand[1], second, fin[second], tal, mtcnt, strong);
        if (fin[second] >= 5 && allele_type(L, second) != 1) {            /* :563 */
            R->cand[1].flt_applied = 1;
            filter_cand(P, L, R, &R->cand[1], tal, mtcnt, strong);
        }
    }
    free(nfrag); free(foff); free(in_bc); free(bc_order); free(ft);
    return 0;
}

/* umi_start may be NULL (no host-applied down-sampling marks); otherwise the array of include/smcounter_hip.h */
int smc_oracle_call_batch_full(const smc_params* P, const smc_locus* loci, int64_t n_loci, const uint32_t* meta,
                               const uint32_t* umi, const uint32_t* frag, const uint32_t* dist, const uint32_t* umi_start,
                               smc_row* rows, int32_t* fragile, double* pi_all /* may be NULL: [n_loci][SMC_MAX_ALLELES] */) {
    for (int64_t l = 0; l < n_loci; ++l) {
        const smc_locus* L = &loci[l];
        if (L->n_alleles > SMC_MAX_ALLELES) return -1;
        call_locus(P, L, meta + 4 * (int64_t)L->read_off4, umi + 4 * (int64_t)L->read_off4, frag + 4 * (int64_t)L->read_off4,
                   dist + 4 * (int64_t)L->read_off4, umi_start ? umi_start + L->umi_off : NULL, &rows[l],
                   fragile ? &fragile[l] : NULL, pi_all ? pi_all + l * SMC_MAX_ALLELES : NULL);
    }
    return 0;
}
int smc_oracle_call_batch_ds(const smc_params* P, const smc_locus* loci, int64_t n_loci, const uint32_t* meta,
                             const uint32_t* umi, const uint32_t* frag, const uint32_t* dist, const uint32_t* umi_start,
                             smc_row* rows, int32_t* fragile) {
    return smc_oracle_call_batch_full(P, loci, n_loci, meta, umi, frag, dist, umi_start, rows, fragile, NULL);
}
int smc_oracle_call_batch(const smc_params* P, const smc_locus* loci, int64_t n_loci, const uint32_t* meta,
                          const uint32_t* umi, const uint32_t* frag, const uint32_t* dist, smc_row* rows) {
    return smc_oracle_call_batch_ds(P, loci, n_loci, meta, umi, frag, dist, NULL, rows, NULL);
}
int smc_oracle_call_batch_ex(const smc_params* P, const smc_locus* loci, int64_t n_loci, const uint32_t* meta,
                             const uint32_t* umi, const uint32_t* frag, const uint32_t* dist, smc_row* rows,
                             int32_t* fragile) {
    return smc_oracle_call_batch_ds(P, loci, n_loci, meta, umi, frag, dist, NULL, rows, fragile);
}

/* exposed for tests: the Fisher restatement against captured scipy calls */
/* ---- the NON-parity down-sampling of loci over the barcode cap (include/smcounter_hip.h: smc_philox_marks), restated: not the
 * reference's random.sample (smCounter.py:496-498 - that one is smcounter_amd/py2compat.py's, from barcode texts) but the documented
 * alternative: a key of bcDict (a barcode with an included read, :467-468) gets the 64-bit value words 0, 1 of
 * Philox4x32-10(counter = (the barcode's identity - the caller's 64-bit value, or its index -, 0, 0), key = the halves of
 * (position ^ seed)); the ds smallest stay, ties by index.
 * Written from the published algorithm (Salmon et al., "Parallel random numbers: as easy as 1, 2, 3", SC'11; the known answers of
 * Random123's kat_vectors are in tests/test_oracle_golden.py), independently of csrc/k_philox_marks.inc. */
static void oracle_philox(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) {
    uint32_t c[4] = {ctr[0], ctr[1], ctr[2], ctr[3]}, k[2] = {key[0], key[1]};
    for (int round = 0; round < 10; ++round) {
        if (round) { k[0] += 0x9E3779B9u; k[1] += 0xBB67AE85u; }          /* the Weyl sequence of the key schedule */
        uint64_t m0 = (uint64_t)0xD2511F53u * c[0], m1 = (uint64_t)0xCD9E8D57u * c[2];
        uint32_t hi0 = (uint32_t)(m0 >> 32), lo0 = (uint32_t)m0, hi1 = (uint32_t)(m1 >> 32), lo1 = (uint32_t)m1;
        uint32_t n[4] = {hi1 ^ c[1] ^ k[0], lo1, hi0 ^ c[3] ^ k[1], lo0};
        memcpy(c, n, sizeof c);
    }
    memcpy(out, c, sizeof c);
}
void smc_oracle_philox(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) { oracle_philox(ctr, key, out); }
typedef struct { uint64_t r; uint32_t u; } phx_key;
static int cmp_phx(const void* a, const void* b) {
    const phx_key *x = (const phx_key*)a, *y = (const phx_key*)b;
    if (x->r != y->r) return x->r < y->r ? -1 : 1;
    return x->u < y->u ? -1 : x->u > y->u;
}
/* marks umi_start / loci[].flags in place, as the host's reference-exact sampling does; meta / umi: the raw-field planes */
int smc_oracle_philox_marks(const smc_params* P, smc_locus* loci, int64_t n_loci, const int64_t* pos, const uint32_t* meta,
                            const uint32_t* umi, uint32_t* umi_start, const uint64_t* ident /* per umi_start entry, or NULL: the index */,
                            uint64_t seed) {
    for (int64_t l = 0; l < n_loci; ++l) {
        smc_locus* L = &loci[l];
        const int nU = L->n_umi, n = L->n_reads;
        if (P->ds <= 0 || nU <= P->ds || (L->flags & SMC_LF_SAMPLED)) continue;
        const uint32_t* m = meta + 4ll * L->read_off4;
        const uint32_t* um = umi + 4ll * L->read_off4;
        uint32_t* us = umi_start + L->umi_off;
        unsigned char* in_bc = (unsigned char*)calloc((size_t)nU + 1, 1);
        for (int i = 0; i < n; ++i) {                                    /* incCond, smCounter.py:378 (an in-deletion read counts with minBQ, :418) */
            int bq = (m[i] >> 8) & 0xff, fl = (m[i] >> 16) & 0xff, mq = m[i] >> 24;
            if (((fl >> SMC_KIND_SHIFT) & 3) == SMC_KIND_GAP) bq = P->min_bq;
            if (bq >= P->min_bq && mq >= P->min_mq && (fl & SMC_FL_MMOK) && um[i] < (uint32_t)nU) in_bc[um[i]] = 1;
        }
        phx_key* keys = (phx_key*)malloc(sizeof(phx_key) * ((size_t)nU + 1));
        int nk = 0;
        const uint64_t kx = (uint64_t)pos[l] ^ seed;
        const uint32_t key[2] = {(uint32_t)kx, (uint32_t)(kx >> 32)};
        for (int u = 0; u < nU; ++u) {
            if (!in_bc[u]) continue;
            const uint64_t id = ident ? ident[L->umi_off + (uint32_t)u] : (uint64_t)(uint32_t)u;
            const uint32_t ctr[4] = {(uint32_t)id, (uint32_t)(id >> 32), 0u, 0u};
            uint32_t x[4];
            oracle_philox(ctr, key, x);
            keys[nk].r = (uint64_t)x[0] << 32 | x[1]; keys[nk].u = (uint32_t)u; ++nk;
        }
        qsort(keys, (size_t)nk, sizeof(phx_key), cmp_phx);
        for (int k = P->ds; k < nk; ++k) us[keys[k].u] |= SMC_USTART_DROPPED;
        L->flags |= SMC_LF_SAMPLED;
        free(keys); free(in_bc);
    }
    return 0;
}

void smc_oracle_fisher(int64_t a, int64_t b, int64_t c, int64_t d, double* oddsratio, double* pvalue) {
    fisher_exact(a, b, c, d, oddsratio, pvalue);
}

int smc_oracle_row_size(void) { return (int)sizeof(smc_row); }
