/*
 * smcounter_hip.h - C ABI of the MI355X (gfx950) implementation of smCounter's per-locus
 * variant-calling hot path.
 *
 * What this replaces in the reference (/root/reference/smCounter.py):
 *   - vc()            smCounter.py:274-600   one call per locus inside a multiprocessing worker
 *   - calProb()       smCounter.py:26-98     per-barcode posterior
 *   - filterVariants  smCounter.py:182-269   FILTER flags (all but the two that read the FASTA)
 *   - the Pool dispatch of main()  smCounter.py:683-685  (one apply_async per locus) becomes one
 *     smc_plan_run() over a structure-of-arrays batch of loci.
 * The reference has no FFI of its own (it is pure Python); the binding a maintainer would add is
 * the ctypes stub shown in INTEGRATION.md.
 *
 * Conventions: plain C, caller owns every buffer, no exceptions cross the boundary; every entry
 * point returns 0 on success or a negative SMC_E_* code, and smc_last_error() returns the
 * message of the last failure on the calling thread.  One host thread per GPU.
 */
#ifndef SMCOUNTER_HIP_H
#define SMCOUNTER_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SMC_ABI_VERSION 8
#define SMC_MAX_ALLELES 64 /* allele ids per locus; ids 0-5 are A,T,G,C,N,'DEL' */

/* error codes */
#define SMC_OK 0
#define SMC_E_ARG (-1)     /* bad argument */
#define SMC_E_HIP (-2)     /* HIP runtime failure (message has the hipError string) */
#define SMC_E_NOGPU (-3)   /* no usable gfx950 device */
#define SMC_E_INPUT (-4)   /* batch violates the layout contract */

/* row.status */
#define SMC_ST_OK 0
#define SMC_ST_ZERO_COVERAGE 1      /* usedMT == 0: the 45-field Zero_Coverage row, smCounter.py:492-494 */
#define SMC_ST_DOWNSAMPLED 0x100    /* more barcodes than ds: reference would random.sample (:496-498) */
#define SMC_ST_BAD_INPUT 0x200      /* an id in the batch was out of the range its descriptor declares */
#define SMC_ST_UNDERFLOW 0x400      /* a barcode of the locus has so many fragments that calProb's products (smCounter.py:62-77:
                                     * 0.9^n for unpaired fragments) left the normal double range (n > ~ 6,700): the reference's own
                                     * posterior there is made of denormal rounding and depends on the order it multiplies in (py2
                                     * dict order of read ids).  The row is computed; its PI / UMT columns are not pinned. */

/* FILTER bits in smc_cand.flt, in the order filterVariants appends them (smCounter.py:187-266) */
#define SMC_F_LM 0x001
#define SMC_F_LSM 0x002
#define SMC_F_HP 0x004   /* set by the host: needs the reference sequence (smCounter.py:195-199) */
#define SMC_F_LOWC 0x008 /* set by the host (smCounter.py:202-203) */
#define SMC_F_DP 0x010
#define SMC_F_SB 0x020
#define SMC_F_LOWQ 0x040
#define SMC_F_R1CP 0x080
#define SMC_F_R2CP 0x100
#define SMC_F_PRIMERCP 0x200

/* per-read flag bits inside meta (bits 16-23); see smcounter_amd/features.py */
#define SMC_FL_R2 1
#define SMC_FL_REV 2
#define SMC_FL_MMOK 4
#define SMC_KIND_SHIFT 3
#define SMC_KIND_BASE 0
#define SMC_KIND_GAP 1      /* 'DEL': inside a deletion, quality forced to minBQ (smCounter.py:416-418) */
#define SMC_KIND_INS 2
#define SMC_KIND_DELSTART 3

/* Read class: bits 27-31 of every `frag` plane word (bits 0-26 hold the fragment slot). What a read adds to its
 * allele's tallies and whether it enters bcDict depends only on a handful of predicates that the feature extraction
 * knows (it has the run's parameters): the class names the combination, the default kernel looks the tally
 * increments up by class and never touches base quality / MAPQ / distance arithmetic for the tallies, nor the
 * `dist` plane at all. incCond = (bq >= minBQ or in-deletion) and mapq >= minMQ and mismatch-ok (smCounter.py:378).
 *   in-deletion ('DEL', kind 1):                  0 + incCond
 *   insertion / deletion start (kind 2, 3):       2 + 2 * reverse + incCond
 *   regular base (kind 0):                        6 + 8 * reverse + sub, with sub =
 *       0 not included, bq >= minBQ   1 not included, bq < minBQ (lowQReads, :428)
 *       2 included read 1, distToBcEnd > 20   3 included read 1, distToBcEnd <= 20            (:432-440)
 *       4 + (distToBcEnd <= 20) + 2 * (distToPrimerEnd <= primerDist)   included read 2       (:441-452)
 * The raw fields stay in the planes (the CPU restatement under oracle/ computes everything from them, so the parity
 * tests check the classes too).  The class bakes in minBQ / minMQ / mismatchThr / primerDist: planes are only valid for the
 * parameter set they were built with, which is why every locus descriptor carries smc_param_fingerprint() of it and
 * smc_plan_run refuses any other (SMC_E_INPUT). */
#define SMC_FRAG_SLOT_MASK 0x07FFFFFFu
#define SMC_FRAG_CLASS_SHIFT 27
#define SMC_N_READ_CLASS 22
/* The read word: what the locus kernels read, ONE uint32 per read (4 of the raw-field planes' 16 bytes cross HBM).
 *   bits 0-7 allele id, 8-15 base quality (<= 126; minBQ for a read inside a deletion) - the low half of the meta word;
 *   bit 16 the read is the first of its fragment at this locus (the fragment slots are dense and ascending, so the slot
 *          number itself carries nothing else); bit 17 the read is included (incCond, smCounter.py:378) and bit 18 its class is
 *          one of the SMC_N_READ_CLASS known ones - both functions of the class (smc_class_bits below), spelled out so that the
 *          scan takes them with the fragment bit in one byte instead of looking the class up; bits 19-26 zero; bits 27-31 the
 *          read class (as in the frag word).
 * smc_build_planes writes it directly; smc_pack_words folds a batch's meta and frag planes into it (and checks the slot
 * contract, which the words can no longer break); smc_plan_run_words runs on it. */
#define SMC_RW_NF 0x00010000u
#define SMC_RW_INC 0x00020000u
#define SMC_RW_OK 0x00040000u
#define SMC_RW_CLASS_SHIFT 27
#if defined(__HIPCC__)
#define SMC_HOST_DEVICE __host__ __device__   /* (the device plane builder evaluates it too) */
#else
#define SMC_HOST_DEVICE
#endif
SMC_HOST_DEVICE static inline uint32_t smc_read_class(int kind, int rev, int r2, int inc, int bq_ok, int le20, int prle) {
    if (kind == 1) return (uint32_t)(0 + (inc ? 1 : 0));
    if (kind != 0) return (uint32_t)(2 + (rev ? 2 : 0) + (inc ? 1 : 0));
    uint32_t sub;
    if (!inc) sub = bq_ok ? 0u : 1u;
    else if (!r2) sub = 2u + (le20 ? 1u : 0u);
    else sub = 4u + (le20 ? 1u : 0u) + (prle ? 2u : 0u);
    return 6u + (rev ? 8u : 0u) + sub;
}

/* bits 17-18 of the read word for a class: included = the incCond half of smc_read_class's encoding, known = class < 22 */
SMC_HOST_DEVICE static inline uint32_t smc_class_bits(uint32_t cls) {
    if (cls >= SMC_N_READ_CLASS) return 0u;
    const int inc = cls < 6u ? (int)(cls & 1u) : (((cls - 6u) & 7u) >= 2u);
    return SMC_RW_OK | (inc ? SMC_RW_INC : 0u);
}

/* The read word in 16 bits (ABI 7): what smc_build_planes_w16 writes and smc_plan_run_words16 reads - the intermediate between the
 * walk over the alignments and the locus kernels at half the bytes (the walk is bound by what it writes).  The same
 * information as the 32-bit word, for runs whose allele ids stay below 16 and whose base qualities below 64 (a run that
 * breaks either is reported - status bit 32 of smc_build_planes_w16 - and built with 32-bit words instead):
 *   bits 0-3 allele id; bit 4 first read of its fragment at this locus; bit 5 included (incCond); bits 6-7 and 14-15 the low and
 *   the high half of a 4-bit class index; bits 8-13 base quality.
 * (included << 4 | class index) is the read class renumbered so that inclusion is a bit of its own (smc_class16 below:
 *   not included: 0 inside a deletion, 1 + reverse at an insertion / deletion start, 3 + 2 * reverse + (bq < minBQ) on a base;
 *   included:     0, 1 + reverse as above, 3 + 6 * reverse + (smc_read_class's sub - 2) on a base);
 * "the class is a known one" (bit 18 of the 32-bit word) has no bit: the builder writes no other. */
SMC_HOST_DEVICE static inline uint32_t smc_class16(uint32_t cls) {          /* smc_read_class's number -> included << 4 | index; 31: no class */
    if (cls < 2u) return (cls & 1u) << 4;
    if (cls < 6u) return ((cls & 1u) << 4) | (1u + ((cls - 2u) >> 1));
    if (cls >= SMC_N_READ_CLASS) return 31u;
    { const uint32_t rev = (cls - 6u) >> 3, sub = (cls - 6u) & 7u;
      return sub < 2u ? 3u + 2u * rev + sub : 16u | (3u + 6u * rev + (sub - 2u)); }
}
SMC_HOST_DEVICE static inline uint32_t smc_class16_inv(uint32_t c16) {      /* and back; 31 for a code smc_class16 never returns */
    const uint32_t inc = (c16 >> 4) & 1u, idx = c16 & 15u;
    if (idx == 0u) return inc;
    if (idx < 3u) return 2u + 2u * (idx - 1u) + inc;
    if (!inc) return idx < 7u ? 6u + 8u * ((idx - 3u) >> 1) + ((idx - 3u) & 1u) : 31u;
    return idx < 15u ? 6u + 8u * ((idx - 3u) / 6u) + 2u + (idx - 3u) % 6u : 31u;
}
/* a 32-bit read word (allele < 16, quality < 64, a known class) as a 16-bit one, and back */
SMC_HOST_DEVICE static inline uint32_t smc_read_word16(uint32_t w) {
    const uint32_t c = smc_class16(w >> SMC_RW_CLASS_SHIFT);
    return (w & 15u) | ((w & SMC_RW_NF) ? 16u : 0u) | ((c >> 4) & 1u) << 5 | (c & 3u) << 6 | ((w >> 8) & 63u) << 8 | ((c >> 2) & 3u) << 14;
}
SMC_HOST_DEVICE static inline uint32_t smc_read_word32(uint32_t h) {
    const uint32_t c16 = ((h >> 5) & 1u) << 4 | ((h >> 6) & 3u) | ((h >> 14) & 3u) << 2, cls = smc_class16_inv(c16);
    return (h & 15u) | ((h >> 8) & 63u) << 8 | ((h & 16u) ? SMC_RW_NF : 0u) | smc_class_bits(cls) | cls << SMC_RW_CLASS_SHIFT;
}

/* The numeric arguments vc() receives (smCounter.py:274) that the device path needs, plus the two
 * values vc() derives before the pileup loop. mismatchThr and hpLen are consumed on the host
 * (feature extraction / reference-sequence test). */
typedef struct smc_params {
    int32_t min_bq;      /* minBQ */
    int32_t min_mq;      /* minMQ */
    int32_t mt_drop;     /* mtDrop */
    int32_t primer_dist; /* primerDist */
    int32_t ds;          /* maxMT if > 0 else int(round(2.0 * mtDepth))   smCounter.py:486 */
    int32_t reserved;
    double smt;          /* strong-MT threshold 2.0 / 3.0 / 4.0 by rpb    smCounter.py:302-308 */
    double mismatch_thr; /* mismatchThr: consumed by the feature extraction (flag bit / read class); here so that a run can
                          * be checked against the parameters its planes were built with */
} smc_params;

/* 15-bit fingerprint (never 0) of the four parameters the feature extraction folds into the planes (read class, the
 * mismatch-ok flag, the in-deletion quality).  Stored in smc_locus.flags bits 1-15 by whoever builds a batch; smc_plan_create
 * requires one common non-zero value over the batch, smc_plan_run compares it with the fingerprint of the run's smc_params
 * and returns SMC_E_INPUT when they differ (planes built for one parameter set give silently wrong rows under another). */
static inline uint16_t smc_param_fingerprint(int32_t min_bq, int32_t min_mq, double mismatch_thr, int32_t primer_dist) {
    union { double d; uint64_t u; } cv;
    cv.d = mismatch_thr;
    uint64_t h = 0x9E3779B97F4A7C15ull;
    const uint64_t w[4] = {(uint64_t)(uint32_t)min_bq, (uint64_t)(uint32_t)min_mq, cv.u, (uint64_t)(uint32_t)primer_dist};
    for (int i = 0; i < 4; ++i) {
        h ^= w[i];
        h *= 0xBF58476D1CE4E5B9ull;
        h ^= h >> 29;
    }
    const uint16_t fp = (uint16_t)((h >> 17) & 0x7FFFu);
    return fp ? fp : (uint16_t)1;
}
#define SMC_LF_FP_SHIFT 1 /* smc_locus.flags bits 1-15: smc_param_fingerprint of the parameters the planes were built with */

/* One per locus, 32 bytes. Reads of locus l occupy plane slots [4*read_off4, 4*read_off4 + n_reads)
 * (every locus starts on a 4-read boundary). Within the locus the reads are SORTED barcode-major:
 * by barcode id, then by fragment slot, then by pileup order (stable), so a barcode's reads are one
 * contiguous run, a fragment's reads are adjacent, and the first-seen mate still comes first
 * (smCounter.py:468-479 depends on that order only within a fragment).
 * umi ids are < n_umi (dense, order of first appearance in the pileup). frag ids are locus-level
 * fragment slots < n_frag, grouped by barcode: the fragments of barcode u occupy one contiguous slot
 * range, ranges ordered by u, each fragment's slot fixed by its first appearance within the barcode;
 * n_frag = number of distinct fragments (= allFrag, smCounter.py:483).
 * umi_start[umi_off + u], u = 0..n_umi, is the index (relative to the locus) of barcode u's first read;
 * the last entry equals n_reads. Every barcode has at least one read. Given barcode-major reads, umi_start
 * determines the umi plane; the kernels use umi_start and the fragment-start bit of the read words and load neither the umi
 * plane nor - with the read class (smc_read_class above) - the dist plane: `umi` and `dist` may be NULL in smc_plan_run /
 * smc_call_batch_host (they are the raw fields the CPU restatement checks the classes and the order against). Checked per
 * locus, violations flag the row SMC_ST_BAD_INPUT: fragment slots dense and ascending, 0 .. n_frag - 1 (smc_pack_words),
 * allele < n_alleles, a known read class, umi_start ascending and covering [0, n_reads), every barcode starting a fragment.
 * Base qualities are Phred values <= 126 (BAM holds 0..93); larger bytes are clamped to 126. */
/* smc_locus.flags */
#define SMC_LF_SAMPLED 1u /* the host has applied the reference's down-sampling (smCounter.py:496-498): barcodes whose
                           * umi_start entry has bit 31 set are keys of bcDict the sample dropped; the number kept must be
                           * min(#keys, ds), else the row is SMC_ST_BAD_INPUT. Without this flag a locus over the cap gets
                           * the non-parity stand-in (ds lowest barcode ids). */
#define SMC_USTART_DROPPED 0x80000000u
typedef struct smc_locus {
    uint32_t read_off4; /* first plane slot / 4 */
    uint32_t umi_off;   /* first entry of this locus in the umi_start array */
    int32_t n_reads;
    int32_t n_umi;
    int32_t n_frag;
    uint8_t ref_allele; /* allele id of the reference base, 255 if it is not a key */
    uint8_t n_alleles;  /* size of the locus's allele table, <= SMC_MAX_ALLELES */
    uint16_t flags;
    uint64_t snp_mask;  /* bit a set: allele a is a single letter (TYPE 'SNP', smCounter.py:107) */
} smc_locus;

/* per-allele tallies of the pileup scan; index names follow the reference's dicts */
enum {
    SMC_T_CNT = 0,   /* alleleCnt        smCounter.py:379,401,459 */
    SMC_T_FWD,       /* forwardCnt       :389,411,457 */
    SMC_T_REV,       /* reverseCnt       :387,409,455 */
    SMC_T_LOWQ,      /* lowQReads        :428-429 */
    SMC_T_R1N,       /* len(r1BcEndPos)  :438-439 */
    SMC_T_R1LE,      /* #r1BcEndPos <= 20          :234 */
    SMC_T_R2N,       /* len(r2BcEndPos)  :449-451 */
    SMC_T_R2BCLE,    /* #r2BcEndPos <= 20          :244 */
    SMC_T_R2PRLE,    /* #r2PrimerEndPos <= primerDist  :256 */
    SMC_T_CONCORD,   /* concordPairCnt   :475-476 */
    SMC_T_DISCORD,   /* discordPairCnt   :479 */
    SMC_T_PAD,
    SMC_NT = 12
};

typedef struct smc_cand {
    int32_t allele;      /* allele id, -1 when the candidate does not exist */
    int32_t flt_applied; /* 1 if filterVariants ran (PI >= 5 and TYPE in SNP/INDEL, :549/:563) */
    uint32_t flt;        /* SMC_F_* bits decided on the device */
    int32_t vmf_lt_099;  /* 1.0*MTCnt/usedMT < 0.99, the gate of HP and LowC (:198,:202) */
    int32_t vdp;         /* alleleCnt[allele] */
    int32_t vmt;         /* MTCnt[allele] */
    int32_t vsm;         /* strongMTCnt[allele] */
    int32_t pad;
    int32_t tal[SMC_NT]; /* tallies of this allele: alleleCnt and the pair counts (SMC_T_CNT, _CONCORD, _DISCORD) always; the eight
                          * that only filterVariants reads (SMC_T_FWD .. SMC_T_R2PRLE) where flt_applied - the device does not
                          * count them for a locus no candidate of which reaches the filters */
    double pi;           /* finalDict[allele], unrounded */
    double p_sb, p_r1, p_r2, p_pr; /* Fisher two-sided p-values of the four tests, NaN if not run */
} smc_cand;

/* One per locus: everything the 45-column row (smCounter.py:575-600) is printed from. 432 bytes. */
typedef struct smc_row {
    int32_t status;
    int32_t n_touched;   /* number of keys in finalDict */
    int32_t cvg, all_frag, all_mt, used_frag, used_mt; /* DP FR MT UFR UMT */
    int32_t mt3, mt5, mt7, mt10;
    int32_t max_allele, second_allele; /* maxBase / secondMaxBase (:535-537) */
    int32_t biallelic;   /* condition of :555 */
    int32_t dp[4], umt[4], vsm[4]; /* A,T,G,C: alleleCnt, MTCnt, strongMTCnt */
    double pi[4];        /* A,T,G,C: finalDict, unrounded */
    uint64_t touched_mask;
    int32_t ref_tal[SMC_NT]; /* tallies of the reference allele: alleleCnt always, the filter-only eight where a candidate has
                              * flt_applied; its pair counts are not kept (nothing reads them: the DP filter looks at the candidate's) */
    smc_cand cand[2];    /* [0] origAlt (:541), [1] secondMaxBase when biallelic */
} smc_row;

/* What travels between GPUs: the part of smc_row that the 45-column row is PRINTED from (smCounter.py:575-600; rows.py),
 * 168 bytes instead of 432.  The tallies and Fisher p-values stay behind: filterVariants has already run on the device
 * and left its verdict in the FILTER bits.  Replaces the pickled result strings of the reference's pool
 * (`[p.get() for p in results]`, smCounter.py:685). */
#define SMC_WIRE_BIALLELIC 0x10000u   /* smc_wire_row.status bit 16: smc_row.biallelic; bits 0-15: smc_row.status */
#define SMC_WIRE_FLT_MASK 0x3FFu      /* smc_wire_cand.flags bits 0-9: SMC_F_* */
#define SMC_WIRE_FLT_APPLIED 0x400u   /* bit 10: smc_cand.flt_applied */
#define SMC_WIRE_VMF_LT_099 0x800u    /* bit 11: smc_cand.vmf_lt_099 */
typedef struct smc_wire_cand {
    int16_t allele;  /* -1: no such candidate */
    uint16_t flags;
    int32_t vdp, vmt, vsm;
    double pi;
} smc_wire_cand;
typedef struct smc_wire_row {
    uint32_t status;
    int32_t cvg, all_frag, all_mt, used_frag, used_mt;
    int32_t mt3, mt5, mt7, mt10;
    int32_t dp[4], umt[4], vsm[4];
    double pi[4];
    smc_wire_cand cand[2];
} smc_wire_row;

/* ---- input of the device plane builder (smc_build_planes): a run's alignments as a structure of arrays, one entry per
 * ALIGNMENT, produced by the decoder (libsmc_bam.so: smc_bam_alignments, smcounter_host.h) */
#define SMC_DA_R1 1u
#define SMC_DA_R2 2u
#define SMC_DA_REV 4u
#define SMC_DA_MMOK 16u /* mismatchPer100b <= mismatchThr (smCounter.py:352-356, third term of incCond :378) */
typedef struct smc_dev_aln {
    int32_t pos, end;          /* 0-based reference span [pos, end) */
    uint32_t cig_off, seq_off; /* first CIGAR word / first base (and quality) in the pools */
    uint16_t n_cig;
    uint8_t oflag, mapq;
    uint16_t left_sp, qalen;   /* leading soft clip, query_alignment_length (:336-349, :434-448) */
    uint16_t l_seq, pad;
    uint32_t bc_gid, pair_gid; /* run-wide ids of the barcode and of (barcode, read id), dense, in file order */
} smc_dev_aln;
typedef struct smc_dev_locus {
    uint32_t w0, w1;           /* alignments [w0, w1) are the candidates that can cover the locus */
    uint32_t slot_off, n;      /* first (4-aligned) read slot of the locus in the run's planes, pileup depth */
} smc_dev_locus;

typedef struct smc_ctx smc_ctx;
typedef struct smc_plan smc_plan;

int smc_abi_version(void);
const char* smc_last_error(void);
int smc_row_size(void);   /* sizeof(smc_row), for binding self-checks */
int smc_locus_size(void); /* sizeof(smc_locus) */

/* Number of gfx950 devices visible; does not initialise any of them. */
int smc_device_count(void);

/* (diagnostic, no GPU needed) the read-class table the kernel uses: out[2c], out[2c+1] for class c < 32 - nine 5-bit
 * tally increments in SMC_T_* order (six in the first word, three in the second); second word bit 31 = incCond. */
void smc_class_table(uint32_t* out);

/* Bind a context to one device (one per process / host thread). */
int smc_create(int device, smc_ctx** out);
void smc_destroy(smc_ctx* ctx);

/* Build a launch plan for a batch: bins loci by on-chip table size, uploads the descriptors and the
 * bin index lists. `loci` is host memory; it is copied. Replaces the construction of the
 * apply_async task list, smCounter.py:684. */
int smc_plan_create(smc_ctx* ctx, const smc_locus* loci, int64_t n_loci, smc_plan** out);
void smc_plan_destroy(smc_plan* plan);
/* The same plan for a batch whose descriptors are already on the DEVICE (smc_build_planes has just written them): the binning
 * runs there, only a small record (and the few loci deep enough to be cut into parts) comes back.  Enqueued on `stream`
 * behind whatever wrote `d_loci`; synchronises on it.  `d_loci` stays the caller's and must outlive the plan.  Plans of one
 * context may be made on different streams one after the other (the record they sum into is the context's: a plan's kernels wait
 * for the plan before it to be through with it); a plan is run by one stream at a time. */
int smc_plan_create_dev(smc_ctx* ctx, const smc_locus* d_loci, int64_t n_loci, void* stream, smc_plan** out);
/* (ABI 8) The same plan WITHOUT the host in the loop: nothing here waits for the device.  The launches are sized from the record of
 * the context's last plan (scaled to this batch's loci, with room) instead of this batch's own, which the device sums while the
 * host goes on; whether the batch FITS those sizes is decided on the device: one that does not launches nothing.  So, after the
 * caller has synchronised the stream for the rows, smc_plan_spec_ok(plan, &ok) says whether they are this batch's (ok = 1) or the
 * plan has to be made again - by this function, which then goes the exact way (as it does for a context's first plan: there is
 * nothing to size it from), or by smc_plan_create_dev - and run again (ok = 0).  For callers that make plan after plan of batches of
 * one kind - the runs of a BAM (smCounter.py:683-685 submits them one after the other), the steps of the bench: the host thread no
 * longer waits for the build of the planes, the device no longer idles around the read-back.  `prm`: the parameters the planes were
 * built with (every descriptor has to carry their fingerprint).  smc_plan_run (raw-field planes) is not for such a plan. */
int smc_plan_create_dev_spec(smc_ctx* ctx, const smc_params* prm, const smc_locus* d_loci, int64_t n_loci, void* stream, smc_plan** out);
int smc_plan_spec_ok(smc_plan* plan, int* ok);
/* Plans made by smc_plan_create_dev_spec in this context, how many of them went the exact way, how many the device found not to
 * fit (reads a counter on the device: synchronises it) - for a caller that destroys its plans unchecked (a benchmark loop). */
int smc_plan_spec_counts(smc_ctx* ctx, int64_t* made, int64_t* exact, int64_t* not_fitting);
/* Forget what the context's next such plan would be sized from (it then goes the exact way): for a caller that knows its next batch
 * is of another kind than the last one. */
int smc_plan_hint_reset(smc_ctx* ctx);

/* (ABI 8) The NON-parity down-sampling of loci over the barcode cap, on the device.  smCounter.py:496-498 seeds Python 2's
 * Mersenne twister with the position STRING and samples bcDict's keys in Python 2's dict order; the parity path reproduces that on
 * the host from the barcode texts and marks the dropped keys in umi_start (SMC_LF_SAMPLED / SMC_USTART_DROPPED above).  This call
 * leaves the same kind of marks without the host's sampler: every key of bcDict of a locus with n_umi > ds (and no marks yet) gets
 * the 64-bit value made of words 0 and 1 of Philox4x32-10(counter = (identity lo, identity hi, 0, 0), key = the two halves of
 * (position ^ seed)); the ds smallest stay (ties: the lower index).  `d_ident`: one 64-bit identity per umi_start entry (read
 * only at loci over the cap) - e.g. a hash of the barcode's text: the sample then depends on (seed, position, barcode) alone, not on
 * how the caller numbered the barcodes or cut the file into batches; NULL: the barcode's index in its locus.  With
 * `d_ident_index` (one uint32 per umi_start entry: what smc_build_planes leaves in u_gid at such loci - the run-wide barcode id)
 * `d_ident` is a table by that index instead.  Counter-based - no
 * state, no order - so the marks do not depend on the launch; oracle/smc_oracle.c restates it (smc_oracle_philox_marks).  NOT what
 * the reference samples: rows of such loci carry SMC_ST_DOWNSAMPLED and differ from smCounter's.  d_pos[l]: the 1-based position
 * of locus l; d_status: a word the call ORs bits into - 1: a locus with more than 2^18 barcodes was left to the stand-in (the ds
 * lowest indices).  Enqueued on `stream` after whatever wrote the words, before the batch's plan runs. */
int smc_philox_marks(smc_ctx* ctx, const smc_params* prm, smc_locus* d_loci, int64_t n_loci, const int64_t* d_pos, const void* d_words,
                     int word_bits, uint32_t* d_umi_start, const uint64_t* d_ident, const uint32_t* d_ident_index, uint64_t seed,
                     uint32_t* d_status, void* stream);
void smc_philox4x32_10_host(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]);
/* number of kernel launches one smc_plan_run issues, and bytes of device scratch it holds */
int smc_plan_info(const smc_plan* plan, int32_t* n_launches, int64_t* scratch_bytes);

/* Optional: bracket the dominant k_call_v2 launch (the bin holding most reads) of each
 * smc_plan_run with a HIP event pair on the run's stream, kept in a ring of `slots` pairs
 * (0 disables). smc_plan_kernel_ms synchronises on the recorded pairs and returns that launch's
 * mean duration over the last min(runs, slots) runs, with the loci and reads one launch covers. */
int smc_plan_set_timing(smc_plan* plan, int slots);
int smc_plan_kernel_ms(smc_plan* plan, float* avg_ms, int32_t* n_samples, int64_t* n_loci, int64_t* n_reads);

/* Run the hot path over the batch. meta/umi/frag/dist, umi_start and rows are DEVICE pointers
 * (planes n_slots x uint32 each, umi and dist may be NULL - the kernels do not read them; umi_start sum(n_umi + 1) x uint32; rows n_loci x smc_row). `stream` is a hipStream_t (NULL = default
 * stream). Asynchronous: returns after enqueueing. */
int smc_plan_run(smc_plan* plan, const smc_params* params, const uint32_t* meta,
                 const uint32_t* umi, const uint32_t* frag, const uint32_t* dist,
                 const uint32_t* umi_start, smc_row* rows, void* stream);
/* The same on read words (smc_read_word above; n_slots x uint32): the form the kernels read.  smc_plan_run is
 * smc_pack_words into a buffer the plan keeps + this call.  smc_locus.n_frag is taken as given (allFrag of the row): the
 * words carry the fragment boundaries, not their count. */
int smc_plan_run_words(smc_plan* plan, const smc_params* params, const uint32_t* words, const uint32_t* umi_start,
                       smc_row* rows, void* stream);
/* The same on 16-bit read words (smc_read_word16; n_slots x uint16, a locus's first word 8-byte aligned: read_off4 counts
 * quads of reads as before) - what smc_build_planes_w16 writes. */
int smc_plan_run_words16(smc_plan* plan, const smc_params* params, const uint16_t* words16, const uint32_t* umi_start,
                         smc_row* rows, void* stream);
/* meta + frag planes of the plan's batch -> read words (device pointers; asynchronous on `stream`) */
int smc_pack_words(smc_plan* plan, const uint32_t* meta, const uint32_t* frag, uint32_t* words, void* stream);

/* Convenience for callers without their own device buffers: host pointers in, host rows out
 * (synchronous; does H2D, smc_plan_run, D2H). */
int smc_call_batch_host(smc_ctx* ctx, const smc_params* params, const smc_locus* loci,
                        int64_t n_loci, const uint32_t* meta, const uint32_t* umi,
                        const uint32_t* frag, const uint32_t* dist, int64_t n_slots,
                        const uint32_t* umi_start, int64_t n_umi_start, smc_row* rows_out);

/* Pack n rows (DEVICE pointers) into wire rows on `stream` (asynchronous), for the gather to the writing rank; and the
 * inverse on the HOST (no GPU needed): the printed fields are restored exactly, everything else is zero (p-values NaN,
 * n_touched / max_allele / second_allele / touched_mask / tallies: not carried). */
int smc_wire_row_size(void);
int smc_pack_rows(smc_ctx* ctx, const smc_row* rows, int64_t n, smc_wire_row* wire, void* stream);
int smc_unpack_rows(const smc_wire_row* wire, int64_t n, smc_row* rows);

/* Build the planes of a run ON THE DEVICE from its alignments (the device half of the feature extraction; the host half is
 * smc_bam_alignments): per locus the covering alignments in file order, per read the CIGAR walk, allele, quality, flags,
 * end distances and read class (smCounter.py:316-366, :371-452), barcode / fragment ids by first appearance (:462-471),
 * the barcode-major order, umi_start and the descriptor - byte for byte what smc_bam_planes builds on the host.
 * Everything in `in` and every output is a DEVICE pointer.  Outputs: the read words (the run's slots start at slot_base) and /
 * or the four raw-field planes (any of the five may be NULL - not written then - as long as `words`, or `meta` and `frag`, is
 * there; the raw-field planes are for checks: with only `words` the walk stages and stores a quarter of the bytes), umi_start / u_gid / u_finc (sized slots + loci of the batch; locus l of the run uses
 * [umi_base + slot_off(l) + l, ... + n_umi(l)]; u_gid / u_finc - run-wide barcode id and first INCLUDED pileup index per
 * barcode - are filled only for loci with more barcodes than params->ds: what the host needs for the reference's
 * down-sampling, :496-498), loci[n_loci] (read_off4 / umi_off already batch-relative), and for every allele beyond the six
 * fixed ones five words in xlist (locus, allele id, alignment, query position, indel; smc_bam_allele_key turns them into the
 * key text).  counters[0] = entries appended to xlist, counters[1] = status bits (0 = fine; see csrc/k_build_planes.inc:
 * 1 depth mismatch (a locus's reads differ from loc[].n, or exceed in->max_depth / smc_build_max_depth()), 2 extras overflow, 4 base quality > 126, 8 more
 * than 64 alleles) - the caller falls back to smc_bam_planes for the run when it is not 0.  Asynchronous on `stream`. */
typedef struct smc_build_in {
    const smc_dev_aln* aln; const uint32_t* cig;
    const uint8_t* bq;   /* the bases and their qualities as ONE stream of (letter, quality) byte pairs: base i of the pool (smc_dev_aln.
                          * seq_off counts bases) has its ASCII letter at byte 2i and its quality at byte 2i + 1.  The walk reads the 64
                          * positions of an alignment under a tile as 128 consecutive bytes - one or two cache lines; as two separate
                          * pools it touched two to four (measured: 1.79 -> 1.41 ms on the 3000x shape).  2-byte aligned, and readable
                          * 128 bytes past the last pair */
    const smc_dev_locus* loc; const uint8_t* refseq;
    int32_t start0, n_loci, n_bc, n_pair;
    int32_t max_depth;   /* reads at the run's deepest locus (the caller counted them for loc[].n); checked against
                          * smc_build_max_depth() */
    int32_t n_aln;       /* entries of aln[] (every loc[].w1 must stay within them); < 0: not checked */
    const smc_dev_locus* loc_host; /* the same loc[] in HOST memory (the decoder fills it there): sizes the sort and the launch
                                    * grids without a round trip; NULL = the library copies loc[] back itself (synchronous) */
} smc_build_in;
int smc_build_max_depth(void);
/* Optional: bracket the walk that writes the planes (k_bp_tiles<true>, the dominant kernel of smc_build_planes) with a HIP event
 * pair on the run's stream, in a ring of `slots` pairs (0 disables); smc_build_kernel_ms synchronises on the recorded pairs and
 * returns that launch's mean duration over the last min(runs, slots) calls. */
int smc_build_set_timing(smc_ctx* ctx, int slots);
int smc_build_kernel_ms(smc_ctx* ctx, float* avg_ms, int32_t* n_samples);
int smc_build_planes(smc_ctx* ctx, const smc_params* params, const smc_build_in* in, uint32_t slot_base, uint32_t umi_base,
                     uint32_t* words, uint32_t* meta, uint32_t* umi, uint32_t* frag, uint32_t* dist, uint32_t* umi_start,
                     uint32_t* u_gid, uint32_t* u_finc, smc_locus* loci, uint32_t* xlist, uint32_t xcap, uint32_t* counters, void* stream);

/* The same run with the read words in 16 bits (smc_read_word16 above) and nothing else written but umi_start / u_gid / u_finc /
 * loci / xlist: the walk stores half the bytes (the 3000x panel shape: 1.18 -> 1.00 ms) and the locus kernels load half.
 * counters[1] has one more status bit, 32: the run has an allele id beyond 15 at some locus or a base quality beyond 63 - the
 * words are not to be used, the caller builds the run again with smc_build_planes.  params->min_bq must lie in 0 .. 63. */
int smc_build_planes_w16(smc_ctx* ctx, const smc_params* params, const smc_build_in* in, uint32_t slot_base, uint32_t umi_base,
                         uint16_t* words16, uint32_t* umi_start, uint32_t* u_gid, uint32_t* u_finc, smc_locus* loci,
                         uint32_t* xlist, uint32_t xcap, uint32_t* counters, void* stream);

/* The context keeps the device blocks of destroyed plans for the next plan (at most 64 blocks / 8 GB; the oldest goes first).
 * smc_pool_trim waits for the plans' last runs and returns every pooled block to the runtime. */
int smc_pool_trim(smc_ctx* ctx);

/* Device memory for callers without a GPU runtime of their own (the Python command line uses these instead of importing
 * PyTorch: about a second of start-up): allocation, synchronous copies, device synchronisation. */
/* smc_mem_alloc is hipMalloc.  (For most of round 5 it backed blocks of 256 MB and more by HIP virtual memory over physical
 * handles of 64 MB: such a range can LOSE what is written to it on this ROCm - csrc/host_abi.inc, vmm_alloc; scripts/vmm_stress.py;
 * SMC_VMM_CHUNK_MB=<MB> in the environment brings it back for measurements.)  Pointers from it are freed with smc_mem_free only. */
int smc_mem_alloc(smc_ctx* ctx, int64_t bytes, void** out);
void smc_mem_free(smc_ctx* ctx, void* p);
/* For an array the plane builder's walk WRITES (the read words of a batch): which physical pages hold it moves that kernel by up to
 * 10 % (the same from launch to launch; no counter of translation, L2 or request counts tells a fast allocation from a slow one,
 * but a write-only kernel with the walk's pattern does: smc_mem_write_probe).  smc_mem_alloc_best makes up to `tries` allocations
 * (hipMalloc blocks, all held until the choice is made so that each gets other pages), times that pattern into each (~ 4 ms per candidate)
 * and keeps the fastest.  info (may be NULL): [0] the kept block's probe time in ms, [1] the slowest candidate's, [2] candidates
 * tried.  Freed with smc_mem_free. */
int smc_mem_alloc_best(smc_ctx* ctx, int64_t bytes, int tries, void** out, float* info);
/* (smc_mem_write_probe OVERWRITES [p, p + bytes) with its pattern: for blocks that hold nothing yet.) */
int smc_mem_write_probe(smc_ctx* ctx, void* p, int64_t bytes, float* ms);
/* page-locked host memory: copies to and from it run at the link's rate (pageable memory goes through a staging copy) */
int smc_mem_alloc_host(smc_ctx* ctx, int64_t bytes, void** out);
void smc_mem_free_host(smc_ctx* ctx, void* p);
int smc_mem_h2d(smc_ctx* ctx, void* dst_device, const void* src_host, int64_t bytes);
int smc_mem_d2h(smc_ctx* ctx, void* dst_host, const void* src_device, int64_t bytes);
int smc_device_sync(smc_ctx* ctx);

/* HIP-event timing helpers so a host language without HIP bindings can time the stream the
 * kernels run on. */
int smc_event_create(void** ev);
int smc_event_record(void* ev, void* stream);
int smc_event_elapsed_ms(void* start, void* stop, float* ms); /* synchronises on `stop` */
void smc_event_destroy(void* ev);

#ifdef __cplusplus
}
#endif
#endif /* SMCOUNTER_HIP_H */
