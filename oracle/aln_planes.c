/* aln_planes.c - CPU restatement of the per-pileup-read part of smCounter's vc() (TEST INFRASTRUCTURE: only tests/, bench.py's
 * parity passes and __graft_entry__.smoke() may load it; the product builds its planes with csrc/k_build_planes.inc on the GPU
 * or csrc/smc_bam.cpp on the host).
 *
 * From a run's ALIGNMENTS (the structure of arrays smc_bam_alignments / the synthetic generator hand to smc_build_planes:
 * smc_dev_aln, CIGAR words, one base letter and one quality per query position, per locus the window of the file that can cover
 * it) to the raw-field planes of include/smcounter_hip.h, one locus at a time, the way the reference does it:
 *   smCounter.py:316      the pileup column: every alignment that spans the position, in file order; per read
 *                         query_position / is_del / indel as samtools' resolve_cigar2 gives them (the peek at the next
 *                         operation from the last base of a match or a deletion);
 *   :336-366              leading soft clip, query_alignment_length, pairOrder, strand (taken from the decoded record);
 *   :371-425              the allele: 'INS|..' / 'DEL|..' keys at an indel start, 'DEL' inside a deletion (quality := minBQ,
 *                         :416-418), else the base letter; ids 0-5 = A,T,G,C,N,'DEL', further keys numbered by first sight;
 *   :378                  incCond; :432-452 the end distances of a regular base;
 *   :462-471              barcodes and, within a barcode, read ids numbered by first appearance at the locus.
 * The reads of a locus are then laid out barcode-major (barcode, fragment, pileup order) with the fragment slot, the read
 * class (smc_read_class), umi_start and the descriptor - the contract of smcounter_hip.h.
 *
 * Pinned by tests/test_aln_planes.py: equal (up to barcode / fragment numbering, which this file shares with the host builders
 * anyway) to smc_bam_planes on the three reference-generated BAM fixtures and on synthetic runs written as BAMs. */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "../include/smcounter_hip.h"

typedef struct { int qpos, indel, isdel; } res_t;

/* query position / in-deletion / indel-follows for reference position p (htslib bam_plp resolve_cigar2 semantics, as the
 * decoders of this repository restate them) */
static res_t resolve(const uint32_t* c, int n_cig, int pos, int p, int l_seq) {
    res_t r = {-1, 0, 0};
    int x = pos, y = 0;
    for (int ci = 0; ci < n_cig; ++ci) {
        const int op = (int)(c[ci] & 15u), len = (int)(c[ci] >> 4);
        const int ref_op = op == 0 || op == 7 || op == 8, gap_op = op == 2 || op == 3;
        if (ref_op || gap_op) {
            if (p >= x && p < x + len) {
                r.qpos = ref_op ? y + (p - x) : y;
                r.isdel = gap_op;
                if (p == x + len - 1 && ci + 1 < n_cig) {
                    const int nop = (int)(c[ci + 1] & 15u), nlen = (int)(c[ci + 1] >> 4);
                    r.indel = nop == 1 ? nlen : nop == 2 ? -nlen : 0;
                }
                break;
            }
            x += len;
            if (ref_op) y += len;
        } else if (op == 1 || op == 4) y += len;
    }
    if (r.isdel && r.indel != 0 && r.qpos >= l_seq) r.indel = 0;
    return r;
}

typedef struct { uint32_t key, val; } slot_t;
static uint32_t* ht_find(slot_t* t, uint32_t mask, uint32_t key, int* fresh) {       /* key + 1 stored: 0 = empty */
    uint32_t h = (key * 0x9E3779B1u) & mask;
    for (;;) {
        if (t[h].key == 0u) { t[h].key = key + 1u; *fresh = 1; return &t[h].val; }
        if (t[h].key == key + 1u) { *fresh = 0; return &t[h].val; }
        h = (h + 1u) & mask;
    }
}

static uint64_t allele_hash(const smc_dev_aln* a, const uint8_t* seq, int qpos, int indel) {
    uint64_t h = 1469598103934665603ull;
    const uint8_t* s = seq + a->seq_off;
#define MIX(b) do { h ^= (uint64_t)((b) & 0xffu); h *= 1099511628211ull; } while (0)
    MIX(indel > 0 ? 'I' : indel < 0 ? 'D' : 'L');
    MIX(s[qpos]);
    if (indel > 0) { for (int k = qpos + 1; k < (int)a->l_seq && k < qpos + 1 + indel; ++k) MIX(s[k]); }
    else if (indel < 0) { const uint32_t l = (uint32_t)(-indel); MIX(l); MIX(l >> 8); MIX(l >> 16); MIX(l >> 24); }
#undef MIX
    return h;
}
static int fixed_allele(uint32_t c) { return c == 'A' ? 0 : c == 'T' ? 1 : c == 'G' ? 2 : c == 'C' ? 3 : c == 'N' ? 4 : -1; }

typedef struct { uint32_t meta, umi, fidx, cls, dist, order; } read_t;

/* planes of loci [l0, l1) of the run.  loc[l].slot_off / .n as the decoder counted them; locus l's reads go to plane slots
 * slot_base + slot_off .., its umi_start entries to umi_base + slot_off + l ..  Returns 0, or -1 out of memory, -2 a locus whose
 * depth differs from loc[l].n, -3 a pileup that begins with an alignment flagged neither READ1 nor READ2, -4 more than 64 alleles, -5 a quality > 126 */
int smc_aln_planes(const smc_dev_aln* aln, const uint32_t* cig, const uint8_t* seq, const uint8_t* qual, const smc_dev_locus* loc,
                   const uint8_t* refseq, int32_t start0, int32_t l0, int32_t l1, int32_t min_bq, int32_t min_mq, int32_t primer_dist,
                   uint32_t fp, uint32_t slot_base, uint32_t umi_base, uint32_t* meta, uint32_t* umi, uint32_t* frag, uint32_t* dist,
                   uint32_t* umi_start, smc_locus* loci) {
    uint32_t cap = 0;
    for (int l = l0; l < l1; ++l) if (loc[l].n > cap) cap = loc[l].n;
    uint32_t tsz = 64;
    while (tsz < 2u * (cap + 1u)) tsz <<= 1;
    read_t* rd = (read_t*)malloc(sizeof(read_t) * (cap + 1u));
    slot_t* tb = (slot_t*)malloc(sizeof(slot_t) * tsz);
    slot_t* tf = (slot_t*)malloc(sizeof(slot_t) * tsz);
    uint32_t* nfr = (uint32_t*)malloc(4u * (cap + 1u));       /* fragments per barcode, then their first slot */
    uint32_t* perm = (uint32_t*)malloc(4u * (cap + 1u));
    if (!rd || !tb || !tf || !nfr || !perm) { free(rd); free(tb); free(tf); free(nfr); free(perm); return -1; }
    int rc = 0;
    for (int l = l0; l < l1 && rc == 0; ++l) {
        const int p = start0 + l;
        const uint32_t so = slot_base + loc[l].slot_off, uo = umi_base + loc[l].slot_off + (uint32_t)l;
        memset(tb, 0, sizeof(slot_t) * tsz); memset(tf, 0, sizeof(slot_t) * tsz);
        uint32_t n = 0, n_umi = 0, n_frag = 0, n_extra = 0;
        int pair_r2 = -1;
        uint64_t xhash[58];
        uint32_t xsingle[58], xletter[58];
        for (uint32_t a = loc[l].w0; a < loc[l].w1 && rc == 0; ++a) {
            const smc_dev_aln* A = aln + a;
            if (!(A->pos <= p && p < A->end)) continue;
            if (n >= loc[l].n) { rc = -2; break; }
            /* pairOrder (smCounter.py:359-362): R2 wins over R1; neither - the value the previous pileup read left behind */
            if (A->oflag & (SMC_DA_R1 | SMC_DA_R2)) pair_r2 = (A->oflag & SMC_DA_R2) != 0;
            else if (pair_r2 < 0) { rc = -3; break; }
            const res_t r = resolve(cig + A->cig_off, A->n_cig, A->pos, p, A->l_seq);
            const int gap = (r.isdel && r.indel == 0) || r.qpos < 0;
            const int kind = r.indel > 0 ? 2 : r.indel < 0 ? 3 : r.isdel ? 1 : 0;
            uint32_t site = 0, bq = 0;
            if (!gap) { site = seq[A->seq_off + (uint32_t)r.qpos]; bq = qual[A->seq_off + (uint32_t)r.qpos]; }
            if (bq > 126u) { rc = -5; break; }
            /* the allele (smCounter.py:371-425) */
            uint32_t al;
            if (gap) al = 5u;
            else if (r.indel == 0 && fixed_allele(site) >= 0) al = (uint32_t)fixed_allele(site);
            else {
                const uint64_t h = allele_hash(A, seq, r.qpos, r.indel);
                uint32_t k = 0;
                while (k < n_extra && xhash[k] != h) ++k;
                if (k == n_extra) {
                    if (n_extra >= 58u) { rc = -4; break; }
                    xhash[k] = h; xsingle[k] = r.indel == 0; xletter[k] = site; ++n_extra;
                }
                al = 6u + k;
            }
            const int r2 = pair_r2, rev = (A->oflag & SMC_DA_REV) != 0, mmok = (A->oflag & SMC_DA_MMOK) != 0;
            uint32_t dbc = 0, dpr = 0;
            if (kind == 0) {                                                  /* :432-452 */
                const int rel = r.qpos - (int)A->left_sp, far = (int)A->qalen - rel;
                int d1 = r2 ? (rev ? rel : far) : (rev ? far : rel), d2 = r2 ? (rev ? far : rel) : 0;
                d1 = d1 < 0 ? 0 : d1 > 65535 ? 65535 : d1; d2 = d2 < 0 ? 0 : d2 > 65535 ? 65535 : d2;
                dbc = (uint32_t)d1; dpr = (uint32_t)d2;
            }
            const int bq_ok = (int)bq >= min_bq;
            const int inc = (bq_ok || kind == 1) && (int)A->mapq >= min_mq && mmok;   /* incCond, :378 */
            const uint32_t flags = (uint32_t)(r2 ? 1 : 0) | (uint32_t)(rev ? 2 : 0) | (uint32_t)(mmok ? 4 : 0) | (uint32_t)kind << 3;
            read_t* R = rd + n;
            R->meta = al | (kind == 1 ? (uint32_t)min_bq : bq) << 8 | flags << 16 | (uint32_t)A->mapq << 24;
            R->dist = dbc | dpr << 16;
            R->cls = smc_read_class(kind, rev, r2, inc, bq_ok, dbc <= 20u, (int)dpr <= primer_dist);
            R->order = n;
            /* barcode / read id by first appearance at the locus (:462-471) */
            int fresh;
            uint32_t* u = ht_find(tb, tsz - 1u, A->bc_gid, &fresh);
            if (fresh) { *u = n_umi; nfr[n_umi] = 0u; ++n_umi; }
            R->umi = *u;
            uint32_t* f = ht_find(tf, tsz - 1u, A->pair_gid, &fresh);
            if (fresh) { *f = nfr[R->umi]++; ++n_frag; }
            R->fidx = *f;
            ++n;
        }
        if (rc) break;
        if (n != loc[l].n) { rc = -2; break; }
        /* fragment slots: the fragments of barcode u occupy [first(u), first(u) + nfr(u)) */
        uint32_t run = 0;
        for (uint32_t u = 0; u < n_umi; ++u) { const uint32_t c = nfr[u]; nfr[u] = run; run += c; }
        /* barcode-major order: by fragment slot, then pileup order (a counting sort over the slots keeps it stable) */
        uint32_t* scnt = (uint32_t*)calloc((size_t)n_frag + 1u, 4u);
        if (!scnt) { rc = -1; break; }
        for (uint32_t i = 0; i < n; ++i) ++scnt[nfr[rd[i].umi] + rd[i].fidx + 1u];
        for (uint32_t s = 0; s < n_frag; ++s) scnt[s + 1u] += scnt[s];
        for (uint32_t i = 0; i < n; ++i) perm[scnt[nfr[rd[i].umi] + rd[i].fidx]++] = i;
        free(scnt);
        uint32_t last_u = 0xFFFFFFFFu;
        for (uint32_t j = 0; j < n; ++j) {
            const read_t* R = rd + perm[j];
            const uint32_t slot = nfr[R->umi] + R->fidx;
            meta[so + j] = R->meta; umi[so + j] = R->umi; frag[so + j] = slot | R->cls << SMC_FRAG_CLASS_SHIFT; dist[so + j] = R->dist;
            if (R->umi != last_u) { umi_start[uo + R->umi] = j; last_u = R->umi; }
        }
        umi_start[uo + n_umi] = n;
        for (uint32_t j = n; j < ((n + 3u) & ~3u); ++j) { meta[so + j] = 0u; umi[so + j] = 0u; frag[so + j] = 0u; dist[so + j] = 0u; }
        smc_locus L;
        memset(&L, 0, sizeof L);
        L.read_off4 = so >> 2; L.umi_off = uo; L.n_reads = (int32_t)n; L.n_umi = (int32_t)n_umi; L.n_frag = (int32_t)n_frag;
        int ra = fixed_allele(refseq[l]);
        uint64_t mask = 0x1f;
        for (uint32_t k = 0; k < n_extra; ++k)
            if (xsingle[k]) { mask |= 1ull << (6u + k); if (ra < 0 && xletter[k] == (uint32_t)refseq[l]) ra = (int)(6u + k); }
        L.ref_allele = (uint8_t)(ra < 0 ? 255 : ra); L.n_alleles = (uint8_t)(6u + n_extra);
        L.flags = (uint16_t)(fp << SMC_LF_FP_SHIFT); L.snp_mask = mask;
        loci[l] = L;
    }
    free(rd); free(tb); free(tf); free(nfr); free(perm);
    return rc;
}
