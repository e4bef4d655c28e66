/* smcounter_host.h - C ABI of the host-side companions of the HIP hot path (plain C++, no GPU, no torch):
 *
 *   libsmc_bam.so     BGZF/BAM decode + pileup -> the SoA planes of smcounter_hip.h.  Replaces what the reference
 *                     gets from pysam for this path: `samfile.pileup(region=..., truncate=True, max_depth=...,
 *                     stepper='nofilter')` and the per-read attribute reads of vc() (smCounter.py:309-366, :405-452).
 *   libsmc_rowfmt.so  the numeric columns of the 45-field row as CPython 2.7 prints them (smCounter.py:575-599).
 *
 * SURVEY.md section 8 rows f2 and f1 (the callers either side of the hot path).  The Python bindings are
 * smcounter_amd/bamio.py (NativeBam) and smcounter_amd/rows.py (format_rows). */
#ifndef SMCOUNTER_HOST_H
#define SMCOUNTER_HOST_H

#include <stdint.h>

#include "smcounter_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---------------------------------------------------------------- libsmc_bam.so */

/* Open a coordinate-sorted BAM with its .bai beside it (pysam.AlignmentFile(bam, 'rb'), smCounter.py:310).
 * 0 on success; otherwise *out still holds a handle whose smc_bam_error() explains (close it). */
int smc_bam_open(const char* path, void** out);
void smc_bam_close(void* h);
int smc_bam_n_refs(void* h);
const char* smc_bam_ref_name(void* h, int i);
int64_t smc_bam_ref_len(void* h, int i);
/* compressed bytes of the file that hold [start0, end0) of `chrom` according to the linear index (16 kb granules; -1: not
 * known) - a coarse volume estimate for sizing a first run */
int64_t smc_bam_span_bytes(void* h, const char* chrom, int64_t start0, int64_t end0);
const char* smc_bam_error(void* h);

/* Pileup of positions start0 .. end0-1 of `chrom` (0-based), per-read attributes kept column-wise inside the handle;
 * stops after the locus at which the batch reaches max_reads pileup reads (*n_loci_done = loci emitted).  Returns
 * the number of pileup reads, or < 0: -3 a read name with fewer than 3 ':' fields (smCounter.py:341-343), -4 no
 * sequence, -5 more than 255 alleles at a locus.  An unknown chromosome gives empty loci (like pysam). */
int64_t smc_bam_pileup(void* h, const char* chrom, int64_t start0, int64_t end0, int64_t max_reads, int64_t* n_loci_done);
int64_t smc_bam_keys_len(void* h);
/* copy the last smc_bam_pileup into caller arrays (n = its return value; n loci = *n_loci_done) */
void smc_bam_copy(void* h, uint32_t* umi, uint32_t* frag, uint8_t* flag, uint8_t* mq, uint32_t* nm, uint32_t* n_indel,
                  uint32_t* left_sp, uint32_t* qlen, uint32_t* qalen, int32_t* qpos, int32_t* indel, uint8_t* is_del,
                  uint8_t* allele, uint8_t* bq, int64_t* read_off, int32_t* n_keys, char* keys);

/* Fused decode -> planes: the same pileup, with the per-read feature arithmetic of vc() (:327-366, :432-452) applied
 * and the batch laid out as smc_plan_run reads it (4-read aligned loci, barcode-major reads, read class in the frag
 * word).  Once the sizes are known `alloc(ctx, n_slots, n_loci, out)` must fill out[0..3] with four
 * uint32[n_slots] buffers (meta, umi, frag, dist) and out[4] with an smc_locus[n_loci] buffer.  `refseq` = reference
 * bases of [start0, end0) (REF column, :488).  Loci with more barcodes than `ds` are listed by smc_bam_ds_info()
 * (text: one line per locus with its barcode strings) so that the host can apply the reference's random.sample.
 * Returns the number of pileup reads or < 0 as smc_bam_pileup. */
typedef void (*smc_planes_alloc)(void* ctx, int64_t n_slots, int64_t n_loci, void** out);
int64_t smc_bam_planes(void* h, const char* chrom, int64_t start0, int64_t end0, int64_t max_reads, double mismatch_thr,
                       const char* refseq, int nthreads, int ds, int min_bq, int min_mq, int primer_dist,
                       smc_planes_alloc alloc, void* alloc_ctx, int64_t* n_loci_done, int64_t* n_slots,
                       int64_t* n_umi_start);
void smc_bam_planes_copy(void* h, uint32_t* umi_start, int32_t* n_keys, char* keys);
const char* smc_bam_ds_info(void* h);

/* ---- device plane builder, host half (the device half is smc_build_planes in smcounter_hip.h): the run's alignments as a
 * structure of arrays - decode only, one entry per alignment; the per-pileup-read work of smCounter.py:316-366, :371-452
 * and :462-471 happens on the GPU.  Replaces, for this path, what the reference takes from pysam's AlignedSegment
 * objects. */
/* (smc_dev_aln / smc_dev_locus: smcounter_hip.h - the HIP library reads them) */
/* `alloc(ctx, n_aln, n_cig, n_seq, n_loci, out)` provides the four arrays: out[0] aln (n_aln x smc_dev_aln), out[1] the CIGAR pool
 * (n_cig x uint32), out[2] the base pool as (ASCII letter, quality) byte pairs (2 x n_seq bytes: smc_build_in.bq - every alignment is
 * placed so that reference position start0 + 64 t falls on pair 64 m, i.e. the builder's tile windows are aligned 128-byte lines; the
 * gaps hold ('A', 0)), out[3] loc (n_loci x smc_dev_locus).  status bit 1: an alignment flagged neither READ1 nor READ2 (the device
 * takes the previous pileup read's pairOrder), bit 2: a field does not fit the packed record (host builder). */
typedef void (*smc_aln_alloc)(void* ctx, int64_t n_aln, int64_t n_cig, int64_t n_seq, int64_t n_loci, void** out);
int64_t smc_bam_alignments(void* h, const char* chrom, int64_t start0, int64_t end0, int64_t max_reads, double mismatch_thr,
                           int nthreads, smc_aln_alloc alloc, void* alloc_ctx, int64_t* n_loci_done, int64_t* n_slots,
                           int32_t* n_bc, int32_t* n_pair, int32_t* status);
int smc_bam_allele_key(void* h, int64_t aln_index, int32_t qpos, int32_t indel, char* out, int cap);
const char* smc_bam_barcode_name(void* h, int32_t gid);
/* FNV-1a (64 bits) of the text of every run-wide barcode id of the last smc_bam_alignments: the identities smc_philox_marks keys on. */
int64_t smc_bam_barcode_idents(void* h, uint64_t* out, int64_t cap);

/* ---------------------------------------------------------------- libsmc_rowfmt.so */

/* Upper bound of one printed line of smc_format_tails. */
int smc_rowfmt_stride(void);
/* For each of the n rows: columns DP .. PI_C (fields 6-44 of the 45, smCounter.py:575-597) TAB-joined and ended by
 * '\n', printed from candidate chosen[i] (0 / 1: the bi-allelic decision :567-573 is the caller's; NULL = all 0).
 * chosen[i] < 0, status != 0, a zero denominator or |value| >= 1e8 leave an EMPTY line: the caller prints that row
 * itself.  `out` holds n * smc_rowfmt_stride() bytes; returns the bytes written. */
int64_t smc_format_tails(const smc_row* rows, const int8_t* chosen, int64_t n, char* out);
/* The same with the whole line where that needs no string work: for a row with alt[i] != 0 (and ref[i] != 0) the line is
 * CHROM, POS, ref[i], alt[i], "SNP", the 39 columns and the raw FILTER ";" of a locus no filter applies to - what vc()
 * returns for it (smCounter.py:599); for the other rows just the 39 columns, as smc_format_tails (the caller adds
 * CHROM..TYPE and FILTER).  Chromosome names: chroms[chrom_off[c] .. chrom_off[c + 1]) for c = chrom_id[i].  pred[i]
 * (optional) = int(float(<the PI column as printed>)) - what the repeat filters and the writers compare (:757, :838) -
 * or INT32_MIN for a row left empty.  `out` holds n * smc_rowfmt_line_stride(longest chromosome name) bytes; up to
 * `nthreads` (<= 16) threads print stretches of rows.  Returns the bytes written. */
int smc_rowfmt_line_stride(int max_chrom_len);
int64_t smc_format_lines(const smc_row* rows, const int8_t* chosen, int64_t n, const char* chroms, const int32_t* chrom_off,
                         const int32_t* chrom_id, const int64_t* pos, const uint8_t* ref, const uint8_t* alt, int max_chrom_len,
                         int nthreads, char* out, int32_t* pred);

#ifdef __cplusplus
}
#endif
#endif
