// smcounter_hip.hip - MI355X (gfx950) kernels and C ABI for smCounter's per-locus hot path.
//
// The whole of vc() up to (not including) string formatting runs on the device (reference:
// /root/reference/smCounter.py); one translation unit, the kernels in the .inc files below:
//
//   k_build_planes.inc   the planes of a run from its alignments (smCounter.py:316-366, :371-452, :462-471): the alignments of a
//                        tile of 64 loci are radix-sorted ONCE by (barcode, fragment, file index); a wavefront (lane = locus)
//                        then takes its part of that list a batch at a time as bit COLUMNS (which alignments cover the lane's
//                        locus; which of them start a fragment / a barcode there) - ranks are popcounts - first counting, then
//                        writing ONE word per read through an LDS staging buffer.
//   k_plan.inc           the launch plan of a batch whose descriptors are in HBM: class and weight bucket per locus, the launch
//                        lists written on the device (smc_plan_create_dev).
//   k_call_v2.inc        scan + group + score + rank, one workgroup per locus (several for a deep one).  The reads arrive
//                        barcode-major, each word saying whether it starts a fragment and carrying a 5-bit read class, so
//                        nothing is looked up:
//                        the scan evaluates its predicates four reads at a time as byte lanes of one register (v_perm_b32,
//                        DPP look-back), leaves ONE flag byte per read in LDS (fragment survives in bcDict / shows the reference
//                        allele / merged pair / included / fragment start, :468-479), and counts alleleCnt (:379,401,459).
//                        A barcode pass turns popcounts over the flag bytes into fragment counts; one-allele barcodes are
//                        scored from a per-count table, the others by the 8-lane general calProb path (:26-98) with
//                        lane-parallel FP64 epilogue; per-allele PI is summed in 64-bit fixed point (order-independent);
//                        the E stage ranks (:534-555), and only for loci whose candidate reaches filterVariants are the eight
//                        filter-only tallies counted, in one more pass over the just-streamed reads.
//   k_filter_loci.inc    filterVariants minus the two FASTA-dependent flags (:182-269) for the loci on the worklist; Fisher
//                        exact two-sided by chunked hypergeometric sums with a log-factorial table.
//   k_pack_rows.inc      432-byte rows -> 168-byte wire rows for the multi-GPU gather.
//   k_pack_words.inc     raw-field planes (meta, frag) -> read words, for batches a host builder made.
//
// Data layout (include/smcounter_hip.h): ONE uint32 per pileup read (allele, quality, fragment start, read
// class - what is left of the 16 B of raw fields per read once the plane builder has digested them) and umi_start; a 32-byte
// descriptor per locus; a 432-byte row out.
// This is integer / branchy, HBM-streaming work: no MFMA.
#include <hip/hip_runtime.h>

#include <limits.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <cmath>
#include <mutex>
#include <string>
#include <type_traits>
#include <vector>

#include "smcounter_hip.h"

#define WAVE 64
#define N_ID 4
#define GAP_ID 5

// One translation unit; the parts:
#include "device_common.inc"   // LDS header, DPP reductions, encodings, k_simple_table, finish_row (E stage)
#include "k_build_planes.inc" // kernel 0: the planes from a run's alignments (device half of the feature extraction)
#include "k_bp_emit2.inc"     //   its walk that writes the planes (lane = alignment)
#include "k_call_v2.inc"       // kernel 1: scan + group + score + rank (whole loci; deep loci in parts and chunks)
#include "k_filter_loci.inc"   // kernel 2: filterVariants / Fisher exact for the loci on the worklist
#include "k_pack_rows.inc"     // kernel 3: rows -> 168-byte wire rows for the multi-GPU gather
#include "k_pack_words.inc"    // kernel 4: raw-field planes -> one word per read (smc_pack_words)
#include "k_philox_marks.inc" // the non-parity down-sampling of loci over the barcode cap, Philox4x32-10 keyed by position
#include "k_plan.inc"          // launch plan of a batch whose descriptors are in HBM (classify + fill)
#include "host_abi.inc"        // the C ABI of include/smcounter_hip.h
