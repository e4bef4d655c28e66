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

#include <limits.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <cmath>
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
#include "k_call_v2.inc"       // kernel 1: scan + group + score + rank (whole loci; deep loci in parts and chunks)
#include "k_filter_loci.inc"   // kernel 2: filterVariants / Fisher exact for the loci on the worklist
#include "k_pack_rows.inc"     // kernel 3: rows -> 168-byte wire rows for the multi-GPU gather
#include "k_plan.inc"          // launch plan of a batch whose descriptors are in HBM (classify + fill)
#include "host_abi.inc"        // the C ABI of include/smcounter_hip.h
