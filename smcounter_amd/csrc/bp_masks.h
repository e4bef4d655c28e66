// bp_masks.h - the bit trick at the centre of the plane builder's walks (k_build_planes.inc: bp_columns), in plain C++ so that the host
// tests can compile it too (tests/test_host_logic.py checks it against a loop over the rows).
#pragma once
#include <stdint.h>
#if defined(__HIPCC__)
#define BP_HD __host__ __device__ __forceinline__
#else
#define BP_HD static inline
#endif
// C: the lane's column; S: bit r = row r belongs to the same run as row r - 1 (wave-uniform); carried: the lane's last covering
// alignment before the batch belongs to row 0's run.  -> the rows that start a run at this locus
BP_HD uint32_t bp_heads(uint32_t C, uint32_t S, bool carried) {
    const uint32_t K1 = S, K2 = K1 & (K1 << 1), K4 = K2 & (K2 << 2), K8 = K4 & (K4 << 4), K16 = K8 & (K8 << 8);
    uint32_t V = C;
    V |= (V << 1) & K1; V |= (V << 2) & K2; V |= (V << 4) & K4; V |= (V << 8) & K8; V |= (V << 16) & K16;
    uint32_t H = C & ~((V << 1) & S);
    const uint32_t t = ~(S | 1u), G0 = t ? (t & (0u - t)) - 1u : 0xFFFFFFFFu;    // the rows of row 0's run
    if (carried) H &= ~G0;
    return H;
}
