// The 16 x 16 diagonal-tile factorisation shared by the cyclic reduction (ba_bcr.h), the blocked dense solver
// (ba_dense_big.h) and the one-workgroup dense solver (ba_dense.h): Cholesky factor and its inverse by one wave.
#pragma once
#include <hip/hip_runtime.h>

#include "ba_banded.h"  // band_readlane, band_rsqrt

namespace ba {

#ifndef BA_BCR_GROUP
#define BA_BCR_GROUP 4
#endif
// Cholesky factor of one 16x16 diagonal tile and the inverse of the factor, by ONE wave.  Lane r < 16 holds row r of the
// tile in registers; lane 16 + c holds, in the SAME registers, the running sums of column c of the inverse X = L^-1.
// The pivot column's entries L[c][j] travel by readlane (two per double; they, not the arithmetic, set the time), and
// every broadcast entry serves both halves with ONE fma:  rows  A[r][c] -= L[r][j] L[c][j];  inverse  s_c += L[c][j] x_j
// with  x_j = (delta_j,col - s_j) / L[j][j]  (forward substitution, column oriented).  Cycles per tile: two separate
// sweeps 10.4 k, one sweep 6 k, broadcasts issued four at a time and one fma per entry 5.7 k -- about 700 instructions
// at the 8 cycles per instruction a lone wave gets issued, which is what is left.
// dt: row (lane & 15) of the tile in LDS (in: A, out: L); sLinv: 16 x 17 in LDS; li: 16 x 16 in global memory or null.
__device__ __forceinline__ int bcr_potrf16(double *dt, double *sLinv, double *li, int lane) {
    const int l15 = lane & 15;
    const bool rows = lane < 16;
    int bad = 0;
    double v[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) v[c] = rows ? dt[c] : 0.0;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        double d = band_readlane(v[j], j);
        asm volatile("" : "+v"(d));   // (tested in vector registers: as a scalar the flag of every column stays live)
        if (!(d > 0.0)) bad = 1;
        const double inv = band_rsqrt(d > 0.0 ? d : 1.0);
        v[j] = (rows ? v[j] : ((j == l15 ? 1.0 : 0.0) - v[j])) * inv;   // L[r][j]  |  x_j of this lane's column
        const double mul = rows ? -v[j] : v[j];
        // four columns at a time: their broadcasts are issued together, so that the wait for a readlane's scalar result
        // is paid once per group, not once per element
#pragma unroll
        for (int c0 = j + 1; c0 < 16; c0 += BA_BCR_GROUP) {
            double lc[BA_BCR_GROUP];
#pragma unroll
            for (int u = 0; u < BA_BCR_GROUP; ++u)
                if (c0 + u < 16) lc[u] = band_readlane(v[j], c0 + u);   // L[c][j], from the row lanes
#pragma unroll
            for (int u = 0; u < BA_BCR_GROUP; ++u)
                if (c0 + u < 16) v[c0 + u] = fma(mul, lc[u], v[c0 + u]);
            // (pinned here: left alone, the optimiser sinks these updates to where v[c] is next read -- a left-looking
            // sweep that keeps every L[c][j] of the tile in scalar registers; they do not fit and get spilled to and
            // reloaded from VGPR lanes, 1200 extra lane moves per tile)
#pragma unroll
            for (int u = 0; u < BA_BCR_GROUP; ++u)
                if (c0 + u < 16) asm volatile("" : "+v"(v[c0 + u]));
        }
    }
    if (rows) {
#pragma unroll
        for (int c = 0; c < 16; ++c)
            if (c <= l15) dt[c] = v[c];
    } else if (lane < 32) {
#pragma unroll
        for (int r = 0; r < 16; ++r) sLinv[r * 17 + l15] = v[r];
        if (li) {
#pragma unroll
            for (int r = 0; r < 16; ++r) li[r * 16 + l15] = v[r];
        }
    }
    return bad;
}

}  // namespace ba
