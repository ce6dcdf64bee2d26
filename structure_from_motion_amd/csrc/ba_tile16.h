// The 16 x 16 diagonal-tile factorisation shared by the cyclic reduction (ba_bcr.h), the blocked dense solver
// (ba_dense_big.h) and the one-workgroup dense solver (ba_dense.h): Cholesky factor and its inverse by one wave.
#pragma once
#include <hip/hip_runtime.h>
#include <type_traits>

#include "ba_banded.h"  // band_readlane, band_rsqrt

namespace ba {

// 64-bit DPP row broadcast (gfx90a+: the only DPP control the fp64 ALU takes): every lane of a row of 16 reads lane L of its row.
// The operand must not have been written by the instruction just before (two wait states: the s_nop; the compiler's hazard
// recogniser does not look into inline assembly).
template <int L>
__device__ __forceinline__ double dpp_row_bcast(double v) {
    double r;
    asm volatile("s_nop 1\n\tv_mov_b64_dpp %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "=v"(r) : "v"(v), "n"(L));
    return r;
}
// acc += (src of lane C of this row) * mul, ONE instruction
template <int C, bool NOP>
__device__ __forceinline__ void dpp_row_fmac(double &acc, double src, double mul) {
    if (NOP) asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(src), "v"(mul), "n"(C));
    else asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(src), "v"(mul), "n"(C));
}
template <int I, int N, class F>
__device__ __forceinline__ void tile_static_for(F &&f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        tile_static_for<I + 1, N>(f);
    }
}

// Cholesky factor of one 16x16 diagonal tile and the inverse of the factor, by ONE wave.  Lane r of every row of 16 lanes holds
// row r of the tile (w) and the running sums of column r of the inverse X = L^-1 (x), in registers; the four rows of the wave do
// the same work on copies.  The pivot column's entries L[c][j] reach the other lanes as the DPP operand of the fma that uses them
// (row_newbcast: one instruction per entry and half --  A[r][c] -= L[r][j] L[c][j]  and  s_c += L[c][j] x_j  with
// x_j = (delta_j,col - s_j) / L[j][j], forward substitution, column oriented).  Round 3: the broadcasts used to travel through
// scalar registers (two v_readlane per entry, one fma serving both halves in the two halves of a row pair): 5.0 k cycles per
// tile, about 700 instructions of a lone wave; the same arithmetic (same operands to every fma: same bits) now takes 464
// instructions and 4.4 k cycles -- an fp64 fma with a DPP operand issues at about a quarter of the plain rate.  (Two 32-bit DPP
// moves and a plain fma per entry: 22 k cycles, measured.)
// dt: row (lane & 15) of the tile in LDS (in: A, out: L); sLinv: 16 x 17 in LDS; li: 16 x 16 in global memory or null.
__device__ __forceinline__ int bcr_potrf16(double *dt, double *sLinv, double *li, int lane) {
    const int l15 = lane & 15;
    int bad = 0;
    double w[16], x[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) { w[c] = dt[c]; x[c] = 0.0; }
    // The pivot chain of column j + 1 (broadcast, reciprocal square root with its Newton steps: serial) is started as soon as
    // the first fma of column j has produced its operand, in front of the other fmas of column j, which do not depend on it.
    double d = dpp_row_bcast<0>(w[0]);
    if (!(d > 0.0)) bad = 1;
    double inv = band_rsqrt(d > 0.0 ? d : 1.0);
    tile_static_for<0, 16>([&](auto jc) {
        constexpr int j = decltype(jc)::value;
        w[j] *= inv;                                        // L[r][j]
        x[j] = ((j == l15 ? 1.0 : 0.0) - x[j]) * inv;       // x_j of this lane's column
        const double nm = -w[j], xj = x[j];
        if constexpr (j + 1 < 16) {
            dpp_row_fmac<j + 1, true>(w[j + 1], w[j], nm);
            dpp_row_fmac<j + 1, false>(x[j + 1], w[j], xj);
            d = dpp_row_bcast<j + 1>(w[j + 1]);
            if (!(d > 0.0)) bad = 1;
            inv = band_rsqrt(d > 0.0 ? d : 1.0);
        }
        tile_static_for<j + 2, 16>([&](auto cc) {
            constexpr int c = decltype(cc)::value;
            dpp_row_fmac<c, false>(w[c], w[j], nm);
            dpp_row_fmac<c, false>(x[c], w[j], xj);
        });
    });
    if (lane < 16) {
#pragma unroll
        for (int c = 0; c < 16; ++c)
            if (c <= l15) dt[c] = w[c];
#pragma unroll
        for (int r = 0; r < 16; ++r) sLinv[r * 17 + l15] = x[r];
        if (li) {
#pragma unroll
            for (int r = 0; r < 16; ++r) li[r * 16 + l15] = x[r];
        }
    }
    return bad;
}

}  // namespace ba
