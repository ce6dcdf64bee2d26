// Multi-workgroup dense fp64 Cholesky solve of the reduced camera system: 768 < 6 Nf <= BA_DENSE_BIG_MAX unknowns
// whose block pattern is (nearly) full -- every camera sees the same landmarks, an object-centric capture: the case
// SURVEY.md 8(d) calls a genuine dense contraction (n^3 / 3 = 4e11 flop at 10.7 k unknowns).  ba_dense.h covers
// n <= 768 in ONE workgroup; here the whole chip works on the trailing update.
//
// Blocked right-looking factorisation A = L L^T, block size 64, in place in the lower triangle of a row-major
// M x M matrix (M = n + 1 rounded up to a multiple of 64):
//   * row n is the right-hand side -g ("augmented" matrix): it goes through the panel solve and the trailing update like
//     any other row, so when the factorisation is done it holds y = L^-1 (-g) -- the forward substitution costs nothing;
//     rows behind it are padding with a unit diagonal;
//   * block columns are taken in GROUPS of two: inside a group the second column first receives the first one's update
//     (k_big_syrk, col_only), each column's panel is solved (k_big_trsm), and the trailing matrix behind the group is
//     updated by both panels in ONE pass (K = 128):
//       k_big_trsm   one workgroup per 64 rows below: panel <- panel Linv^T on the fp64 matrix cores;
//       k_big_syrk   one workgroup per 64 x 64 tile: tile -= P_I P_J^T, K / 4 MFMAs (v_mfma_f64_16x16x4_f64) per wave and
//                    16 x 16 piece, operand tiles staged through LDS 32 columns of K at a time; the workgroup of tile (0, 0)
//                    -- the diagonal block of the next column, complete with this update -- keeps it in LDS and goes on to
//                    FACTOR it while the rest of the grid is still updating (big_potrf_block: Cholesky in 16-column steps
//                    with the serial chain on one wave, as k_bcr_factor, and the explicit inverse of the factor, so that
//                    the panel solve is a matrix product).  Only column 0 has a launch of its own for that (k_big_potrf).
//   * backward substitution L^T x = y, four block columns per launch from the last to the first: every workgroup solves the
//     group's columns itself (redundantly) and takes their x out of one earlier block of y.
// Round 2 had three launches per block column (the diagonal factorisation alone on the chip for 19 us), one pass over the
// trailing matrix per block column, and one backward launch per column whose matrix-vector products waited for their loads
// one row at a time: 18.6 ms per LM step at 10 326 unknowns; round 3: 13.7 (tools/ab_big.sh, tools/trace_big_dense.sh;
// groups of four columns: 13.5 there, but slower at 6 k unknowns).
// About 2.3 n / 64 launches, queued without host synchronisation; a non-positive pivot sets
// *info (the LM step is then void and repeated with PCG, as for the other direct solvers).
#pragma once
#include <hip/hip_runtime.h>

#include "ba_bcr.h"    // bcr_potrf16, bcr_static_for
#include "ba_dense.h"

#define BA_DENSE_BIG_MAX 12288   // unknowns: 2048 free cameras; the matrix takes 1.2 GB
#define BA_BIG_NB 64
#ifndef BA_BIG_GROUP
#define BA_BIG_GROUP 2   // block columns per pass over the trailing matrix
#endif

namespace ba {

inline int big_padded(int n) { return ((n + 1 + BA_BIG_NB - 1) / BA_BIG_NB) * BA_BIG_NB; }
inline size_t big_matrix_doubles(int n) { return (size_t)big_padded(n) * big_padded(n); }
inline size_t big_linv_doubles(int n) { return (size_t)(big_padded(n) / BA_BIG_NB) * BA_BIG_NB * BA_BIG_NB; }

// S (upper 6x6 blocks) -> lower triangle of the M x M matrix (leading dimension M); A was zeroed before
__global__ void k_big_assemble(int n_upper, const long long *__restrict__ keys, const double *__restrict__ S, int M,
                               double *__restrict__ A) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n_upper * 36) return;
    const int k = e / 36, ab = e - k * 36, a = ab / 6, b = ab - a * 6;
    const int i = (int)(keys[k] >> 32), j = (int)(keys[k] & 0xffffffffLL);   // i <= j: upper block (i, j)
    const double v = S[e];
    const size_t r = (size_t)i * 6 + a, c = (size_t)j * 6 + b;
    if (i == j) {
        if (a >= b) A[r * M + c] = 0.5 * (v + S[(size_t)k * 36 + b * 6 + a]);   // symmetrised lower half of the diagonal block
    } else {
        A[c * M + r] = v;   // lower block (j, i) = upper block transposed
    }
}

// right-hand-side row, its (never used) diagonal entry, unit diagonal of the padding
__global__ void k_big_init(int n, int M, const double *__restrict__ g, double *__restrict__ A) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n) A[(size_t)n * M + t] = -g[t];
    else if (t == n) A[(size_t)n * M + n] = 1e300;   // (larger than any |y|^2: the row's own pivot stays positive)
    else if (t < M) A[(size_t)t * M + t] = 1.0;
}

// Cholesky of the 64 x 64 diagonal block at k0 and the inverse of its factor: ONE workgroup of four waves, the block in
// LDS, factored right-looking in 16-column steps the way k_bcr_factor does it (ba_bcr.h) -- wave 0 owns the serial chain
// (bcr_potrf16: the 16 x 16 diagonal tile and the inverse of its factor in registers, DPP row broadcasts), then its own
// panel tile (s+1, s) and the update of tile (s+1, s+1), and is already factoring the next diagonal tile while waves
// 1..3 apply step s to the rest (panel tiles = products with the tile's inverse, trailing tiles, all MFMA).  The inverse
// X = L^-1 of the whole block follows tile by tile, by distance from the diagonal:
//   X_ii = (from the chain),   X_ij = -X_ii sum_{k = j .. i-1} L_ik X_kj
// -- the accumulator of the inner sum IS the second MFMA's B operand (element q of a lane = row l4 + 4q, which the
// k index of an MFMA lane is free to mean), so nothing is staged in between.
// (History: the block in LDS with three barriers per column, 87 us per block; one wave with a row of 64 per lane in
// registers, 54 us -- a third of the whole solve at 10 k unknowns; this version: see DESIGN.md.)
// (the factorisation proper: the 64 x 64 block is in sD; called by every thread of a 256-thread workgroup)
__device__ __forceinline__ void big_potrf_block(double *__restrict__ sD, double *__restrict__ sLinv, int k0, int M,
                                                double *__restrict__ A, double *__restrict__ Linv, int *__restrict__ info) {
    constexpr int NB = BA_BIG_NB, NT = NB / 16, LDD = NB + 1;
    // X tiles: the diagonal ones in sLinv (16 x 17 each); tile (i, j), i > j, in the never-used upper tile (j, i) of sD
    auto xel = [&](int i, int j, int a, int b) -> double & { return i == j ? sLinv[i * 272 + a * 17 + b] : sD[(16 * j + a) * LDD + 16 * i + b]; };
    const int tid = threadIdx.x, lane = tid & 63, l15 = lane & 15, l4 = lane >> 4;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    int bad = 0;
    auto panel_tile = [&](int i, int s) {   // (i, s) <- (i, s) Linv_s^T
        mfma_d4 w = mfma_d4{0.0, 0.0, 0.0, 0.0};
        double aq[4], bq[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) { aq[q] = sD[(16 * i + l15) * LDD + 16 * s + 4 * q + l4]; bq[q] = sLinv[s * 272 + l15 * 17 + 4 * q + l4]; }
#pragma unroll
        for (int q = 0; q < 4; ++q) w = __builtin_amdgcn_mfma_f64_16x16x4f64(aq[q], bq[q], w, 0, 0, 0);
#pragma unroll
        for (int v = 0; v < 4; ++v) sD[(16 * i + l4 + 4 * v) * LDD + 16 * s + l15] = w[v];
    };
    auto trailing_tile = [&](int i, int j, int s) {   // (i, j) -= (i, s) (j, s)^T
        double aq[4], bq[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            aq[q] = -sD[(16 * i + l15) * LDD + 16 * s + 4 * q + l4];
            bq[q] = sD[(16 * j + l15) * LDD + 16 * s + 4 * q + l4];
        }
        mfma_d4 w;
#pragma unroll
        for (int v = 0; v < 4; ++v) w[v] = sD[(16 * i + l4 + 4 * v) * LDD + 16 * j + l15];
#pragma unroll
        for (int q = 0; q < 4; ++q) w = __builtin_amdgcn_mfma_f64_16x16x4f64(aq[q], bq[q], w, 0, 0, 0);
#pragma unroll
        for (int v = 0; v < 4; ++v) sD[(16 * i + l4 + 4 * v) * LDD + 16 * j + l15] = w[v];
    };
    // two barriers per step: B1 releases the factor + inverse of tile (s, s), B2 the panel (i, s)
    if (wv == 0) {
        bcr_static_for<0, NT>([&](auto step_c) {
            constexpr int s = decltype(step_c)::value;
            if (bcr_potrf16(&sD[(16 * s + l15) * LDD + 16 * s], sLinv + s * 272, nullptr, lane)) bad = 1;
            __syncthreads();
            if (s + 1 < NT) {
                panel_tile(s + 1, s);
                band_wave_sync();
                trailing_tile(s + 1, s + 1, s);
            }
            __syncthreads();
        });
    } else {
        bcr_static_for<0, NT>([&](auto step_c) {
            constexpr int s = decltype(step_c)::value;
            __syncthreads();
#pragma unroll
            for (int i = s + 2; i < NT; ++i)
                if ((i - s - 2) % 3 == wv - 1) panel_tile(i, s);
            __syncthreads();
            int task = 0;
#pragma unroll
            for (int i = s + 1; i < NT; ++i) {
#pragma unroll
                for (int j = s + 1; j <= i; ++j) {
                    if (i == s + 1 && j == s + 1) continue;   // (wave 0 has done it)
                    if (wv == 1 + task % 3) trailing_tile(i, j, s);
                    ++task;
                }
            }
        });
    }
    __syncthreads();
#pragma unroll
    for (int d = 1; d < NT; ++d) {
        if (wv < NT - d) {
            const int j = wv, i = wv + d;
            mfma_d4 w = mfma_d4{0.0, 0.0, 0.0, 0.0};
            for (int k = j; k < i; ++k) {
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    w = __builtin_amdgcn_mfma_f64_16x16x4f64(sD[(16 * i + l15) * LDD + 16 * k + 4 * q + l4],
                                                             xel(k, j, 4 * q + l4, l15), w, 0, 0, 0);
            }
            mfma_d4 x = mfma_d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int q = 0; q < 4; ++q)
                x = __builtin_amdgcn_mfma_f64_16x16x4f64(-sLinv[i * 272 + l15 * 17 + 4 * q + l4], w[q], x, 0, 0, 0);
#pragma unroll
            for (int v = 0; v < 4; ++v) xel(i, j, l4 + 4 * v, l15) = x[v];
        }
        __syncthreads();
    }
    for (int e = tid; e < NB * NB; e += 256) {
        const int r = e / NB, c = e - r * NB;
        if (c <= r) A[(size_t)(k0 + r) * M + k0 + c] = sD[r * LDD + c];
        Linv[e] = c <= r ? xel(r >> 4, c >> 4, r & 15, c & 15) : 0.0;
    }
    if (tid == 0 && bad) atomicCAS(info, 0, k0 + 1);
}

__global__ __launch_bounds__(256) void k_big_potrf(int k0, int M, double *__restrict__ A, double *__restrict__ Linv,
                                                   int *__restrict__ info) {
    constexpr int NB = BA_BIG_NB, NT = NB / 16, LDD = NB + 1;
    __shared__ double sD[NB * LDD];
    __shared__ double sLinv[NT * 272];                  // diagonal tiles of the inverse, 16 x 17 each
    const int tid = threadIdx.x;
    for (int e = tid; e < NB * NB; e += 256) {
        const int r = e / NB, c = e - r * NB;
        sD[r * LDD + c] = A[(size_t)(k0 + r) * M + k0 + c];   // (entries right of the diagonal are never used)
    }
    __syncthreads();
    big_potrf_block(sD, sLinv, k0, M, A, Linv, info);
}

// panel rows r0 .. r0+63 (r0 = k0 + 64 (1 + blockIdx.x)): P <- P Linv^T.  256 threads = 4 waves of 16 rows.
__global__ __launch_bounds__(256) void k_big_trsm(int k0, int M, double *__restrict__ A, const double *__restrict__ Linv) {
    constexpr int NB = BA_BIG_NB, LD = NB + 1;
    __shared__ double sP[NB * LD];
    __shared__ double sI[NB * LD];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, l15 = lane & 15, l4 = lane >> 4;
    const int r0 = k0 + NB * (1 + blockIdx.x);
    for (int e = tid; e < NB * NB; e += 256) {
        const int r = e / NB, c = e - r * NB;
        sP[r * LD + c] = A[(size_t)(r0 + r) * M + k0 + c];
        sI[r * LD + c] = Linv[e];
    }
    __syncthreads();
    // out[i][c] = sum_k P[i][k] Linv[c][k]: operand A = P[row l15][k], operand B = Linv[row l15 (= output column)][k]
    mfma_d4 acc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[t] = mfma_d4{0.0, 0.0, 0.0, 0.0};
    const double *pa = &sP[(wv * 16 + l15) * LD + l4];
#pragma unroll
    for (int kc = 0; kc < NB; kc += 4) {
        const double a = pa[kc];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const double b = sI[(t * 16 + l15) * LD + kc + l4];
            acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[t], 0, 0, 0);
        }
    }
    // accumulator element v of a lane: row l4 + 4 v, column l15 of the 16 x 16 tile
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int v = 0; v < 4; ++v) A[(size_t)(r0 + wv * 16 + l4 + 4 * v) * M + k0 + t * 16 + l15] = acc[t][v];
}

// trailing update: tile (I, J), I >= J, of the lower triangle behind block column k:  C -= P_I P_J^T.
// blockIdx.x enumerates the tile pairs row by row; 256 threads = 4 waves of 16 rows x 64 columns.  The two 64 x 64
// operand tiles are staged through LDS in two halves of K (coalesced 256-byte row pieces): read straight from L2, one
// MFMA operand load touches 16 cache lines for 512 useful bytes and the vector cache becomes the limit (the first
// version ran at 16 % of the MFMA rate).  34 KB of LDS per workgroup: four workgroups per CU.
// FUSE: the workgroup of tile (0, 0) -- the diagonal block of the NEXT block column, complete once this update is applied --
// goes on to factor it (big_potrf_block) while the rest of the grid is still updating: no launch of its own for the
// diagonal factorisation, and not 19 us per block column with one workgroup on the chip.  (Its LDS -- the block,
// the diagonal tiles of the inverse -- lies over and behind the operand buffers: 42.5 KB per workgroup instead of 34.)
template <bool FUSE>
__global__ __launch_bounds__(256) void k_big_syrk(int kp, int K, int rb, int col_only, int M, double *__restrict__ A, double *__restrict__ Linv_next,
                                                  int *__restrict__ info) {
    // tiles (I, J), I >= J, of the lower triangle whose first row / column is rb:  C -= P_I P_J^T with the panels
    // P = A[., kp .. kp + K) (K = 64: one block column; K = 128: two at once -- half the passes over the trailing matrix).
    // col_only: only the tiles (I, 0) -- the block column that has to be complete before the second panel can be solved.
    constexpr int NB = BA_BIG_NB, KH = 32, LD = KH + 1, NT = NB / 16, LDD = NB + 1;
    __shared__ double sAB[2 * NB * LD > NB * LDD ? 2 * NB * LD : NB * LDD];
    __shared__ double sLinv[FUSE ? NT * 272 : 1];
    double *sA = sAB, *sB = sAB + NB * LD;
    const int t = blockIdx.x;
    int I, J;
    if (col_only) {
        I = t; J = 0;
    } else {   // (I, J) from the linear index t = I (I + 1) / 2 + J
        I = (int)((sqrt(8.0 * (double)t + 1.0) - 1.0) * 0.5);
        while ((I + 1) * (I + 2) / 2 <= t) ++I;
        while (I * (I + 1) / 2 > t) --I;
        J = t - I * (I + 1) / 2;
    }
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, l15 = lane & 15, l4 = lane >> 4;
    const int rI = rb + NB * I, rJ = rb + NB * J;
    mfma_d4 acc[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) acc[q] = mfma_d4{0.0, 0.0, 0.0, 0.0};
    const int lr = tid >> 2, lc = (tid & 3) * 8;   // loader: row lr, eight consecutive k
    for (int h = 0; h < K / KH; ++h) {
        const double *ga = &A[(size_t)(rI + lr) * M + kp + h * KH + lc], *gb = &A[(size_t)(rJ + lr) * M + kp + h * KH + lc];
        double va[8], vb[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) { va[k] = ga[k]; vb[k] = gb[k]; }
        if (h) __syncthreads();   // everybody is done with the previous piece of K
#pragma unroll
        for (int k = 0; k < 8; ++k) { sA[lr * LD + lc + k] = va[k]; sB[lr * LD + lc + k] = vb[k]; }
        __syncthreads();
        const double *pa = &sA[(wv * 16 + l15) * LD + l4];
#pragma unroll
        for (int kc = 0; kc < KH; kc += 4) {
            const double a = pa[kc];
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, sB[(q * 16 + l15) * LD + kc + l4], acc[q], 0, 0, 0);
        }
    }
    if (FUSE && I == 0 && J == 0) {
        // the updated diagonal block of the next column: into LDS (over the operand buffers) instead of back to HBM,
        // factored in place; big_potrf_block stores the factor and its inverse
        __syncthreads();   // everybody is done with the operands
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const double *cp = &A[(size_t)(rI + wv * 16 + l4 + 4 * v) * M + rJ + l15];
            double old[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) old[q] = cp[q * 16];
#pragma unroll
            for (int q = 0; q < 4; ++q) sAB[(wv * 16 + l4 + 4 * v) * LDD + q * 16 + l15] = old[q] - acc[q][v];
        }
        __syncthreads();
        big_potrf_block(sAB, sLinv, rI, M, A, Linv_next, info);
        return;
    }
#pragma unroll
    for (int v = 0; v < 4; ++v) {
        double *cp = &A[(size_t)(rI + wv * 16 + l4 + 4 * v) * M + rJ + l15];
        double old[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) old[q] = cp[q * 16];
#pragma unroll
        for (int q = 0; q < 4; ++q) cp[q * 16] = old[q] - acc[q][v];
    }
}

// backward substitution, BA_BIG_BACK_G block columns per launch (c_hi down to c_lo), launched from the last group to the
// first with grid c_lo + 1: EVERY workgroup solves the group's columns itself (redundantly, same arithmetic, same bits):
//   for c = c_hi .. c_lo:  x_c = Linv_c^T (y_c - sum over the group's later columns c' of L[c', c]^T x_c')
// -- ten 64 x 64 matrix-vector products out of L2 for a group of four -- then workgroup j < c_lo takes the group's x out of
// y_j (y lives in row n of A), and workgroup c_lo stores x.  One launch per block column (the first version) spent 14 us
// per column with most of the chip idle: 2.3 of 16 ms at 10 k unknowns.  Only indices < n count (the right-hand-side row
// and the padding carry no unknowns).
#ifndef BA_BIG_BACK_G
#define BA_BIG_BACK_G 4
#endif
__global__ __launch_bounds__(256) void k_big_back(int c_hi, int n, int M, double *__restrict__ A, const double *__restrict__ Linv_all,
                                                  double *__restrict__ x) {
    constexpr int NB = BA_BIG_NB, G = BA_BIG_BACK_G;
    __shared__ double sx[G][NB];
    __shared__ double sy[NB];
    __shared__ double sred[4][NB];
    const int tid = threadIdx.x, col = tid & 63, part = tid >> 6;
    const int c_lo = c_hi - G + 1 > 0 ? c_hi - G + 1 : 0, ng = c_hi - c_lo + 1;
    double *y = A + (size_t)n * M;
    for (int gi = 0; gi < ng; ++gi) {
        const int c = c_hi - gi, k0 = c * NB;
        const int rmax = n - k0 < NB ? n - k0 : NB;   // unknowns in this block
        // y_c less the columns of the group already solved
        double s = 0.0;
        for (int gj = 0; gj < gi; ++gj) {
            const int k1 = (c_hi - gj) * NB, rm1 = n - k1 < NB ? n - k1 : NB;
            (void)rm1;   // (x beyond the unknowns is zero, the rows behind them are padding: every load is unconditional)
            double v[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) v[r] = A[(size_t)(k1 + part * 16 + r) * M + k0 + col];
#pragma unroll
            for (int r = 0; r < 16; ++r) s += v[r] * sx[gj][part * 16 + r];
        }
        sred[part][col] = s;
        __syncthreads();
        if (tid < NB) sy[tid] = tid < rmax ? y[k0 + tid] - ((sred[0][tid] + sred[1][tid]) + (sred[2][tid] + sred[3][tid])) : 0.0;
        __syncthreads();
        // x_c[col] = sum_r Linv[r][col] y[r] (Linv is lower triangular: r >= col); thread (col, part) takes r = part mod 4
        const double *Linv = Linv_all + (size_t)c * NB * NB;
        s = 0.0;
        {
            double v[16];   // (y beyond the unknowns is zero)
#pragma unroll
            for (int r = 0; r < 16; ++r) v[r] = Linv[(part + 4 * r) * NB + col];
#pragma unroll
            for (int r = 0; r < 16; ++r) s += v[r] * sy[part + 4 * r];
        }
        sred[part][col] = s;
        __syncthreads();
        if (tid < NB) sx[gi][tid] = tid < rmax ? (sred[0][tid] + sred[1][tid]) + (sred[2][tid] + sred[3][tid]) : 0.0;
        __syncthreads();
    }
    const int j = blockIdx.x;
    if (j == c_lo) {
        for (int gi = 0; gi < ng; ++gi) {
            const int k0 = (c_hi - gi) * NB, rmax = n - k0 < NB ? n - k0 : NB;
            if (tid < rmax) x[k0 + tid] = sx[gi][tid];
        }
        return;
    }
    // y_j[col] -= sum over the group's columns c and rows r of L[64 c + r][64 j + col] x_c[r]
    const int j0 = j * NB;
    double s = 0.0;
    for (int gi = 0; gi < ng; ++gi) {
        const int k0 = (c_hi - gi) * NB, rmax = n - k0 < NB ? n - k0 : NB;
        (void)rmax;
        double v[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) v[r] = A[(size_t)(k0 + part * 16 + r) * M + j0 + col];
#pragma unroll
        for (int r = 0; r < 16; ++r) s += v[r] * sx[gi][part * 16 + r];
    }
    sred[part][col] = s;
    __syncthreads();
    if (tid < NB) y[j0 + tid] -= (sred[0][tid] + sred[1][tid]) + (sred[2][tid] + sred[3][tid]);
}

// x = -S^{-1} g.  A: big_matrix_doubles(n), Linv: big_linv_doubles(n); *info (device) = 0 on success.
// (Tried: look-ahead over a second stream -- the diagonal factorisation of column k + 1 beside the bulk of column k's
// update.  The streams did overlap, but the lone factorisation wave needs 320 registers and only became resident when
// the update's waves had drained from a SIMD: no gain.  CU-masked streams, which would have given it CUs of its own,
// crash inside hipExtStreamCreateWithCUMask on this ROCm 7.2 / gfx950 stack, tools/cumask_probe.hip.
// Tried again with the four-wave k_big_potrf (104 registers, resident beside the update): the next column's tiles first on
// st, the bulk on a side stream behind an event, st waiting for the bulk in front of its next update -- 18.4 ms per solve
// against 18.6: the two event hand-overs per block column cost what the overlap of 29 us gains.)
inline hipError_t big_solve_launch(hipStream_t st, int n, int n_upper, const long long *keys, const double *S, const double *g,
                                   double *x, double *A, double *Linv, int *info) {
    const int M = big_padded(n), nblk = M / BA_BIG_NB;
    hipError_t e;
    if ((e = hipMemsetAsync(A, 0, big_matrix_doubles(n) * sizeof(double), st)) != hipSuccess) return e;
    if ((e = hipMemsetAsync(info, 0, sizeof(int), st)) != hipSuccess) return e;
    const int ne = n_upper * 36;
    hipLaunchKernelGGL(k_big_assemble, dim3((ne + 255) / 256), dim3(256), 0, st, n_upper, keys, S, M, A);
    hipLaunchKernelGGL(k_big_init, dim3((M + 255) / 256), dim3(256), 0, st, n, M, g, A);
    hipLaunchKernelGGL(k_big_potrf, dim3(1), dim3(256), 0, st, 0, M, A, Linv, info);
    constexpr int NB = BA_BIG_NB;
    for (int kb = 0; kb < nblk;) {
        // a GROUP of g block columns kb .. kb + g - 1 at a time: inside the group column c first receives the update of the
        // group's earlier panels (col_only, K = 64 c; its tile (0, 0) is column c's diagonal block -- factored by that
        // workgroup), then its panel is solved; the trailing matrix behind the group is then updated by all g panels in ONE
        // pass (K = 64 g): the pass over the trailing matrix, which is what the factorisation streams from HBM, happens once
        // per group instead of once per block column.
        const int g = nblk - kb < BA_BIG_GROUP ? nblk - kb : BA_BIG_GROUP, kp = kb * NB;
        for (int c = 0; c < g; ++c) {
            const int col = kb + c, below = nblk - 1 - col;
            double *li = Linv + (size_t)col * NB * NB;
            if (c > 0) hipLaunchKernelGGL((k_big_syrk<true>), dim3(below + 1), dim3(256), 0, st, kp, c * NB, col * NB, 1, M, A, li, info);
            if (below > 0) hipLaunchKernelGGL(k_big_trsm, dim3(below), dim3(256), 0, st, col * NB, M, A, li);
        }
        const int T = nblk - (kb + g);
        if (T > 0)
            hipLaunchKernelGGL((k_big_syrk<true>), dim3(T * (T + 1) / 2), dim3(256), 0, st, kp, g * NB, (kb + g) * NB, 0, M, A,
                               Linv + (size_t)(kb + g) * NB * NB, info);
        kb += g;
    }
    for (int c_hi = (n - 1) / BA_BIG_NB; c_hi >= 0; c_hi -= BA_BIG_BACK_G) {
        const int c_lo = c_hi - BA_BIG_BACK_G + 1 > 0 ? c_hi - BA_BIG_BACK_G + 1 : 0;
        hipLaunchKernelGGL(k_big_back, dim3(c_lo + 1), dim3(256), 0, st, c_hi, n, M, A, Linv, x);
    }
    return hipGetLastError();
}

}  // namespace ba
