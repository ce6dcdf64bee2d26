// Dense fp64 Cholesky solve of the reduced camera system for SMALL camera counts
// (6 * free cameras <= BA_DENSE_MAX: the reference's own workloads -- 9 to 46 free cameras).
// One workgroup, blocked right-looking factorisation (32x32 diagonal blocks factored by one wave, trailing
// update on the fp64 matrix cores) with the forward substitution fused in, then the backward substitution.  Larger systems use the banded
// Cholesky or the block-Jacobi PCG (block-sparse S); a multi-workgroup factorisation for 10k-unknown dense systems is
// not written.
#pragma once
#include <hip/hip_runtime.h>

#include <type_traits>

#include "ba_device.h"
#include "ba_banded.h"  // band_wave_sync
#include "ba_tile16.h"  // bcr_potrf16

#define BA_DENSE_MAX 768
#define BA_NB 32

namespace ba {

typedef double mfma_d4 __attribute__((ext_vector_type(4)));

// S (upper 6x6 blocks) -> full symmetric dense matrix, row-major, leading dimension n
__global__ void k_dense_assemble(int n_upper, const long long *__restrict__ keys, const double *__restrict__ S, int n,
                                 double *__restrict__ A) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n_upper * 36) return;
    const int k = e / 36, ab = e - k * 36, a = ab / 6, b = ab - a * 6;
    const int i = (int)(keys[k] >> 32), j = (int)(keys[k] & 0xffffffffLL);
    const double v = S[e];
    const size_t r = (size_t)i * 6 + a, c = (size_t)j * 6 + b;
    if (i == j) {
        // the accumulated diagonal block is symmetric up to the rounding order of the atomics: symmetrise
        A[r * n + c] = 0.5 * (v + S[(size_t)k * 36 + b * 6 + a]);
    } else {
        A[r * n + c] = v;
        A[c * n + r] = v;
    }
}

#define BA_DENSE_THREADS 512
#define BA_PS 34                 // doubles per panel row in LDS (16-byte aligned rows, 4 banks apart)
#define BA_DENSE_LDS_ROWS 448     // panel rows that fit in LDS next to the static arrays (448 * 34 * 8 = 119 KB)
#ifdef BA_DENSE_TIMING
#define BA_DT(i) do { if (tid == 0) { const long long t_ = __builtin_amdgcn_s_memtime(); dt[i] += t_ - t_prev; t_prev = t_; } } while (0)
#else
#define BA_DT(i) do { } while (0)
#endif

// x = -A^{-1} g by Cholesky; A (row-major, lower triangle used) is overwritten by its lower factor.  info = 0 or the
// 1-based index of the first non-positive pivot (then x is not valid).  One workgroup of 8 waves, blocked right-looking:
//   1. the 32x32 diagonal block is factored by ONE wave as two 16x16 tiles (ba_tile16.h: rows in registers, readlane
//      broadcasts, MFMA for the tile below and its update), overlapped with the trailing update of the column before (look-ahead),
//   2. the panel below it: P <- P X^T with X = L_kk^-1 (from the two tile inverses) on the matrix core, a strip of 16
//      rows per wave,
//   3. the trailing update: every wave takes 32x32 blocks of the lower triangle as 2x2 fp64 MFMA tiles (16x16x4), operands
//      read from the panel's copy in LDS (or in L2 when it does not fit),
// the forward substitution rides along in the trailing phase (wave 1, panel still in LDS); then the blocked backward
// substitution with one-wave diagonal solves.  n is a multiple of 6 (even), so every row starts 16-byte aligned.
__global__ __launch_bounds__(BA_DENSE_THREADS) void k_dense_cholesky_solve(int n, double *__restrict__ A, const double *__restrict__ g,
                                                                           double *__restrict__ x, int *__restrict__ info, int lds_rows) {
    extern __shared__ __attribute__((aligned(16))) double sPanel[];  // [lds_rows][BA_PS]: the panel of the current block column
    // [kb & 1]: factored diagonal block (lower), for the panel solve.  (One row and half a column vector of slack: the
    // row update below reads a fixed number of entries per lane at constant offsets, some of them past the diagonal.)
    __shared__ double sLd[2][BA_NB + 1][BA_NB + 1];
    __shared__ double sInvd[2][BA_NB];           // 1 / diagonal of that block
    __shared__ double sX[BA_NB][BA_NB + 1];   // inverse of the factor of the current diagonal block (lower; the panel solve is a product with it)
    __shared__ double sTileInv[16 * 17];      // inverse of the factor of the 16x16 tile just factored (diagonal block, wave 0)
    __shared__ double sY[BA_DENSE_MAX];
    __shared__ double sInvAll[BA_DENSE_MAX];     // 1 / diagonal of L, kept for the backward substitution
    __shared__ int sBad;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, tx = tid & 31;
    constexpr int NW = BA_DENSE_THREADS / 64;
    const int nb = (n + BA_NB - 1) / BA_NB;
    if (tid == 0) sBad = 0;
    for (int i = tid; i < n; i += BA_DENSE_THREADS) sY[i] = -g[i];
    __syncthreads();
#ifdef BA_DENSE_TIMING
    long long dt[8] = {0, 0, 0, 0, 0, 0, 0, 0}, t_prev = __builtin_amdgcn_s_memtime();
#endif

    // ---- 1. diagonal block (wave 0 only): staged in LDS (lanes r and r + 32 share the loads of row r), factored as two
    // 16x16 tiles with the rows in registers, wave-level synchronisation only.  (A register-resident row of 32 per lane
    // turns into ~370 spilled registers; rows of 16 fit.)
    auto factor_diag = [&](const int kbx, const bool in_place) {
        const int k0 = kbx * BA_NB;
        const int kw = (n - k0 < BA_NB) ? n - k0 : BA_NB;
        double (&sL)[BA_NB + 1][BA_NB + 1] = sLd[kbx & 1];
        double (&sInv)[BA_NB] = sInvd[kbx & 1];
        const int r = tx, hi = lane >> 5;
        if (!in_place) {  // (in place: this wave has just written the updated block into sL, see the trailing update)
            // (all sixteen loads in flight: a rolled loop would wait for each of them in turn)
            const double *ap = &A[(size_t)(k0 + (r < kw ? r : 0)) * n + k0];
            double v[BA_NB / 2];
#pragma unroll
            for (int t = 0; t < BA_NB / 2; ++t) v[t] = ap[(hi + 2 * t) < kw ? hi + 2 * t : 0];
#pragma unroll
            for (int t = 0; t < BA_NB / 2; ++t) {
                const int c = hi + 2 * t;
                sL[r][c] = (r < kw && c < kw && c <= r) ? v[t] : ((c == r) ? 1.0 : 0.0);
            }
        }
        band_wave_sync();
        int bad = 0;
        // Two 16x16 tiles (the tile chain of the cyclic reduction, ba_bcr.h): tile (0,0) factored and its factor inverted
        // in registers, a row per lane, the pivot column's entries broadcast as DPP operands (bcr_potrf16); tile (1,0) <- (1,0)
        // Linv^T and (1,1) -= (1,0) (1,0)^T on the matrix core; then tile (1,1).  13 k cycles per block -- the version
        // before worked in LDS, four columns per step with every lane factoring the 4x4 pivot block redundantly: 31 k.
        {
            const int l15 = lane & 15, l4 = lane >> 4;
            if (bcr_potrf16(&sL[l15][0], sTileInv, nullptr, lane)) bad = k0 + 1;
            band_wave_sync();
#pragma unroll
            for (int e = lane; e < 256; e += 64) sX[e >> 4][e & 15] = sTileInv[(e >> 4) * 17 + (e & 15)];   // X00
            mfma_d4 w = mfma_d4{0.0, 0.0, 0.0, 0.0};
            double aq[4], bq[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) { aq[q] = sL[16 + l15][4 * q + l4]; bq[q] = sTileInv[l15 * 17 + 4 * q + l4]; }
#pragma unroll
            for (int q = 0; q < 4; ++q) w = __builtin_amdgcn_mfma_f64_16x16x4f64(aq[q], bq[q], w, 0, 0, 0);
            band_wave_sync();   // (every lane has read the old tile)
#pragma unroll
            for (int v = 0; v < 4; ++v) sL[16 + l4 + 4 * v][l15] = w[v];
            band_wave_sync();
            mfma_d4 t;
#pragma unroll
            for (int v = 0; v < 4; ++v) t[v] = sL[16 + l4 + 4 * v][16 + l15];
#pragma unroll
            for (int q = 0; q < 4; ++q) { aq[q] = -sL[16 + l15][4 * q + l4]; bq[q] = sL[16 + l15][4 * q + l4]; }
#pragma unroll
            for (int q = 0; q < 4; ++q) t = __builtin_amdgcn_mfma_f64_16x16x4f64(aq[q], bq[q], t, 0, 0, 0);
#pragma unroll
            for (int v = 0; v < 4; ++v) sL[16 + l4 + 4 * v][16 + l15] = t[v];
            band_wave_sync();
            if (bcr_potrf16(&sL[16 + l15][16], sTileInv, nullptr, lane) && bad == 0) bad = k0 + 17;
            band_wave_sync();
            // the inverse of the whole block for the panel product: X11, and X10 = -X11 (L10 X00) -- the accumulator of the
            // inner product is the outer one's B operand as it stands (element q of a lane = row l4 + 4 q)
#pragma unroll
            for (int e = lane; e < 256; e += 64) sX[16 + (e >> 4)][16 + (e & 15)] = sTileInv[(e >> 4) * 17 + (e & 15)];
            w = mfma_d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int q = 0; q < 4; ++q) w = __builtin_amdgcn_mfma_f64_16x16x4f64(sL[16 + l15][4 * q + l4], sX[4 * q + l4][l15], w, 0, 0, 0);
            t = mfma_d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int q = 0; q < 4; ++q) t = __builtin_amdgcn_mfma_f64_16x16x4f64(-sTileInv[l15 * 17 + 4 * q + l4], w[q], t, 0, 0, 0);
#pragma unroll
            for (int v = 0; v < 4; ++v) sX[16 + l4 + 4 * v][l15] = t[v];
            band_wave_sync();
        }
        if (hi == 0) {
            const double iv = 1.0 / sL[r][r];
            sInv[r] = iv;
            if (r < kw) {
                sInvAll[k0 + r] = iv;
                double *op = &A[(size_t)(k0 + r) * n + k0];
                double w[BA_NB];
#pragma unroll
                for (int c = 0; c < BA_NB; ++c) w[c] = sL[r][c];
#pragma unroll
                for (int c = 0; c < BA_NB; ++c)
                    if (c <= r) op[c] = w[c];
            }
        }
        if (lane == 0 && bad != 0 && sBad == 0) sBad = bad;
    };
    // The diagonal block of column kb + 1 is factored by wave 0 WHILE the other waves finish the trailing update of
    // column kb (wave 0 updates that block first): the two serial parts of a block column overlap.
    if (wave == 0) factor_diag(0, false);
    __syncthreads();
    BA_DT(0);
    for (int kb = 0; kb < nb; ++kb) {
        const int k0 = kb * BA_NB;
        const int kw = (n - k0 < BA_NB) ? n - k0 : BA_NB;
        double (&sL)[BA_NB + 1][BA_NB + 1] = sLd[kb & 1];
        double (&sInv)[BA_NB] = sInvd[kb & 1];
        // ---- 2. panel below the diagonal block: row_i <- row_i L_kk^{-T}, one row per lane.  Rows that have a place in the
        // LDS panel travel through it in both directions: a lane reading or writing its own 256-byte row in L2 touches 64
        // cache lines per wave instruction, the staged copies are coalesced (16 lanes per row).
        const int prow = n - (k0 + kw), srow = prow < lds_rows ? prow : lds_rows;  // panel rows, staged panel rows
        if (prow > 0) {  // (then kw == BA_NB: only the last block column is short, and it has no panel)
#pragma unroll 4
            for (int idx = tid; idx < srow * (BA_NB / 2); idx += BA_DENSE_THREADS) {
                const int pr = idx >> 4, c = (idx & 15) * 2;
                *reinterpret_cast<double2 *>(&sPanel[pr * BA_PS + c]) = *reinterpret_cast<const double2 *>(&A[(size_t)(k0 + kw + pr) * n + k0 + c]);
            }
            __syncthreads();
        }
        // P <- P X^T with X = L_kk^-1 (factor_diag): a product on the matrix core, one strip of 16 panel rows per wave and
        // trip -- 12 MFMAs (X is lower triangular: three 16x16 tiles) in place of a 32-column substitution per row, whose
        // serial FMA chain was 19 k cycles per block column at 276 unknowns.
        {
            const int l15 = lane & 15, l4 = lane >> 4;
            const int nstrip = (prow + 15) >> 4;
            for (int st = wave; st < nstrip; st += NW) {
                const int pr0 = 16 * st;
                const int pra = pr0 + l15 < prow ? pr0 + l15 : prow - 1;   // (rows past the end: a copy of the last one, not stored)
                double aq[8];
                if (pra < srow) {
#pragma unroll
                    for (int q = 0; q < 8; ++q) aq[q] = sPanel[pra * BA_PS + 4 * q + l4];
                } else {
                    const double *ap = &A[(size_t)(k0 + kw + pra) * n + k0];
#pragma unroll
                    for (int q = 0; q < 8; ++q) aq[q] = ap[4 * q + l4];
                }
                mfma_d4 o0 = mfma_d4{0.0, 0.0, 0.0, 0.0}, o1 = mfma_d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int q = 0; q < 4; ++q) o0 = __builtin_amdgcn_mfma_f64_16x16x4f64(aq[q], sX[l15][4 * q + l4], o0, 0, 0, 0);
#pragma unroll
                for (int q = 0; q < 4; ++q) o1 = __builtin_amdgcn_mfma_f64_16x16x4f64(aq[q], sX[16 + l15][4 * q + l4], o1, 0, 0, 0);
#pragma unroll
                for (int q = 0; q < 4; ++q) o1 = __builtin_amdgcn_mfma_f64_16x16x4f64(aq[4 + q], sX[16 + l15][16 + 4 * q + l4], o1, 0, 0, 0);
                band_wave_sync();   // (every lane holds its part of the strip before anyone overwrites it)
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    const int pr = pr0 + l4 + 4 * v;
                    if (pr < srow) {
                        sPanel[pr * BA_PS + l15] = o0[v];
                        sPanel[pr * BA_PS + 16 + l15] = o1[v];
                    } else if (pr < prow) {
                        double *ap = &A[(size_t)(k0 + kw + pr) * n + k0];
                        ap[l15] = o0[v];
                        ap[16 + l15] = o1[v];
                    }
                }
            }
        }
        __syncthreads();
        // the staged rows go back to L2 (the backward substitution reads L there); nobody waits for these stores
#pragma unroll 4
        for (int idx = tid; idx < srow * (BA_NB / 2); idx += BA_DENSE_THREADS) {
            const int pr = idx >> 4, c = (idx & 15) * 2;
            *reinterpret_cast<double2 *>(&A[(size_t)(k0 + kw + pr) * n + k0 + c]) = *reinterpret_cast<const double2 *>(&sPanel[pr * BA_PS + c]);
        }
        BA_DT(1);
        // ---- 3. trailing update A[ib, jb] -= P_ib P_jb^T, kb < jb <= ib: one 32x32 block per wave and trip
        {
            const int m = nb - kb - 1;             // block rows / columns left
            // One wave computes the 32x32 product as 2x2 MFMA tiles (v_mfma_f64_16x16x4_f64): operand A is P_ib[l15][4 kc + l4],
            // operand B is P_jb[l15][4 kc + l4] -- ONE double per lane, tile and k-step (a 4x4 register tile on the vector
            // ALUs reads four times as much LDS per FMA and was bound by those reads) -- and the result tile has
            // col = lane & 15, row = (lane >> 4) + 4 v in element v of the accumulator.
            const int l15 = lane & 15, l4 = lane >> 4;
            auto do_pair = [&](const int ib, const int jb, double (*const next_diag)[BA_NB + 1]) {
                const int rb = (kb + 1 + ib) * BA_NB, cb = (kb + 1 + jb) * BA_NB;
                const bool diag = ib == jb;  // (its upper right tile is not needed)
                // the tile's old values are fetched up front (sixteen read-modify-writes at the end would be sixteen L2
                // round trips in a row: the stores may alias the loads), their latency hides behind the products
                mfma_d4 acc[2][2];
                double old[2][2][4];
                const int ccl = cb + l15;
#pragma unroll
                for (int ri = 0; ri < 2; ++ri)
#pragma unroll
                    for (int ci = 0; ci < 2; ++ci) {
                        acc[ri][ci] = mfma_d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
                        for (int v = 0; v < 4; ++v) {
                            const int rr = rb + ri * 16 + l4 + 4 * v, cc = ccl + ci * 16;
                            old[ri][ci][v] = A[(size_t)(rr < n ? rr : n - 1) * n + (cc < n ? cc : n - 1)];
                        }
                    }
                // operands from LDS when the panel fits there (two separate loops, so that the LDS one is compiled to
                // ds_read and not to generic loads), else from the panel's copy in L2
                auto products = [&](const double *const (&pa)[2], const double *const (&pb)[2]) {
#pragma unroll
                    for (int kc = 0; kc < BA_NB; kc += 4) {
                        const double a0 = pa[0][kc], a1 = pa[1][kc], b0 = pb[0][kc], b1 = pb[1][kc];
                        acc[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, acc[0][0], 0, 0, 0);
                        acc[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b0, acc[1][0], 0, 0, 0);
                        acc[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, acc[1][1], 0, 0, 0);
                        if (!diag) acc[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b1, acc[0][1], 0, 0, 0);
                    }
                };
                int ra[2], rc[2];
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    ra[q] = rb + q * 16 + l15 < n ? rb + q * 16 + l15 : n - 1;
                    rc[q] = cb + q * 16 + l15 < n ? cb + q * 16 + l15 : n - 1;
                }
                // (per block row: with more rows than the LDS panel holds, the first lds_rows of them are still read there)
                const int o = k0 + BA_NB;
                const bool a_lds = (ib + 1) * BA_NB <= lds_rows || n - o <= lds_rows, b_lds = (jb + 1) * BA_NB <= lds_rows || n - o <= lds_rows;
                const double *const la[2] = {&sPanel[(a_lds ? ra[0] - o : 0) * BA_PS + l4], &sPanel[(a_lds ? ra[1] - o : 0) * BA_PS + l4]};
                const double *const lb[2] = {&sPanel[(b_lds ? rc[0] - o : 0) * BA_PS + l4], &sPanel[(b_lds ? rc[1] - o : 0) * BA_PS + l4]};
                const double *const ga[2] = {&A[(size_t)ra[0] * n + k0 + l4], &A[(size_t)ra[1] * n + k0 + l4]};
                const double *const gb[2] = {&A[(size_t)rc[0] * n + k0 + l4], &A[(size_t)rc[1] * n + k0 + l4]};
                if (a_lds) products(la, lb);            // (jb <= ib: b is in LDS whenever a is)
                else if (b_lds) products(ga, lb);
                else products(ga, gb);
                // (columns of the strip beyond kw belong to no panel: the last block is the only short one, and it has no trailing part)
#pragma unroll
                for (int ri = 0; ri < 2; ++ri)
#pragma unroll
                    for (int ci = 0; ci < 2; ++ci)
#pragma unroll
                        for (int v = 0; v < 4; ++v) {
                            const int rr = rb + ri * 16 + l4 + 4 * v, cc = ccl + ci * 16;
                            const bool in = rr < n && cc <= rr;
                            if (next_diag != nullptr)  // the next diagonal block goes straight to where it is factored (padded to 32x32 by the identity)
                                next_diag[rr - rb][cc - cb] = in ? old[ri][ci][v] - acc[ri][ci][v] : ((cc == rr) ? 1.0 : 0.0);
                            else if (in)
                                A[(size_t)rr * n + cc] = old[ri][ci][v] - acc[ri][ci][v];
                        }
            };
            // Forward substitution L y = -g rides along (wave 1): y of this block column from the factored diagonal block
            // in LDS, then its contribution leaves the right-hand side of the panel rows -- the panel is still in LDS, a
            // separate forward pass would fetch all of L from L2 again, two round trips per block column.
            if (wave == 1) {
                const int r = tx;
                double lrow[BA_NB], yb[BA_NB];
#pragma unroll
                for (int c = 0; c < BA_NB; ++c) lrow[c] = (c < r) ? sL[r][c] : 0.0;
                const double inv = sInv[r];
                double y = (r < kw) ? sY[k0 + r] : 0.0;
#pragma unroll
                for (int j = 0; j < BA_NB; ++j) {
                    const double yj = readlane_f64(y * inv, j);
                    yb[j] = yj;
                    y = (r == j) ? yj : fma(-lrow[j], yj, y);  // (lrow[j] = 0 for the lanes above row j)
                }
                if (lane < kw) sY[k0 + r] = y;
                const int rows = n - (k0 + kw), rl = rows < lds_rows ? rows : lds_rows;
                for (int pr = lane; pr < rl; pr += 64) {
                    const double *pp = &sPanel[pr * BA_PS];
                    double s0 = 0.0, s1 = 0.0;
#pragma unroll
                    for (int c = 0; c < BA_NB; c += 2) {
                        const double2 v = *reinterpret_cast<const double2 *>(pp + c);
                        s0 = fma(v.x, yb[c], s0);
                        s1 = fma(v.y, yb[c + 1], s1);
                    }
                    sY[k0 + kw + pr] -= s0 + s1;
                }
                for (int pr = rl + lane; pr < rows; pr += 64) {
                    const double *pp = &A[(size_t)(k0 + kw + pr) * n + k0];
                    double s0 = 0.0, s1 = 0.0;
#pragma unroll
                    for (int c = 0; c < BA_NB; c += 2) {
                        const double2 v = *reinterpret_cast<const double2 *>(pp + c);
                        s0 = fma(v.x, yb[c], s0);
                        s1 = fma(v.y, yb[c + 1], s1);
                    }
                    sY[k0 + kw + pr] -= s0 + s1;
                }
            }
            if (m > 0) {
                if (wave == 0) {
                    // the next diagonal block first, then its factorisation while the other waves do the rest
                    do_pair(0, 0, sLd[(kb + 1) & 1]);
                    factor_diag(kb + 1, true);
                } else {
                    int ib = 0, jb = 0;  // block pairs 1, 2, ... in row-major triangular order, dealt over waves 1 .. NW-1
                    for (int skip = wave; skip > 0; --skip) { if (++jb > ib) { ++ib; jb = 0; } }
                    while (ib < m) {
                        do_pair(ib, jb, nullptr);
                        for (int skip = NW - 1; skip > 0; --skip) { if (++jb > ib) { ++ib; jb = 0; } }
                    }
                }
            }
        }
        __syncthreads();
        BA_DT(2);
    }
    if (tid == 0) *info = sBad;
    if (sBad != 0) return;

    BA_DT(3);
    // ---- backward substitution L^T x = y, right-looking: the solved block leaves the rows above it at once (rows of L,
    // coalesced, all loads of a thread in flight together); wave 0 fetches the next diagonal block meanwhile
    {
        double lcol[BA_NB], inv = 1.0;  // wave 0: column tx of the diagonal block to solve next (L(j, tx), j > tx)
        auto fetch_diag = [&](const int kbx) {
            const int k0 = kbx * BA_NB;
            const int kw = (n - k0 < BA_NB) ? n - k0 : BA_NB;
            const int r = tx;
#pragma unroll
            for (int j = 0; j < BA_NB; ++j) {
                const double v = A[(size_t)(k0 + (j < kw ? j : 0)) * n + k0 + (r < kw ? r : 0)];
                lcol[j] = (r < kw && j < kw && j > r) ? v : 0.0;
            }
            inv = (r < kw) ? sInvAll[k0 + r] : 1.0;
        };
        if (wave == 0) fetch_diag(nb - 1);
        for (int kb = nb - 1; kb >= 0; --kb) {
            const int k0 = kb * BA_NB;
            const int kw = (n - k0 < BA_NB) ? n - k0 : BA_NB;
            if (wave == 0) {
                const int r = tx;
                double xv = (r < kw) ? sY[k0 + r] : 0.0;
#pragma unroll
                for (int j = BA_NB - 1; j >= 0; --j) {
                    const double xj = readlane_f64(xv * inv, j);
                    xv = (r == j) ? xj : fma(-lcol[j], xj, xv);
                }
                if (lane < kw) sY[k0 + r] = xv;
            }
            __syncthreads();
            if (kb == 0) break;
            if (wave == 0) fetch_diag(kb - 1);
            for (int i = tid; i < k0; i += BA_DENSE_THREADS) {
                double v[BA_NB];
#pragma unroll
                for (int c = 0; c < BA_NB; ++c) v[c] = A[(size_t)(k0 + (c < kw ? c : 0)) * n + i];
                double s0 = 0.0, s1 = 0.0;
#pragma unroll
                for (int c = 0; c < BA_NB; c += 2) {
                    s0 = fma(v[c], c < kw ? sY[k0 + c] : 0.0, s0);
                    s1 = fma(v[c + 1], c + 1 < kw ? sY[k0 + c + 1] : 0.0, s1);
                }
                sY[i] -= s0 + s1;
            }
            __syncthreads();
        }
    }
    for (int i = tid; i < n; i += BA_DENSE_THREADS) x[i] = sY[i];
    BA_DT(4);
#ifdef BA_DENSE_TIMING
    if (tid == 0) printf("[dense n=%d] diag %lld panel %lld trailing+forward %lld backward %lld (s_memtime ticks)\n", n, dt[0], dt[1], dt[2], dt[4]);
#endif
}

}  // namespace ba
