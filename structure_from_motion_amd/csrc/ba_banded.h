// Block-banded fp64 Cholesky solve of the reduced camera system.
//
// Image sequences (the BAL-shaped street problems, and any registration order in which a point is
// seen by a short run of consecutive cameras) give a reduced camera system whose 6x6 blocks all lie
// within a narrow band |i - j| <= B.  Block-Jacobi PCG needs hundreds of iterations on such chains
// (their low-frequency bending modes are badly conditioned); the band factorisation is exact and
// costs N sequential block-column steps.  One workgroup: the active (B+1) x (B+1) window of blocks
// lives in LDS (circular indexing), the next block row is prefetched into registers while the
// current column is eliminated, and the forward substitution is fused into the factorisation.
//
// Band storage: Lb[j][d][36], d = 0..B = block (j + d, j) of the LOWER triangle, row-major 6x6.
// After the factorisation Lb[j][0] holds the INVERSE of the diagonal Cholesky factor (lower) and
// Lb[j][d], d > 0, the blocks of L.
#pragma once
#include <hip/hip_runtime.h>

#define BA_BAND_MAX 20       // largest half bandwidth (in blocks): (B+1)^2 * 288 B of LDS
#define BA_BAND_THREADS 768

namespace ba {

__global__ void k_band_assemble(int n_upper, const long long *__restrict__ keys, const double *__restrict__ S, int B,
                                double *__restrict__ Lb) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n_upper * 36) return;
    const int k = e / 36, ab = e - k * 36, a = ab / 6, b = ab - a * 6;
    const int i = (int)(keys[k] >> 32), j = (int)(keys[k] & 0xffffffffLL);  // i <= j, upper block (i, j)
    const int d = j - i;
    // lower block (j, i)[a][b] = S(i, j)[b][a]; column i of the band, offset d
    double v = S[(size_t)k * 36 + b * 6 + a];
    if (d == 0) v = 0.5 * (v + S[(size_t)k * 36 + ab]);
    Lb[((size_t)i * (B + 1) + d) * 36 + ab] = v;
}

// x = -A^{-1} g.  info = 0, or 1 + the block column of the first non-positive pivot.
__global__ __launch_bounds__(BA_BAND_THREADS) void k_band_cholesky_solve(int N, int B, double *__restrict__ Lb,
                                                                         const double *__restrict__ g,
                                                                         double *__restrict__ x, double *__restrict__ ywork,
                                                                         int *__restrict__ info) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int P = B + 1;
    double *sW = smem;                       // P*P*36: window of A / partially updated blocks
    double *sCol = sW + (size_t)P * P * 36;  // B*36: current column of L
    double *sB = sCol + (size_t)(B > 0 ? B : 1) * 36;  // P*6: window of the right-hand side
    double *sDi = sB + (size_t)P * 6;        // 36: inverse of the diagonal factor
    double *sY = sDi + 36;                   // 6
    __shared__ int sBad;
    __shared__ unsigned char sD1[BA_BAND_MAX * (BA_BAND_MAX + 1) / 2], sD2[BA_BAND_MAX * (BA_BAND_MAX + 1) / 2];
    const int tid = threadIdx.x;
    const int npair = B * (B + 1) / 2;
    if (tid == 0) sBad = 0;
    for (int q = tid; q < npair; q += BA_BAND_THREADS) {
        // q -> (d1, d2), 1 <= d2 <= d1 <= B, row-major over d1
        int d1 = 1, rem = q;
        while (rem >= d1) { rem -= d1; ++d1; }
        sD1[q] = (unsigned char)d1;
        sD2[q] = (unsigned char)(rem + 1);
    }
    // initial window: rows 0..B, lower part
    for (int e = tid; e < P * P * 36; e += BA_BAND_THREADS) {
        const int blk = e / 36, ab = e - blk * 36, r = blk / P, c = blk - r * P;
        sW[e] = (r >= c && r < N) ? Lb[((size_t)c * P + (r - c)) * 36 + ab] : 0.0;
    }
    for (int e = tid; e < P * 6; e += BA_BAND_THREADS) sB[e] = (e / 6 < N) ? -g[e] : 0.0;
    __syncthreads();

    for (int j = 0; j < N; ++j) {
        const int js = j % P;
        // 0. prefetch the block row that enters the window after this step: r = j + P, columns r-B..r
        const int rn = j + P;
        double pre = 0.0, preb = 0.0;
        if (rn < N) {
            if (tid < P * 36) {
                const int d = tid / 36, ab = tid - d * 36;  // block (rn, rn - d) = Lb[rn - d][d]
                pre = Lb[((size_t)(rn - d) * P + d) * 36 + ab];
            } else if (tid < P * 36 + 6) {
                preb = -g[(size_t)rn * 6 + (tid - P * 36)];
            }
        }
        // 1. diagonal block: Cholesky and inverse of the factor (one lane; 6x6)
        if (tid == 0) {
            const double *A = &sW[(size_t)(js * P + js) * 36];
            double L[6][6], Li[6][6];
            bool ok = true;
#pragma unroll
            for (int c = 0; c < 6; ++c) {
                double d = A[c * 6 + c];
#pragma unroll
                for (int k = 0; k < c; ++k) d -= L[c][k] * L[c][k];
                if (!(d > 0.0)) { ok = false; d = 1.0; }
                const double l = sqrt(d), il = 1.0 / l;
                L[c][c] = l;
#pragma unroll
                for (int r = c + 1; r < 6; ++r) {
                    double s = A[r * 6 + c];
#pragma unroll
                    for (int k = 0; k < c; ++k) s -= L[r][k] * L[c][k];
                    L[r][c] = s * il;
                }
            }
#pragma unroll
            for (int c = 0; c < 6; ++c) {
                Li[c][c] = 1.0 / L[c][c];
#pragma unroll
                for (int r = c + 1; r < 6; ++r) {
                    double s = 0.0;
#pragma unroll
                    for (int k = c; k < r; ++k) s -= L[r][k] * Li[k][c];
                    Li[r][c] = s / L[r][r];
                }
            }
#pragma unroll
            for (int r = 0; r < 6; ++r)
#pragma unroll
                for (int c = 0; c < 6; ++c) sDi[r * 6 + c] = (c <= r) ? Li[r][c] : 0.0;
            if (!ok && sBad == 0) sBad = j + 1;
        }
        __syncthreads();
        // 2. column of L: L(j+d, j) = A(j+d, j) Ljj^-T = A Linv^T;  y_j = Linv b_j
        if (tid < B * 36) {
            const int d = tid / 36 + 1, ab = tid - (d - 1) * 36, a = ab / 6, b = ab - a * 6;
            double v = 0.0;
            if (j + d < N) {
                const double *X = &sW[(size_t)(((j + d) % P) * P + js) * 36 + a * 6];
#pragma unroll
                for (int k = 0; k < 6; ++k) v += X[k] * sDi[b * 6 + k];
            }
            sCol[tid] = v;
            if (j + d < N) Lb[((size_t)j * P + d) * 36 + ab] = v;
        } else if (tid < B * 36 + 36) {
            const int ab = tid - B * 36;
            Lb[(size_t)j * P * 36 + ab] = sDi[ab];
        } else if (tid < B * 36 + 42) {
            const int a = tid - B * 36 - 36;
            double v = 0.0;
#pragma unroll
            for (int k = 0; k < 6; ++k) v += sDi[a * 6 + k] * sB[js * 6 + k];
            sY[a] = v;
            ywork[(size_t)j * 6 + a] = v;
        }
        __syncthreads();
        // 3. trailing update of the window, right-hand side update, and the new block row slides in
        for (int e = tid; e < npair * 36; e += BA_BAND_THREADS) {
            const int q = e / 36, ab = e - q * 36, a = ab / 6, b = ab - a * 6;
            const int d1 = sD1[q], d2 = sD2[q];
            if (j + d1 < N) {
                const double *l1 = &sCol[(d1 - 1) * 36 + a * 6], *l2 = &sCol[(d2 - 1) * 36 + b * 6];
                double s = 0.0;
#pragma unroll
                for (int k = 0; k < 6; ++k) s += l1[k] * l2[k];
                sW[(size_t)(((j + d1) % P) * P + ((j + d2) % P)) * 36 + ab] -= s;
            }
        }
        if (tid >= BA_BAND_THREADS - 128 && tid < BA_BAND_THREADS - 128 + B * 6) {
            const int t = tid - (BA_BAND_THREADS - 128), d = t / 6 + 1, a = t - (d - 1) * 6;
            if (j + d < N) {
                double s = 0.0;
#pragma unroll
                for (int k = 0; k < 6; ++k) s += sCol[(d - 1) * 36 + a * 6 + k] * sY[k];
                sB[((j + d) % P) * 6 + a] -= s;
            }
        }
        if (rn < N) {
            if (tid < P * 36) {
                const int d = tid / 36, ab = tid - d * 36;  // block (rn, rn - d) -> slot (rn % P = js, (rn - d) % P)
                sW[(size_t)(js * P + ((rn - d) % P)) * 36 + ab] = pre;
            } else if (tid < P * 36 + 6) {
                sB[js * 6 + (tid - P * 36)] = preb;
            }
        }
        __syncthreads();
    }
    if (tid == 0) *info = sBad;
    if (sBad != 0) return;

    // backward substitution  x_j = Linv_j^T (y_j - sum_d L(j+d, j)^T x_{j+d}),  window of x in sB
    double *sPart = sW;  // B*36 partial products
    for (int e = tid; e < P * 6; e += BA_BAND_THREADS) sB[e] = 0.0;
    __syncthreads();
    double nxt = (tid < P * 36 && N > 0) ? Lb[((size_t)(N - 1) * P) * 36 + tid] : 0.0;
    for (int j = N - 1; j >= 0; --j) {
        const int js = j % P;
        const double cur = nxt;
        if (j > 0 && tid < P * 36) nxt = Lb[((size_t)(j - 1) * P) * 36 + tid];
        if (tid < P * 36) {
            const int d = tid / 36, ab = tid - d * 36, a = ab / 6;
            if (d == 0) sDi[ab] = cur;
            else sPart[tid - 36] = (j + d < N) ? cur * sB[((j + d) % P) * 6 + a] : 0.0;  // L(j+d,j)[a][k] * x_{j+d}[a]
        }
        __syncthreads();
        if (tid < 6) {
            double s = ywork[(size_t)j * 6 + tid];
            for (int d = 0; d < B; ++d)
#pragma unroll
                for (int a = 0; a < 6; ++a) s -= sPart[d * 36 + a * 6 + tid];
            sY[tid] = s;
        }
        __syncthreads();
        if (tid < 6) {
            double v = 0.0;
#pragma unroll
            for (int k = 0; k < 6; ++k) v += sDi[k * 6 + tid] * sY[k];  // Linv^T
            sB[js * 6 + tid] = v;
            x[(size_t)j * 6 + tid] = v;
        }
        __syncthreads();
    }
}

}  // namespace ba
