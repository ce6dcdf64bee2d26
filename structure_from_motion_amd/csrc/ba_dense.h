// Dense fp64 Cholesky solve of the reduced camera system for SMALL camera counts
// (6 * free cameras <= BA_DENSE_MAX: the reference's own workloads -- 9 to 46 free cameras).
// One workgroup, blocked right-looking factorisation with the 32x32 panels staged in LDS,
// followed by the two triangular solves.  Larger systems use the block-Jacobi PCG
// (block-sparse S); a multi-workgroup MFMA f64 factorisation for 10k-unknown systems is not
// written yet.
#pragma once
#include <hip/hip_runtime.h>

#define BA_DENSE_MAX 768
#define BA_NB 32

namespace ba {

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

// x = -A^{-1} g by Cholesky; A is overwritten by its lower factor.  info = 0 or the 1-based index of
// the first non-positive pivot (then x is not valid).
__global__ __launch_bounds__(1024) void k_dense_cholesky_solve(int n, double *__restrict__ A, const double *__restrict__ g,
                                                               double *__restrict__ x, int *__restrict__ info) {
    __shared__ double sL[BA_NB][BA_NB + 1];
    __shared__ double sA[BA_NB][BA_NB + 1];
    __shared__ double sB[BA_NB][BA_NB + 1];
    __shared__ double sY[BA_DENSE_MAX];
    __shared__ int sBad;
    const int tid = threadIdx.x, tx = tid & 31, ty = tid >> 5;
    const int nb = (n + BA_NB - 1) / BA_NB;
    if (tid == 0) sBad = 0;
    __syncthreads();

    for (int kb = 0; kb < nb; ++kb) {
        const int k0 = kb * BA_NB;
        const int kw = (n - k0 < BA_NB) ? n - k0 : BA_NB;
        // 1. diagonal block -> LDS, unblocked Cholesky there
        sL[ty][tx] = (ty < kw && tx <= ty) ? A[(size_t)(k0 + ty) * n + k0 + tx] : ((ty == tx) ? 1.0 : 0.0);
        __syncthreads();
        for (int j = 0; j < kw; ++j) {
            if (tid == 0) {
                const double d = sL[j][j];
                if (!(d > 0.0) && sBad == 0) sBad = k0 + j + 1;
                sL[j][j] = sqrt(d > 0.0 ? d : 1.0);
            }
            __syncthreads();
            if (ty == 0 && tx > j && tx < kw) sL[tx][j] /= sL[j][j];
            __syncthreads();
            if (ty > j && ty < kw && tx > j && tx <= ty) sL[ty][tx] -= sL[ty][j] * sL[tx][j];
            __syncthreads();
        }
        if (ty < kw && tx <= ty) A[(size_t)(k0 + ty) * n + k0 + tx] = sL[ty][tx];
        // 2. panel below the diagonal block: row_i <- row_i L_kk^{-T}, one row per lane
        for (int i = k0 + kw + tid; i < n; i += 1024) {
            double row[BA_NB];
            double *ap = &A[(size_t)i * n + k0];
#pragma unroll
            for (int c = 0; c < BA_NB; ++c) row[c] = (c < kw) ? ap[c] : 0.0;
#pragma unroll
            for (int c = 0; c < BA_NB; ++c) {
                double v = row[c];
#pragma unroll
                for (int k = 0; k < c; ++k) v -= row[k] * sL[c][k];
                row[c] = v / sL[c][c];
            }
#pragma unroll
            for (int c = 0; c < BA_NB; ++c)
                if (c < kw) ap[c] = row[c];
        }
        __syncthreads();
        // 3. trailing update A[ib, jb] -= P_ib P_jb^T for kb < jb <= ib
        for (int ib = kb + 1; ib < nb; ++ib) {
            const int r = ib * BA_NB + ty;
            sA[ty][tx] = (r < n && tx < kw) ? A[(size_t)r * n + k0 + tx] : 0.0;
            for (int jb = kb + 1; jb <= ib; ++jb) {
                const int rj = jb * BA_NB + ty;
                __syncthreads();
                sB[ty][tx] = (rj < n && tx < kw) ? A[(size_t)rj * n + k0 + tx] : 0.0;
                __syncthreads();
                double acc = 0.0;
#pragma unroll
                for (int k = 0; k < BA_NB; ++k) acc += sA[ty][k] * sB[tx][k];
                const int c = jb * BA_NB + tx;
                if (r < n && c < n && c <= r) A[(size_t)r * n + c] -= acc;
            }
            __syncthreads();
        }
    }
    if (tid == 0) *info = sBad;
    if (sBad != 0) return;

    // forward substitution L y = -g
    for (int i = tid; i < n; i += 1024) sY[i] = -g[i];
    __syncthreads();
    for (int kb = 0; kb < nb; ++kb) {
        const int k0 = kb * BA_NB;
        const int kw = (n - k0 < BA_NB) ? n - k0 : BA_NB;
        sL[ty][tx] = (ty < kw && tx <= ty) ? A[(size_t)(k0 + ty) * n + k0 + tx] : ((ty == tx) ? 1.0 : 0.0);
        __syncthreads();
        for (int j = 0; j < kw; ++j) {
            if (tid == 0) sY[k0 + j] /= sL[j][j];
            __syncthreads();
            if (ty == 0 && tx > j && tx < kw) sY[k0 + tx] -= sL[tx][j] * sY[k0 + j];
            __syncthreads();
        }
        for (int i = k0 + kw + tid; i < n; i += 1024) {
            const double *ap = &A[(size_t)i * n + k0];
            double s = 0.0;
            for (int c = 0; c < kw; ++c) s += ap[c] * sY[k0 + c];
            sY[i] -= s;
        }
        __syncthreads();
    }
    // backward substitution L^T x = y
    for (int kb = nb - 1; kb >= 0; --kb) {
        const int k0 = kb * BA_NB;
        const int kw = (n - k0 < BA_NB) ? n - k0 : BA_NB;
        // subtract the already solved part: column tx of the panel, rows strided by ty
        double part = 0.0;
        if (tx < kw)
            for (int i = k0 + kw + ty; i < n; i += BA_NB) part += A[(size_t)i * n + k0 + tx] * sY[i];
        sA[ty][tx] = part;
        sL[ty][tx] = (ty < kw && tx <= ty) ? A[(size_t)(k0 + ty) * n + k0 + tx] : ((ty == tx) ? 1.0 : 0.0);
        __syncthreads();
        if (ty == 0 && tx < kw) {
            double s = 0.0;
            for (int k = 0; k < BA_NB; ++k) s += sA[k][tx];
            sY[k0 + tx] -= s;
        }
        __syncthreads();
        for (int j = kw - 1; j >= 0; --j) {
            if (tid == 0) sY[k0 + j] /= sL[j][j];
            __syncthreads();
            if (ty == 0 && tx < j) sY[k0 + tx] -= sL[j][tx] * sY[k0 + j];
            __syncthreads();
        }
    }
    for (int i = tid; i < n; i += 1024) x[i] = sY[i];
}

}  // namespace ba
