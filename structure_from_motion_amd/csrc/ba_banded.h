// Block-banded fp64 Cholesky solve of the reduced camera system.
//
// Image sequences (the BAL-shaped street problems, and any registration order in which a point is
// seen by a short run of consecutive cameras) give a reduced camera system whose 6x6 blocks all lie
// within a narrow band |i - j| <= B.  Block-Jacobi PCG needs hundreds of iterations on such chains
// (their low-frequency bending modes are badly conditioned); the band factorisation is exact and
// costs N sequential block-column steps.
//
// One workgroup.  The live window of the right-looking factorisation is the triangle of blocks
// (row, col), j <= col <= row <= j + B.  It is kept in REGISTERS: a thread pair owns the unordered
// pair {s1, s2} of circular slots (slot = block row mod (B+1)) and holds half (3 x 6) of the one
// block of that pair that is currently in the lower triangle; when one of the two rows retires the
// pair is refilled from the block row that slides in.  Per block column there are two barriers:
//   A  every wave that needs it factors the 6x6 diagonal block redundantly (serial chain, rsqrt +
//      Newton, no divisions), lanes (d, a) solve one row of L(j+d, j) by forward substitution
//      (static register indexing), the forward substitution of the right-hand side is fused in;
//   B  rank-6 update of the window from the column of L in LDS (54 LDS doubles per 108 FMAs per
//      thread), the updated column j+1 is written back to LDS for the next step.
// Block rows are prefetched from HBM two steps ahead.  The backward substitution runs in one wave.
//
// Band storage: Lb[j][d][36], d = 0..B = block (j + d, j) of the LOWER triangle, row-major 6x6.
// After the factorisation Lb[j][0] holds the updated diagonal block (its Cholesky factor is L_jj) and
// Lb[j][d], d > 0, the blocks of L; k_band_scale then replaces L(j+d, j) by L(j+d, j) L_jj^{-1}.
#pragma once
#include <hip/hip_runtime.h>

#define BA_BAND_MAX 16       // largest half bandwidth (in blocks)
#define BA_BAND_THREADS 768  // upper bound of the workgroup size (see band_threads)

namespace ba {

// One elimination front.  A long chain is eliminated from BOTH ends at once (two workgroups): the top side runs over
// cameras 0 .. m-1 in natural order, the bottom side over cameras N-1 .. m+B in reversed order; both stop in front
// of the separator (cameras m .. m+B-1), add their Schur complements and right-hand sides into one dense 6B x 6B
// system (itself a banded system of B cameras with half bandwidth B - 1: the same kernels solve it, one-sided) and
// the two backward substitutions run outwards from the separator.
struct BandSide {
    double *Lb;       // band storage of this side's system, in this side's camera order
    double *g;        // its right-hand side (the separator rows are zero on the bottom side: counted once)
    double *ywork;
    int N;            // block columns of this side's system (eliminated part + separator)
    int n_elim;       // block columns to eliminate (== N: no separator, the whole system is factored)
    int flip;         // 1: reversed camera order; camera of column j = cam0 - j, else cam0 + j
    int cam0;
};
struct BandArgs {
    BandSide side[2];
    int B;
    double *mid_l;    // separator system (B cameras, dense = band B - 1) in band storage, and
    double *mid_g;    //   minus its right-hand side: solved by the same kernels as a one-sided system
    int sep;          // first separator camera
    double *x;        // solution, camera order
    int *info;        // 0, or 1 + a block column with a non-positive pivot (zeroed before the launches)
};

#ifdef BA_BAND_TIMING  // tools/band_bench.hip: cycle stamps of one block column, per wave
__device__ long long g_band_t[16 * 8];
#define BA_BAND_STAMP(k) do { if (j == BA_BAND_TIMING && (threadIdx.x & 63) == 0 && (threadIdx.x >> 6) < 16) g_band_t[(threadIdx.x >> 6) * 8 + (k)] = clock64(); } while (0)
#else
#define BA_BAND_STAMP(k) do { } while (0)
#endif

// Window threads come in groups of 16 lanes = (slot c) x (8 consecutive slots s2 <= c) x (block half): within a group
// one operand of the rank-6 update is shared (LDS broadcast) and the other falls into 16 disjoint bank quads, so the
// ds_read_b128 of the update are conflict-free (blocks are 288 B apart = 8 banks mod 64, halves 36 banks).
__host__ __device__ inline int band_window_groups(int P) {
    int g = 0;
    for (int c = 0; c < P; ++c) g += c / 8 + 1;
    return g;
}
// thread layout of k_band_cholesky_solve: [solver waves: 6 B row lanes + the y lane | window threads | >= 128 auxiliary lanes]
__host__ __device__ inline int band_first_window_thread(int B) { return ((B * 6 + 1 + 63) / 64) * 64; }
__host__ __device__ inline int band_first_aux_thread(int B) {
    return band_first_window_thread(B) + ((16 * band_window_groups(B + 1) + 63) / 64) * 64;
}
inline int band_threads(int B) {
    const int P = B + 1;
    int n = band_first_aux_thread(B) + 128;
    if (P * 36 + 6 > n) n = P * 36 + 6;
    return ((n + 63) / 64) * 64;
}

// Upper blocks S(i, j), i <= j, of the reduced camera system -> band storage of the one or two sides.
// m < 0: one side (Lb0).  Otherwise cameras [0, m + B) go to the top side and cameras [m, N) reversed to the bottom
// side; blocks inside the separator [m, m + B) are stored on the top side only.
__global__ void k_band_assemble(int n_upper, const long long *__restrict__ keys, const double *__restrict__ S, int B, int N,
                                int m, double *__restrict__ Lb0, double *__restrict__ Lb1) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n_upper * 36) return;
    const int k = e / 36, ab = e - k * 36, a = ab / 6, b = ab - a * 6;
    const int i = (int)(keys[k] >> 32), j = (int)(keys[k] & 0xffffffffLL);  // i <= j, upper block (i, j)
    const int d = j - i, P = B + 1;
    const double vt = S[(size_t)k * 36 + b * 6 + a];  // transposed element
    const double v0 = S[(size_t)k * 36 + ab];
    if (m < 0 || j < m + B) {
        // lower block (j, i)[a][b] = S(i, j)[b][a]; column i of the band, offset d
        Lb0[((size_t)i * P + d) * 36 + ab] = d == 0 ? 0.5 * (vt + v0) : vt;
    }
    if (m >= 0 && i >= m && !(j < m + B)) {
        // reversed order: cameras i' = N-1-i >= j' = N-1-j, lower block (i', j') = rows of camera i, columns of camera j
        const int jr = N - 1 - j;
        Lb1[((size_t)jr * P + d) * 36 + ab] = d == 0 ? 0.5 * (vt + v0) : v0;
    }
}

// right-hand sides of the two sides (m >= 0): top = g[0 .. m+B), bottom = g reversed with zero separator rows
__global__ void k_band_rhs(int N, int B, int m, const double *__restrict__ g, double *__restrict__ g0, double *__restrict__ g1) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= N * 6) return;
    const int c = e / 6, k = e - c * 6;
    if (c < m + B) g0[e] = g[e];
    if (c >= m) g1[(size_t)(N - 1 - c) * 6 + k] = (c < m + B) ? 0.0 : g[e];
}

__device__ __forceinline__ double band_rsqrt(double d) {
    double y = __builtin_amdgcn_rsq(d);
    const double e = fma(-d * y, y, 1.0);  // v_rsq_f64 is good to ~2^-26: one Newton step reaches fp64 rounding level
    return fma(0.5 * y, e, y);
}

// Cholesky factor of the symmetric 6x6 block A (lower part read): L[r][c], r > c, and il[c] = 1 / L[c][c]
__device__ __forceinline__ bool band_chol6(const double *A, double (&L)[6][6], double (&il)[6]) {
    bool ok = true;
#pragma unroll
    for (int c = 0; c < 6; ++c) {
        double d = A[c * 6 + c];
#pragma unroll
        for (int k = 0; k < c; ++k) d = fma(-L[c][k], L[c][k], d);
        if (!(d > 0.0)) { ok = false; d = 1.0; }
        il[c] = band_rsqrt(d);
#pragma unroll
        for (int r = c + 1; r < 6; ++r) {
            double s = A[r * 6 + c];
#pragma unroll
            for (int k = 0; k < c; ++k) s = fma(-L[r][k], L[c][k], s);
            L[r][c] = s * il[c];
        }
    }
    return ok;
}

// LDS written by some lanes of a wave and read by others of the SAME wave: order the accesses
__device__ __forceinline__ void band_wave_sync() {
    // (a workgroup-scope fence would also drain the outstanding global prefetches: wait for LDS only)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
}


// Factorisation A = L L^T and forward substitution y = L^{-1} (-g).  info = 0, or 1 + the block column of the
// first non-positive pivot.  Per block column j there are two phases, separated by barriers:
//   phase 1  solver lanes (d, a): Cholesky of the diagonal block (every wave redundantly), one row of L(j+d, j) each
//            -> sL[j & 1]; y wave: y_j.   IN PARALLEL the window threads apply column j-1 (from sL[(j-1) & 1]) to
//            their register blocks -- this rank-6 update of the whole window is off the serial chain -- and publish
//            the blocks of column j+1 to sA2.
//   phase 2  column j+1 for the next step: sA = sA2 - L(., j) L(j+1, j)^T (36 P threads, 6 FMAs each); the pairs of
//            the slot that left the window are refilled from the block row that slides in; right-hand side; L -> HBM.
// The register window holds the blocks (r, c), j+2 <= c <= r <= j+P+1; slot of a row = row mod P.
// With a separator (n_elim < N) the side stops after n_elim columns and adds what is left of its window -- the Schur
// complement on the separator cameras -- and of its right-hand side to the dense separator system.
__global__ __launch_bounds__(BA_BAND_THREADS) void k_band_cholesky_solve(const BandArgs args) {
    const BandSide side = args.side[blockIdx.x];
    const int N = side.N, B = args.B, n_elim = side.n_elim;
    double *__restrict__ Lb = side.Lb;
    const double *__restrict__ g = side.g;
    double *__restrict__ ywork = side.ywork;
    constexpr int PM = BA_BAND_MAX + 1;
    __shared__ __attribute__((aligned(16))) double sA[PM * 36];       // sA[d] = A(j + d, j), the column being eliminated
    __shared__ __attribute__((aligned(16))) double sA2[2][PM * 36];   // [c & 1][d]: A(c + d, c) of the next column(s), one update short
    __shared__ __attribute__((aligned(16))) double sL[2][PM * 36];    // [j & 1][d - 1] = L(j + d, j)
    __shared__ __attribute__((aligned(16))) double sNew[PM * 36 + 8];  // block row rn = j+P+1: sNew[d] = A(rn, rn - d); then -g[rn]
    __shared__ __attribute__((aligned(16))) double sB[(PM + 1) * 6];  // right-hand side rows j .. j+P, slot = row mod (P + 1)
    __shared__ __attribute__((aligned(16))) double sY[8];
    __shared__ int sBad;
    const int P = B + 1, Pb = P + 1;
    const int tid = threadIdx.x, nthr = blockDim.x;
    const int WS = band_first_window_thread(B), T0 = band_first_aux_thread(B);
    if (tid == 0) sBad = 0;

    // ---- roles ----
    const int wave0 = tid & ~63;
    const bool is_solver = wave0 < WS;  // waves holding the (d, a) row lanes and the y lane (tid = 6 B)
    const bool is_loader = tid < P * 36 + 6;
    const int aux = tid - T0, naux = nthr - T0;  // auxiliary lane index (< 0: not one)
    bool is_win = false;
    int s1 = 0, s2 = 0;  // the pair of slots of a window thread, s1 >= s2
    const int h3 = (tid & 1) * 3;
    if (tid >= WS && tid < T0) {
        int gq = (tid - WS) >> 4, c = 0;
        while (c < P && gq >= c / 8 + 1) { gq -= c / 8 + 1; ++c; }
        if (c < P) {
            s1 = c;
            s2 = 8 * gq + ((tid & 15) >> 1);
            is_win = s2 <= c;
        }
    }
    auto clampN = [&](long long r) { return r < N ? r : (long long)N - 1; };
    // initial window: rows 2 .. P+1 (slot s holds row ((s - 2) mod P) + 2); row P+1 arrives through sNew in step 0
    double acc[3][6];
    {
        int r1 = s1 - 2, r2 = s2 - 2;
        if (r1 < 0) r1 += P;
        if (r2 < 0) r2 += P;
        r1 += 2; r2 += 2;
        const int r = r1 > r2 ? r1 : r2, c = r1 > r2 ? r2 : r1;
        const bool live = is_win && r < N && r <= P;
        const double *src = &Lb[(clampN(c) * P + (live ? r - c : 0)) * 36 + h3 * 6];
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int b = 0; b < 6; ++b) acc[i][b] = live ? src[i * 6 + b] : 0.0;
    }
    for (int e = tid; e < P * 36; e += nthr) {
        sA[e] = (e / 36 < N) ? Lb[e] : 0.0;                                         // column 0
        sA2[1][e] = (1 + e / 36 < N) ? Lb[(size_t)P * 36 + e] : 0.0;                // column 1, raw
        sA2[0][e] = 0.0;
    }
    for (int e = tid; e < Pb * 6; e += nthr) sB[e] = (e / 6 < N) ? -g[e] : 0.0;
    // loader element of block row r: block (r, r - d) = Lb[r - d][d] (tid = d * 36 + ab), or g[r] (negated when staged).
    // Loads are unconditional (clamped row, zeroed when staged): a predicated load would put a branch around it and
    // the compiler then drains ALL outstanding loads (vmcnt(0)) at every use instead of counting them.
    const double *ld_base = Lb;
    long long ld_stride = 0;
    double ld_sign = 0.0;
    {
        const int d = tid / 36, ab = tid - d * 36;
        if (tid < P * 36) { ld_base = Lb + ((long long)d - (long long)d * P) * 36 + ab; ld_stride = (long long)P * 36; ld_sign = 1.0; }
        else if (is_loader) { ld_base = g + (tid - P * 36); ld_stride = 6; ld_sign = -1.0; }
    }
    auto load_row = [&](int r) -> double { return ld_base[clampN(r) * ld_stride]; };
    // two prefetch registers used alternately (the column loop is unrolled by two): no register moves, so the
    // wait for a row sits two block columns after its load
    double pa = load_row(P + 1), pb = load_row(P + 2);
    __syncthreads();

    // rank-6 update of this thread's block with column jp (operands in sL[jp & 1]); publishes column jp + 2
    auto window_update = [&](const int jp, const int jm /* (jp + 2) mod P */) {
        int m1 = s1 - jm, m2 = s2 - jm;
        if (m1 < 0) m1 += P;
        if (m2 < 0) m2 += P;
        m1 += 2; m2 += 2;  // offsets of the two rows from jp, in [2, P + 1]
        if (m1 == P + 1 || m2 == P + 1) return;  // refilled in phase 2 of step jp: nothing to apply yet
        const int dr = m1 > m2 ? m1 : m2, dc = m1 > m2 ? m2 : m1;
        if (dr <= B) {
            const double *l1 = &sL[jp & 1][(dr - 1) * 36 + h3 * 6], *l2 = &sL[jp & 1][(dc - 1) * 36];
            // each operand row is fetched one row ahead of the FMAs that use it (the LDS latency hides behind 18 FMAs)
            double r1[3][6], r2[2][6];
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int k = 0; k < 6; ++k) r1[i][k] = l1[i * 6 + k];
#pragma unroll
            for (int k = 0; k < 6; ++k) r2[0][k] = l2[k];
#pragma unroll
            for (int b = 0; b < 6; ++b) {
                if (b < 5) {
#pragma unroll
                    for (int k = 0; k < 6; ++k) r2[(b + 1) & 1][k] = l2[(b + 1) * 6 + k];
                }
#pragma unroll
                for (int i = 0; i < 3; ++i)
#pragma unroll
                    for (int k = 0; k < 6; ++k) acc[i][b] = fma(-r1[i][k], r2[b & 1][k], acc[i][b]);
            }
        }
        if (dc == 2) {
            double *o = &sA2[jp & 1][(dr - 2) * 36 + h3 * 6];  // column jp + 2
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int b = 0; b < 6; ++b) o[i * 6 + b] = acc[i][b];
        }
    };

    auto copy_column = [&](const int jc) {
        constexpr int NC = (BA_BAND_MAX * 36 + 127) / 128;  // naux >= 128
        double cv[NC];
#pragma unroll
        for (int k = 0; k < NC; ++k) {
            const int e = aux + k * naux;
            cv[k] = sL[jc & 1][e < B * 36 ? e : 0];
        }
#pragma unroll
        for (int k = 0; k < NC; ++k) {
            const int e = aux + k * naux;
            if (e < B * 36) Lb[((size_t)jc * P + 1) * 36 + e] = cv[k];
        }
    };
    // j mod P and j mod (P + 1) are carried from column to column (P is a run-time value: a modulo costs a division)
    int mP = 0, mPb = 0;
    auto wrapP = [&](int v) { return v >= P ? v - P : v; };
    auto column_step = [&](const int j, double &pre) {
        // ---- phase 1 ----
        BA_BAND_STAMP(0);
        if (is_loader) sNew[tid] = (j + P + 1 < N) ? ld_sign * pre : 0.0;
        pre = load_row(j + P + 3);
        if (aux >= 0) {
            // off the serial chain: the diagonal block (k_band_scale factors it again) and the previous column of L to HBM
            if (aux < 36) Lb[(size_t)j * P * 36 + aux] = sA[aux];
            if (j > 0) copy_column(j - 1);
        }
        if (is_solver) {
            double L[6][6], il[6];
            const bool ok = band_chol6(sA, L, il);
            BA_BAND_STAMP(1);
            if (!ok && tid == 0 && sBad == 0) sBad = j + 1;
            if (tid <= B * 6) {
                // lane (d, a): row a of L(j+d, j) = A(j+d, j) L_jj^{-T}; lane 6 B: y_j = L_jj^{-1} b_j -- the same
                // forward substitution, on a different row
                const bool yl = tid == B * 6;
                const int d = tid / 6 + 1, a = tid - (d - 1) * 6;
                double xr[6];
                const double *Ar = yl ? &sB[mPb * 6] : &sA[d * 36 + a * 6];
#pragma unroll
                for (int b = 0; b < 6; ++b) {
                    double s = Ar[b];
#pragma unroll
                    for (int k = 0; k < b; ++k) s = fma(-xr[k], L[b][k], s);
                    xr[b] = s * il[b];
                }
                if (!yl && j + d >= N) {
#pragma unroll
                    for (int b = 0; b < 6; ++b) xr[b] = 0.0;
                }
                double *o = yl ? sY : &sL[j & 1][(d - 1) * 36 + a * 6];
#pragma unroll
                for (int b = 0; b < 6; ++b) o[b] = xr[b];
            }
        } else if (is_win && j > 0) {
            window_update(j - 1, wrapP(mP + 1));
        }
        BA_BAND_STAMP(2);
        __syncthreads();
        // ---- phase 2 ----
        BA_BAND_STAMP(3);
        if (tid < P * 36) {
            // column j + 1: A(j+1+d', j+1) = sA2 - L(j+1+d', j) L(j+1, j)^T  (d' = B: nothing to subtract)
            const int dd = tid / 36, ab = tid - dd * 36, a = ab / 6, b = ab - a * 6;
            double v = sA2[(j + 1) & 1][tid];
            if (dd < B) {
                const double *l1 = &sL[j & 1][dd * 36 + a * 6], *l2 = &sL[j & 1][b * 6];
#pragma unroll
                for (int k = 0; k < 6; ++k) v = fma(-l1[k], l2[k], v);
            }
            sA[tid] = v;
        }
        if (is_win) {
            // pairs of the slot that left the window (row j + 1) take the blocks of row rn = j + P + 1
            const int sn = wrapP(mP + 1);
            if (s1 == sn || s2 == sn) {
                const int so = (s1 == sn) ? s2 : s1;
                int mo = so - wrapP(wrapP(mP + 1) + 1);
                if (mo < 0) mo += P;
                mo += 2;                     // offset of the other row from j, in [2, P + 1] (P + 1: the diagonal pair)
                const int d = P + 1 - mo;    // block (rn, rn - d)
                const double *src = &sNew[d * 36 + h3 * 6];
#pragma unroll
                for (int i = 0; i < 3; ++i)
#pragma unroll
                    for (int b = 0; b < 6; ++b) acc[i][b] = src[i * 6 + b];
                if (d == B) {                // block (rn, j + 2): last block of column j + 2, no update to come
                    double *o = &sA2[j & 1][B * 36 + h3 * 6];
#pragma unroll
                    for (int i = 0; i < 3; ++i)
#pragma unroll
                        for (int b = 0; b < 6; ++b) o[i * 6 + b] = acc[i][b];
                }
            }
        } else if (aux >= 0) {
            // right-hand-side update b_{j+d} -= L(j+d, j) y_j; y_j to HBM
            if (aux < B * 6) {
                const int d = aux / 6 + 1, a = aux - (d - 1) * 6;
                const int sl = mPb + d >= Pb ? mPb + d - Pb : mPb + d;
                const double *l = &sL[j & 1][(d - 1) * 36 + a * 6];
                double s = sB[sl * 6 + a];
#pragma unroll
                for (int k = 0; k < 6; ++k) s = fma(-l[k], sY[k], s);
                sB[sl * 6 + a] = s;
            } else if (aux >= 120 && aux < 126) {
                sB[mPb * 6 + (aux - 120)] = sNew[P * 36 + (aux - 120)];  // slot (j + P + 1) mod (P + 1) = j mod (P + 1)
                ywork[(size_t)j * 6 + (aux - 120)] = sY[aux - 120];
            }
        }
        BA_BAND_STAMP(4);
        __syncthreads();
        BA_BAND_STAMP(5);
        mP = wrapP(mP + 1);
        mPb = mPb + 1 >= Pb ? 0 : mPb + 1;
    };
    for (int j = 0; j < n_elim; j += 2) {
        column_step(j, pa);
        if (j + 1 < n_elim) column_step(j + 1, pb);
    }
    if (aux >= 0 && n_elim > 0) copy_column(n_elim - 1);
    if (tid == 0 && sBad != 0) atomicMax(args.info, sBad);
    if (n_elim < N) {
        // separator: rows / columns je .. je+B-1.  Column je is in sA, column je+1 in sA2 (complete once the last
        // eliminated column has been applied to the window), the rest in the registers of the window threads.
        const int je = n_elim;
        double *mid = args.mid_l, *mid_g = args.mid_g;
        if (is_win && je > 0) window_update(je - 1, wrapP(mP + 1));  // (mP = je mod P here)
        __syncthreads();
        // block (je + tr, je + tc) of THIS side, tr >= tc -> band storage of the separator system (column = lower camera)
        auto emit = [&](int tr, int tc, int a, int b, double v) {
            if (!side.flip) {
                atomicAdd(&mid[((size_t)tc * B + (tr - tc)) * 36 + a * 6 + b], v);
            } else {  // reversed side: its lower block (tr, tc) holds rows of camera B-1-tr, columns of camera B-1-tc
                atomicAdd(&mid[((size_t)(B - 1 - tr) * B + (tr - tc)) * 36 + b * 6 + a], v);
            }
        };
        if (tid < B * 36) {
            const int d = tid / 36, ab = tid - d * 36;
            emit(d, 0, ab / 6, ab % 6, sA[tid]);
        }
        if (tid < (B - 1) * 36) {
            const int d = tid / 36, ab = tid - d * 36;
            emit(d + 1, 1, ab / 6, ab % 6, sA2[(je + 1) & 1][tid]);
        }
        if (is_win && je > 0) {
            const int jm = (je + 1) % P;
            int m1 = s1 - jm, m2 = s2 - jm;
            if (m1 < 0) m1 += P;
            if (m2 < 0) m2 += P;
            m1 += 2; m2 += 2;  // offsets of the two rows from je - 1
            if (m1 >= 3 && m1 <= B && m2 >= 3 && m2 <= B) {
                const int dr = m1 > m2 ? m1 : m2, dc = m1 > m2 ? m2 : m1;
#pragma unroll
                for (int i = 0; i < 3; ++i)
#pragma unroll
                    for (int b = 0; b < 6; ++b) emit(dr - 1, dc - 1, h3 + i, b, acc[i][b]);
            }
        }
        if (tid < B * 6) {  // (the solver kernels take g and solve for -g: store minus the right-hand side)
            const int t = tid / 6, k = tid - t * 6;
            atomicAdd(&mid_g[(side.flip ? B - 1 - t : t) * 6 + k], -sB[((je + t) % Pb) * 6 + k]);
        }
    }
}

// Pre-scaling for the backward substitution (parallel over block columns, after the factorisation):
//   Lb[j][d] <- W(j+d, j) = L(j+d, j) L_jj^{-1}  (d >= 1),   ywork[j] <- L_jj^{-T} y_j,
// which turns the backward recurrence into  x_j = yhat_j - sum_d W(j+d, j)^T x_{j+d}  with no triangular solve
// on its serial chain.  grid = N, block >= 6 B + 1 threads: thread (d, a) rescales one block row.
__global__ void k_band_scale(const BandArgs args) {
    if (*args.info != 0) return;
    const BandSide side = args.side[blockIdx.y];
    const int j = blockIdx.x, t = threadIdx.x, B = args.B, P = B + 1;
    if (t > B * 6 || j >= side.n_elim) return;
    double *__restrict__ Lb = side.Lb;
    double *__restrict__ ywork = side.ywork;
    double L[6][6], il[6], Li[6][6];  // Li = inverse of the lower Cholesky factor of the stored diagonal block
    band_chol6(&Lb[(size_t)j * P * 36], L, il);
#pragma unroll
    for (int c = 0; c < 6; ++c) {
        Li[c][c] = il[c];
#pragma unroll
        for (int r = c + 1; r < 6; ++r) {
            double s = 0.0;
#pragma unroll
            for (int k = c; k < r; ++k) s = fma(-L[r][k], Li[k][c], s);
            Li[r][c] = s * il[r];
        }
    }
    if (t < B * 6) {
        double *row = &Lb[((size_t)j * P + 1) * 36 + t * 6], v[6], w[6];
#pragma unroll
        for (int k = 0; k < 6; ++k) v[k] = row[k];
#pragma unroll
        for (int b = 0; b < 6; ++b) {
            double s = 0.0;
#pragma unroll
            for (int k = b; k < 6; ++k) s = fma(v[k], Li[k][b], s);
            w[b] = s;
        }
#pragma unroll
        for (int k = 0; k < 6; ++k) row[k] = w[k];
    } else {
        double *y = &ywork[(size_t)j * 6], v[6], w[6];
#pragma unroll
        for (int k = 0; k < 6; ++k) v[k] = y[k];
#pragma unroll
        for (int b = 0; b < 6; ++b) {
            double s = 0.0;
#pragma unroll
            for (int k = b; k < 6; ++k) s = fma(Li[k][b], v[k], s);
            w[b] = s;
        }
#pragma unroll
        for (int k = 0; k < 6; ++k) y[k] = w[k];
    }
}

template <int CTRL>
__device__ __forceinline__ double band_dpp(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xf, 0xf, true);
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double band_readlane(double v, int lane) {
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), lane),
                            __builtin_amdgcn_readlane(__double2loint(v), lane));
}

#define BA_BAND_BACK_THREADS 64
#define BA_BAND_BK 15  // pairs (d >= 2, a) per lane: ceil(6 (BA_BAND_MAX - 1) / 8)
// doubles read past the last block column by the masked prefetch lanes: the band buffer is allocated this much longer
#define BA_BAND_PAD (72 + BA_BAND_BK * 48 + 8)

// Backward substitution  x_j = yhat_j - sum_{d=1..B} W(j+d, j)^T x_{j+d}  (after k_band_scale): ONE wave, the
// recurrence is serial in j.  lane = k * 8 + g, k = output component.  Only the d = 1 term is on the serial
// chain: lane (k, .) sums it in-lane over a with x_{j+1} broadcast through SGPRs (readlane).  The d >= 2 terms
// of column j - 1 (pairs q' = g + 8 i spread over the 8 lanes of a group, reduced with DPP) are evaluated one
// column ahead, off the chain, from the solution window in LDS.  The window is stored twice (slots s and s + P)
// so that a lane's reads are base + constant offsets without wrap-around; columns are prefetched three ahead.
// With a separator the recurrence of a side starts below it (column n_elim - 1) from the separator's solution.
__global__ __launch_bounds__(BA_BAND_BACK_THREADS) void k_band_backward(const BandArgs args) {
    constexpr int PM = BA_BAND_MAX + 1;
    __shared__ __attribute__((aligned(16))) double sX[2 * PM * 6 + 8 * BA_BAND_BK + 24];
    if (*args.info != 0) return;
    const BandSide side = args.side[blockIdx.x];
    const int N = side.N, B = args.B, P = B + 1, start = side.n_elim;
    const double *__restrict__ Lb = side.Lb;
    const double *__restrict__ ywork = side.ywork;
    double *__restrict__ x = args.x;
    const long long xdir = side.flip ? -6 : 6, xorg = (long long)side.cam0 * 6;  // x index of (column j, k) = xorg + j xdir + k
    const int lane = threadIdx.x, bk = lane >> 3, bg = lane & 7;
    for (int e = lane; e < 2 * PM * 6 + 8 * BA_BAND_BK + 24; e += 64) sX[e] = 0.0;
    band_wave_sync();
    double xs[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};  // x_{j+1}, wave-uniform
    if (start < N) {
        // separator rows start .. start+B-1 of this side = separator cameras in this side's order
        const double *xsep = args.x + (size_t)args.sep * 6;  // solved before this launch
        for (int e = lane; e < B * 6; e += 64) {
            const int t = e / 6, k = e - t * 6, sl = (start + t) % P;
            const double v = xsep[(side.flip ? B - 1 - t : t) * 6 + k];
            sX[sl * 6 + k] = v;
            sX[(sl + P) * 6 + k] = v;
        }
#pragma unroll
        for (int a = 0; a < 6; ++a) xs[a] = xsep[(side.flip ? B - 1 : 0) * 6 + a];
    }
    double msk[BA_BAND_BK];
#pragma unroll
    for (int i = 0; i < BA_BAND_BK; ++i) msk[i] = (bg + 8 * i < 6 * (B - 1)) ? 1.0 : 0.0;
    const double msk1 = B >= 1 ? 1.0 : 0.0;
    const int kk = bk < 6 ? bk : 0;
    struct Col { double w1[6], wq[BA_BAND_BK], y; };
    auto load_col = [&](int j, Col &c) {
        const long long jc = j > 0 ? j : 0;
        const double *col = Lb + (jc * P + 1) * 36 + kk;
#pragma unroll
        for (int a = 0; a < 6; ++a) c.w1[a] = col[a * 6];
        const double *cq = col + 36 + bg * 6;
#pragma unroll
        for (int i = 0; i < BA_BAND_BK; ++i) c.wq[i] = cq[i * 48];
        c.y = ywork[jc * 6 + kk];
    };
    // window values for column jc: x_{jc+d}[a], d >= 2, i.e. flattened index (jc % P) * 6 + 12 + q'
    auto window_read = [&](int jc, double (&xv)[BA_BAND_BK]) {
        int jm = jc % P;
        if (jm < 0) jm += P;
        const double *xw = &sX[jm * 6 + 12 + bg];
#pragma unroll
        for (int i = 0; i < BA_BAND_BK; ++i) xv[i] = xw[8 * i];
    };
    // base_jc = yhat_jc - (d >= 2 terms), reduced over the 8 lanes of a group
    auto window_base = [&](const Col &cm, const double (&xv)[BA_BAND_BK]) -> double {
        double p0 = 0.0, p1 = 0.0, p2 = 0.0;
#pragma unroll
        for (int i = 0; i + 2 < BA_BAND_BK; i += 3) {
            p0 = fma(cm.wq[i] * msk[i], xv[i], p0);
            p1 = fma(cm.wq[i + 1] * msk[i + 1], xv[i + 1], p1);
            p2 = fma(cm.wq[i + 2] * msk[i + 2], xv[i + 2], p2);
        }
        double p = (p0 + p1) + p2;
        p += band_dpp<0xB1>(p);   // quad_perm [1,0,3,2]
        p += band_dpp<0x4E>(p);   // quad_perm [2,3,0,1]
        p += band_dpp<0x141>(p);  // row_half_mirror: the other quad of the 8-lane group
        return cm.y - p;
    };
    // Two banks of three columns: a bank is refilled right after its last use, three columns before its next
    // use.  (The compiler drains all outstanding loads at the loop head; with this order the youngest of them
    // is three columns old there.)
    Col x0, x1, x2, y0, y1, y2;
    load_col(start - 1, x0);
    load_col(start - 2, x1);
    load_col(start - 3, x2);
    band_wave_sync();
    double base;
    {
        double xv[BA_BAND_BK];
        window_read(start - 1, xv);
        base = window_base(x0, xv);
    }
    // step(j): cj = column j (its d = 1 part), cm = column j - 1 (its d >= 2 part and yhat)
    auto back_step = [&](const int j, const Col &cj, const Col &cm) {
        const int js = j % P;
        double xv[BA_BAND_BK];
        window_read(j - 1, xv);
        // chain: x_j[k] = base_j[k] - sum_a W(j+1, j)[a][k] x_{j+1}[a]
        double s0 = base, s1 = 0.0;
#pragma unroll
        for (int a = 0; a < 6; a += 2) {
            s0 = fma(-cj.w1[a] * msk1, xs[a], s0);
            s1 = fma(-cj.w1[a + 1] * msk1, xs[a + 1], s1);
        }
        const double xk = s0 + s1;
#pragma unroll
        for (int a = 0; a < 6; ++a) xs[a] = band_readlane(xk, a * 8);
        if (bg == 0 && bk < 6) {
            sX[js * 6 + bk] = xk;
            sX[(js + P) * 6 + bk] = xk;
            x[xorg + (long long)j * xdir + bk] = xk;
        }
        base = window_base(cm, xv);  // off the chain: base_{j-1}
        band_wave_sync();
    };
    for (int j = start - 1; j >= 0; j -= 6) {
        load_col(j - 3, y0);
        load_col(j - 4, y1);
        load_col(j - 5, y2);
        back_step(j, x0, x1);
        if (j >= 1) back_step(j - 1, x1, x2);
        if (j >= 2) back_step(j - 2, x2, y0);
        if (j < 3) break;
        load_col(j - 6, x0);
        load_col(j - 7, x1);
        load_col(j - 8, x2);
        back_step(j - 3, y0, y1);
        if (j >= 4) back_step(j - 4, y1, y2);
        if (j >= 5) back_step(j - 5, y2, x0);
    }
}

// ------------------------------------------------------------------------------------------ host side
// Device buffers of one solve (owned by the caller).  sizes in doubles for N cameras, half bandwidth B:
struct BandWork {
    double *l, *y, *g, *mid;
    int *info;
};
inline size_t band_l_doubles(int N, int B) { return ((size_t)N + B) * (B + 1) * 36 + 2 * BA_BAND_PAD; }
inline size_t band_vec_doubles(int N, int B) { return ((size_t)N + B) * 6 + 16; }
inline size_t band_mid_doubles(int B) {  // separator system: band storage, right-hand side, work vector
    return B < 1 ? 16 : band_l_doubles(B, B - 1) + 2 * band_vec_doubles(B, B - 1);
}
// chains at least this long are eliminated from both ends
inline bool band_two_sided(int N, int B) { return B >= 1 && N >= 6 * (B + 1); }

// x = -S^{-1} g for the block-banded reduced camera system given by its upper blocks; *info (device) = 0 on success
inline hipError_t band_solve_launch(hipStream_t st, int N, int B, int n_upper, const long long *keys, const double *S,
                                    const double *g, double *x, const BandWork &w) {
    const int P = B + 1;
    hipError_t e;
    if ((e = hipMemsetAsync(w.l, 0, band_l_doubles(N, B) * sizeof(double), st)) != hipSuccess) return e;
    if ((e = hipMemsetAsync(w.info, 0, sizeof(int), st)) != hipSuccess) return e;
    BandArgs a{};
    a.B = B; a.x = x; a.info = w.info;
    const bool two = band_two_sided(N, B);
    const int m = two ? (N - B) / 2 : -1;
    int sides = 1;
    if (!two) {
        a.side[0] = BandSide{w.l, const_cast<double *>(g), w.y, N, N, 0, 0};
        a.side[1] = a.side[0];
    } else {
        sides = 2;
        const int Nt = m + B, Nb = N - m;
        a.side[0] = BandSide{w.l, w.g, w.y, Nt, m, 0, 0};
        a.side[1] = BandSide{w.l + (size_t)Nt * P * 36 + BA_BAND_PAD, w.g + (size_t)Nt * 6, w.y + (size_t)Nt * 6, Nb, Nb - B, 1, N - 1};
        a.mid_l = w.mid;
        a.mid_g = w.mid + band_l_doubles(B, B - 1);
        a.sep = m;
        if ((e = hipMemsetAsync(w.mid, 0, band_mid_doubles(B) * sizeof(double), st)) != hipSuccess) return e;
        hipLaunchKernelGGL(k_band_rhs, dim3((N * 6 + 255) / 256), dim3(256), 0, st, N, B, m, g, a.side[0].g, a.side[1].g);
    }
    const int ne = n_upper * 36;
    hipLaunchKernelGGL(k_band_assemble, dim3((ne + 255) / 256), dim3(256), 0, st, n_upper, keys, S, B, N, m, a.side[0].Lb,
                       a.side[1].Lb);
    hipLaunchKernelGGL(k_band_cholesky_solve, dim3(sides), dim3(band_threads(B)), 0, st, a);
    if (two) {
        // the separator system: B cameras, half bandwidth B - 1, one-sided, solution straight into x
        BandArgs c{};
        c.B = B - 1; c.x = x; c.info = w.info;
        c.side[0] = BandSide{a.mid_l, a.mid_g, a.mid_g + band_vec_doubles(B, B - 1), B, B, 0, m};
        c.side[1] = c.side[0];
        hipLaunchKernelGGL(k_band_cholesky_solve, dim3(1), dim3(band_threads(B - 1)), 0, st, c);
        hipLaunchKernelGGL(k_band_scale, dim3(B, 1), dim3((((B - 1) * 6 + 1 + 63) / 64) * 64), 0, st, c);
        hipLaunchKernelGGL(k_band_backward, dim3(1), dim3(BA_BAND_BACK_THREADS), 0, st, c);
    }
    const int ncol = two ? (a.side[0].n_elim > a.side[1].n_elim ? a.side[0].n_elim : a.side[1].n_elim) : N;
    hipLaunchKernelGGL(k_band_scale, dim3(ncol, sides), dim3(((B * 6 + 1 + 63) / 64) * 64), 0, st, a);
    hipLaunchKernelGGL(k_band_backward, dim3(sides), dim3(BA_BAND_BACK_THREADS), 0, st, a);
    return hipGetLastError();
}

}  // namespace ba
