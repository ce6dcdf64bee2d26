// The "rows" mode of the build pass: the Schur complement from GLOBAL per-block entry lists.
//
// k_build (ba_kernels.h) eliminates the points of a chunk inside one workgroup and relies on structure INSIDE the chunk:
// neighbouring points see the same cameras, so a chunk has few camera pairs with many entries each, and one flush per
// chunk.  Unstructured co-visibility has none of that -- every point of a chunk brings its own cameras, the pair groups
// hold one entry of eight, and each of them is a 288-byte read-modify-write at the memory side (the object-centric
// shape orbit1024: 2.6 M groups for 4.3 M entries, 746 MB of atomics per pass).  Here the pass is cut the other way:
//   * k_lin: one chunk of whole points per workgroup (no camera slots, no pairs): residuals, weights, the point blocks
//     V, g_p (gathered per point in LDS), their damped factor, and per observation a row of 16 doubles (128 bytes: one cache line per gather) and a
//     tail of 8 in HBM, written coalesced:
//         T' = Wt L^-T (9), Y = R X (3), w = T' u (3), and what the camera sums need: the projection derivatives a, b, p02,
//         p12, the curvature weights wh (2) and the weighted residuals w r (2);
//   * k_rows_cross: one lane pair per off-diagonal block (i < j) of S that has entries: it walks the block's entry list
//     (observation i, observation j), gathers the two rows from L2 / HBM, forms sum T_i T_j^T in registers exactly as
//     k_build's Schur phase does, and STORES the block -- every block is written once, by its owner, no atomics;
//   * k_rows_diag: one workgroup per free camera: its observations (the entries of its diagonal block) give U, g_c, its part
//     of the reduced gradient and the diagonal block, summed over the lanes in a fixed order.
// (k_rows_diag and k_rows_cross are the two kinds of workgroup of ONE launch, k_rows_schur: they only depend on k_lin.)
// Every floating-point sum of these three kernels has a fixed order (k_lin gathers each point's sums over its consecutive
// lanes): without heavy points two passes over the same state give the same bits.
// The host builds the lists once per structure (analyze_structure(..., rows); Symbolic::rw_ent / rw_ptr).  Heavy points (tracks
// no chunk holds) get their rows from k_lin_heavy and sit in the same lists; in the CHUNK mode the lists hold the heavy points
// only and the ADD variants of k_rows_* run behind k_build.
#pragma once
#include "ba_heavy.h"
#include "ba_kernels.h"

namespace ba {

constexpr int LIN_ROW = 17;     // doubles per observation in k_lin's store staging (odd)
constexpr int CROSS_ROW = 37;   // doubles per block in the store staging of rows_cross_block (odd: no bank conflicts)
constexpr int ROW_A = 16, ROW_B = 8, ROW_DOUBLES = ROW_A + ROW_B;   // rows of 128 B (one cache line per gather) + the camera-sum tails

struct RowsArgs {
    double *rows;                 // [n_obs * ROW_A]: T' (9), Y (3), w (3), pad -- internal observation order, 128-byte rows
    double *tails;                // [n_obs * ROW_B]: a, b, p02, p12, wh (2), w r (2)
    const int *ent;               // 2 per entry
    const long long *ptr;         // [n_keys + 1]
    const int *key_slot;          // block slot of each key (Symbolic::rw_slot)
    long long row_off;            // rows / tails are indexed by (observation - row_off): 0 in the rows mode, the first heavy
                                  // observation in the chunk mode (rows of the heavy points only)
    const int *xblk;              // keys of the off-diagonal blocks
    const int *diag_key;          // [n_free] key of the camera's diagonal block, or -1
    int n_x;
};

template <int LOSS>
__global__ __launch_bounds__(CHUNK_OBS) void k_lin(const BuildArgs A, double *__restrict__ rows, double *__restrict__ tails) {
    // per observation: its part of V (6), g_p (3); afterwards the staging of the rows and tails on their way out (LIN_ROW per lane)
    __shared__ __attribute__((aligned(16))) double sV[CHUNK_OBS * LIN_ROW];
    __shared__ __attribute__((aligned(16))) double sP[CHUNK_PTS * 9];      // per point: L^-1 (6), u (3)
    __shared__ int sStart[CHUNK_PTS + 1];                                  // first lane of each point's observations
    __shared__ double sRed[8];
    const int ck = blockIdx.x, tid = threadIdx.x;
    const int ob = A.ck_obs[ck], nobs = A.ck_obs[ck + 1] - ob;
    const int pb = A.ck_pt[ck], npts = A.ck_pt[ck + 1] - pb;
    const double lam = A.state->lambda, floor_now = A.state->floor;
    const bool act = tid < nobs;
    const int oo = ob + (act ? tid : 0);
    const int cam = A.o_cam[oo];
    const int p = A.o_pt[oo];
    const int p_prev = A.o_pt[oo > ob ? oo - 1 : oo];
    const double2 xy = reinterpret_cast<const double2 *>(A.o_xy)[oo];
    CamAux ca;
    load_aux_local(A.aux, cam, ca);
    const double X0 = A.points[3 * (size_t)p], X1 = A.points[3 * (size_t)p + 1], X2 = A.points[3 * (size_t)p + 2];
    const bool free_cam = A.cam_free[cam] >= 0;

    double Wt[9], Yr[3], tail[8];
    const int lp = act ? p - pb : 0;
    // the observations of a point are consecutive lanes: each point's sums are gathered by ONE lane in lane order (no
    // atomics -- with the fixed order of the other two kernels every pass over the same state gives the same bits)
    if (act && (tid == 0 || p != p_prev)) sStart[lp] = tid;
    if (tid == 0) sStart[npts] = nobs;
    double cost_loc = 0.0;
    if (act) {
        ObsLin L;
        linearize(ca, X0, X1, X2, xy.x, xy.y, LOSS >= 0 ? LOSS : A.loss, A.inv_fs2, floor_now, L);
        cost_loc = A.half_fs2 * L.rho_sum;
        Yr[0] = L.Y[0]; Yr[1] = L.Y[1]; Yr[2] = L.Y[2];
        double wJp[2][3], vs[9];
#pragma unroll
        for (int j = 0; j < 3; ++j) { wJp[0][j] = L.wh[0] * L.Jp[0][j]; wJp[1][j] = L.wh[1] * L.Jp[1][j]; }
        vs[0] = wJp[0][0] * L.Jp[0][0] + wJp[1][0] * L.Jp[1][0];
        vs[1] = wJp[0][0] * L.Jp[0][1] + wJp[1][0] * L.Jp[1][1];
        vs[2] = wJp[0][0] * L.Jp[0][2] + wJp[1][0] * L.Jp[1][2];
        vs[3] = wJp[0][1] * L.Jp[0][1] + wJp[1][1] * L.Jp[1][1];
        vs[4] = wJp[0][1] * L.Jp[0][2] + wJp[1][1] * L.Jp[1][2];
        vs[5] = wJp[0][2] * L.Jp[0][2] + wJp[1][2] * L.Jp[1][2];
        const double wr0 = L.w[0] * L.r[0], wr1 = L.w[1] * L.r[1];
        vs[6] = L.Jp[0][0] * wr0 + L.Jp[1][0] * wr1;
        vs[7] = L.Jp[0][1] * wr0 + L.Jp[1][1] * wr1;
        vs[8] = L.Jp[0][2] * wr0 + L.Jp[1][2] * wr1;
        // translation rows of W = Jc'^T wh Jp (k_build's Wt)
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j) Wt[i * 3 + j] = (L.wh[0] * L.Jc[0][3 + i]) * L.Jp[0][j] + (L.wh[1] * L.Jc[1][3 + i]) * L.Jp[1][j];
        // P = [[a, 0, p02], [0, b, p12]] (linearize): the camera block of the observation is P [-[Y]x | I]
        tail[0] = L.Jc[0][3]; tail[1] = L.Jc[1][4]; tail[2] = L.Jc[0][5]; tail[3] = L.Jc[1][5];
        tail[4] = L.wh[0]; tail[5] = L.wh[1]; tail[6] = wr0; tail[7] = wr1;
        double *vp = &sV[tid * 9];
#pragma unroll
        for (int k = 0; k < 9; ++k) vp[k] = vs[k];
    }
    lds_barrier();

    double gmax_loc = 0.0;
    if (tid < npts) {
        double *vp = &sP[tid * 9];
        double a[9];
#pragma unroll
        for (int k = 0; k < 9; ++k) a[k] = 0.0;
        for (int o = sStart[tid], o1 = sStart[tid + 1]; o < o1; ++o) {
#pragma unroll
            for (int k = 0; k < 9; ++k) a[k] += lds_f64(&sV[o * 9 + k]);
        }
        const double v00 = a[0], v01 = a[1], v02 = a[2], v11 = a[3], v12 = a[4], v22 = a[5];
        const double g0 = a[6], g1 = a[7], g2 = a[8];
        gmax_loc = fmax(fabs(g0), fmax(fabs(g1), fabs(g2)));
        if (A.dbg_v) {
            double *dv = &A.dbg_v[(size_t)(pb + tid) * 6];
            dv[0] = v00; dv[1] = v01; dv[2] = v02; dv[3] = v11; dv[4] = v12; dv[5] = v22;
            double *dg = &A.dbg_gp[(size_t)(pb + tid) * 3];
            dg[0] = g0; dg[1] = g1; dg[2] = g2;
        }
        // (k_build's phase 2a, the same arithmetic)
        const double a00 = v00 + lam * fmin(fmax(v00, BA_D_MIN), BA_D_MAX);
        const double a11 = v11 + lam * fmin(fmax(v11, BA_D_MIN), BA_D_MAX);
        const double a22 = v22 + lam * fmin(fmax(v22, BA_D_MIN), BA_D_MAX);
        const double i00 = rsqrt_nr(a00);
        const double l10 = v01 * i00, l20 = v02 * i00;
        const double i11 = rsqrt_nr(fmax(a11 - l10 * l10, 1e-15 * a11));
        const double l21 = (v12 - l20 * l10) * i11;
        const double i22 = rsqrt_nr(fmax(a22 - l20 * l20 - l21 * l21, 1e-15 * a22));
        const double i10 = -l10 * i00 * i11;
        const double i21 = -l21 * i11 * i22;
        const double i20 = -(l20 * i00 + l21 * i10) * i22;
        const double u0 = i00 * g0, u1 = i10 * g0 + i11 * g1, u2 = i20 * g0 + i21 * g1 + i22 * g2;
        vp[0] = i00; vp[1] = i10; vp[2] = i11; vp[3] = i20; vp[4] = i21; vp[5] = i22;
        vp[6] = u0; vp[7] = u1; vp[8] = u2;
        double *gl = &A.pt_linv[(size_t)(pb + tid) * 6];
        gl[0] = i00; gl[1] = i10; gl[2] = i11; gl[3] = i20; gl[4] = i21; gl[5] = i22;
        double *gu = &A.pt_u[(size_t)(pb + tid) * 3];
        gu[0] = u0; gu[1] = u1; gu[2] = u2;
    }
    {
        const double cs = wave_sum_dpp(cost_loc), gm = wave_max_dpp(gmax_loc);
        if ((tid & 63) == 0) { sRed[tid >> 6] = cs; sRed[4 + (tid >> 6)] = gm; }
    }
    lds_barrier();
    if (tid == 0) {
        A.ck_cost[ck] = (sRed[0] + sRed[1]) + (sRed[2] + sRed[3]);
        A.ck_gmax[ck] = fmax(fmax(sRed[4], sRed[5]), fmax(sRed[6], sRed[7]));
    }
    // rows and tails leave through LDS: a wave's 64 rows are ONE run of 8 KB (its tails one of 4 KB), stored with consecutive lanes
    // on consecutive 16-byte pieces.  (sV is free: the per-point gather is behind the barrier above.)  Observations of fixed
    // cameras get zeros: no entry ever names them.
    {
        double t[9], w0 = 0.0, w1 = 0.0, w2 = 0.0;
#pragma unroll
        for (int k = 0; k < 9; ++k) t[k] = 0.0;
        const bool on = act && free_cam;
        if (on) {
            const double *vp = &sP[lp * 9];
            const double i00 = vp[0], i10 = vp[1], i11 = vp[2], i20 = vp[3], i21 = vp[4], i22 = vp[5];
            const double u0 = vp[6], u1 = vp[7], u2 = vp[8];
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                t[i * 3] = Wt[i * 3] * i00;
                t[i * 3 + 1] = Wt[i * 3] * i10 + Wt[i * 3 + 1] * i11;
                t[i * 3 + 2] = Wt[i * 3] * i20 + Wt[i * 3 + 1] * i21 + Wt[i * 3 + 2] * i22;
            }
            w0 = t[0] * u0 + t[1] * u1 + t[2] * u2; w1 = t[3] * u0 + t[4] * u1 + t[5] * u2; w2 = t[6] * u0 + t[7] * u1 + t[8] * u2;
        }
        const int lane = tid & 63, wv = tid >> 6;
        const int n_w = nobs - wv * 64 < 64 ? nobs - wv * 64 : 64;      // observations of this wave (may be <= 0)
        double *stg = &sV[wv * 64 * LIN_ROW];
        double *mine = stg + lane * LIN_ROW;
#pragma unroll
        for (int k = 0; k < 9; ++k) mine[k] = t[k];
        mine[9] = on ? Yr[0] : 0.0; mine[10] = on ? Yr[1] : 0.0; mine[11] = on ? Yr[2] : 0.0;
        mine[12] = w0; mine[13] = w1; mine[14] = w2; mine[15] = 0.0;
        wave_lds_sync();
        double2 *out = reinterpret_cast<double2 *>(rows + (size_t)(ob + wv * 64) * ROW_A);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int idx = lane + 64 * k, b = idx >> 3, c = idx & 7;
            if (b < n_w) out[idx] = make_double2(stg[b * LIN_ROW + 2 * c], stg[b * LIN_ROW + 2 * c + 1]);
        }
        wave_lds_sync();
#pragma unroll
        for (int k = 0; k < 8; ++k) mine[k] = on ? tail[k] : 0.0;
        wave_lds_sync();
        double2 *outt = reinterpret_cast<double2 *>(tails + (size_t)(ob + wv * 64) * ROW_B);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int idx = lane + 64 * k, b = idx >> 2, c = idx & 3;
            if (b < n_w) outt[idx] = make_double2(stg[b * LIN_ROW + 2 * c], stg[b * LIN_ROW + 2 * c + 1]);
        }
    }
}

// acc += T_i T_j^T for T = [[Y]x T'; T'] (k_build's phase 3: bottom rows B = [N [Yj]x^T | N], N = T'_i T'_j^T, top rows [Yi]x B)
__device__ __forceinline__ void schur_entry(const double *__restrict__ Ti, const double *__restrict__ Yi, const double *__restrict__ Tj,
                                            const double *__restrict__ Yj, double *__restrict__ acc) {
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        const double a0 = Ti[3 * r], a1 = Ti[3 * r + 1], a2 = Ti[3 * r + 2];
        double B[6];
        B[3] = a0 * Tj[0] + a1 * Tj[1] + a2 * Tj[2];
        B[4] = a0 * Tj[3] + a1 * Tj[4] + a2 * Tj[5];
        B[5] = a0 * Tj[6] + a1 * Tj[7] + a2 * Tj[8];
        B[0] = Yj[1] * B[5] - Yj[2] * B[4];
        B[1] = Yj[2] * B[3] - Yj[0] * B[5];
        B[2] = Yj[0] * B[4] - Yj[1] * B[3];
        const int s1 = (r + 1) % 3, s2 = (r + 2) % 3;
#pragma unroll
        for (int c = 0; c < 6; ++c) {
            acc[(3 + r) * 6 + c] += B[c];
            acc[s1 * 6 + c] += Yi[s2] * B[c];
            acc[s2 * 6 + c] -= Yi[s1] * B[c];
        }
    }
}

__device__ __forceinline__ void load_row12(const double *__restrict__ rows, long long o, double *__restrict__ T, double *__restrict__ Y) {
    const double2 *q = reinterpret_cast<const double2 *>(rows + (size_t)o * ROW_A);
    const double2 v0 = q[0], v1 = q[1], v2 = q[2], v3 = q[3], v4 = q[4], v5 = q[5];
    T[0] = v0.x; T[1] = v0.y; T[2] = v1.x; T[3] = v1.y; T[4] = v2.x; T[5] = v2.y; T[6] = v3.x; T[7] = v3.y; T[8] = v4.x;
    Y[0] = v4.y; Y[1] = v5.x; Y[2] = v5.y;
}

// One lane pair per off-diagonal block with entries; lane h of the pair takes entries 4 h .. 4 h + 3 of every run of eight.
// ADD: the blocks are added to (chunk mode: the heavy points' pairs on top of what the chunk kernel accumulated; every
// block still has ONE owner here, and the launch is ordered behind the chunk kernel: plain read-modify-write).
template <bool ADD>
__device__ __forceinline__ void rows_cross_block(const RowsArgs &R, double *__restrict__ S, const int blk) {
    const int tid = threadIdx.x, half = tid & 1;
    const int g = blk * 128 + (tid >> 1);
    const bool valid = g < R.n_x;
    const int key = R.xblk[valid ? g : 0];
    const long long e0 = R.ptr[key], e1 = valid ? R.ptr[key + 1] : e0;
    double acc[36];
#pragma unroll
    for (int k = 0; k < 36; ++k) acc[k] = 0.0;
    for (long long base = e0 + 4 * half; base < e1; base += 8) {
        int oi[4], oj[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {   // (clamped: the loads are unconditional)
            const int2 v = reinterpret_cast<const int2 *>(R.ent)[base + e < e1 ? base + e : e0];
            oi[e] = v.x; oj[e] = v.y;
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            double Ti[9], Yi[3], Tj[9], Yj[3];
            load_row12(R.rows, oi[e] - R.row_off, Ti, Yi);
            load_row12(R.rows, oj[e] - R.row_off, Tj, Yj);
            if (base + e < e1) schur_entry(Ti, Yi, Tj, Yj, acc);
        }
    }
#pragma unroll
    for (int k = 0; k < 36; ++k) acc[k] += dpp_f64<0xB1>(acc[k]);
    const int slot = valid ? R.key_slot[key] : -1;
    {
        // A wave whose 32 blocks sit side by side in S (the usual case: the keys are ascending, only a diagonal block in between
        // breaks the run) stores them as ONE run of 9 KB through LDS, consecutive lanes on consecutive 16-byte pieces: 144 full
        // lines instead of 18 store instructions of 64 scattered 8-byte pieces each (ADD: read-modify-write of the run, the same way).
        __shared__ __attribute__((aligned(16))) double sRun[4][32 * CROSS_ROW];
        const int lane = tid & 63, w = tid >> 6;
        const int slot0 = __builtin_amdgcn_readfirstlane(slot);
        if (__all(valid && slot == slot0 + (lane >> 1))) {
            double *mine = &sRun[w][(lane >> 1) * CROSS_ROW + 18 * half];
            if (half == 0) {
#pragma unroll
                for (int k = 0; k < 18; ++k) mine[k] = -acc[k];
            } else {
#pragma unroll
                for (int k = 0; k < 18; ++k) mine[k] = -acc[18 + k];
            }
            wave_lds_sync();
            double2 *run = reinterpret_cast<double2 *>(S + (size_t)slot0 * 36);
#pragma unroll
            for (int k = 0; k < 9; ++k) {
                const int idx = lane + 64 * k, b = idx / 18, c = idx - b * 18;
                const double2 was = ADD ? run[idx] : make_double2(0.0, 0.0);
                run[idx] = make_double2(was.x + sRun[w][b * CROSS_ROW + 2 * c], was.y + sRun[w][b * CROSS_ROW + 2 * c + 1]);
            }
            return;
        }
    }
    if (valid) {
        double *d = S + (size_t)slot * 36;
        if (half == 0) {
#pragma unroll
            for (int k = 0; k < 18; ++k) d[k] = (ADD ? d[k] : 0.0) - acc[k];
        } else {
#pragma unroll
            for (int k = 18; k < 36; ++k) d[k] = (ADD ? d[k] : 0.0) - acc[k];
        }
    }
}

// One workgroup per free camera: the entries of its diagonal block are its observations (i, i) -- each also gives the
// camera's U, g_c and its part of the reduced gradient -- and, for a camera that sees a point twice, the pairs (i, j), (j, i).
constexpr int DIAG_VALS = 36 + 21 + 6 + 6;
template <bool ADD>
__device__ __forceinline__ void rows_diag_block(const RowsArgs &R, double *__restrict__ S, double *__restrict__ U, double *__restrict__ gc,
                                                double *__restrict__ gred, const int f) {
    __shared__ double sW[4][DIAG_VALS];
    const int tid = threadIdx.x;
    const int key = R.diag_key[f];
    if (key < 0) return;   // (no observation outside the heavy points: the cleared accumulators stand)
    const long long e0 = R.ptr[key], e1 = R.ptr[key + 1];
    double acc[36], u[21], g[6], gr[6];
#pragma unroll
    for (int k = 0; k < 36; ++k) acc[k] = 0.0;
#pragma unroll
    for (int k = 0; k < 21; ++k) u[k] = 0.0;
#pragma unroll
    for (int k = 0; k < 6; ++k) { g[k] = 0.0; gr[k] = 0.0; }
    for (long long e = e0 + tid; e < e1; e += 256) {
        const int2 v = reinterpret_cast<const int2 *>(R.ent)[e];
        double Ti[9], Yi[3], Tj[9], Yj[3];
        load_row12(R.rows, v.x - R.row_off, Ti, Yi);
        load_row12(R.rows, v.y - R.row_off, Tj, Yj);
        schur_entry(Ti, Yi, Tj, Yj, acc);
        if (v.x == v.y) {
            const double2 *q = reinterpret_cast<const double2 *>(R.rows + (size_t)(v.x - R.row_off) * ROW_A);
            const double2 *qt = reinterpret_cast<const double2 *>(R.tails + (size_t)(v.x - R.row_off) * ROW_B);
            const double2 t6 = q[6], t7 = q[7], t8 = qt[0], t9 = qt[1], t10 = qt[2], t11 = qt[3];
            const double w0 = t6.x, w1 = t6.y, w2 = t7.x;
            gr[0] += Yi[1] * w2 - Yi[2] * w1;
            gr[1] += Yi[2] * w0 - Yi[0] * w2;
            gr[2] += Yi[0] * w1 - Yi[1] * w0;
            gr[3] += w0; gr[4] += w1; gr[5] += w2;
            const double a = t8.x, b = t8.y, p02 = t9.x, p12 = t9.y, wh0 = t10.x, wh1 = t10.y, wr0 = t11.x, wr1 = t11.y;
            // the camera block of the observation as linearize forms it in the local rotation tangent: P (-[Y]x) | P
            double Jc[2][6];
            Jc[0][0] = p02 * Yi[1]; Jc[0][1] = a * Yi[2] - p02 * Yi[0]; Jc[0][2] = -a * Yi[1];
            Jc[1][0] = -b * Yi[2] + p12 * Yi[1]; Jc[1][1] = -p12 * Yi[0]; Jc[1][2] = b * Yi[0];
            Jc[0][3] = a; Jc[0][4] = 0.0; Jc[0][5] = p02;
            Jc[1][3] = 0.0; Jc[1][4] = b; Jc[1][5] = p12;
#pragma unroll
            for (int i = 0; i < 6; ++i) {
                const double x0 = wh0 * Jc[0][i], x1 = wh1 * Jc[1][i];
#pragma unroll
                for (int j = i; j < 6; ++j) u[ut6(i, j)] += x0 * Jc[0][j] + x1 * Jc[1][j];
                g[i] += Jc[0][i] * wr0 + Jc[1][i] * wr1;
            }
        }
    }
    // fixed-order sum over the workgroup: DPP wave sums, then the four waves in wave order
    const int w = tid >> 6;
#pragma unroll
    for (int k = 0; k < 36; ++k) { const double s = wave_sum_dpp(acc[k]); if ((tid & 63) == 0) sW[w][k] = s; }
#pragma unroll
    for (int k = 0; k < 21; ++k) { const double s = wave_sum_dpp(u[k]); if ((tid & 63) == 0) sW[w][36 + k] = s; }
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        const double s = wave_sum_dpp(g[k]), s2 = wave_sum_dpp(gr[k]);
        if ((tid & 63) == 0) { sW[w][57 + k] = s; sW[w][63 + k] = s2; }
    }
    __syncthreads();
    if (tid < DIAG_VALS) {
        const double v = (sW[0][tid] + sW[1][tid]) + (sW[2][tid] + sW[3][tid]);
        double *dst = tid < 36 ? &S[(size_t)R.key_slot[key] * 36 + tid] : tid < 57 ? &U[(size_t)f * 21 + (tid - 36)]
                                                                           : tid < 63 ? &gc[(size_t)f * 6 + (tid - 57)] : &gred[(size_t)f * 6 + (tid - 63)];
        const double sv = (tid < 36 || tid >= 63) ? -v : v;
        *dst = ADD ? *dst + sv : sv;
    }
}

// Both in ONE launch (they only depend on k_lin): the first n_diag workgroups are the cameras', the rest the off-diagonal blocks'.
template <bool ADD>
__global__ __launch_bounds__(256) void k_rows_schur(const RowsArgs R, double *__restrict__ S, double *__restrict__ U, double *__restrict__ gc,
                                                    double *__restrict__ gred, const int n_diag) {
    if ((int)blockIdx.x < n_diag) rows_diag_block<ADD>(R, S, U, gc, gred, (int)blockIdx.x);
    else rows_cross_block<ADD>(R, S, (int)blockIdx.x - n_diag);
}

// k_lin for a HEAVY point (a track no chunk holds): one workgroup, the point's observations in strides of 256.  Pass 1 sums
// V, g_p and the cost (fixed order), every thread factors the damped block, pass 2 writes the observations' rows and tails;
// the point's pairs then go through k_rows_* like everybody else's (the first version of the heavy path added every pair's
// block to S with 36 atomics from one workgroup per point: 20 ms per landmark seen by 1 778 cameras).
template <int LOSS>
__global__ __launch_bounds__(256) void k_lin_heavy(const BuildArgs A, const HeavyArgs H, double *__restrict__ rows, double *__restrict__ tails,
                                                   long long row_off) {
    __shared__ double sRed[4];
    const int hp = blockIdx.x, tid = threadIdx.x;
    const long long ob = H.hv_obs[hp];
    const int L = (int)(H.hv_obs[hp + 1] - ob);
    const int p = H.pt0 + hp;
    const int loss = LOSS >= 0 ? LOSS : A.loss;
    const double lam = A.state->lambda, floor_now = A.state->floor;
    const double X0 = A.points[3 * (size_t)p], X1 = A.points[3 * (size_t)p + 1], X2 = A.points[3 * (size_t)p + 2];
    double vs[9], cost = 0.0;
#pragma unroll
    for (int k = 0; k < 9; ++k) vs[k] = 0.0;
    for (int o = tid; o < L; o += 256) {
        const int cam = A.o_cam[ob + o];
        const double2 xy = reinterpret_cast<const double2 *>(A.o_xy)[ob + o];
        CamAux ca;
        load_aux_local(A.aux, cam, ca);
        ObsLin Ln;
        linearize(ca, X0, X1, X2, xy.x, xy.y, loss, A.inv_fs2, floor_now, Ln);
        cost += A.half_fs2 * Ln.rho_sum;
        double wJp[2][3];
#pragma unroll
        for (int j = 0; j < 3; ++j) { wJp[0][j] = Ln.wh[0] * Ln.Jp[0][j]; wJp[1][j] = Ln.wh[1] * Ln.Jp[1][j]; }
        vs[0] += wJp[0][0] * Ln.Jp[0][0] + wJp[1][0] * Ln.Jp[1][0];
        vs[1] += wJp[0][0] * Ln.Jp[0][1] + wJp[1][0] * Ln.Jp[1][1];
        vs[2] += wJp[0][0] * Ln.Jp[0][2] + wJp[1][0] * Ln.Jp[1][2];
        vs[3] += wJp[0][1] * Ln.Jp[0][1] + wJp[1][1] * Ln.Jp[1][1];
        vs[4] += wJp[0][1] * Ln.Jp[0][2] + wJp[1][1] * Ln.Jp[1][2];
        vs[5] += wJp[0][2] * Ln.Jp[0][2] + wJp[1][2] * Ln.Jp[1][2];
        const double wr0 = Ln.w[0] * Ln.r[0], wr1 = Ln.w[1] * Ln.r[1];
        vs[6] += Ln.Jp[0][0] * wr0 + Ln.Jp[1][0] * wr1;
        vs[7] += Ln.Jp[0][1] * wr0 + Ln.Jp[1][1] * wr1;
        vs[8] += Ln.Jp[0][2] * wr0 + Ln.Jp[1][2] * wr1;
    }
#pragma unroll
    for (int k = 0; k < 9; ++k) vs[k] = block_sum256(vs[k], sRed);
    cost = block_sum256(cost, sRed);
    const double v00 = vs[0], v01 = vs[1], v02 = vs[2], v11 = vs[3], v12 = vs[4], v22 = vs[5];
    const double g0 = vs[6], g1 = vs[7], g2 = vs[8];
    const double a00 = v00 + lam * fmin(fmax(v00, BA_D_MIN), BA_D_MAX);
    const double a11 = v11 + lam * fmin(fmax(v11, BA_D_MIN), BA_D_MAX);
    const double a22 = v22 + lam * fmin(fmax(v22, BA_D_MIN), BA_D_MAX);
    const double i00 = rsqrt_nr(a00);
    const double l10 = v01 * i00, l20 = v02 * i00;
    const double i11 = rsqrt_nr(fmax(a11 - l10 * l10, 1e-15 * a11));
    const double l21 = (v12 - l20 * l10) * i11;
    const double i22 = rsqrt_nr(fmax(a22 - l20 * l20 - l21 * l21, 1e-15 * a22));
    const double i10 = -l10 * i00 * i11;
    const double i21 = -l21 * i11 * i22;
    const double i20 = -(l20 * i00 + l21 * i10) * i22;
    const double u0 = i00 * g0, u1 = i10 * g0 + i11 * g1, u2 = i20 * g0 + i21 * g1 + i22 * g2;
    if (tid == 0) {
        double *gl = &A.pt_linv[(size_t)p * 6];
        gl[0] = i00; gl[1] = i10; gl[2] = i11; gl[3] = i20; gl[4] = i21; gl[5] = i22;
        double *gu = &A.pt_u[(size_t)p * 3];
        gu[0] = u0; gu[1] = u1; gu[2] = u2;
        if (A.dbg_v) {
            double *dv = &A.dbg_v[(size_t)p * 6];
            dv[0] = v00; dv[1] = v01; dv[2] = v02; dv[3] = v11; dv[4] = v12; dv[5] = v22;
            double *dg = &A.dbg_gp[(size_t)p * 3];
            dg[0] = g0; dg[1] = g1; dg[2] = g2;
        }
        A.ck_cost[H.n_chunks + hp] = cost;
        A.ck_gmax[H.n_chunks + hp] = fmax(fabs(g0), fmax(fabs(g1), fabs(g2)));
    }
    for (int o = tid; o < L; o += 256) {
        const int cam = A.o_cam[ob + o];
        if (A.cam_free[cam] < 0) continue;
        const double2 xy = reinterpret_cast<const double2 *>(A.o_xy)[ob + o];
        CamAux ca;
        load_aux_local(A.aux, cam, ca);
        ObsLin Ln;
        linearize(ca, X0, X1, X2, xy.x, xy.y, loss, A.inv_fs2, floor_now, Ln);
        double t[9];
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const double x0 = (Ln.wh[0] * Ln.Jc[0][3 + i]) * Ln.Jp[0][0] + (Ln.wh[1] * Ln.Jc[1][3 + i]) * Ln.Jp[1][0];
            const double x1 = (Ln.wh[0] * Ln.Jc[0][3 + i]) * Ln.Jp[0][1] + (Ln.wh[1] * Ln.Jc[1][3 + i]) * Ln.Jp[1][1];
            const double x2 = (Ln.wh[0] * Ln.Jc[0][3 + i]) * Ln.Jp[0][2] + (Ln.wh[1] * Ln.Jc[1][3 + i]) * Ln.Jp[1][2];
            t[i * 3] = x0 * i00;
            t[i * 3 + 1] = x0 * i10 + x1 * i11;
            t[i * 3 + 2] = x0 * i20 + x1 * i21 + x2 * i22;
        }
        const double w0 = t[0] * u0 + t[1] * u1 + t[2] * u2, w1 = t[3] * u0 + t[4] * u1 + t[5] * u2,
                     w2 = t[6] * u0 + t[7] * u1 + t[8] * u2;
        const double wr0 = Ln.w[0] * Ln.r[0], wr1 = Ln.w[1] * Ln.r[1];
        double2 *d = reinterpret_cast<double2 *>(rows + (size_t)(ob + o - row_off) * ROW_A);
        d[0] = make_double2(t[0], t[1]); d[1] = make_double2(t[2], t[3]); d[2] = make_double2(t[4], t[5]);
        d[3] = make_double2(t[6], t[7]); d[4] = make_double2(t[8], Ln.Y[0]); d[5] = make_double2(Ln.Y[1], Ln.Y[2]);
        d[6] = make_double2(w0, w1); d[7] = make_double2(w2, 0.0);
        double2 *q = reinterpret_cast<double2 *>(tails + (size_t)(ob + o - row_off) * ROW_B);
        q[0] = make_double2(Ln.Jc[0][3], Ln.Jc[1][4]); q[1] = make_double2(Ln.Jc[0][5], Ln.Jc[1][5]);
        q[2] = make_double2(Ln.wh[0], Ln.wh[1]); q[3] = make_double2(wr0, wr1);
    }
}

}  // namespace ba
