// HIP kernels of the bundle-adjustment engine (gfx950 / CDNA4, wave64, fp64).
// Included once by ba_engine.hip.
//
//   k_cam_prep        per camera: rotation matrix + SO(3) left Jacobian            (tiny)
//   k_build           FUSED per chunk of points: residual, analytic Jacobian blocks, robust
//                     weights, per-point V / g_p in LDS, 3x3 Cholesky in registers, per-camera
//                     U / g_c and Schur-complement blocks accumulated in LDS tiles, one flush of
//                     the tiles to HBM per chunk.              HBM-bound: 40 B/obs + 96 B/point
//   k_finalize        per free camera: damping, diagonal block, block-Jacobi inverse (tiny)
//   k_pcg_*           block-Jacobi preconditioned CG on the block-sparse reduced camera system
//   k_cam_update      trial camera poses
//   k_backsub         FUSED per chunk: back-substitution of the point steps, model (predicted)
//                     reduction, trial points, trial cost
//   k_reduce_* / k_decide   deterministic reductions + the LM accept / damping logic on device
//   k_eval_*          building blocks in the caller's observation order (parity probes)
#pragma once
#include "ba_device.h"
#include "ba_symbolic.h"

namespace ba {

// ------------------------------------------------------------------ LM state kept on device
struct LmState {
    double lambda, nu;
    double F;         // cost at the current state
    double F_trial;   // cost at the trial state
    double pred;      // model reduction of the last step
    double rho;       // gain ratio
    double step_norm, x_norm;
    double gmax;      // ||g||_inf at the current state
    double dF;
    int accept, status, iter, n_accept;
    int terminate;
    int solve_info;   // pivot flag of the direct solver behind the last k_decide (!= 0: the step was void)
    int pad1, pad2;
    double floor;     // current curvature floor in [floor_min, 1]: 1 = IRLS (far from the minimum), see k_decide
    double floor_min;
};

struct BuildArgs {
    const double *aux;      // [n_cams * BA_CAM_AUX]
    const double *points;   // [n_points*3] internal order
    const int *o_cam;
    const int *o_pt;
    const short *o_lcam;
    const double *o_xy;
    const int *cam_free;
    const int *ck_obs, *ck_pt, *ck_cam, *ck_cam_list, *ck_pair, *ck_pair_slot;
    const long long *ck_pe;
    const int *ck_pgrp;             // per chunk (n_local_pairs + 1) group offsets, at ck_pair[ck] + ck
    const unsigned *pair_entries;
    double *S, *U, *gc, *gred;      // reduced-system accumulators (zeroed before the launch)
    double *pt_linv, *pt_u;         // [n_pts_obs*6], [n_pts_obs*3]
    double *ck_cost, *ck_gmax;      // per-chunk partials
    double *dbg_v, *dbg_gp;         // optional [n_pts_obs*6], [n_pts_obs*3]
    const LmState *state;           // lambda is read from here
    int loss;
    double inv_fs2, half_fs2;
    unsigned long long *stamps;     // diagnostic build only
};

// slabs of the deterministic build (ba_det.h; layout: DetTables in ba_symbolic.h); cam_slab == nullptr: atomic mode
struct DetArgs {
    double *cam_slab;             // rows of 33 doubles
    double *blk_slab;             // rows of 36 doubles
    const long long *ck_lg;       // [n_chunks + 1] later-round rows of each chunk, relative to n_raw
    long long n_cl, n_raw, n_lg;
};

__global__ void k_cam_prep(int n_cams, const double *__restrict__ pose, const double *__restrict__ intr,
                           double *__restrict__ aux) {
    int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c < n_cams) cam_aux_from_pose(pose + 6 * (size_t)c, intr + 4 * (size_t)c, aux + (size_t)c * BA_CAM_AUX);
}

// sum / max of one value per thread over a 256-thread block; result valid in thread 0
__device__ __forceinline__ double block_sum256(double v, double *scratch) {
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) scratch[w] = v;
    __syncthreads();
    return scratch[0] + scratch[1] + scratch[2] + scratch[3];
}
__device__ __forceinline__ double block_max256(double v, double *scratch) {
    v = wave_max(v);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) scratch[w] = v;
    __syncthreads();
    return fmax(fmax(scratch[0], scratch[1]), fmax(scratch[2], scratch[3]));
}

// STAMP = true is a diagnostic build only (tools/stamps.py): thread 0 records s_memtime at the phase
// boundaries into A.stamps (a buffer nothing else reads); never used by the product path.
#define BA_STAMP(i)                                                               \
    do {                                                                          \
        if (STAMP && tid == 0) A.stamps[(size_t)ck * 12 + (i)] = __builtin_amdgcn_s_memtime(); \
    } while (0)
// LDS row of an observation in k_build: T' (3x3, row-major) and Y = R X; one pad double makes the row stride 26 banks,
// so 32 lanes reading the same column of 32 different rows hit 32 different bank pairs (stride 24 would repeat after 8)
#ifndef BA_T_STRIDE
#define BA_T_STRIDE 13
#endif
constexpr int T_STRIDE = BA_T_STRIDE;
constexpr int U_REPLICAS = 4;
constexpr int U_REP_STRIDE = CHUNK_CAMS * 27 + 20;  // doubles; 776 = 16 banks (of 4 bytes) past a multiple of 64
constexpr int P_REP_BASE = U_REPLICAS * U_REP_STRIDE + 6;  // second replica of the per-point sums, behind the camera replicas
static_assert(P_REP_BASE + CHUNK_PTS * 9 <= CHUNK_OBS * 18, "the replicas live in the T buffer");
static_assert(CHUNK_CAMS_WIDE * 27 <= P_REP_BASE, "a wide chunk's single replicas end in front of the point replicas");
constexpr int STG_WAVE = 8 * 36 + 16;  // doubles per wave behind the T rows: eight staged Schur blocks + 32 block slots (phase 3)
static_assert(CHUNK_OBS * T_STRIDE + (CHUNK_OBS / 64) * STG_WAVE <= CHUNK_OBS * 18, "the staging area lives behind the T rows");
// LOSS: the robust loss as a compile-time constant (huber, the only kind the reference uses), or -1 for A.loss (the
// generic switch carries the polynomial constants of log1p / atan through the kernel's registers).
template <bool STAMP, int LOSS>
__global__ __launch_bounds__(CHUNK_OBS) __attribute__((amdgpu_waves_per_eu(3, 3))) void k_build(const BuildArgs A) {
    // phase 1: replicas of the camera / point sums; then per observation T' (3x3) and Y = R X at stride T_STRIDE -- the
    // 6x3 block T = W L^-T of the observation is [[Y]x T'; T'] in the local rotation tangent (load_aux_local), so 12
    // doubles stand for 18; after phase 3 the parked partial Schur blocks (36 doubles per lane pair)
    __shared__ __attribute__((aligned(16))) double sT[CHUNK_OBS * 18];
    __shared__ __attribute__((aligned(16))) double sP[CHUNK_PTS * 9];    // per point: V(6) g_p(3), then L^-1(6) u(3)
    __shared__ __attribute__((aligned(16))) double sC[CHUNK_CAMS * 33];  // per local camera: U(21) g_c(6) g_red part(6)
    __shared__ double sRed[8];
    __shared__ int sCam[CHUNK_CAMS_WIDE];                                // free-camera index of each local camera
    __shared__ unsigned char sLp[CHUNK_OBS];                             // local point of each observation
    __shared__ unsigned char sLc[CHUNK_OBS];                             // local camera of each observation
    __shared__ int sPg[CHUNK_OBS / 2 + 2], sSlot[CHUNK_OBS / 2 + 1];    // group offsets / S block slots of the first pairs

    const int ck = blockIdx.x, tid = threadIdx.x;
    const int ob = A.ck_obs[ck], nobs = A.ck_obs[ck + 1] - ob;
    const int pb = A.ck_pt[ck], npts = A.ck_pt[ck + 1] - pb;
    const int cb = A.ck_cam[ck], ncam = A.ck_cam[ck + 1] - cb;
    const int qb = A.ck_pair[ck], npair = A.ck_pair[ck + 1] - qb;
    const double lam = A.state->lambda;
    // A "wide" chunk has more cameras than full tiles (unstructured visibility: a new camera with nearly every
    // observation): ONE replica per camera at sT[slot * 27], and the cameras from slot CHUNK_CAMS on have no tile in sC --
    // their U, g_c leave for HBM right after phase 1 (consecutive lanes on consecutive elements), their part of the reduced
    // gradient goes the way of the cameras without any slot (phase 2b).
    const bool wide = ncam > CHUNK_CAMS;
    const int ntile = wide ? CHUNK_CAMS : ncam;

    BA_STAMP(0);
    // All global loads of the chunk are issued here, unconditionally (clamped indices: a branch around a load makes
    // the compiler drain every outstanding load at its use), oldest first in the order they are needed:
    //   1. the small per-chunk lists that are staged in LDS before the first barrier, and the Schur phase's pair
    //      entries of this lane pair (one group of PAIR_RUN words, first round; used four barriers later),
    //   2. this lane's observation (camera / point index, pixel),
    //   3. its camera block and point (dependent on 2; only phase 1, after the barrier, waits for them).
    const bool act = tid < nobs;
    const int oo = ob + (act ? tid : 0);
    const int st_cam = A.ck_cam_list[tid < ncam ? cb + tid : 0];
    const bool has_pg = tid <= npair && tid <= CHUNK_OBS / 2, has_slot = tid < npair && tid <= CHUNK_OBS / 2;
    const int st_pg = A.ck_pgrp[qb + ck + (has_pg ? tid : 0)];  // (pairs beyond CHUNK_OBS / 2 hold no first-round group)
    const int st_slot = A.ck_pair_slot[has_slot ? qb + tid : 0];
    const long long peb = A.ck_pe[ck];
    const int rounds = (int)((A.ck_pe[ck + 1] - peb) / ((CHUNK_OBS / 2) * PAIR_RUN));
    const int half = tid & 1;
    unsigned ents[PAIR_RUN / 2];  // lane `half` of lane pair g takes entries 4 half .. 4 half + 3 of group g
    {
        // (a chunk without pair entries reads the next chunk's first words, or the padding at the end of the array)
        const uint4 e0 = reinterpret_cast<const uint4 *>(A.pair_entries + peb)[tid];
        const bool any = rounds > 0;
        ents[0] = any ? e0.x : PAIR_NULL; ents[1] = any ? e0.y : PAIR_NULL; ents[2] = any ? e0.z : PAIR_NULL; ents[3] = any ? e0.w : PAIR_NULL;
    }
    const int cam = A.o_cam[oo];
    const int p = A.o_pt[oo];
    const int lc_ld = A.o_lcam[oo];
    const double2 xy = reinterpret_cast<const double2 *>(A.o_xy)[oo];
    CamAux ca;
    load_aux_local(A.aux, cam, ca);
    const double X0 = A.points[3 * (size_t)p], X1 = A.points[3 * (size_t)p + 1], X2 = A.points[3 * (size_t)p + 2];
    const double floor_now = A.state->floor;
    for (int i = tid; i < npts * 9; i += CHUNK_OBS) sP[i] = 0.0;
    for (int i = tid; i < ntile * 33; i += CHUNK_OBS) sC[i] = 0.0;
    // four replicas of the cameras' U / g_c tiles in the T buffer, which is idle until phase 2b (see phase 1)
    if (!wide) {
        for (int i = tid; i < ncam * 27; i += CHUNK_OBS) {
#pragma unroll
            for (int r = 0; r < U_REPLICAS; ++r) sT[r * U_REP_STRIDE + i] = 0.0;
        }
    } else {
        for (int i = tid; i < ncam * 27; i += CHUNK_OBS) sT[i] = 0.0;
    }
    for (int i = tid; i < npts * 9; i += CHUNK_OBS) sT[P_REP_BASE + i] = 0.0;  // second replica of the point sums
    if (tid < ncam) sCam[tid] = st_cam;
    if (has_pg) sPg[tid] = st_pg;
    if (has_slot) sSlot[tid] = st_slot;
    lds_barrier();

    BA_STAMP(1);
    // ---- phase 1: one observation per lane ------------------------------------------------
    double Wt[9];   // rows 3..5 of W = Jc'^T wh Jp (the translation rows: P^T wh Jp)
    double Yr[3];   // R X
    double vs[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) vs[k] = 0.0;
    int lp = 0, lc = -2, fcam = -1;
    double cost_loc = 0.0;
    if (act) {
        lp = p - pb;
        sLp[tid] = (unsigned char)lp;
        lc = lc_ld;
        sLc[tid] = (unsigned char)lc;
        ObsLin L;
        linearize(ca, X0, X1, X2, xy.x, xy.y, LOSS >= 0 ? LOSS : A.loss, A.inv_fs2, floor_now, L);
        cost_loc = A.half_fs2 * L.rho_sum;
        Yr[0] = L.Y[0]; Yr[1] = L.Y[1]; Yr[2] = L.Y[2];
        double wJp[2][3];
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
        if (lc != -2) {  // free camera
            double wJc[2][6];
#pragma unroll
            for (int i = 0; i < 6; ++i) { wJc[0][i] = L.wh[0] * L.Jc[0][i]; wJc[1][i] = L.wh[1] * L.Jc[1][i]; }
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int j = 0; j < 3; ++j) Wt[i * 3 + j] = wJc[0][3 + i] * L.Jp[0][j] + wJc[1][3 + i] * L.Jp[1][j];
            if (lc >= 0) {
                // per-camera U (21) and g_c (6): LDS atomics into the camera's tile.  (A gather through a camera-sorted
                // observation list, without atomics, won at two workgroups per CU; at three the VALU is the scarcer
                // unit and the plain atomics are faster.)
                // Same-address fp64 LDS atomics serialise (1.5 lane-ops per clock for this pattern against 8.6 for
                // distinct addresses, tools/lds_bench.hip): the lanes of one camera are spread over four bank-shifted
                // replicas of its tile by their point index (4.3 lane-ops per clock); summed after the barrier.
                double *cp = &sT[(wide ? 0 : (lp & (U_REPLICAS - 1)) * U_REP_STRIDE) + lc * 27];
                if (lc >= CHUNK_CAMS) fcam = sCam[lc];   // (phase 2b: its reduced gradient goes to HBM directly)
#pragma unroll
                for (int i = 0; i < 6; ++i) {
#pragma unroll
                    for (int j = i; j < 6; ++j) atomicAdd(&cp[ut6(i, j)], wJc[0][i] * L.Jc[0][j] + wJc[1][i] * L.Jc[1][j]);
                    atomicAdd(&cp[21 + i], L.Jc[0][i] * wr0 + L.Jc[1][i] * wr1);
                }
            } else {  // camera without a local slot in this chunk: straight to HBM
                fcam = A.cam_free[cam];
                double *up = &A.U[(size_t)fcam * 21], *gp = &A.gc[(size_t)fcam * 6];
#pragma unroll
                for (int i = 0; i < 6; ++i) {
#pragma unroll
                    for (int j = i; j < 6; ++j) atomicAdd(&up[ut6(i, j)], wJc[0][i] * L.Jc[0][j] + wJc[1][i] * L.Jc[1][j]);
                    atomicAdd(&gp[i], L.Jc[0][i] * wr0 + L.Jc[1][i] * wr1);  // gradient weights w, not the curvature weights
                }
            }
        }
    }
    BA_STAMP(2);
    // per-point sums V (6), g_p (3): LDS atomics (same reasoning; the segmented wave reduction they replace cost
    // more VALU and shuffle time than the atomics cost LDS time)
    if (act) {
        double *vp = (tid & 1) ? &sT[P_REP_BASE + lp * 9] : &sP[lp * 9];  // neighbouring lanes: alternate replicas
#pragma unroll
        for (int k = 0; k < 9; ++k) atomicAdd(&vp[k], vs[k]);
    }
    lds_barrier();

    BA_STAMP(3);
    // camera tiles = sum of their replicas (highest lanes first: the low lanes have a point each in phase 2a)
    if (!wide) {
        for (int item = CHUNK_OBS - 1 - tid; item < ncam * 27; item += CHUNK_OBS) {
            const int l = item / 27;
            double v = 0.0;
#pragma unroll
            for (int r = 0; r < U_REPLICAS; ++r) v += sT[r * U_REP_STRIDE + item];
            sC[l * 33 + (item - l * 27)] = v;
        }
    } else {
        for (int item = CHUNK_OBS - 1 - tid; item < ncam * 27; item += CHUNK_OBS) {
            const int l = item / 27, k = item - l * 27;
            const double v = sT[item];
            if (l < CHUNK_CAMS) {
                sC[l * 33 + k] = v;
            } else {
                const size_t f = (size_t)sCam[l];
                if (k < 21) atomicAdd(&A.U[f * 21 + k], v);
                else atomicAdd(&A.gc[f * 6 + (k - 21)], v);
            }
        }
    }
    BA_STAMP(4);
    // ---- phase 2a: one point per lane: damp, 3x3 Cholesky, inverse factor in registers ------
    double gmax_loc = 0.0;
    if (tid < npts) {
        double *vp = &sP[tid * 9];
        const double *rp = &sT[P_REP_BASE + tid * 9];
        // (single 8-byte reads, here and below: lds_f64, ba_device.h)
        const double v00 = lds_f64(vp) + lds_f64(rp), v01 = lds_f64(vp + 1) + lds_f64(rp + 1), v02 = lds_f64(vp + 2) + lds_f64(rp + 2),
                     v11 = lds_f64(vp + 3) + lds_f64(rp + 3), v12 = lds_f64(vp + 4) + lds_f64(rp + 4), v22 = lds_f64(vp + 5) + lds_f64(rp + 5);
        const double g0 = lds_f64(vp + 6) + lds_f64(rp + 6), g1 = lds_f64(vp + 7) + lds_f64(rp + 7), g2 = lds_f64(vp + 8) + lds_f64(rp + 8);
        gmax_loc = fmax(fabs(g0), fmax(fabs(g1), fabs(g2)));
        if (A.dbg_v) {
            double *dv = &A.dbg_v[(size_t)(pb + tid) * 6];
            dv[0] = v00; dv[1] = v01; dv[2] = v02; dv[3] = v11; dv[4] = v12; dv[5] = v22;
            double *dg = &A.dbg_gp[(size_t)(pb + tid) * 3];
            dg[0] = g0; dg[1] = g1; dg[2] = g2;
        }
        const double a00 = v00 + lam * fmin(fmax(v00, BA_D_MIN), BA_D_MAX);
        const double a11 = v11 + lam * fmin(fmax(v11, BA_D_MIN), BA_D_MAX);
        const double a22 = v22 + lam * fmin(fmax(v22, BA_D_MIN), BA_D_MAX);
        // reciprocal square roots (v_rsq_f64 + Newton) instead of sqrt and divide: the chain is serial per lane
        const double i00 = rsqrt_nr(a00);
        const double l10 = v01 * i00, l20 = v02 * i00;
        // (V + lambda D is positive definite; the floors only guard the rounding of rank-deficient V)
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
    BA_STAMP(5);
    {
        // chunk cost and max |g_p|: DPP wave reductions, the four wave results cross the barrier that 2b needs anyway
        const double cs = wave_sum_dpp(cost_loc), gm = wave_max_dpp(gmax_loc);
        if ((tid & 63) == 0) { sRed[tid >> 6] = cs; sRed[4 + (tid >> 6)] = gm; }
    }
    lds_barrier();
    if (tid == 0) {
        A.ck_cost[ck] = (sRed[0] + sRed[1]) + (sRed[2] + sRed[3]);
        A.ck_gmax[ck] = fmax(fmax(sRed[4], sRed[5]), fmax(sRed[6], sRed[7]));
    }

    BA_STAMP(6);
    // ---- phase 2b: T' = Wt L^-T and Y into LDS (overwrites the phase-1 staging) ---------------------------
    if (act && lc != -2) {
        const double *vp = &sP[lp * 9];
        const double i00 = lds_f64(vp), i10 = lds_f64(vp + 1), i11 = lds_f64(vp + 2), i20 = lds_f64(vp + 3), i21 = lds_f64(vp + 4), i22 = lds_f64(vp + 5);
        double *tp = &sT[tid * T_STRIDE];
        double t[9];
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            t[i * 3] = Wt[i * 3] * i00;
            t[i * 3 + 1] = Wt[i * 3] * i10 + Wt[i * 3 + 1] * i11;
            t[i * 3 + 2] = Wt[i * 3] * i20 + Wt[i * 3 + 1] * i21 + Wt[i * 3 + 2] * i22;
        }
#pragma unroll
        for (int k = 0; k < 9; ++k) tp[k] = t[k];
        tp[9] = Yr[0]; tp[10] = Yr[1]; tp[11] = Yr[2];
        if (lc == -1 || lc >= CHUNK_CAMS) {  // camera without a tile: its part of the reduced gradient, -T u = -[Y x w; w], w = T' u
            const double u0 = lds_f64(vp + 6), u1 = lds_f64(vp + 7), u2 = lds_f64(vp + 8);
            const double w0 = t[0] * u0 + t[1] * u1 + t[2] * u2, w1 = t[3] * u0 + t[4] * u1 + t[5] * u2,
                         w2 = t[6] * u0 + t[7] * u1 + t[8] * u2;
            double *d = &A.gred[(size_t)fcam * 6];
            atomicAdd(&d[0], -(Yr[1] * w2 - Yr[2] * w1));
            atomicAdd(&d[1], -(Yr[2] * w0 - Yr[0] * w2));
            atomicAdd(&d[2], -(Yr[0] * w1 - Yr[1] * w0));
            atomicAdd(&d[3], -w0); atomicAdd(&d[4], -w1); atomicAdd(&d[5], -w2);
        }
    }
    lds_barrier();
    BA_STAMP(7);
    BA_STAMP(8);

    // ---- phase 3: Schur-complement blocks and reduced gradient ----------------------------------
    // The chunk's (observation i, observation j) entries come in groups of PAIR_RUN entries of ONE camera pair
    // (ba_symbolic.cpp).  Lane pair g reduces group g: each lane forms the whole 6x6 products sum T_i T_j^T of FOUR
    // entries in registers.  With T = [[Y]x T'; T'] the bottom rows of T_i T_j^T are B = [N [Yj]x^T | N], N = T'_i T'_j^T
    // (row r of N [Yj]x^T is Yj x n_r), and the top rows are [Yi]x B, streamed row by row so that only one row of B is
    // live: 27 doubles read from LDS and 99 FMA-class instructions per entry, where the full 6x3 rows took 39 and 117.
    // Then the two lanes are summed by DPP and each parks one half of the block in sT once every lane is done with T
    // (exactly 36 doubles per lane pair); phase 4 sums the groups of each pair.  No LDS atomics.  Self entries (i == j)
    // also carry -T_i u_p of the reduced gradient (camera tile, LDS atomics: few).  Groups beyond the first
    // CHUNK_OBS / 2 of a chunk (rare: long tracks) are done first and go to HBM straight from the registers.
    {
        double acc[36];  // + sum T_i T_j^T (the sign is applied when the block leaves the registers)
        for (int round = (rounds > 0 ? rounds : 1) - 1; round >= 0; --round) {
            unsigned en[PAIR_RUN / 2];
            if (round > 0) {
                const uint4 e0 = reinterpret_cast<const uint4 *>(A.pair_entries + peb)[(size_t)round * CHUNK_OBS + tid];
                en[0] = e0.x; en[1] = e0.y; en[2] = e0.z; en[3] = e0.w;
            } else {
#pragma unroll
                for (int e = 0; e < PAIR_RUN / 2; ++e) en[e] = ents[e];
            }
            double gacc[6];
            int self_lc = -1;
#pragma unroll
            for (int k = 0; k < 36; ++k) acc[k] = 0.0;
#pragma unroll
            for (int k = 0; k < 6; ++k) gacc[k] = 0.0;
#pragma unroll
            for (int e = 0; e < PAIR_RUN / 2; ++e) {
                const unsigned ent = en[e];
                const unsigned ec = ent == PAIR_NULL ? 0u : ent;
                const int oi = ec & 255u, oj = (ec >> 8) & 255u;
                const double *ti = &sT[oi * T_STRIDE], *tj = &sT[oj * T_STRIDE], *uu = &sP[(int)sLp[oi] * 9 + 6];
                const bool real = ent != PAIR_NULL, self = real && oi == oj;
                if (self) self_lc = sLc[oi];
                double Tj[9], Yj[3], Yi[3], tu[3];
#pragma unroll
                for (int k = 0; k < 9; ++k) Tj[k] = lds_f64(tj + k);
#pragma unroll
                for (int k = 0; k < 3; ++k) { Yj[k] = lds_f64(tj + 9 + k); Yi[k] = lds_f64(ti + 9 + k); tu[k] = lds_f64(uu + k); }
#pragma unroll
                for (int r = 0; r < 3; ++r) {
                    const double a0 = lds_f64(ti + 3 * r), a1 = lds_f64(ti + 3 * r + 1), a2 = lds_f64(ti + 3 * r + 2);
                    if (real) {
                        double B[6];
                        B[3] = a0 * Tj[0] + a1 * Tj[1] + a2 * Tj[2];
                        B[4] = a0 * Tj[3] + a1 * Tj[4] + a2 * Tj[5];
                        B[5] = a0 * Tj[6] + a1 * Tj[7] + a2 * Tj[8];
                        B[0] = Yj[1] * B[5] - Yj[2] * B[4];
                        B[1] = Yj[2] * B[3] - Yj[0] * B[5];
                        B[2] = Yj[0] * B[4] - Yj[1] * B[3];
                        const int s1 = (r + 1) % 3, s2 = (r + 2) % 3;   // (Yi x b)_{r+1} gets +Yi_{r+2} b_r, (Yi x b)_{r+2} gets -Yi_{r+1} b_r
#pragma unroll
                        for (int c = 0; c < 6; ++c) {
                            acc[(3 + r) * 6 + c] += B[c];
                            acc[s1 * 6 + c] += Yi[s2] * B[c];
                            acc[s2 * 6 + c] -= Yi[s1] * B[c];
                        }
                        if (self) {
                            const double w = a0 * tu[0] + a1 * tu[1] + a2 * tu[2];
                            gacc[3 + r] += w;
                            gacc[s1] += Yi[s2] * w;
                            gacc[s2] -= Yi[s1] * w;
                        }
                    }
                }
            }
            if (self_lc >= 0 && self_lc < CHUNK_CAMS) {   // (both lanes add their own four entries)
                double *d = &sC[self_lc * 33 + 27];
#pragma unroll
                for (int k = 0; k < 6; ++k) atomicAdd(&d[k], -gacc[k]);
            }
            // sum of the two lanes; each lane keeps one half of the block
#pragma unroll
            for (int k = 0; k < 36; ++k) acc[k] += dpp_f64<0xB1>(acc[k]);
            if (round > 0) {
                // Straight to HBM, but not from the registers: a lane's atomics would go to 64 different blocks per wave
                // instruction, 64 memory-side requests of 8 bytes each (unstructured visibility -- hardly two points of a
                // chunk share a camera pair -- spends its time there).  Each wave turns its 32 blocks through a staging
                // area behind the T rows, eight blocks at a time, and adds them with consecutive lanes on consecutive
                // elements: one request per 64-byte line.
                double *stg = &sT[CHUNK_OBS * T_STRIDE + (tid >> 6) * STG_WAVE];
                int *stg_slot = reinterpret_cast<int *>(stg + 8 * 36);
                const int lane = tid & 63, pl = lane >> 1;
                // (the group's first entry sits in the EVEN lane; the odd lane has none when the group holds at most four)
                if (half == 0) stg_slot[pl] = en[0] != PAIR_NULL ? A.ck_pair_slot[qb + (int)(en[0] >> 16)] : -1;
#pragma unroll 1
                for (int b = 0; b < 4; ++b) {
                    if ((pl >> 3) == b) {
                        double *d = &stg[(pl & 7) * 36];
                        if (half == 0) {
#pragma unroll
                            for (int k = 0; k < 18; ++k) d[k] = -acc[k];
                        } else {
#pragma unroll
                            for (int k = 18; k < 36; ++k) d[k] = -acc[k];
                        }
                    }
                    wave_lds_sync();
#pragma unroll 1
                    for (int i = 0; i < 5; ++i) {
                        const int e = lane + 64 * i;
                        if (e < 8 * 36) {
                            const int blk = e / 36, k = e - blk * 36;
                            const int slot = stg_slot[8 * b + blk];
                            if (slot >= 0) atomicAdd(&A.S[(size_t)slot * 36 + k], stg[e]);
                        }
                    }
                    wave_lds_sync();
                }
            }
        }
        lds_barrier();  // every lane is done reading T
        // (the lane pairs take the groups window by window; the flush wants the groups of a pair side by side: ck_park)
        double *park = &sT[(tid >> 1) * 36];
        if (half == 0) {
#pragma unroll
            for (int k = 0; k < 18; ++k) park[k] = -acc[k];
        } else {
#pragma unroll
            for (int k = 18; k < 36; ++k) park[k] = -acc[k];
        }
    }
    lds_barrier();

    BA_STAMP(9);
    // ---- phase 4: one flush per chunk (fp64 atomics to HBM) ------------------------------------------
    // Element i = tid + 256 t of the flat lists (camera tile l = i / 33, k = i % 33; pair l = i / 36, k = i % 36): l and k are
    // stepped, not divided, and TWO elements are in flight per trip -- their index words, then their partial blocks, are
    // requested together (one element per trip with a division was a chain of three dependent LDS round trips and some fifty
    // instructions, fifteen times per lane: a sixth of the kernel).
    {
        const int n_items = ntile * 33;
        int l = tid / 33, k = tid - l * 33;   // 256 = 7 * 33 + 25
        for (int i = tid; i < n_items; i += CHUNK_OBS) {
            const size_t f = (size_t)sCam[l];
            const double v = sC[i];
            double *dst = k < 21 ? &A.U[f * 21 + k] : (k < 27 ? &A.gc[f * 6 + (k - 21)] : &A.gred[f * 6 + (k - 27)]);
            atomicAdd(dst, v);
            l += 7; k += 25;
            if (k >= 33) { k -= 33; l += 1; }
        }
    }
    {
        // Schur blocks: element k of pair l = sum over the pair's groups of the parked partial blocks.  Only pairs
        // with a group in the first round are parked; their offsets and block slots were staged in phase 0.
        const int np1 = npair < CHUNK_OBS / 2 ? npair : CHUNK_OBS / 2;
        const int n_items = np1 * 36;
        int l = tid / 36, k = tid - l * 36;   // 256 = 7 * 36 + 4; the second element of a trip is 256 further
        for (int i = tid; i < n_items; i += 2 * CHUNK_OBS) {
            int l2 = l + 7, k2 = k + 4;
            if (k2 >= 36) { k2 -= 36; l2 += 1; }
            const bool on2 = i + CHUNK_OBS < n_items;
            if (!on2) { l2 = l; k2 = k; }
            const int a0 = sPg[l], b0 = sPg[l2], slot_a = sSlot[l], slot_b = sSlot[l2];
            int a1 = sPg[l + 1], b1 = sPg[l2 + 1];
            if (a1 > CHUNK_OBS / 2) a1 = CHUNK_OBS / 2;  // later rounds went to HBM directly
            if (b1 > CHUNK_OBS / 2) b1 = CHUNK_OBS / 2;
            // the first two groups of each element unconditionally (a pair seldom has more), the rest in a loop
            // (a pair whose groups all lie in later rounds has a0 >= a1: nothing is added, the reads stay inside the buffer)
            const int a0c = a0 < a1 ? a0 : 0, b0c = b0 < b1 ? b0 : 0;
            const double xa0 = sT[a0c * 36 + k], xa1 = sT[(a0 + 1 < a1 ? a0 + 1 : a0c) * 36 + k];
            const double xb0 = sT[b0c * 36 + k2], xb1 = sT[(b0 + 1 < b1 ? b0 + 1 : b0c) * 36 + k2];
            double va = xa0 + (a0 + 1 < a1 ? xa1 : 0.0), vb = xb0 + (b0 + 1 < b1 ? xb1 : 0.0);
            for (int g = a0 + 2; g < a1; ++g) va += sT[g * 36 + k];
            for (int g = b0 + 2; g < b1; ++g) vb += sT[g * 36 + k2];
            if (a1 > a0) atomicAdd(&A.S[(size_t)slot_a * 36 + k], va);
            if (on2 && b1 > b0) atomicAdd(&A.S[(size_t)slot_b * 36 + k2], vb);
            l += 14; k += 8;   // 512 = 14 * 36 + 8
            if (k >= 36) { k -= 36; l += 1; }
        }
    }
    BA_STAMP(10);
}

// ------------------------------------------------------------------ finalize the reduced system
// 6x6 SPD inverse by Cholesky in registers; returns false if not positive definite.
__device__ __forceinline__ bool spd6_inverse(const double *__restrict__ Ain, double *__restrict__ out) {
    double L[6][6];
    bool ok = true;
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        double d = Ain[j * 6 + j];
#pragma unroll
        for (int k = 0; k < j; ++k) d -= L[j][k] * L[j][k];
        if (!(d > 0.0)) ok = false;
        const double ljj = sqrt(d);
        L[j][j] = ljj;
        const double inv = 1.0 / ljj;
#pragma unroll
        for (int i = j + 1; i < 6; ++i) {
            double s = Ain[i * 6 + j];
#pragma unroll
            for (int k = 0; k < j; ++k) s -= L[i][k] * L[j][k];
            L[i][j] = s * inv;
        }
    }
    if (!ok) return false;
    double Li[6][6];  // inverse of L (lower)
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        Li[j][j] = 1.0 / L[j][j];
#pragma unroll
        for (int i = j + 1; i < 6; ++i) {
            double s = 0.0;
#pragma unroll
            for (int k = j; k < i; ++k) s -= L[i][k] * Li[k][j];
            Li[i][j] = s / L[i][i];
        }
    }
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
        for (int j = 0; j <= i; ++j) {
            double s = 0.0;
#pragma unroll
            for (int k = i; k < 6; ++k) s += Li[k][i] * Li[k][j];
            out[i * 6 + j] = s;
            out[j * 6 + i] = s;
        }
    return true;
}

// The build kernels linearise in the local rotation tangent (load_aux_local).  This brings the accumulated reduced
// system to the rvec parameters the solver steps in, once per build pass: with D = diag(Jl, I) per camera,
// U = D^T U' D, g = D^T g', S_ij = D_i^T S'_ij D_j.  One thread per upper block, then one per free camera.
__device__ __forceinline__ void apply_jl_camera(int f, const int *__restrict__ free_cam, const double *__restrict__ aux,
                                                double *__restrict__ U, double *__restrict__ gc, double *__restrict__ gred) {
    const double *J = aux + (size_t)free_cam[f] * BA_CAM_AUX + 9;
    double jl[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) jl[k] = J[k];
    double *u = &U[(size_t)f * 21];
    double B[36];
#pragma unroll
    for (int r = 0; r < 6; ++r)
#pragma unroll
        for (int c = r; c < 6; ++c) B[r * 6 + c] = B[c * 6 + r] = u[ut6(r, c)];
    double T3[18];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 6; ++c) T3[r * 6 + c] = jl[r] * B[c] + jl[3 + r] * B[6 + c] + jl[6 + r] * B[12 + c];
#pragma unroll
    for (int k = 0; k < 18; ++k) B[k] = T3[k];
#pragma unroll
    for (int r = 0; r < 6; ++r) {
        const double b0 = B[r * 6], b1 = B[r * 6 + 1], b2 = B[r * 6 + 2];
#pragma unroll
        for (int c = 0; c < 3; ++c) B[r * 6 + c] = b0 * jl[c] + b1 * jl[3 + c] + b2 * jl[6 + c];
    }
#pragma unroll
    for (int r = 0; r < 6; ++r)
#pragma unroll
        for (int c = r; c < 6; ++c) u[ut6(r, c)] = (r < 3 && c < 3 && r != c) ? 0.5 * (B[r * 6 + c] + B[c * 6 + r]) : B[r * 6 + c];
    double *g1 = &gc[(size_t)f * 6], *g2 = &gred[(size_t)f * 6];
    const double a0 = g1[0], a1 = g1[1], a2 = g1[2], b0 = g2[0], b1 = g2[1], b2 = g2[2];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        g1[r] = jl[r] * a0 + jl[3 + r] * a1 + jl[6 + r] * a2;
        g2[r] = jl[r] * b0 + jl[3 + r] * b1 + jl[6 + r] * b2;
    }
}
// S_ij <- Jl_i^T S_ij Jl_j (rotation parts) for the 64 blocks of a wave at once (t0 = the wave's first block, a multiple of 64).  A lane's block is 288
// contiguous bytes, so the wave's blocks are ONE run of 18 KB: the wave copies the run into LDS with consecutive lanes on
// consecutive 16-byte pieces (one request per 64-byte line where 36 strided 8-byte loads per lane asked for every line nine times
// over), each lane transforms its block there (row stride 37 doubles: odd, no bank conflicts), and the run goes back the same way.
// SKIP_DIAG: diagonal blocks are left alone (neither loaded nor stored: another thread owns them).
constexpr int FIN_ROW = 37;
template <bool SKIP_DIAG>
__device__ __forceinline__ void apply_jl_run(const int t, const int n_upper, const long long *__restrict__ keys, const int *__restrict__ free_cam,
                                             const double *__restrict__ aux, double *__restrict__ S, double *__restrict__ sBw,
                                             unsigned char *__restrict__ sSkipw) {
    const int lane = threadIdx.x & 63, t0 = t - lane;
    const bool isblk = t < n_upper;
    const long long key = keys[isblk ? t : t0];
    const int i = (int)(key >> 32), j = (int)(key & 0xffffffff);
    const bool mine = isblk && !(SKIP_DIAG && i == j);
    sSkipw[lane] = mine ? 0 : 1;
    // (the two cameras' left Jacobians: requested before the run is copied)
    const double *Ji = aux + (size_t)free_cam[i] * BA_CAM_AUX + 9, *Jj = aux + (size_t)free_cam[j] * BA_CAM_AUX + 9;
    double ji[9], jj[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) { ji[k] = Ji[k]; jj[k] = Jj[k]; }
    wave_lds_sync();
    double *run = S + (size_t)t0 * 36;
    double2 piece[18];
#pragma unroll
    for (int k = 0; k < 18; ++k) {
        const int idx = lane + 64 * k, b = idx / 18;
        piece[k] = sSkipw[b] ? make_double2(0.0, 0.0) : reinterpret_cast<const double2 *>(run)[idx];
    }
#pragma unroll
    for (int k = 0; k < 18; ++k) {
        const int idx = lane + 64 * k, b = idx / 18, c = idx - b * 18;
        sBw[b * FIN_ROW + 2 * c] = piece[k].x;
        sBw[b * FIN_ROW + 2 * c + 1] = piece[k].y;
    }
    wave_lds_sync();
    if (mine) {
        double *blk = &sBw[lane * FIN_ROW];
        double B[36];
#pragma unroll
        for (int k = 0; k < 36; ++k) B[k] = lds_f64(blk + k);
        double T3[18];  // Ji^T * (top three rows)
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int c = 0; c < 6; ++c) T3[r * 6 + c] = ji[r] * B[c] + ji[3 + r] * B[6 + c] + ji[6 + r] * B[12 + c];
#pragma unroll
        for (int k = 0; k < 18; ++k) B[k] = T3[k];
#pragma unroll
        for (int r = 0; r < 6; ++r) {  // (left three columns) * Jj
            const double b0 = B[r * 6], b1 = B[r * 6 + 1], b2 = B[r * 6 + 2];
#pragma unroll
            for (int c = 0; c < 3; ++c) B[r * 6 + c] = b0 * jj[c] + b1 * jj[3 + c] + b2 * jj[6 + c];
        }
#pragma unroll
        for (int k = 0; k < 36; ++k) blk[k] = B[k];
    }
    wave_lds_sync();
#pragma unroll
    for (int k = 0; k < 18; ++k) {
        const int idx = lane + 64 * k, b = idx / 18, c = idx - b * 18;
        if (!sSkipw[b]) reinterpret_cast<double2 *>(run)[idx] = make_double2(sBw[b * FIN_ROW + 2 * c], sBw[b * FIN_ROW + 2 * c + 1]);
    }
}
__global__ __launch_bounds__(128) void k_apply_jl(int nf, int n_upper, const long long *__restrict__ keys, const int *__restrict__ free_cam,
                           const double *__restrict__ aux, double *__restrict__ S, double *__restrict__ U,
                           double *__restrict__ gc, double *__restrict__ gred) {
    __shared__ __attribute__((aligned(16))) double sB[2][64 * FIN_ROW];
    __shared__ unsigned char sSkip[2][64];
    const int t = blockIdx.x * blockDim.x + threadIdx.x, w = threadIdx.x >> 6;
    if (t - (int)(threadIdx.x & 63) < n_upper) apply_jl_run<false>(t, n_upper, keys, free_cam, aux, S, sB[w], sSkip[w]);
    if (t >= n_upper && t < n_upper + nf) apply_jl_camera(t - n_upper, free_cam, aux, U, gc, gred);
}

__device__ __forceinline__ void finalize_camera(int f, const double *__restrict__ U, const double *__restrict__ gc, double *__restrict__ gred,
                           double *__restrict__ S, const int *__restrict__ diag_slot, double *__restrict__ Minv,
                           double *__restrict__ cam_gmax, const LmState *__restrict__ st) {
    const double lam = st->lambda;
    double *blk = &S[(size_t)diag_slot[f] * 36];
    const double *u = &U[(size_t)f * 21];
    double A[36];
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
        for (int j = i; j < 6; ++j) {
            double v = u[ut6(i, j)];
            if (i == j) v += lam * fmin(fmax(v, BA_D_MIN), BA_D_MAX);
            const double a = blk[i * 6 + j] + v;
            const double bsym = (i == j) ? a : blk[j * 6 + i] + v;
            // the Schur part is symmetric up to the rounding of the atomics; keep both halves as accumulated
            A[i * 6 + j] = a;
            A[j * 6 + i] = bsym;
        }
#pragma unroll
    for (int k = 0; k < 36; ++k) blk[k] = A[k];
    double Ms[36];
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
        for (int j = 0; j < 6; ++j) Ms[i * 6 + j] = 0.5 * (A[i * 6 + j] + A[j * 6 + i]);
    double Mi[36];
    if (!spd6_inverse(Ms, Mi)) {
#pragma unroll
        for (int k = 0; k < 36; ++k) Mi[k] = 0.0;
#pragma unroll
        for (int i = 0; i < 6; ++i) Mi[i * 6 + i] = 1.0 / fmax(fabs(Ms[i * 6 + i]), 1e-300);
    }
#pragma unroll
    for (int k = 0; k < 36; ++k) Minv[(size_t)f * 36 + k] = Mi[k];
    // g_red = g_c - sum_i T_i u_p: the build kernel accumulated the second term only
    double gm = 0.0;
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        const double g = gc[(size_t)f * 6 + i];
        gred[(size_t)f * 6 + i] += g;
        gm = fmax(gm, fabs(g));
    }
    cam_gmax[f] = gm;
}

__global__ void k_finalize(int nf, const double *__restrict__ U, const double *__restrict__ gc, double *__restrict__ gred,
                           double *__restrict__ S, const int *__restrict__ diag_slot, double *__restrict__ Minv,
                           double *__restrict__ cam_gmax, const LmState *__restrict__ st) {
    const int f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f < nf) finalize_camera(f, U, gc, gred, S, diag_slot, Minv, cam_gmax, st);
}
// the block transform of apply_jl_run (diagonal block) + apply_jl_camera + finalize_camera of ONE camera with everything kept in registers
// between the three: every operand is requested once, up front, and every result stored once -- the three functions one
// after the other wait for their own stores to come back from memory three times (the camera threads' chain of
// dependent round trips set the time of k_apply_finalize: 18 us).  Same arithmetic, same order.
__device__ __forceinline__ void apply_finalize_camera(int f, const int *__restrict__ free_cam, const double *__restrict__ aux,
                                                      double *__restrict__ S, double *__restrict__ U, double *__restrict__ gc,
                                                      double *__restrict__ gred, const int *__restrict__ diag_slot,
                                                      double *__restrict__ cam_gmax, const LmState *__restrict__ st) {
    const int slot = diag_slot[f], cam = free_cam[f];
    const double lam = st->lambda;
    double *blk = &S[(size_t)slot * 36], *u = &U[(size_t)f * 21], *g1 = &gc[(size_t)f * 6], *g2 = &gred[(size_t)f * 6];
    const double *J = aux + (size_t)cam * BA_CAM_AUX + 9;
    double B[36], ub[21], ga[6], gb[6], jl[9];
#pragma unroll
    for (int k = 0; k < 36; ++k) B[k] = blk[k];
#pragma unroll
    for (int k = 0; k < 21; ++k) ub[k] = u[k];
#pragma unroll
    for (int k = 0; k < 6; ++k) { ga[k] = g1[k]; gb[k] = g2[k]; }
#pragma unroll
    for (int k = 0; k < 9; ++k) jl[k] = J[k];
    double T3[18];
    // --- the diagonal block of S: D^T B D
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 6; ++c) T3[r * 6 + c] = jl[r] * B[c] + jl[3 + r] * B[6 + c] + jl[6 + r] * B[12 + c];
#pragma unroll
    for (int k = 0; k < 18; ++k) B[k] = T3[k];
#pragma unroll
    for (int r = 0; r < 6; ++r) {
        const double b0 = B[r * 6], b1 = B[r * 6 + 1], b2 = B[r * 6 + 2];
#pragma unroll
        for (int c = 0; c < 3; ++c) B[r * 6 + c] = b0 * jl[c] + b1 * jl[3 + c] + b2 * jl[6 + c];
    }
    // --- U = D^T U' D, g = D^T g'
    double Ub[36];
#pragma unroll
    for (int r = 0; r < 6; ++r)
#pragma unroll
        for (int c = r; c < 6; ++c) Ub[r * 6 + c] = Ub[c * 6 + r] = ub[ut6(r, c)];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 6; ++c) T3[r * 6 + c] = jl[r] * Ub[c] + jl[3 + r] * Ub[6 + c] + jl[6 + r] * Ub[12 + c];
#pragma unroll
    for (int k = 0; k < 18; ++k) Ub[k] = T3[k];
#pragma unroll
    for (int r = 0; r < 6; ++r) {
        const double b0 = Ub[r * 6], b1 = Ub[r * 6 + 1], b2 = Ub[r * 6 + 2];
#pragma unroll
        for (int c = 0; c < 3; ++c) Ub[r * 6 + c] = b0 * jl[c] + b1 * jl[3 + c] + b2 * jl[6 + c];
    }
#pragma unroll
    for (int r = 0; r < 6; ++r)
#pragma unroll
        for (int c = r; c < 6; ++c) {
            ub[ut6(r, c)] = (r < 3 && c < 3 && r != c) ? 0.5 * (Ub[r * 6 + c] + Ub[c * 6 + r]) : Ub[r * 6 + c];
            u[ut6(r, c)] = ub[ut6(r, c)];
        }
    {
        const double a0 = ga[0], a1 = ga[1], a2 = ga[2], b0 = gb[0], b1 = gb[1], b2 = gb[2];
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            ga[r] = jl[r] * a0 + jl[3 + r] * a1 + jl[6 + r] * a2;
            gb[r] = jl[r] * b0 + jl[3 + r] * b1 + jl[6 + r] * b2;
        }
    }
    // --- finalize: damped U onto the diagonal block, block-Jacobi inverse, reduced gradient
    double A[36];
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
        for (int j = i; j < 6; ++j) {
            double v = ub[ut6(i, j)];
            if (i == j) v += lam * fmin(fmax(v, BA_D_MIN), BA_D_MAX);
            const double a = B[i * 6 + j] + v;
            const double bsym = (i == j) ? a : B[j * 6 + i] + v;
            A[i * 6 + j] = a;
            A[j * 6 + i] = bsym;
        }
#pragma unroll
    for (int k = 0; k < 36; ++k) blk[k] = A[k];
    // (the block-Jacobi inverse is the PCG's business: k_block_jacobi computes it from the stored block when a PCG
    // solve is about to run -- with a direct solver nobody reads it)
    double gm = 0.0;
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        g1[i] = ga[i];
        g2[i] = gb[i] + ga[i];
        gm = fmax(gm, fabs(ga[i]));
    }
    cam_gmax[f] = gm;
}

// k_apply_jl and k_finalize in one launch (single-GPU build pass: nothing has to happen between them).  A camera's
// thread transforms its own diagonal block (the block threads leave the diagonal keys alone) and finalises it.
// One thread per upper block of S (apply_jl_run; the diagonal blocks belong to the camera threads), then one per free camera.
__global__ __launch_bounds__(128) void k_apply_finalize(int nf, int n_upper, const long long *__restrict__ keys, const int *__restrict__ free_cam,
                                 const double *__restrict__ aux, double *__restrict__ S, double *__restrict__ U,
                                 double *__restrict__ gc, double *__restrict__ gred, const int *__restrict__ diag_slot,
                                 double *__restrict__ cam_gmax, const LmState *__restrict__ st) {
    __shared__ __attribute__((aligned(16))) double sB[2][64 * FIN_ROW];
    __shared__ unsigned char sSkip[2][64];
    const int t = blockIdx.x * blockDim.x + threadIdx.x, w = threadIdx.x >> 6;
    if (t - (int)(threadIdx.x & 63) < n_upper) apply_jl_run<true>(t, n_upper, keys, free_cam, aux, S, sB[w], sSkip[w]);
    if (t >= n_upper && t < n_upper + nf) apply_finalize_camera(t - n_upper, free_cam, aux, S, U, gc, gred, diag_slot, cam_gmax, st);
}

// block-Jacobi preconditioner of the PCG: inverse of the (symmetrised) finalised diagonal block of every free camera,
// the same arithmetic as in finalize_camera.  Launched by do_pcg when the single-GPU build pass left it out.
__global__ void k_block_jacobi(int nf, const double *__restrict__ S, const int *__restrict__ diag_slot, double *__restrict__ Minv) {
    const int f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= nf) return;
    const double *blk = &S[(size_t)diag_slot[f] * 36];
    double A[36], Ms[36], Mi[36];
#pragma unroll
    for (int k = 0; k < 36; ++k) A[k] = blk[k];
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
        for (int j = 0; j < 6; ++j) Ms[i * 6 + j] = 0.5 * (A[i * 6 + j] + A[j * 6 + i]);
    if (!spd6_inverse(Ms, Mi)) {
#pragma unroll
        for (int k = 0; k < 36; ++k) Mi[k] = 0.0;
#pragma unroll
        for (int i = 0; i < 6; ++i) Mi[i * 6 + i] = 1.0 / fmax(fabs(Ms[i * 6 + i]), 1e-300);
    }
#pragma unroll
    for (int k = 0; k < 36; ++k) Minv[(size_t)f * 36 + k] = Mi[k];
}

// ------------------------------------------------------------------ PCG on S x = -g_red
// Block-Jacobi preconditioned CG on the block-sparse reduced camera system, two launches per
// iteration, every scalar (alpha, beta, convergence) on the device:
//   k_pcg_a(it): p = z + beta p_old (recomputed on the fly for the gathered columns, p is
//                double-buffered), q = S p with one wave per block row, pq = p.q
//   k_pcg_b(it): x += alpha p, r -= alpha q, z = M^-1 r, rz = r.z, rn = r.r   (one lane per camera)
// The dot products are NOT accumulated by atomics (whose order would vary from run to run): every workgroup stores
// its partial sum in a slot of its own, and every workgroup of the NEXT launch adds the partials up itself, in the same
// fixed order -- all workgroups see the same bits, and so does every run.  rz / rn rotate through three sets of
// partials and pq through two so that a launch never overwrites a set that another workgroup of the same launch reads.
struct PcgScalars {
    double done, iters, pad0, pad1;
};
struct PcgPartials {
    double *rz[3], *rn[3], *bb;   // [nb_cam] each: one slot per workgroup of k_pcg_init / k_pcg_b
    double *pq[2];                // [nb_row] each: one slot per workgroup of k_pcg_a
    int nb_cam, nb_row;
};
// sum of n partials, the same bits in every thread of every workgroup: thread t adds elements t, t + 256, ... in
// order, then the fixed tree of block_sum256
__device__ __forceinline__ double pcg_total(const double *__restrict__ part, int n, double *red) {
    double s = 0.0;
    for (int i = threadIdx.x; i < n; i += 256) s += part[i];
    s = wave_sum_dpp(s);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) red[w] = s;
    __syncthreads();
    return (red[0] + red[1]) + (red[2] + red[3]);
}

__global__ __launch_bounds__(256) void k_pcg_init(int nf, const double *__restrict__ gred, const double *__restrict__ Minv, double *__restrict__ x,
                           double *__restrict__ r, double *__restrict__ z, PcgPartials pp) {
    __shared__ double red[4];
    const int f = blockIdx.x * blockDim.x + threadIdx.x;
    double rz = 0.0, bb = 0.0;
    if (f < nf) {
        double rv[6];
#pragma unroll
        for (int b = 0; b < 6; ++b) rv[b] = -gred[(size_t)f * 6 + b];
#pragma unroll
        for (int a = 0; a < 6; ++a) {
            double zz = 0.0;
#pragma unroll
            for (int b = 0; b < 6; ++b) zz += Minv[(size_t)f * 36 + a * 6 + b] * rv[b];
            x[(size_t)f * 6 + a] = 0.0;
            r[(size_t)f * 6 + a] = rv[a];
            z[(size_t)f * 6 + a] = zz;
            rz += rv[a] * zz;
            bb += rv[a] * rv[a];
        }
    }
    rz = block_sum256(rz, red);
    bb = block_sum256(bb, red);
    if (threadIdx.x == 0) { pp.rz[0][blockIdx.x] = rz; pp.bb[blockIdx.x] = bb; pp.rn[0][blockIdx.x] = bb; }
}

__global__ __launch_bounds__(256) void k_pcg_a(int nf, int it, double tol2, const int *__restrict__ bsr_row,
                                               const int *__restrict__ bsr_col, const int *__restrict__ bsr_src,
                                               const double *__restrict__ S, const double *__restrict__ z,
                                               const double *__restrict__ p_in, double *__restrict__ p_out,
                                               double *__restrict__ q, PcgScalars *sc, PcgPartials pp) {
    __shared__ double red[4];
    if (sc->done != 0.0) return;
    const double rz_cur = pcg_total(pp.rz[it % 3], pp.nb_cam, red);
    const double rn_cur = pcg_total(pp.rn[it % 3], pp.nb_cam, red);
    const double bnorm2 = pcg_total(pp.bb, pp.nb_cam, red);
    if (!(rn_cur > tol2 * bnorm2) && it > 0) {  // converged after iteration it-1 (same test, same bits, in every workgroup)
        if (blockIdx.x == 0 && threadIdx.x == 0) { sc->done = 1.0; sc->iters = it; }
        return;
    }
    double beta = 0.0;
    if (it > 0) {
        const double rz_prev = pcg_total(pp.rz[(it + 2) % 3], pp.nb_cam, red);
        if (rz_prev > 0.0) beta = rz_cur / rz_prev;
    }
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int f = blockIdx.x * 4 + wv;
    const int a = lane & 7, slot = lane >> 3;
    double acc = 0.0, pq = 0.0;
    if (f < nf) {
        if (a < 6) {
            for (int k = bsr_row[f] + slot; k < bsr_row[f + 1]; k += 8) {
                const int src = bsr_src[k];
                const double *blk = &S[(size_t)(src & 0x7fffffff) * 36];
                const size_t j6 = (size_t)bsr_col[k] * 6;
                double pj[6];
#pragma unroll
                for (int b = 0; b < 6; ++b) pj[b] = (it > 0) ? z[j6 + b] + beta * p_in[j6 + b] : z[j6 + b];
                if (src >= 0) {
#pragma unroll
                    for (int b = 0; b < 6; ++b) acc += blk[a * 6 + b] * pj[b];
                } else {
#pragma unroll
                    for (int b = 0; b < 6; ++b) acc += blk[b * 6 + a] * pj[b];
                }
            }
        }
        acc += __shfl_xor(acc, 8, 64);
        acc += __shfl_xor(acc, 16, 64);
        acc += __shfl_xor(acc, 32, 64);
        if (slot == 0 && a < 6) {
            const size_t i = (size_t)f * 6 + a;
            const double pn = (it > 0) ? z[i] + beta * p_in[i] : z[i];
            p_out[i] = pn;
            q[i] = acc;
            pq = pn * acc;
        }
    }
    pq = block_sum256(pq, red);
    if (threadIdx.x == 0) pp.pq[it & 1][blockIdx.x] = pq;
}

__global__ __launch_bounds__(256) void k_pcg_b(int nf, int it, const double *__restrict__ Minv, const double *__restrict__ p,
                                               const double *__restrict__ q, double *__restrict__ x,
                                               double *__restrict__ r, double *__restrict__ z, PcgScalars *sc, PcgPartials pp) {
    __shared__ double red[4];
    if (sc->done != 0.0) return;
    const int f = blockIdx.x * blockDim.x + threadIdx.x;
    const double pqv = pcg_total(pp.pq[it & 1], pp.nb_row, red);
    const double rz_cur = pcg_total(pp.rz[it % 3], pp.nb_cam, red);
    const double alpha = (pqv > 0.0) ? rz_cur / pqv : 0.0;
    double rz = 0.0, rn = 0.0;
    if (f < nf) {
        double rv[6];
#pragma unroll
        for (int b = 0; b < 6; ++b) {
            const size_t i = (size_t)f * 6 + b;
            x[i] += alpha * p[i];
            rv[b] = r[i] - alpha * q[i];
            r[i] = rv[b];
            rn += rv[b] * rv[b];
        }
        if (Minv) {   // block-Jacobi: z = M^-1 r here; another preconditioner (do_bcr_pcg): z and r.z by the kernels that follow
#pragma unroll
            for (int a = 0; a < 6; ++a) {
                double zz = 0.0;
#pragma unroll
                for (int b = 0; b < 6; ++b) zz += Minv[(size_t)f * 36 + a * 6 + b] * rv[b];
                z[(size_t)f * 6 + a] = zz;
                rz += rv[a] * zz;
            }
        }
    }
    rz = block_sum256(rz, red);
    rn = block_sum256(rn, red);
    if (threadIdx.x == 0) {
        if (Minv) pp.rz[(it + 1) % 3][blockIdx.x] = rz;
        pp.rn[(it + 1) % 3][blockIdx.x] = rn;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        sc->iters = it + 1;
        if (!(pqv > 0.0)) sc->done = 1.0;  // S p . p <= 0: stop (x keeps the last good iterate)
    }
}

// PCG with a preconditioner of its own (the cyclic reduction of the in-band part of S, do_bcr_pcg):
//   k_pcgx_init: x = 0, r = -g_red, and with z = M^-1 r already there: r.z and |r|^2 into the partials of slot 0
//   k_pcgx_dot:  r.z of iteration it + 1 once z = M^-1 r has been applied
__global__ __launch_bounds__(256) void k_pcgx_init(int nf, const double *__restrict__ gred, const double *__restrict__ z,
                                                   double *__restrict__ x, double *__restrict__ r, PcgPartials pp) {
    __shared__ double red[4];
    const int f = blockIdx.x * blockDim.x + threadIdx.x;
    double rz = 0.0, bb = 0.0;
    if (f < nf) {
#pragma unroll
        for (int a = 0; a < 6; ++a) {
            const size_t i = (size_t)f * 6 + a;
            const double rv = -gred[i];
            x[i] = 0.0;
            r[i] = rv;
            rz += rv * z[i];
            bb += rv * rv;
        }
    }
    rz = block_sum256(rz, red);
    bb = block_sum256(bb, red);
    if (threadIdx.x == 0) { pp.rz[0][blockIdx.x] = rz; pp.bb[blockIdx.x] = bb; pp.rn[0][blockIdx.x] = bb; }
}
__global__ __launch_bounds__(256) void k_pcgx_dot(int nf, int it, const double *__restrict__ r, const double *__restrict__ z,
                                                  const PcgScalars *sc, PcgPartials pp) {
    __shared__ double red[4];
    if (sc->done != 0.0) return;
    const int f = blockIdx.x * blockDim.x + threadIdx.x;
    double rz = 0.0;
    if (f < nf) {
#pragma unroll
        for (int a = 0; a < 6; ++a) rz += r[(size_t)f * 6 + a] * z[(size_t)f * 6 + a];
    }
    rz = block_sum256(rz, red);
    if (threadIdx.x == 0) pp.rz[(it + 1) % 3][blockIdx.x] = rz;
}

// ------------------------------------------------------------------ trial cameras
// dx_loc (may be null): the step in the local rotation tangent, (Jl dr, dt) per free camera -- what the chunk
// back-substitution multiplies its local-tangent camera blocks with (one 3x3 product per camera here instead of one per
// observation there)
__global__ void k_cam_update(int n_cams, const int *__restrict__ cam_free, const double *__restrict__ pose,
                             const double *__restrict__ intr, const double *__restrict__ dx, double *__restrict__ pose_t,
                             double *__restrict__ aux_t, double *__restrict__ cam_part, const double *__restrict__ aux,
                             double *__restrict__ dx_loc) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= n_cams) return;
    const int f = cam_free[c];
    double ps[6], dd[6], d2 = 0.0, x2 = 0.0;
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        const double v = pose[6 * (size_t)c + i];
        double d = 0.0;
        if (f >= 0) { d = dx[(size_t)f * 6 + i]; d2 += d * d; x2 += v * v; }
        dd[i] = d;
        ps[i] = v + d;
        pose_t[6 * (size_t)c + i] = ps[i];
    }
    if (f >= 0 && dx_loc) {
        const double *J = aux + (size_t)c * BA_CAM_AUX + 9;
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            dx_loc[(size_t)f * 6 + r] = J[3 * r] * dd[0] + J[3 * r + 1] * dd[1] + J[3 * r + 2] * dd[2];
            dx_loc[(size_t)f * 6 + 3 + r] = dd[3 + r];
        }
    }
    cam_aux_from_pose(ps, intr + 4 * (size_t)c, aux_t + (size_t)c * BA_CAM_AUX);
    cam_part[2 * (size_t)c] = d2;
    cam_part[2 * (size_t)c + 1] = x2;
}

// ------------------------------------------------------------------ back-substitution + trial cost
struct BacksubArgs {
    const double *aux, *aux_t;
    const double *points;
    double *points_t;
    const int *o_cam, *o_pt;
    const double *o_xy;
    const int *cam_free;
    const int *ck_obs, *ck_pt;
    const double *dx;               // camera step [n_free*6]: in the local rotation tangent for k_backsub, in rvec
                                    // coordinates for k_backsub_heavy (dx_rvec)
    const double *dx_rvec;
    const double *pt_linv, *pt_u;
    double *ck_part;                // [n_chunks*4]: pred, F_trial, |dp|^2, |x_p|^2
    int loss;
    double inv_fs2, half_fs2;
    const LmState *state;
};

// LOSS: the robust loss as a compile-time constant (huber), or -1 for A.loss -- as in k_build.  The chunk's loads are
// issued unconditionally (clamped indices), and every barrier orders LDS only: the stores of the trial points and the
// loads of the next phase stay in flight across them.
template <int LOSS>
__global__ __launch_bounds__(CHUNK_OBS) __attribute__((amdgpu_waves_per_eu(5, 5))) void k_backsub(const BacksubArgs A) {
    __shared__ double sP[CHUNK_PTS * 3];
    __shared__ double sRed[4][4];
    const int ck = blockIdx.x, tid = threadIdx.x;
    const int ob = A.ck_obs[ck], nobs = A.ck_obs[ck + 1] - ob;
    const int pb = A.ck_pt[ck], npts = A.ck_pt[ck + 1] - pb;
    const int loss = LOSS >= 0 ? LOSS : A.loss;
    const bool act = tid < nobs;
    const int o = ob + (act ? tid : 0);
    const int cam = A.o_cam[o];
    const int p = A.o_pt[o];
    const double2 xy = reinterpret_cast<const double2 *>(A.o_xy)[o];
    CamAux ca;
    load_aux_local(A.aux, cam, ca);   // (local rotation tangent; A.dx is the step in the same coordinates, k_cam_update)
    const double X0 = A.points[3 * (size_t)p], X1 = A.points[3 * (size_t)p + 1], X2 = A.points[3 * (size_t)p + 2];
    const int f = A.cam_free[cam];
    const double floor_now = A.state->floor;
    for (int i = tid; i < npts * 3; i += CHUNK_OBS) sP[i] = 0.0;
    lds_barrier();
    const int lp = act ? p - pb : 0;
    const double uo = xy.x, vo = xy.y;
    double r0 = 0, r1 = 0, w0 = 0, w1 = 0, h0 = 0, h1 = 0, a0 = 0, a1 = 0, Jp[2][3];
    {
        ObsLin L;
        linearize(ca, X0, X1, X2, uo, vo, loss, A.inv_fs2, floor_now, L);
        r0 = L.r[0]; r1 = L.r[1]; w0 = L.w[0]; w1 = L.w[1]; h0 = L.wh[0]; h1 = L.wh[1];
#pragma unroll
        for (int j = 0; j < 3; ++j) { Jp[0][j] = L.Jp[0][j]; Jp[1][j] = L.Jp[1][j]; }
        if (f >= 0) {
#pragma unroll
            for (int i = 0; i < 6; ++i) {
                const double d = A.dx[(size_t)f * 6 + i];
                a0 += L.Jc[0][i] * d; a1 += L.Jc[1][i] * d;
            }
        }
        if (act) {
            const double wa0 = h0 * a0, wa1 = h1 * a1;
#pragma unroll
            for (int j = 0; j < 3; ++j) atomicAdd(&sP[lp * 3 + j], Jp[0][j] * wa0 + Jp[1][j] * wa1);
        }
    }
    // the per-point operands of the middle phase are requested in front of the barrier that precedes their use (not at the
    // top of the kernel: twelve more live doubles through the linearisation cost the fifth wave per SIMD)
    double li[6], uu[3], pp[3];
    {
        const size_t q = (size_t)(pb + (tid < npts ? tid : 0));
#pragma unroll
        for (int k = 0; k < 6; ++k) li[k] = A.pt_linv[q * 6 + k];
#pragma unroll
        for (int k = 0; k < 3; ++k) { uu[k] = A.pt_u[q * 3 + k]; pp[k] = A.points[3 * q + k]; }
    }
    lds_barrier();
    double dp2 = 0.0, xp2 = 0.0;
    if (tid < npts) {
        const size_t q = (size_t)(pb + tid);
        const double i00 = li[0], i10 = li[1], i11 = li[2], i20 = li[3], i21 = li[4], i22 = li[5];
        const double s0 = sP[tid * 3], s1 = sP[tid * 3 + 1], s2 = sP[tid * 3 + 2];
        const double y0 = uu[0] + i00 * s0;
        const double y1 = uu[1] + i10 * s0 + i11 * s1;
        const double y2 = uu[2] + i20 * s0 + i21 * s1 + i22 * s2;
        const double d0 = -(i00 * y0 + i10 * y1 + i20 * y2);
        const double d1 = -(i11 * y1 + i21 * y2);
        const double d2 = -(i22 * y2);
        sP[tid * 3] = d0; sP[tid * 3 + 1] = d1; sP[tid * 3 + 2] = d2;
        const double p0 = pp[0], p1 = pp[1], p2 = pp[2];
        A.points_t[3 * q] = p0 + d0; A.points_t[3 * q + 1] = p1 + d1; A.points_t[3 * q + 2] = p2 + d2;
        dp2 = d0 * d0 + d1 * d1 + d2 * d2;
        xp2 = p0 * p0 + p1 * p1 + p2 * p2;
    }
    // the trial camera of the last phase (R, t, intrinsics), requested in front of the barrier (the point operands are dead
    // by now; any earlier and the kernel spills at five waves per SIMD)
    double2 tq[9];
    {
        const double2 *q = reinterpret_cast<const double2 *>(A.aux_t + (size_t)cam * BA_CAM_AUX);
#pragma unroll
        for (int k = 0; k < 5; ++k) tq[k] = q[k];
#pragma unroll
        for (int k = 0; k < 4; ++k) tq[5 + k] = q[9 + k];
    }
    lds_barrier();
    double pred = 0.0, ft = 0.0;
    if (act) {
        const double d0 = sP[lp * 3], d1 = sP[lp * 3 + 1], d2 = sP[lp * 3 + 2];
        const double jd0 = a0 + Jp[0][0] * d0 + Jp[0][1] * d1 + Jp[0][2] * d2;
        const double jd1 = a1 + Jp[1][0] * d0 + Jp[1][1] * d1 + Jp[1][2] * d2;
        pred = -(jd0 * (w0 * r0 + 0.5 * h0 * jd0) + jd1 * (w1 * r1 + 0.5 * h1 * jd1));
        const double Z0 = X0 + d0, Z1 = X1 + d1, Z2 = X2 + d2;
        const double x = tq[0].x * Z0 + tq[0].y * Z1 + tq[1].x * Z2 + tq[5].x;
        const double y = tq[1].y * Z0 + tq[2].x * Z1 + tq[2].y * Z2 + tq[5].y;
        const double z = tq[3].x * Z0 + tq[3].y * Z1 + tq[4].x * Z2 + tq[6].x;
        const double iz = (z != 0.0) ? 1.0 / z : 1.0;  // (project_only's arithmetic, operands loaded above)
        const double u = tq[6].y * (x * iz) + tq[7].y, v = tq[7].x * (y * iz) + tq[8].x;
        const double e0 = u - uo, e1 = v - vo;
        double rho0, rho1, wd;
        loss_eval(loss, e0 * e0 * A.inv_fs2, rho0, wd);
        loss_eval(loss, e1 * e1 * A.inv_fs2, rho1, wd);
        ft = A.half_fs2 * (rho0 + rho1);
    }
    // four block sums behind ONE barrier: DPP wave sums (fixed order), then the four wave results per quantity
    pred = wave_sum_dpp(pred);
    ft = wave_sum_dpp(ft);
    dp2 = wave_sum_dpp(dp2);
    xp2 = wave_sum_dpp(xp2);
    if ((tid & 63) == 0) {
        double *r = sRed[tid >> 6];
        r[0] = pred; r[1] = ft; r[2] = dp2; r[3] = xp2;
    }
    lds_barrier();
    if (tid < 4) A.ck_part[(size_t)ck * 4 + tid] = (sRed[0][tid] + sRed[1][tid]) + (sRed[2][tid] + sRed[3][tid]);
}

// ------------------------------------------------------------------ deterministic reductions + LM control
// out[k] = sum over i of in[i*stride + k] (k < stride <= 4), fixed order -> bitwise reproducible.  ONE block of 1024
// threads; all loads of a trip (two rows, every column) are issued before any is used.
__device__ __forceinline__ void reduce_sum_1024(int n, int stride, const double *__restrict__ in, double *__restrict__ out,
                                                double (*red)[4]) {
    double acc[4] = {0.0, 0.0, 0.0, 0.0};
    for (int i = threadIdx.x; i < n; i += 4096) {   // four rows per trip: sixteen independent loads in flight
        double a[4][4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int iu = i + 1024 * u, ju = iu < n ? iu : i;
#pragma unroll
            for (int k = 0; k < 4; ++k) a[u][k] = in[(size_t)ju * stride + (k < stride ? k : 0)];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int k = 0; k < 4; ++k) acc[k] += i + 1024 * u < n ? a[u][k] : 0.0;
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const double v = wave_sum_dpp(acc[k]);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][k] = v;
    }
    __syncthreads();
    if ((int)threadIdx.x < stride) {
        double t = 0.0;
        for (int w = 0; w < 16; ++w) t += red[w][threadIdx.x];
        out[threadIdx.x] = t;
    }
}
__global__ __launch_bounds__(1024) void k_reduce_sum(int n, int stride, const double *__restrict__ in, double *__restrict__ out) {
    __shared__ double red[16][4];
    reduce_sum_1024(n, stride, in, out, red);
}
__global__ void k_reduce_max(int n, const double *__restrict__ in, double *__restrict__ out) {
    __shared__ double red[4];
    double s = 0.0;
    for (int i = threadIdx.x; i < n; i += 256) s = fmax(s, in[i]);
    s = block_max256(s, red);
    if (threadIdx.x == 0) out[0] = s;
}

__global__ void k_add_scalar(double *p, double v) { *p += v; }
__global__ void k_negate(int n, const double *__restrict__ in, double *__restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = -in[i];
}
__global__ void k_copy_scalar(const double *src, double *dst) { *dst = *src; }
// *out = NaN when *bad is set (an index was out of range), else unchanged
__global__ void k_nan_if(const int *bad, double *out) {
    if (*bad) *out = __longlong_as_double(0x7ff8000000000000LL);
}

// (single-GPU build pass) k_chunk_totals and k_state_after_build in one launch, after k_apply_finalize: same fixed-order
// sums as k_chunk_totals, then F and ||g||_inf into the state
__global__ __launch_bounds__(1024) void k_totals_state(int n, const double *__restrict__ ck_cost, const double *__restrict__ ck_gmax,
                                                       double *__restrict__ sum_out, double *__restrict__ gm, LmState *st, int nf,
                                                       const double *__restrict__ cam_gmax) {
    __shared__ double rs[16], rm[16], rc[16];
    double s = 0.0, m = 0.0, mc = 0.0;
    for (int i = threadIdx.x; i < n; i += 8192) {   // (sixteen loads in flight per lane: the kernel is one workgroup's trips to memory)
        double c[8], g[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int idx = i + u * 1024;
            const int j = idx < n ? idx : i;
            c[u] = ck_cost[j];
            g[u] = ck_gmax[j];
            if (idx >= n) c[u] = g[u] = 0.0;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            s += c[u];
            m = fmax(m, g[u]);
        }
    }
    for (int i = threadIdx.x; i < nf; i += 1024) mc = fmax(mc, cam_gmax[i]);
    s = wave_sum(s);
    m = wave_max(m);
    mc = wave_max(mc);
    if ((threadIdx.x & 63) == 0) {
        rs[threadIdx.x >> 6] = s;
        rm[threadIdx.x >> 6] = m;
        rc[threadIdx.x >> 6] = mc;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        double ts = 0.0, tm = 0.0, tc = 0.0;
        for (int w = 0; w < 16; ++w) {
            ts += rs[w];
            tm = fmax(tm, rm[w]);
            tc = fmax(tc, rc[w]);
        }
        *sum_out = ts;
        gm[0] = tm;
        gm[1] = tc;
        st->F = ts;
        st->gmax = fmax(tm, tc);
    }
}

// after k_build, ONE 1024-thread block: *sum_out = sum of the chunk costs (fixed order -> bitwise reproducible),
// *max_out = max of the chunks' point-gradient maxima.  A one-block reduction is bound by load latency, so every
// trip issues eight independent loads before it uses any.
__global__ __launch_bounds__(1024) void k_chunk_totals(int n, const double *__restrict__ ck_cost, const double *__restrict__ ck_gmax,
                                                       double *__restrict__ sum_out, double *__restrict__ max_out) {
    __shared__ double rs[16], rm[16];
    double s = 0.0, m = 0.0;
    for (int i = threadIdx.x; i < n; i += 4096) {
        double c[4], g[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int idx = i + u * 1024;
            const int j = idx < n ? idx : i;
            c[u] = ck_cost[j];
            g[u] = ck_gmax[j];
            if (idx >= n) c[u] = g[u] = 0.0;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            s += c[u];
            m = fmax(m, g[u]);
        }
    }
    s = wave_sum(s);
    m = wave_max(m);
    if ((threadIdx.x & 63) == 0) {
        rs[threadIdx.x >> 6] = s;
        rm[threadIdx.x >> 6] = m;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        double ts = 0.0, tm = 0.0;
        for (int w = 0; w < 16; ++w) {
            ts += rs[w];
            tm = fmax(tm, rm[w]);
        }
        *sum_out = ts;
        *max_out = tm;
    }
}

// after a build (256 threads): F and ||g||_inf into the state.  sums[0] = F (already all-reduced), gm[0] = point
// maximum (all-reduced), camera maximum = max of cam_gmax[0..nf) (replicated), kept in gm[1].
__global__ __launch_bounds__(256) void k_state_after_build(LmState *st, const double *sums, double *gm, int nf,
                                                           const double *__restrict__ cam_gmax) {
    __shared__ double red[4];
    double m = 0.0;
    for (int i = threadIdx.x; i < nf; i += 256) m = fmax(m, cam_gmax[i]);
    m = block_max256(m, red);
    if (threadIdx.x == 0) {
        gm[1] = m;
        st->F = sums[0];
        st->gmax = fmax(gm[0], m);
    }
}

// after a trial step: gain ratio, accept / reject, damping update, termination (scipy _lsq/common.py:705-717)
// step[0..3] = pred, F_trial, |dp|^2, |x_p|^2 (all-reduced); cams[0..1] = |dc|^2, |x_c|^2 (replicated)
// solve_info (may be null): != 0 when the direct solver met a non-positive pivot -- the step is void, the state stays
// as it is and the host repeats the step with PCG
__device__ __forceinline__ void decide_step(LmState *st, const double *step, const double *cams, double x2_const, double ftol,
                                            double xtol, double gtol, const int *solve_info) {
    // scipy's gtol test on the gradient of the linearisation this step started from.  It sits here, not on the host, so
    // that the host does not have to read the state back after every build: a step queued behind a build whose gradient
    // was already small is void (state untouched), terminate = 1 tells the host.
    if (st->gmax < gtol) {
        st->solve_info = 0;
        st->accept = 0;
        st->terminate = 1;
        return;
    }
    st->solve_info = solve_info ? *solve_info : 0;
    if (st->solve_info != 0) return;
    const double pred = step[0], Ft = step[1];
    const double F = st->F;
    const double dF = F - Ft;
    const double step_norm = sqrt(step[2] + cams[0]);
    const double x_norm = sqrt(step[3] + cams[1] + x2_const);
    double rho = (pred > 0.0) ? dF / pred : -1.0;
    const bool finite = (Ft == Ft) && (fabs(Ft) < 1e300) && (pred == pred);
    if (!finite) rho = -1.0;
    st->F_trial = Ft; st->pred = pred; st->rho = rho; st->dF = dF;
    st->step_norm = step_norm; st->x_norm = x_norm;
    st->iter += 1;
    const bool ftol_ok = finite && dF < ftol * F && rho > 0.25;
    const bool xtol_ok = finite && step_norm < xtol * (xtol + x_norm);
    const bool accept = finite && rho > 0.0;
    st->accept = accept ? 1 : 0;
    if (accept) {
        st->n_accept += 1;
        const double t = 2.0 * rho - 1.0;
        st->lambda *= fmax(1.0 / 3.0, 1.0 - t * t * t);
        st->nu = 2.0;
        st->F = Ft;
        // curvature floor: plain IRLS weights (floor 1, a majoriser: safe far from the minimum) until the
        // steps become small and well predicted, then towards the exact curvature rho' + 2 rho'' z
        if (rho > 0.5 && dF < 0.05 * F) st->floor = fmax(st->floor_min, 0.25 * st->floor);
    } else {
        st->lambda *= st->nu;
        st->nu *= 2.0;
        st->floor = fmin(1.0, 4.0 * st->floor);
    }
    st->lambda = fmin(fmax(st->lambda, 1e-12), 1e12);
    int term = 0;
    if (ftol_ok && xtol_ok) term = 4;
    else if (ftol_ok) term = 2;
    else if (xtol_ok) term = 3;
    st->terminate = term;
}
__global__ void k_decide(LmState *st, const double *step, const double *cams, double x2_const, double ftol, double xtol,
                         double gtol, const int *solve_info) {
    decide_step(st, step, cams, x2_const, ftol, xtol, gtol, solve_info);
}
// (single GPU) the tail of a trial step in one launch: the two fixed-order reductions of the per-chunk and per-camera
// partial sums (same order as k_reduce_sum), |x|^2 of the unobserved points, and the decision
__global__ __launch_bounds__(1024) void k_reduce_decide(int n_ck, const double *__restrict__ ck_part, int n_cams,
                                                        const double *__restrict__ cam_part, double *__restrict__ step_sums,
                                                        double *__restrict__ cam_sums, LmState *st, double x2_const, double ftol,
                                                        double xtol, double gtol, const int *solve_info) {
    __shared__ double red[16][4];
    reduce_sum_1024(n_ck, 4, ck_part, step_sums, red);
    __syncthreads();
    reduce_sum_1024(n_cams, 2, cam_part, cam_sums, red);
    __syncthreads();   // (thread 0 reads what threads 0 .. 3 wrote: same workgroup, behind a barrier)
    if (threadIdx.x == 0) {
        step_sums[3] += x2_const;
        decide_step(st, step_sums, cam_sums, 0.0, ftol, xtol, gtol, solve_info);
    }
}

// ------------------------------------------------------------------ evaluation kernels, caller's order
__global__ void k_eval_residuals(long long n_obs, const double *__restrict__ aux, const double *__restrict__ points,
                                 const int *__restrict__ obs_cam, const int *__restrict__ obs_pt,
                                 const double *__restrict__ obs_xy, double *__restrict__ out) {
    const long long o = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (o >= n_obs) return;
    const int p = obs_pt[o];
    double u, v;
    project_only(aux, obs_cam[o], points[3 * (size_t)p], points[3 * (size_t)p + 1], points[3 * (size_t)p + 2], u, v);
    out[2 * o] = u - obs_xy[2 * o];
    out[2 * o + 1] = v - obs_xy[2 * o + 1];
}

// CHECK (the no-allocation entry point, whose indices were never seen by the host): an observation whose camera or
// point index is out of range is skipped and raises *bad instead of faulting; the arithmetic is the same.
template <bool CHECK>
__global__ void k_eval_cost(long long n_obs, const double *__restrict__ aux, const double *__restrict__ points,
                            const int *__restrict__ obs_cam, const int *__restrict__ obs_pt,
                            const double *__restrict__ obs_xy, int loss, double inv_fs2, double half_fs2,
                            double *__restrict__ blk_cost, double *__restrict__ blk_rsum, int n_cams, int n_points,
                            int *__restrict__ bad) {
    __shared__ double red[4];
    const long long o = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    double c = 0.0, rs = 0.0;
    bool in = o < n_obs;
    if (CHECK && in) {
        const int p = obs_pt[o], cm = obs_cam[o];
        if (p < 0 || p >= n_points || cm < 0 || cm >= n_cams) {
            *bad = 1;
            in = false;
        }
    }
    if (in) {
        const int p = obs_pt[o];
        double u, v;
        project_only(aux, obs_cam[o], points[3 * (size_t)p], points[3 * (size_t)p + 1], points[3 * (size_t)p + 2], u, v);
        const double e0 = u - obs_xy[2 * o], e1 = v - obs_xy[2 * o + 1];
        double rho0, rho1, wd;
        loss_eval(loss, e0 * e0 * inv_fs2, rho0, wd);
        loss_eval(loss, e1 * e1 * inv_fs2, rho1, wd);
        c = half_fs2 * (rho0 + rho1);
        rs = e0 + e1;
    }
    c = block_sum256(c, red);
    rs = block_sum256(rs, red);
    if (threadIdx.x == 0) { blk_cost[blockIdx.x] = c; blk_rsum[blockIdx.x] = rs; }
}

__global__ void k_eval_jacobian(long long n_obs, const double *__restrict__ aux, const double *__restrict__ points,
                                const int *__restrict__ obs_cam, const int *__restrict__ obs_pt,
                                double *__restrict__ jc, double *__restrict__ jp) {
    const long long o = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (o >= n_obs) return;
    const int p = obs_pt[o];
    CamAux ca;
    load_aux(aux, obs_cam[o], ca);
    ObsLin L;
    linearize(ca, points[3 * (size_t)p], points[3 * (size_t)p + 1], points[3 * (size_t)p + 2], 0.0, 0.0, 0, 1.0, 1.0, L);
#pragma unroll
    for (int k = 0; k < 2; ++k) {
#pragma unroll
        for (int i = 0; i < 6; ++i) jc[12 * o + 6 * k + i] = L.Jc[k][i];
#pragma unroll
        for (int j = 0; j < 3; ++j) jp[6 * o + 3 * k + j] = L.Jp[k][j];
    }
}

// per-camera mean L2 reprojection error with the full K (utils.py:271-276): sums by atomics.  The pose comes from
// rows of `rt` (stride doubles per camera, R row-major at 0, t at t_off): the engine's aux rows (stride BA_CAM_AUX,
// t at 18) or rotation matrices exactly as the caller holds them (stride 12, t at 9).
__global__ void k_eval_reproj(long long n_obs, const double *__restrict__ rt, int stride, int t_off,
                              const double *__restrict__ cam_k,
                              const double *__restrict__ points, const int *__restrict__ obs_cam,
                              const int *__restrict__ obs_pt, const double *__restrict__ obs_xy,
                              double *__restrict__ cam_sum, double *__restrict__ cam_cnt) {
    const long long o = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (o >= n_obs) return;
    const int p = obs_pt[o], c = obs_cam[o];
    const double *a = rt + (size_t)c * stride, *tt = a + t_off;
    const double X0 = points[3 * (size_t)p], X1 = points[3 * (size_t)p + 1], X2 = points[3 * (size_t)p + 2];
    const double x = a[0] * X0 + a[1] * X1 + a[2] * X2 + tt[0];
    const double y = a[3] * X0 + a[4] * X1 + a[5] * X2 + tt[1];
    const double z = a[6] * X0 + a[7] * X1 + a[8] * X2 + tt[2];
    const double *K = cam_k + 9 * (size_t)c;
    const double h0 = K[0] * x + K[1] * y + K[2] * z;
    const double h1 = K[3] * x + K[4] * y + K[5] * z;
    const double h2 = K[6] * x + K[7] * y + K[8] * z;
    const double e0 = obs_xy[2 * o] - h0 / h2, e1 = obs_xy[2 * o + 1] - h1 / h2;
    atomicAdd(&cam_sum[c], sqrt(e0 * e0 + e1 * e1));
    atomicAdd(&cam_cnt[c], 1.0);
}

// K = [[fx 0 cx] [0 fy cy] [0 0 1]] per camera from the engine's (fx, fy, cx, cy)
__global__ void k_intr_to_k(int n_cams, const double *__restrict__ intr, double *__restrict__ cam_k) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= n_cams) return;
    double *K = cam_k + 9 * (size_t)c;
    const double *in = intr + 4 * (size_t)c;
    K[0] = in[0]; K[1] = 0.0; K[2] = in[2];
    K[3] = 0.0; K[4] = in[1]; K[5] = in[3];
    K[6] = 0.0; K[7] = 0.0; K[8] = 1.0;
}
__global__ void k_mean_or_nan(int n, const double *__restrict__ sum, const double *__restrict__ cnt, double *__restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = cnt[i] > 0.0 ? sum[i] / cnt[i] : __longlong_as_double(0x7ff8000000000000LL);
}

}  // namespace ba
