// Heavy points: tracks with more observations than a chunk holds (> CHUNK_OBS).  Rare (a landmark seen from
// hundreds of cameras), so these kernels are written for correctness at any track length, not for speed: one
// 256-thread workgroup per heavy point, observations in strides of 256, every quantity recomputed per pass instead
// of kept.  Same arithmetic as k_build / k_backsub (damping, factor, signs), same outputs: U, g_c, S, g_red,
// pt_linv, pt_u, and one row of the per-chunk partials (row n_chunks + heavy index).
#pragma once
#include "ba_kernels.h"

namespace ba {

struct HeavyArgs {
    int n_chunks;                 // rows of ck_cost / ck_gmax / ck_part that belong to chunks
    int pt0;                      // internal index of the first heavy point
    const long long *hv_obs;      // [n_heavy+1] observation range
    const long long *hv_pair;     // [n_heavy+1] into hv_pair_slot
    const int *hv_pair_slot;      // triangular (a <= c) list per point: slot | HV_SLOT_* or -1
    double *hv_t;                 // scratch: T = W L^-T per heavy observation [(obs - hv_obs[0]) * 18]
};

// D.cam_slab != nullptr: the deterministic build (ba_det.h) -- every sum that the atomic mode adds into U, g_c, g_red
// and S goes, with plain stores, to a slab row of its own instead (one per observation, one per observation pair);
// k_det_reduce adds the rows up in a fixed order.
__global__ __launch_bounds__(256) void k_build_heavy(const BuildArgs A, const HeavyArgs H, const DetArgs D) {
    __shared__ double sRed[4];
    const int hp = blockIdx.x, tid = threadIdx.x;
    const long long ob = H.hv_obs[hp];
    const int L = (int)(H.hv_obs[hp + 1] - ob);
    const long long tb = (ob - H.hv_obs[0]) * 18;
    const int p = H.pt0 + hp;
    const double lam = A.state->lambda, floor_now = A.state->floor;
    const double X0 = A.points[3 * (size_t)p], X1 = A.points[3 * (size_t)p + 1], X2 = A.points[3 * (size_t)p + 2];

    // ---- pass 1: V, g_p, cost of the point; U, g_c of its cameras --------------------------------
    double vs[9], cost = 0.0;
#pragma unroll
    for (int k = 0; k < 9; ++k) vs[k] = 0.0;
    for (int o = tid; o < L; o += 256) {
        const int cam = A.o_cam[ob + o];
        const double2 xy = reinterpret_cast<const double2 *>(A.o_xy)[ob + o];
        CamAux ca;
        load_aux_local(A.aux, cam, ca);  // (local rotation tangent: see load_aux_local)
        ObsLin Ln;
        linearize(ca, X0, X1, X2, xy.x, xy.y, A.loss, A.inv_fs2, floor_now, Ln);
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
        const int f = A.cam_free[cam];
        if (f >= 0 && D.cam_slab) {
            double *row = &D.cam_slab[(size_t)(D.n_cl + (ob - H.hv_obs[0]) + o) * 33];
#pragma unroll
            for (int i = 0; i < 6; ++i) {
#pragma unroll
                for (int j = i; j < 6; ++j)
                    row[ut6(i, j)] = Ln.wh[0] * Ln.Jc[0][i] * Ln.Jc[0][j] + Ln.wh[1] * Ln.Jc[1][i] * Ln.Jc[1][j];
                row[21 + i] = Ln.Jc[0][i] * wr0 + Ln.Jc[1][i] * wr1;
            }
        } else if (f >= 0) {
            double *up = &A.U[(size_t)f * 21], *gp = &A.gc[(size_t)f * 6];
#pragma unroll
            for (int i = 0; i < 6; ++i) {
#pragma unroll
                for (int j = i; j < 6; ++j)
                    atomicAdd(&up[ut6(i, j)], Ln.wh[0] * Ln.Jc[0][i] * Ln.Jc[0][j] + Ln.wh[1] * Ln.Jc[1][i] * Ln.Jc[1][j]);
                atomicAdd(&gp[i], Ln.Jc[0][i] * wr0 + Ln.Jc[1][i] * wr1);
            }
        }
    }
#pragma unroll
    for (int k = 0; k < 9; ++k) vs[k] = block_sum256(vs[k], sRed);
    cost = block_sum256(cost, sRed);
    // ---- damp, 3x3 Cholesky, inverse factor (every thread; k_build phase 2a) -----------------------
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
    // ---- pass 2: T = W L^-T per observation (global scratch), g_red -= T u ------------------------
    for (int o = tid; o < L; o += 256) {
        const int cam = A.o_cam[ob + o];
        const int f = A.cam_free[cam];
        double *tp = &H.hv_t[tb + (long long)o * 18];
        if (f < 0) {
#pragma unroll
            for (int k = 0; k < 18; ++k) tp[k] = 0.0;
            continue;
        }
        const double2 xy = reinterpret_cast<const double2 *>(A.o_xy)[ob + o];
        CamAux ca;
        load_aux_local(A.aux, cam, ca);  // (local rotation tangent: see load_aux_local)
        ObsLin Ln;
        linearize(ca, X0, X1, X2, xy.x, xy.y, A.loss, A.inv_fs2, floor_now, Ln);
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            const double wj0 = Ln.wh[0] * Ln.Jc[0][i], wj1 = Ln.wh[1] * Ln.Jc[1][i];
            const double w0 = wj0 * Ln.Jp[0][0] + wj1 * Ln.Jp[1][0];
            const double w1 = wj0 * Ln.Jp[0][1] + wj1 * Ln.Jp[1][1];
            const double w2 = wj0 * Ln.Jp[0][2] + wj1 * Ln.Jp[1][2];
            const double t0 = w0 * i00, t1 = w0 * i10 + w1 * i11, t2 = w0 * i20 + w1 * i21 + w2 * i22;
            tp[i * 3] = t0; tp[i * 3 + 1] = t1; tp[i * 3 + 2] = t2;
            if (D.cam_slab) D.cam_slab[(size_t)(D.n_cl + (ob - H.hv_obs[0]) + o) * 33 + 27 + i] = -(t0 * u0 + t1 * u1 + t2 * u2);
            else atomicAdd(&A.gred[(size_t)f * 6 + i], -(t0 * u0 + t1 * u1 + t2 * u2));
        }
    }
    __threadfence();
    __syncthreads();
    // ---- pass 3: S -= T_a T_c^T for every pair a <= c of the point's observations -------------------
    const int *slots = H.hv_pair_slot + H.hv_pair[hp];
    for (int a = 0; a < L; ++a) {
        const long long row = (long long)a * L - ((long long)a * (a - 1)) / 2;
        const double *ta = &H.hv_t[tb + (long long)a * 18];
        for (int c = a + tid; c < L; c += 256) {
            const int sw = slots[row + (c - a)];
            if (sw < 0) continue;
            const double *tc = &H.hv_t[tb + (long long)c * 18];
            double *d = &A.S[(size_t)(sw & HV_SLOT_MASK) * 36];
            const bool tr = (sw & HV_SLOT_TRANSPOSED) != 0, twice = (sw & HV_SLOT_TWICE) != 0;
            double tcr[18];
#pragma unroll
            for (int k = 0; k < 18; ++k) tcr[k] = tc[k];
            if (D.blk_slab) {
                // the pair's block as it is (-T_a T_c^T); the contribution table carries the transposition / the second,
                // transposed use of a camera that observes the point twice
                double *row_out = &D.blk_slab[(size_t)(D.n_raw + D.n_lg + H.hv_pair[hp] + row + (c - a)) * 36];
#pragma unroll
                for (int r = 0; r < 6; ++r) {
                    const double a0 = ta[r * 3], a1 = ta[r * 3 + 1], a2 = ta[r * 3 + 2];
#pragma unroll
                    for (int q = 0; q < 6; ++q) row_out[r * 6 + q] = -(a0 * tcr[q * 3] + a1 * tcr[q * 3 + 1] + a2 * tcr[q * 3 + 2]);
                }
                continue;
            }
#pragma unroll
            for (int r = 0; r < 6; ++r) {
                const double a0 = ta[r * 3], a1 = ta[r * 3 + 1], a2 = ta[r * 3 + 2];
#pragma unroll
                for (int q = 0; q < 6; ++q) {
                    const double v = -(a0 * tcr[q * 3] + a1 * tcr[q * 3 + 1] + a2 * tcr[q * 3 + 2]);
                    if (twice) {  // one camera observing the point twice: both orders land in its diagonal block
                        atomicAdd(&d[r * 6 + q], v);
                        atomicAdd(&d[q * 6 + r], v);
                    } else {
                        atomicAdd(&d[tr ? q * 6 + r : r * 6 + q], v);
                    }
                }
            }
        }
    }
}

// k_backsub for a heavy point: row n_chunks + heavy index of ck_part
__global__ __launch_bounds__(256) void k_backsub_heavy(const BacksubArgs A, const HeavyArgs H) {
    __shared__ double sRed[4];
    const int hp = blockIdx.x, tid = threadIdx.x;
    const long long ob = H.hv_obs[hp];
    const int L = (int)(H.hv_obs[hp + 1] - ob);
    const size_t p = (size_t)(H.pt0 + hp);
    const double floor_now = A.state->floor;
    const double X0 = A.points[3 * p], X1 = A.points[3 * p + 1], X2 = A.points[3 * p + 2];
    double s0 = 0.0, s1 = 0.0, s2 = 0.0;
    for (int o = tid; o < L; o += 256) {
        const int cam = A.o_cam[ob + o];
        const int f = A.cam_free[cam];
        if (f < 0) continue;
        const double2 xy = reinterpret_cast<const double2 *>(A.o_xy)[ob + o];
        CamAux ca;
        load_aux(A.aux, cam, ca);
        ObsLin Ln;
        linearize(ca, X0, X1, X2, xy.x, xy.y, A.loss, A.inv_fs2, floor_now, Ln);
        double a0 = 0.0, a1 = 0.0;
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            const double d = A.dx_rvec[(size_t)f * 6 + i];
            a0 += Ln.Jc[0][i] * d; a1 += Ln.Jc[1][i] * d;
        }
        const double wa0 = Ln.wh[0] * a0, wa1 = Ln.wh[1] * a1;
        s0 += Ln.Jp[0][0] * wa0 + Ln.Jp[1][0] * wa1;
        s1 += Ln.Jp[0][1] * wa0 + Ln.Jp[1][1] * wa1;
        s2 += Ln.Jp[0][2] * wa0 + Ln.Jp[1][2] * wa1;
    }
    s0 = block_sum256(s0, sRed);
    s1 = block_sum256(s1, sRed);
    s2 = block_sum256(s2, sRed);
    const double *li = &A.pt_linv[p * 6], *uu = &A.pt_u[p * 3];
    const double i00 = li[0], i10 = li[1], i11 = li[2], i20 = li[3], i21 = li[4], i22 = li[5];
    const double y0 = uu[0] + i00 * s0;
    const double y1 = uu[1] + i10 * s0 + i11 * s1;
    const double y2 = uu[2] + i20 * s0 + i21 * s1 + i22 * s2;
    const double d0 = -(i00 * y0 + i10 * y1 + i20 * y2);
    const double d1 = -(i11 * y1 + i21 * y2);
    const double d2 = -(i22 * y2);
    if (tid == 0) {
        A.points_t[3 * p] = X0 + d0; A.points_t[3 * p + 1] = X1 + d1; A.points_t[3 * p + 2] = X2 + d2;
    }
    double pred = 0.0, ft = 0.0;
    for (int o = tid; o < L; o += 256) {
        const int cam = A.o_cam[ob + o];
        const int f = A.cam_free[cam];
        const double2 xy = reinterpret_cast<const double2 *>(A.o_xy)[ob + o];
        CamAux ca;
        load_aux(A.aux, cam, ca);
        ObsLin Ln;
        linearize(ca, X0, X1, X2, xy.x, xy.y, A.loss, A.inv_fs2, floor_now, Ln);
        double a0 = 0.0, a1 = 0.0;
        if (f >= 0) {
#pragma unroll
            for (int i = 0; i < 6; ++i) {
                const double d = A.dx_rvec[(size_t)f * 6 + i];
                a0 += Ln.Jc[0][i] * d; a1 += Ln.Jc[1][i] * d;
            }
        }
        const double jd0 = a0 + Ln.Jp[0][0] * d0 + Ln.Jp[0][1] * d1 + Ln.Jp[0][2] * d2;
        const double jd1 = a1 + Ln.Jp[1][0] * d0 + Ln.Jp[1][1] * d1 + Ln.Jp[1][2] * d2;
        pred -= jd0 * (Ln.w[0] * Ln.r[0] + 0.5 * Ln.wh[0] * jd0) + jd1 * (Ln.w[1] * Ln.r[1] + 0.5 * Ln.wh[1] * jd1);
        double u, v;
        project_only(A.aux_t, cam, X0 + d0, X1 + d1, X2 + d2, u, v);
        const double e0 = u - xy.x, e1 = v - xy.y;
        double rho0, rho1, wd;
        loss_eval(A.loss, e0 * e0 * A.inv_fs2, rho0, wd);
        loss_eval(A.loss, e1 * e1 * A.inv_fs2, rho1, wd);
        ft += A.half_fs2 * (rho0 + rho1);
    }
    pred = block_sum256(pred, sRed);
    ft = block_sum256(ft, sRed);
    if (tid == 0) {
        double *o = &A.ck_part[(size_t)(H.n_chunks + hp) * 4];
        o[0] = pred; o[1] = ft; o[2] = d0 * d0 + d1 * d1 + d2 * d2; o[3] = X0 * X0 + X1 * X1 + X2 * X2;
    }
}

}  // namespace ba
