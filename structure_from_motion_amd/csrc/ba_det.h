// The deterministic (atomic-free) build pass: BaOptions.deterministic = 1.
//
// The reference's solve is single-threaded scipy: the same inputs give the same bits (BundleAdjustment.py:210-220).
// k_build / k_backsub are not: their per-camera and per-point sums are fp64 LDS atomics (the order in which the four
// waves of a workgroup reach a tile varies from run to run) and their flush is fp64 atomics in HBM (the order in which
// the chunks reach a block of S varies).  Here NO floating-point sum has an order that depends on timing:
//   * k_build_det: the chunk kernel with every sum gathered by ONE owner lane in a fixed order -- per point over its
//     observations (consecutive lanes), per local camera over the chunk's observations in lane order -- and every
//     partial result stored, with plain stores, into a slab row of its own: one row of 33 doubles per (chunk, local
//     camera), one row of 36 per (chunk, local camera pair) and per later-round group;
//   * k_build_heavy with a slab (ba_heavy.h): the same for the heavy points (rows per observation and per pair);
//   * k_det_reduce: each element of U, g_c, g_red and S = the sum of its slab rows in the order of the host-built
//     contribution table (DetTables, ba_symbolic.h) -- a fixed-order segmented sum that also writes every element of
//     the reduced system (no memset of S);
//   * k_backsub_det: the back-substitution with the per-point sums gathered the same way.
// The structure is analysed for it (analyze_structure(..., deterministic)): no chunk has more cameras than local slots.
// Per observation the arithmetic is that of the atomic kernels (same linearize, same Schur phase); only the ORDER of
// the sums differs, so the two modes agree to rounding and each is checked against the oracle by the same tests.
#pragma once
#include "ba_heavy.h"
#include "ba_kernels.h"

namespace ba {

constexpr int DET_CS = 17;   // staged per observation for the camera gather: Jc (2 x 6), wh (2), w r (2), one pad

template <int LOSS>
__global__ __launch_bounds__(CHUNK_OBS) void k_build_det(const BuildArgs A, const DetArgs D) {
    __shared__ __attribute__((aligned(16))) double sT[CHUNK_OBS * 18];
    __shared__ __attribute__((aligned(16))) double sP[CHUNK_PTS * 9];    // per point: L^-1 (6), u (3)
    __shared__ double sRed[8];
    __shared__ int sStart[CHUNK_PTS + 1];                                // first lane of each point's observations
    __shared__ unsigned char sLp[CHUNK_OBS];
    __shared__ unsigned char sLc[CHUNK_OBS];                             // local camera of each observation; 255 = fixed camera
    __shared__ int sPg[CHUNK_OBS / 2 + 2];
    __shared__ unsigned long long sMask[CHUNK_CAMS_WIDE * 4];            // per (local camera, wave): the lanes that observe it
    static_assert(CHUNK_OBS * DET_CS <= CHUNK_OBS * 18 && CHUNK_CAMS_WIDE < 255 && CHUNK_OBS == 256, "staging fits the T buffer");

    const int ck = blockIdx.x, tid = threadIdx.x;
    const int ob = A.ck_obs[ck], nobs = A.ck_obs[ck + 1] - ob;
    const int pb = A.ck_pt[ck], npts = A.ck_pt[ck + 1] - pb;
    const int cb = A.ck_cam[ck], ncam = A.ck_cam[ck + 1] - cb;
    const int qb = A.ck_pair[ck], npair = A.ck_pair[ck + 1] - qb;
    const double lam = A.state->lambda, floor_now = A.state->floor;
    const bool act = tid < nobs;
    const int oo = ob + (act ? tid : 0);
    const bool has_pg = tid <= npair && tid <= CHUNK_OBS / 2;
    const int st_pg = A.ck_pgrp[qb + ck + (has_pg ? tid : 0)];
    const int n_groups = A.ck_pgrp[qb + ck + npair];
    const long long peb = A.ck_pe[ck];
    const int rounds = (int)((A.ck_pe[ck + 1] - peb) / ((CHUNK_OBS / 2) * PAIR_RUN));
    const int half = tid & 1;
    const int cam = A.o_cam[oo];
    const int p = A.o_pt[oo];
    const int p_prev = A.o_pt[oo > ob ? oo - 1 : oo];
    const int lc_ld = A.o_lcam[oo];
    const double2 xy = reinterpret_cast<const double2 *>(A.o_xy)[oo];
    CamAux ca;
    load_aux_local(A.aux, cam, ca);
    const double X0 = A.points[3 * (size_t)p], X1 = A.points[3 * (size_t)p + 1], X2 = A.points[3 * (size_t)p + 2];
    if (has_pg) sPg[tid] = st_pg;

    // ---- phase 1: one observation per lane; its point-block terms are staged for the per-point gather -------------
    double Wt[9], Yr[3], cs[16];
    int lp = 0, lc = -2;
    double cost_loc = 0.0;
    if (act) {
        lp = p - pb;
        lc = lc_ld;
        sLp[tid] = (unsigned char)lp;
        sLc[tid] = (unsigned char)(lc >= 0 ? lc : 255);
        if (tid == 0 || p_prev != p) sStart[lp] = tid;   // (observations are sorted by point: one head lane per point)
        ObsLin L;
        linearize(ca, X0, X1, X2, xy.x, xy.y, LOSS >= 0 ? LOSS : A.loss, A.inv_fs2, floor_now, L);
        cost_loc = A.half_fs2 * L.rho_sum;
        Yr[0] = L.Y[0]; Yr[1] = L.Y[1]; Yr[2] = L.Y[2];
        double wJp[2][3];
#pragma unroll
        for (int j = 0; j < 3; ++j) { wJp[0][j] = L.wh[0] * L.Jp[0][j]; wJp[1][j] = L.wh[1] * L.Jp[1][j]; }
        const double wr0 = L.w[0] * L.r[0], wr1 = L.w[1] * L.r[1];
        double *vs = &sT[tid * 9];
        vs[0] = wJp[0][0] * L.Jp[0][0] + wJp[1][0] * L.Jp[1][0];
        vs[1] = wJp[0][0] * L.Jp[0][1] + wJp[1][0] * L.Jp[1][1];
        vs[2] = wJp[0][0] * L.Jp[0][2] + wJp[1][0] * L.Jp[1][2];
        vs[3] = wJp[0][1] * L.Jp[0][1] + wJp[1][1] * L.Jp[1][1];
        vs[4] = wJp[0][1] * L.Jp[0][2] + wJp[1][1] * L.Jp[1][2];
        vs[5] = wJp[0][2] * L.Jp[0][2] + wJp[1][2] * L.Jp[1][2];
        vs[6] = L.Jp[0][0] * wr0 + L.Jp[1][0] * wr1;
        vs[7] = L.Jp[0][1] * wr0 + L.Jp[1][1] * wr1;
        vs[8] = L.Jp[0][2] * wr0 + L.Jp[1][2] * wr1;
        if (lc >= 0) {  // free camera (in this structure every free camera of a chunk has a local slot)
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int j = 0; j < 3; ++j)
                    Wt[i * 3 + j] = (L.wh[0] * L.Jc[0][3 + i]) * L.Jp[0][j] + (L.wh[1] * L.Jc[1][3 + i]) * L.Jp[1][j];
#pragma unroll
            for (int i = 0; i < 6; ++i) { cs[i] = L.Jc[0][i]; cs[6 + i] = L.Jc[1][i]; }
            cs[12] = L.wh[0]; cs[13] = L.wh[1]; cs[14] = wr0; cs[15] = wr1;
        }
    }
    if (tid == 0) sStart[npts] = nobs;
    // which lanes of this wave observe local camera l (lane order = observation order: the gathers below walk the set
    // bits from the lowest up, wave 0 first -- a fixed order)
    for (int l = 0; l < ncam; ++l) {
        const unsigned long long m = __ballot(act && lc == l);
        if ((tid & 63) == 0) sMask[l * 4 + (tid >> 6)] = m;
    }
    lds_barrier();

    // ---- phase 1b: one point per lane: V, g_p = the sum over its observations in lane order; damp, factor ---------
    double gmax_loc = 0.0;
    if (tid < npts) {
        double v[9];
#pragma unroll
        for (int k = 0; k < 9; ++k) v[k] = 0.0;
        const int o1 = sStart[tid + 1];
        for (int o = sStart[tid]; o < o1; ++o) {
#pragma unroll
            for (int k = 0; k < 9; ++k) v[k] += lds_f64(&sT[o * 9 + k]);
        }
        const double v00 = v[0], v01 = v[1], v02 = v[2], v11 = v[3], v12 = v[4], v22 = v[5];
        const double g0 = v[6], g1 = v[7], g2 = v[8];
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
        const double i00 = rsqrt_nr(a00);
        const double l10 = v01 * i00, l20 = v02 * i00;
        const double i11 = rsqrt_nr(fmax(a11 - l10 * l10, 1e-15 * a11));
        const double l21 = (v12 - l20 * l10) * i11;
        const double i22 = rsqrt_nr(fmax(a22 - l20 * l20 - l21 * l21, 1e-15 * a22));
        const double i10 = -l10 * i00 * i11;
        const double i21 = -l21 * i11 * i22;
        const double i20 = -(l20 * i00 + l21 * i10) * i22;
        const double u0 = i00 * g0, u1 = i10 * g0 + i11 * g1, u2 = i20 * g0 + i21 * g1 + i22 * g2;
        double *vp = &sP[tid * 9];
        vp[0] = i00; vp[1] = i10; vp[2] = i11; vp[3] = i20; vp[4] = i21; vp[5] = i22;
        vp[6] = u0; vp[7] = u1; vp[8] = u2;
        double *gl = &A.pt_linv[(size_t)(pb + tid) * 6];
        gl[0] = i00; gl[1] = i10; gl[2] = i11; gl[3] = i20; gl[4] = i21; gl[5] = i22;
        double *gu = &A.pt_u[(size_t)(pb + tid) * 3];
        gu[0] = u0; gu[1] = u1; gu[2] = u2;
    }
    {
        const double csum = wave_sum_dpp(cost_loc), gm = wave_max_dpp(gmax_loc);
        if ((tid & 63) == 0) { sRed[tid >> 6] = csum; sRed[4 + (tid >> 6)] = gm; }
    }
    lds_barrier();   // the staged point terms are consumed; sP and sRed are complete
    if (tid == 0) {
        A.ck_cost[ck] = (sRed[0] + sRed[1]) + (sRed[2] + sRed[3]);
        A.ck_gmax[ck] = fmax(fmax(sRed[4], sRed[5]), fmax(sRed[6], sRed[7]));
    }

    // ---- phase 1c: per-camera U, g_c gathered by ONE lane per (local camera, row) over the chunk's observations ----
    if (act && lc >= 0) {
        double *c = &sT[tid * DET_CS];
#pragma unroll
        for (int k = 0; k < 16; ++k) c[k] = cs[k];
    }
    lds_barrier();
    // item = (local camera, row of U, wave whose lanes it walks): four neighbouring lanes per (camera, row), their
    // partial sums joined by two DPP steps -- ((w0 + w1) + (w2 + w3)), the same order every time
    for (int base = 0; base < ncam * 24; base += CHUNK_OBS) {
        const int item = base + tid;
        const bool on = item < ncam * 24;
        const int l = on ? item / 24 : 0, rem = item - l * 24, row = on ? rem >> 2 : 0, w = rem & 3;
        double acc[6], g = 0.0;
#pragma unroll
        for (int j = 0; j < 6; ++j) acc[j] = 0.0;
        unsigned long long m = on ? sMask[l * 4 + w] : 0ull;
        while (m) {
            const int o = w * 64 + __builtin_ctzll(m);
            m &= m - 1;
            const double *c = &sT[o * DET_CS];
            double cc[16];   // (single 8-byte reads: the row stride is odd, lds_f64, ba_device.h)
#pragma unroll
            for (int k = 0; k < 16; ++k) cc[k] = lds_f64(c + k);
            const double cr0 = lds_f64(c + row), cr1 = lds_f64(c + 6 + row);
            const double a0 = cc[12] * cr0, a1 = cc[13] * cr1;
#pragma unroll
            for (int j = 0; j < 6; ++j) acc[j] += a0 * cc[j] + a1 * cc[6 + j];
            g += cr0 * cc[14] + cr1 * cc[15];
        }
#pragma unroll
        for (int j = 0; j < 6; ++j) { acc[j] += dpp_f64<0xB1>(acc[j]); acc[j] += dpp_f64<0x4E>(acc[j]); }
        g += dpp_f64<0xB1>(g);
        g += dpp_f64<0x4E>(g);
        if (on && w == 0) {
            double *out = &D.cam_slab[(size_t)(cb + l) * 33];
#pragma unroll
            for (int j = 0; j < 6; ++j)
                if (j >= row) out[ut6(row, j)] = acc[j];
            out[21 + row] = g;
        }
    }
    lds_barrier();   // every lane is done with the camera staging

    // ---- phase 2b: T' = Wt L^-T and Y into LDS ----------------------------------------------------------------------
    if (act && lc >= 0) {
        const double *vp = &sP[lp * 9];
        const double i00 = lds_f64(vp), i10 = lds_f64(vp + 1), i11 = lds_f64(vp + 2), i20 = lds_f64(vp + 3), i21 = lds_f64(vp + 4), i22 = lds_f64(vp + 5);
        double *tp = &sT[tid * T_STRIDE];
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            tp[i * 3] = Wt[i * 3] * i00;
            tp[i * 3 + 1] = Wt[i * 3] * i10 + Wt[i * 3 + 1] * i11;
            tp[i * 3 + 2] = Wt[i * 3] * i20 + Wt[i * 3 + 1] * i21 + Wt[i * 3 + 2] * i22;
        }
        tp[9] = Yr[0]; tp[10] = Yr[1]; tp[11] = Yr[2];
    }
    lds_barrier();
    // reduced gradient: -sum T_i u_p = -[Y x w; w], w = T' u, per local camera over its observations in lane order
    for (int base = 0; base < ncam * 4; base += CHUNK_OBS) {
        const int item = base + tid;
        const bool on = item < ncam * 4;
        const int l = on ? item >> 2 : 0, w = item & 3;
        double g[6];
#pragma unroll
        for (int k = 0; k < 6; ++k) g[k] = 0.0;
        unsigned long long m = on ? sMask[l * 4 + w] : 0ull;
        while (m) {
            const int o = w * 64 + __builtin_ctzll(m);
            m &= m - 1;
            const double *t = &sT[o * T_STRIDE], *u = &sP[(int)sLp[o] * 9 + 6];
            double tt[12], uu[3];
#pragma unroll
            for (int k = 0; k < 12; ++k) tt[k] = lds_f64(t + k);
#pragma unroll
            for (int k = 0; k < 3; ++k) uu[k] = lds_f64(u + k);
            const double w0 = tt[0] * uu[0] + tt[1] * uu[1] + tt[2] * uu[2], w1 = tt[3] * uu[0] + tt[4] * uu[1] + tt[5] * uu[2],
                         w2 = tt[6] * uu[0] + tt[7] * uu[1] + tt[8] * uu[2];
            const double y0 = tt[9], y1 = tt[10], y2 = tt[11];
            g[0] += -(y1 * w2 - y2 * w1);
            g[1] += -(y2 * w0 - y0 * w2);
            g[2] += -(y0 * w1 - y1 * w0);
            g[3] += -w0; g[4] += -w1; g[5] += -w2;
        }
#pragma unroll
        for (int k = 0; k < 6; ++k) { g[k] += dpp_f64<0xB1>(g[k]); g[k] += dpp_f64<0x4E>(g[k]); }
        if (on && w == 0) {
            double *out = &D.cam_slab[(size_t)(cb + l) * 33 + 27];
#pragma unroll
            for (int k = 0; k < 6; ++k) out[k] = g[k];
        }
    }

    // ---- phase 3: Schur-complement blocks (k_build's phase 3 without the reduced-gradient part) -----------------------
    {
        double acc[36];
        for (int round = (rounds > 0 ? rounds : 1) - 1; round >= 0; --round) {
            unsigned en[PAIR_RUN / 2];
            {
                const uint4 e0 = reinterpret_cast<const uint4 *>(A.pair_entries + peb)[(size_t)round * CHUNK_OBS + tid];
                const bool any = rounds > 0;
                en[0] = any ? e0.x : PAIR_NULL; en[1] = any ? e0.y : PAIR_NULL; en[2] = any ? e0.z : PAIR_NULL; en[3] = any ? e0.w : PAIR_NULL;
            }
#pragma unroll
            for (int k = 0; k < 36; ++k) acc[k] = 0.0;
#pragma unroll
            for (int e = 0; e < PAIR_RUN / 2; ++e) {
                const unsigned ent = en[e];
                const unsigned ec = ent == PAIR_NULL ? 0u : ent;
                const int oi = ec & 255u, oj = (ec >> 8) & 255u;
                const double *ti = &sT[oi * T_STRIDE], *tj = &sT[oj * T_STRIDE];
                const bool real = ent != PAIR_NULL;
                double Tj[9], Yj[3], Yi[3];
#pragma unroll
                for (int k = 0; k < 9; ++k) Tj[k] = lds_f64(tj + k);   // (single 8-byte reads: lds_f64, ba_device.h)
#pragma unroll
                for (int k = 0; k < 3; ++k) { Yj[k] = lds_f64(tj + 9 + k); Yi[k] = lds_f64(ti + 9 + k); }
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
                        const int s1 = (r + 1) % 3, s2 = (r + 2) % 3;
#pragma unroll
                        for (int c = 0; c < 6; ++c) {
                            acc[(3 + r) * 6 + c] += B[c];
                            acc[s1 * 6 + c] += Yi[s2] * B[c];
                            acc[s2 * 6 + c] -= Yi[s1] * B[c];
                        }
                    }
                }
            }
#pragma unroll
            for (int k = 0; k < 36; ++k) acc[k] += dpp_f64<0xB1>(acc[k]);   // (a + b = b + a: both lanes hold the same sum)
            if (round > 0) {
                // later rounds: the group's partial block goes to a slab row of its own, each lane its half
                const int g = round * (CHUNK_OBS / 2) + (tid >> 1);
                if (g < n_groups) {
                    double *d = &D.blk_slab[(size_t)(D.n_raw + D.ck_lg[ck] + (g - CHUNK_OBS / 2)) * 36];
                    if (half == 0) {
#pragma unroll
                        for (int k = 0; k < 18; ++k) d[k] = -acc[k];
                    } else {
#pragma unroll
                        for (int k = 18; k < 36; ++k) d[k] = -acc[k];
                    }
                }
            }
        }
        lds_barrier();  // every lane is done reading T
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

    // ---- phase 4: per local pair the sum of its first-round groups, in group order, to the pair's slab row -----------
    {
        const int np1 = npair < CHUNK_OBS / 2 ? npair : CHUNK_OBS / 2;
        for (int i = tid; i < np1 * 36; i += CHUNK_OBS) {
            const int l = i / 36, k = i - l * 36;
            const int g0 = sPg[l];
            int g1 = sPg[l + 1];
            if (g1 > CHUNK_OBS / 2) g1 = CHUNK_OBS / 2;
            double v = 0.0;
            for (int g = g0; g < g1; ++g) v += sT[g * 36 + k];
            if (g1 > g0) D.blk_slab[(size_t)(qb + l) * 36 + k] = v;
        }
    }
}

// Each accumulator of the reduced system = the sum of its slab rows in table order.  One thread per element of S (36
// per upper block), then per element of (U, g_c, g_red part) (33 per free camera); writes EVERY element.
__global__ __launch_bounds__(256) void k_det_reduce(int nf, int n_upper, const int *__restrict__ c_ptr, const int *__restrict__ c_idx,
                                                    const int *__restrict__ s_ptr, const unsigned *__restrict__ s_idx,
                                                    const double *__restrict__ cam_slab, const double *__restrict__ blk_slab,
                                                    double *__restrict__ S, double *__restrict__ U, double *__restrict__ gc,
                                                    double *__restrict__ gred) {
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long nS = (long long)n_upper * 36;
    if (t < nS) {
        const int slot = (int)(t / 36), k = (int)(t - (long long)slot * 36), kt = (k % 6) * 6 + k / 6;
        double v = 0.0;
        const int e0 = s_ptr[slot], e1 = s_ptr[slot + 1];
        for (int e = e0; e < e1; e += 4) {   // four rows per trip, their loads issued together; added in table order
            unsigned r[4];
            double x[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) r[q] = s_idx[e + q < e1 ? e + q : e0];
#pragma unroll
            for (int q = 0; q < 4; ++q) x[q] = blk_slab[(size_t)(r[q] & 0x7fffffffu) * 36 + ((r[q] >> 31) ? kt : k)];
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (e + q < e1) v += x[q];
        }
        S[t] = v;
        return;
    }
    const long long t2 = t - nS;
    if (t2 >= (long long)nf * 33) return;
    const int f = (int)(t2 / 33), k = (int)(t2 - (long long)f * 33);
    double v = 0.0;
    const int e0 = c_ptr[f], e1 = c_ptr[f + 1];
    for (int e = e0; e < e1; e += 4) {
        int r[4];
        double x[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) r[q] = c_idx[e + q < e1 ? e + q : e0];
#pragma unroll
        for (int q = 0; q < 4; ++q) x[q] = cam_slab[(size_t)r[q] * 33 + k];
#pragma unroll
        for (int q = 0; q < 4; ++q)
            if (e + q < e1) v += x[q];
    }
    if (k < 21) U[(size_t)f * 21 + k] = v;
    else if (k < 27) gc[(size_t)f * 6 + (k - 21)] = v;
    else gred[(size_t)f * 6 + (k - 27)] = v;
}

// k_backsub with the per-point sums gathered in lane order instead of added by LDS atomics; everything else -- the
// arithmetic, the fixed-order partial sums per chunk -- is k_backsub's.
template <int LOSS>
__global__ __launch_bounds__(CHUNK_OBS) void k_backsub_det(const BacksubArgs A) {
    __shared__ double sS[CHUNK_OBS * 3];
    __shared__ double sP[CHUNK_PTS * 3];
    __shared__ double sRed[4][4];
    __shared__ int sStart[CHUNK_PTS + 1];
    const int ck = blockIdx.x, tid = threadIdx.x;
    const int ob = A.ck_obs[ck], nobs = A.ck_obs[ck + 1] - ob;
    const int pb = A.ck_pt[ck], npts = A.ck_pt[ck + 1] - pb;
    const int loss = LOSS >= 0 ? LOSS : A.loss;
    const bool act = tid < nobs;
    const int o = ob + (act ? tid : 0);
    const int cam = A.o_cam[o];
    const int p = A.o_pt[o];
    const int p_prev = A.o_pt[o > ob ? o - 1 : o];
    const double2 xy = reinterpret_cast<const double2 *>(A.o_xy)[o];
    CamAux ca;
    load_aux_local(A.aux, cam, ca);
    const double X0 = A.points[3 * (size_t)p], X1 = A.points[3 * (size_t)p + 1], X2 = A.points[3 * (size_t)p + 2];
    const int f = A.cam_free[cam];
    const double floor_now = A.state->floor;
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
            for (int j = 0; j < 3; ++j) sS[tid * 3 + j] = Jp[0][j] * wa0 + Jp[1][j] * wa1;
            if (tid == 0 || p_prev != p) sStart[lp] = tid;
        }
    }
    if (tid == 0) sStart[npts] = nobs;
    lds_barrier();
    double dp2 = 0.0, xp2 = 0.0;
    if (tid < npts) {
        const size_t q = (size_t)(pb + tid);
        double s0 = 0.0, s1 = 0.0, s2 = 0.0;
        const int o1 = sStart[tid + 1];
        for (int k = sStart[tid]; k < o1; ++k) { s0 += sS[k * 3]; s1 += sS[k * 3 + 1]; s2 += sS[k * 3 + 2]; }
        const double *li = &A.pt_linv[q * 6], *uu = &A.pt_u[q * 3];
        const double i00 = li[0], i10 = li[1], i11 = li[2], i20 = li[3], i21 = li[4], i22 = li[5];
        const double y0 = uu[0] + i00 * s0;
        const double y1 = uu[1] + i10 * s0 + i11 * s1;
        const double y2 = uu[2] + i20 * s0 + i21 * s1 + i22 * s2;
        const double d0 = -(i00 * y0 + i10 * y1 + i20 * y2);
        const double d1 = -(i11 * y1 + i21 * y2);
        const double d2 = -(i22 * y2);
        sP[tid * 3] = d0; sP[tid * 3 + 1] = d1; sP[tid * 3 + 2] = d2;
        const double p0 = A.points[3 * q], p1 = A.points[3 * q + 1], p2 = A.points[3 * q + 2];
        A.points_t[3 * q] = p0 + d0; A.points_t[3 * q + 1] = p1 + d1; A.points_t[3 * q + 2] = p2 + d2;
        dp2 = d0 * d0 + d1 * d1 + d2 * d2;
        xp2 = p0 * p0 + p1 * p1 + p2 * p2;
    }
    lds_barrier();
    double pred = 0.0, ft = 0.0;
    if (act) {
        const double d0 = sP[lp * 3], d1 = sP[lp * 3 + 1], d2 = sP[lp * 3 + 2];
        const double jd0 = a0 + Jp[0][0] * d0 + Jp[0][1] * d1 + Jp[0][2] * d2;
        const double jd1 = a1 + Jp[1][0] * d0 + Jp[1][1] * d1 + Jp[1][2] * d2;
        pred = -(jd0 * (w0 * r0 + 0.5 * h0 * jd0) + jd1 * (w1 * r1 + 0.5 * h1 * jd1));
        double u, v;
        project_only(A.aux_t, cam, X0 + d0, X1 + d1, X2 + d2, u, v);
        const double e0 = u - uo, e1 = v - vo;
        double rho0, rho1, wd;
        loss_eval(loss, e0 * e0 * A.inv_fs2, rho0, wd);
        loss_eval(loss, e1 * e1 * A.inv_fs2, rho1, wd);
        ft = A.half_fs2 * (rho0 + rho1);
    }
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

}  // namespace ba
