// Developer: LDS bank conflicts of k_build's Schur phase (first round), simulated on the host from the analysis' entry layout with
// the bank model of ds_read_b64 (MI355X_MICROARCH.md, LDS: two 32-lane groups per wave, bank pair = (8-byte address) mod 32,
// identical addresses broadcast, every further distinct address on a bank pair costs one cycle).
// usage: bank_sim <problem.bin>   (problem.bin: int64 n_cams, n_points, n_obs; u8 cam_fixed[]; i32 obs_cam[]; i32 obs_pt[])
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cstdint>
#include <algorithm>
#include "../structure_from_motion_amd/csrc/ba_symbolic.h"
using namespace ba;
int main(int argc, char **argv) {
    FILE *f = fopen(argv[1], "rb");
    if (!f) return 1;
    int64_t hd[3];
    if (fread(hd, 8, 3, f) != 3) return 1;
    std::vector<uint8_t> fixed(hd[0]);
    std::vector<int32_t> oc(hd[2]), op(hd[2]);
    if (fread(fixed.data(), 1, hd[0], f) != (size_t)hd[0] || fread(oc.data(), 4, hd[2], f) != (size_t)hd[2] || fread(op.data(), 4, hd[2], f) != (size_t)hd[2]) return 1;
    fclose(f);
    Symbolic s;
    const char *msg = "";
    if (analyze_structure((int)hd[0], (int)hd[1], hd[2], fixed.data(), oc.data(), op.data(), s, &msg, false, false) != 0) { fprintf(stderr, "%s\n", msg); return 1; }
    const int T_STRIDE = 13;
    const size_t nck = s.ck_obs.size() - 1;
    long long ideal = 0, cycles = 0, real_ent = 0, slots = 0;
    for (size_t c = 0; c < nck; ++c) {
        const int64_t pe0 = s.ck_pe[c], pe1 = s.ck_pe[c + 1];
        if (pe1 == pe0) continue;
        const int ob = s.ck_obs[c];
        // first round: 256 lanes x 4 words
        for (int wave = 0; wave < 4; ++wave)
            for (int e = 0; e < 4; ++e)
                for (int half = 0; half < 2; ++half) {
                    int oi[32], oj[32], lp[32];
                    bool any = false, isnull[32];
                    static const bool nullfree = getenv("NULLFREE") != nullptr;
                    for (int l = 0; l < 32; ++l) {
                        const int tid = wave * 64 + half * 32 + l;
                        const uint32_t ent = s.pair_entries[pe0 + (size_t)tid * 4 + e];
                        const uint32_t ec = ent == PAIR_NULL ? 0u : ent;   // (k_build: a padding entry reads observation 0)
                        oi[l] = ec & 255u; oj[l] = (ec >> 8) & 255u;
                        lp[l] = s.o_pt[ob + oi[l]] - s.ck_pt[c];
                        isnull[l] = nullfree && ent == PAIR_NULL;
                        if (ent != PAIR_NULL) { any = true; ++real_ent; }
                        ++slots;
                    }
                    if (!any) continue;   // (a wave without work still issues the reads; counted as ideal)
                    auto cost = [&](auto addr) {
                        int best = 0;
                        for (int b = 0; b < 32; ++b) {
                            int a[32], n = 0;
                            for (int l = 0; l < 32; ++l) {
                                if (isnull[l]) continue;
                                const int ad = addr(l);
                                if ((ad & 31) != b) continue;
                                bool seen = false;
                                for (int q = 0; q < n; ++q) seen |= a[q] == ad;
                                if (!seen) a[n++] = ad;
                            }
                            best = std::max(best, n);
                        }
                        return best;
                    };
                    for (int k = 0; k < 12; ++k) {
                        cycles += cost([&](int l) { return oi[l] * T_STRIDE + k; });
                        cycles += cost([&](int l) { return oj[l] * T_STRIDE + k; });
                        ideal += 2;
                    }
                    for (int k = 0; k < 3; ++k) { cycles += cost([&](int l) { return 4608 + lp[l] * 9 + 6 + k; }); ideal += 1; }
                }
    }
    printf("chunks %zu: Schur-phase reads, first round: %lld LDS cycles against %lld without conflicts (x %.2f); entry slots filled %.1f %%\n", nck, cycles, ideal,
           (double)cycles / ideal, 100.0 * real_ent / slots);
    return 0;
}
