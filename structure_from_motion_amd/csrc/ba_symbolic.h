// Host-side symbolic analysis of a bundle-adjustment problem (plain C++, no HIP).
//
// The reference builds one boolean Jacobian sparsity matrix per solve with a Python loop
// (BundleAdjustment.py:154-198).  The engine needs a different structure, built once per
// problem and reused by every LM iteration:
//   * points re-ordered so that neighbouring points see the same cameras, observations sorted
//     by (point, camera);
//   * "chunks": runs of consecutive points (<= CHUNK_OBS observations) that one workgroup
//     linearises and eliminates, with the chunk's cameras and camera pairs numbered locally so
//     the per-camera blocks and Schur-complement blocks can be accumulated in LDS;
//   * the block-sparse pattern of the reduced camera system (upper triangle + full CSR view).
#pragma once
#include <cstdint>
#include <vector>

namespace ba {

constexpr int CHUNK_OBS = 256;   // observations per chunk = threads per workgroup
constexpr int CHUNK_PTS = 100;   // points per chunk (LDS slots)
constexpr int CHUNK_CAMS = 28;   // cameras with a full LDS tile per chunk (U, g_c in four replicas, reduced gradient)
constexpr int CHUNK_CAMS_WIDE = 112;  // local camera slots per chunk: a chunk with more than CHUNK_CAMS cameras is "wide" -- one
                                      // replica per camera, and the cameras from slot CHUNK_CAMS on flush U, g_c right after phase 1
constexpr int PAIR_RUN = 8;      // pair entries per group = per lane pair of the build kernel (see Symbolic::pair_entries)
constexpr uint32_t PAIR_NULL = 0xFFFFFFFFu;  // padding entry (local pair ids stay below 0xFFFF: at most 256 * 257 / 2 pairs)

// heavy points (tracks longer than CHUNK_OBS): flags on the pair keys (free-camera indices stay below 2^30) and on the
// resolved block slots (fewer than 2^29 blocks)
constexpr int64_t HV_KEY_TRANSPOSED = (int64_t)1 << 62, HV_KEY_TWICE = (int64_t)1 << 61;
constexpr int64_t HV_KEY_MASK = ~(HV_KEY_TRANSPOSED | HV_KEY_TWICE);
constexpr int32_t HV_SLOT_TRANSPOSED = 1 << 30, HV_SLOT_TWICE = 1 << 29, HV_SLOT_MASK = (1 << 29) - 1;

struct Symbolic {
    int n_cams = 0, n_free = 0, n_points = 0, n_pts_obs = 0;
    int n_pts_chunked = 0;           // internal points [0, n_pts_chunked) are covered by chunks,
    int n_heavy = 0;                 // [n_pts_chunked, n_pts_chunked + n_heavy) are heavy, the rest (up to n_pts_obs .. n_points) unobserved
    int64_t n_obs = 0;
    int max_track = 0;
    double mean_track = 0;
    std::vector<int32_t> cam_free;   // [n_cams] free-camera index or -1
    std::vector<int32_t> free_cam;   // [n_free] camera index
    std::vector<int32_t> pt_order;   // internal point -> caller's point index (observed first)
    std::vector<int64_t> obs_order;  // internal observation -> caller's observation index
    std::vector<int32_t> o_cam;      // [n_obs] camera index, internal order
    std::vector<int32_t> o_pt;       // [n_obs] internal point index
    std::vector<int16_t> o_lcam;     // [n_obs] chunk-local camera slot (< CHUNK_CAMS_WIDE), -1 = none (fixed cameras: -2)
    // chunks
    std::vector<int32_t> ck_obs;     // [n_chunks+1]
    std::vector<int32_t> ck_pt;      // [n_chunks+1] (internal point range)
    std::vector<int32_t> ck_cam;     // [n_chunks+1] into ck_cam_list
    std::vector<int32_t> ck_cam_list;   // free-camera index per local slot
    std::vector<int32_t> ck_pair;    // [n_chunks+1] into ck_pair_slot
    std::vector<int32_t> ck_pair_slot;  // upper-triangle block slot per local pair
    std::vector<int64_t> raw_keys;      // (i<<32 | j) per local pair, same order as ck_pair_slot
    std::vector<int64_t> ck_pe;      // [n_chunks+1] into pair_entries
    std::vector<uint32_t> pair_entries; // oi | oj<<8 | local_pair<<16 (oi, oj: observation index inside the chunk), in groups
                                        // of PAIR_RUN words of one pair, PAIR_NULL-padded; per chunk whole rounds of CHUNK_OBS/2 groups
    std::vector<int32_t> ck_pgrp;       // per chunk (n_local_pairs + 1) group offsets, at ck_pair[ck] + ck
    int64_t n_pair_real = 0;            // entries that are not padding
    // heavy points: observation ranges, and per point the triangular list of (a <= c) observation pairs
    std::vector<int64_t> hv_obs;        // [n_heavy+1] internal observation range of each heavy point
    std::vector<int64_t> hv_pair;       // [n_heavy+1] into hv_keys / hv_pair_slot; row a of a point with L observations
                                        // starts at a L - a (a - 1) / 2, entry (a, c) sits at row start + (c - a)
    std::vector<int64_t> hv_keys;       // (min<<32 | max) | flags, or -1 when one of the two cameras is fixed
    std::vector<int32_t> hv_pair_slot;  // block slot | HV_SLOT_* flags, or -1
    // reduced camera system pattern
    std::vector<int64_t> upper_keys; // sorted unique (i<<32 | j), i <= j, free-camera indices
    std::vector<int32_t> diag_slot;  // [n_free]
    int band = 0;                    // half bandwidth of the pattern in blocks: max (j - i)
    std::vector<int32_t> bsr_row;    // [n_free+1] full (both triangles) pattern
    std::vector<int32_t> bsr_col;    // [nnz_full]
    std::vector<int32_t> bsr_src;    // [nnz_full] upper slot; bit 31 set = use the transposed block
    int64_t n_direct_pairs = 0, n_direct_cam_obs = 0;
    int max_chunk_cams = 0, max_chunk_pairs = 0;
    bool deterministic = false;      // analysed for the atomic-free build (see analyze_structure)
    int64_t n_groups = 0;            // groups of pair entries over all chunks (chunk mode)
    // Global per-block entry lists (ba_rows.h): rw_keys holds, in ascending order, every block (i <= j) that has an entry,
    // rw_slot its slot in the pattern, and entries rw_ptr[k] .. rw_ptr[k + 1] of rw_ent the (observation i, observation j)
    // pairs of block k, in point order (internal observation indices; T_i T_j^T goes to the block as stored).
    //   * "rows" mode: the lists of EVERY point -- the Schur complement is accumulated from them instead of inside the chunks,
    //     for co-visibility without structure inside a chunk (hardly two points of a chunk share a camera pair).  Chunks
    //     then only carry points and observations (no camera slots, no pairs).
    //   * chunk mode (not deterministic): the lists of the HEAVY points only (tracks no chunk can hold); their blocks are
    //     added on top of what the chunk kernel accumulated.
    bool rows = false;
    std::vector<int64_t> rw_keys;
    std::vector<int32_t> rw_slot;
    std::vector<int32_t> rw_ent;     // 2 per entry
    std::vector<int64_t> rw_ptr;     // [rw_keys.size() + 1]
};

// Contribution tables of the deterministic (atomic-free) build: every partial result of the build kernels has a slab
// row of its own, written with plain stores, and each accumulator of the reduced system is the sum of its rows in the
// fixed order listed here (ascending row = chunk order, then later-round groups, then heavy points).
//   camera slab, rows of 33 doubles (U 21, g_c 6, reduced-gradient part 6):
//       row ck_cam[ck] + l            local camera l of chunk ck
//       row n_cl + (o - hv_obs[0])    observation o of a heavy point
//   block slab, rows of 36 doubles (one 6x6 partial Schur block, already signed):
//       row ck_pair[ck] + l           local pair l (< CHUNK_OBS / 2) of chunk ck: the sum of its first-round groups
//       row n_raw + ck_lg[ck] + g - CHUNK_OBS / 2     group g >= CHUNK_OBS / 2 of chunk ck (later rounds)
//       row n_raw + n_lg + k          pair k of the heavy points' triangular lists
struct DetTables {
    std::vector<int64_t> ck_lg;      // [n_chunks + 1] first later-round row of each chunk (relative to n_raw)
    int64_t n_cl = 0, n_raw = 0, n_lg = 0;
    int64_t cam_rows = 0, blk_rows = 0;
    std::vector<int32_t> c_ptr;      // [n_free + 1]
    std::vector<int32_t> c_idx;      // camera-slab rows per free camera
    std::vector<int32_t> s_ptr;      // [n_upper + 1]
    std::vector<uint32_t> s_idx;     // block-slab rows per upper block; bit 31: add the row's block TRANSPOSED
};
void build_det_tables(const Symbolic &s, DetTables &t);

// Phase A: ordering + chunking + this problem's own pair keys.  Returns 0 or a BA_ERR_* code.
// deterministic: the structure for the atomic-free build -- no chunk holds more cameras than it has local slots
// (CHUNK_CAMS_WIDE), and a point seen by more cameras than that is a heavy point.
// rows: the "rows" mode of the Schur complement (Symbolic::rows); its kernels have no atomics, so it goes with deterministic as it is.
int analyze_structure(int n_cams, int n_points, int64_t n_obs, const uint8_t *cam_fixed,
                      const int32_t *obs_cam, const int32_t *obs_pt, Symbolic &s, const char **err, bool deterministic = false,
                      bool rows = false);
// AUTO's rule for the rows mode, from a chunk-mode analysis: the chunk-local groups are mostly padding (fewer than three
// real entries per group of eight on average) and the problem is large enough for it to matter.
inline bool wants_rows_mode(const Symbolic &s) {
    return !s.rows && s.n_pair_real >= 200000 && s.n_pair_real < 3 * s.n_groups;
}
// Phase B: final pattern = own keys U extra keys; resolves chunk-local pairs to block slots.
void finalize_pattern(Symbolic &s, const std::vector<int64_t> &extra_keys);

}  // namespace ba
