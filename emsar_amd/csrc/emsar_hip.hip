// emsar_hip.hip -- MI355X (gfx950 / CDNA4) abundance-estimation core behind include/emsar_hip.h.
//
// Replaces run_MLE_threads() (/root/reference/src/emsar_main.c:446; MLE/Fp/lambdap,
// emsar_functions.c:2946-3126) by an EM on the same segment Poisson likelihood:
//     E-step  w_c = R_c / S_c ,  S_c = sum_t m_ct theta_t        (rows with E_c == 0 are outside F)
//     M-step  theta_t <- theta_t * (sum_c m_ct w_c) / den_t ,    den_t = sum_c m_ct E_c
// and compute_iEUMA / the TPM + iReadcount arithmetic of print_FPKMfinal (emsar_functions.c:3176-3232).
//
// One pass is HBM-bound integer streaming plus FP64 adds: ~2 flop per nonzero -- no MFMA.
// This file: the context, the launch logic (launch_pass, enqueue_cycles, the set solver's driver) and the C ABI.
// Kernels (one translation unit, included below):
//   kernels_tiled.hpp     k_pass_tiled / k_pass_tiled_multi<2>   the hot ones: one workgroup per tile (or pair of tiles) of the
//                         TILED layout, dictionary of theta/acc in LDS, 10-bit ids, per-slice transposed index
//   kernels_csr.hpp       k_pass_csr                             the caller's CSR as it is (layout 1), leftover rows of TILED
//   kernels_vector.hpp    k_update, k_update_p2/p3, k_sq_extrap_ll (SQUAREM extrapolation / acceptance on the device),
//                         k_normalise, k_adj_euma, small reductions
//   kernels_sets.hpp      k_solve_sets                           one workgroup solves one connected set out of LDS
//   collapse.hip          read-level rows -> weighted segments (own translation unit)
#include <hip/hip_runtime.h>

#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "../../include/emsar_hip.h"
#include "layout.hpp"
#include "layout_tiled.hpp"
#include "sets.hpp"
#include "internal.hpp"

#include "kernels_common.hpp"
#include "kernels_csr.hpp"
#include "kernels_tiled.hpp"
#include "kernels_vector.hpp"
#include "kernels_sets.hpp"
#include "kernels_cluster.hpp"

// ==================================================================================================
// context
// ==================================================================================================
constexpr int64_t kPairMinTiles = 2048;   // 256 CUs x 4 resident workgroups x 2 tiles

struct emsar_hip_ctx {
    int device = 0;
    int n_cu = 64;               // compute units of the device (cluster launches: one workgroup per CU at most)
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr, ev2 = nullptr;
    hipStream_t side[3] = {nullptr, nullptr, nullptr};     // the 256- and 512-thread classes of the set solver and the clusters run next to the 64-thread class
    hipEvent_t ev_fork = nullptr, ev_join[3] = {nullptr, nullptr, nullptr};
    std::string err;
    // structure
    bool have_structure = false, have_sample = false;
    int layout = EMSAR_LAYOUT_CSR;
    int64_t n_rows = 0, nnz = 0;
    int32_t n_tx = 0;
    bool ptr64 = false;
    // CSR layout (device)
    void *d_row_ptr = nullptr;   // uint32 or uint64
    int32_t *d_col = nullptr;
    // TILED layout
    emsar::TiledLayout TL;       // host copy keeps slot_row / single_* / left_row (index arrays freed after upload)
    Tile *d_tiles = nullptr;
    Tile *d_utiles = nullptr; int unit_stride = 1;   // emsar::UnitTables
    uint32_t *d_units = nullptr; int64_t n_units = 0;     // units of one or two tiles that share a dictionary (k_pass_tiled_unit)
    uint32_t *d_fwd = nullptr, *d_bwd = nullptr;
    int32_t *d_far = nullptr;
    uint64_t *d_left_ptr = nullptr; int32_t *d_left_col = nullptr; int32_t *d_left_wgt = nullptr; double *d_left_val = nullptr;
    int64_t n_left = 0, n_tiles = 0, n_slots = 0;
    double *d_u = nullptr;       // folded single-tid rows: per-transcript weight sum
    // sample
    bool weighted = false;
    int32_t *d_wgt = nullptr;    // row weights in layout order (0 = row outside F)
    double *d_rowval = nullptr;  // scratch for scatter passes (den, iEUMA)
    double loglik_const = 0.0;   // sum_c R_c log E_c over rows inside F
    // vectors [n_tx]
    double *d_den = nullptr, *d_acc = nullptr;
    double *d_th[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};  // th0 th1 th2 thx thn
    double *d_tmp[3] = {nullptr, nullptr, nullptr};
    int32_t *d_itmp = nullptr;
    Scal *d_scal = nullptr;
    Scal *h_scal = nullptr;      // pinned
    int64_t bytes_formula = 0, bytes_stored = 0;
    int64_t tl_fwd_slots = 0, tl_n_fslices = 0;
    double count_floor = 0.0;    // stopping-rule floor in reads for the current solve (emsar_em_params.count_floor)
    double zero_cut = 0.0;       // emsar_em_params.zero_cut of the current solve
    bool use_graph = true;       // replay check_every cycles of the streaming solve from one hipGraph (EMSAR_HIP_GRAPH=0: launch each kernel)
    int64_t graph_launches = 0;  // of the last solve (debug: EMSAR_HIP_DEBUG)
    bool det = false;            // deterministic mode (emsar_hip_set_deterministic / EMSAR_HIP_DETERMINISTIC): fixed-point sums, kernels_common.hpp
    double fx_mass = 0.0, fx_ll = 0.0;   // its scales for the current sample (upload_sample)
    double *d_sqpart = nullptr;  // per-workgroup partial sums of the SQUAREM vector kernels [4][kSqPart]
    int update_grid = 1024;       // workgroups of k_update (EMSAR_HIP_UPDATE_GRID)
    int sq_grid = 256;           // workgroups of the SQUAREM vector kernels (EMSAR_HIP_SQ_GRID)
    int weighted_unit = 1;       // EMSAR_HIP_WEIGHTED_UNIT: weighted rows on k_pass_tiled_unit -- 1: the plain EM pass, 2: the likelihood passes too, 0: never
    int tiled_multi = 1;         // EMSAR_HIP_TILED_MULTI 1: two tiles per workgroup (k_pass_tiled_multi) above kPairMinTiles tiles, else one
                                 // (k_pass_tiled); 2: always two; 0: always one
    const uint8_t *delta_mask = nullptr;   // d_kind while the streaming solve runs next to resident sets
    // set-resident solver (sets.hpp): host copy of the CSR and of the sample's row weights, built lazily by solve
    std::vector<uint64_t> h_row_ptr;
    std::vector<int32_t> h_col, h_wgt;
    bool sets_ready = false;
    emsar::ResidentSets RS;      // index vectors are freed after the upload, counters stay
    emsar::SetDesc *d_sdesc[emsar::kSetClasses] = {nullptr, nullptr, nullptr};
    SetStat *d_sstat = nullptr; SetStat *h_sstat = nullptr; int64_t n_sstat = 0;
    // workgroup-cluster sets (kernels_cluster.hpp)
    emsar::ClusterDesc *d_cdesc = nullptr; uint32_t *d_cblk = nullptr, *d_crp = nullptr, *d_ccp = nullptr, *d_cpart = nullptr;
    uint16_t *d_cent = nullptr, *d_ccrow = nullptr; int32_t *d_cg_tid = nullptr; double *d_cg_u = nullptr, *d_crow_w = nullptr, *d_cscratch = nullptr;
    unsigned *d_cbar = nullptr;          // [2 n]: barrier words, then abort words
    ClusterStat *d_cstat = nullptr, *h_cstat = nullptr;
    int64_t n_cstat = 0;
    hipEvent_t ev_c0 = nullptr, ev_c1 = nullptr;   // around the cluster launches (stats)
    int32_t *d_g_tid = nullptr; double *d_g_u = nullptr, *d_row_w = nullptr, *d_usum = nullptr;
    uint16_t *d_srp = nullptr, *d_sent = nullptr, *d_scp = nullptr, *d_scrow = nullptr;
    uint8_t *d_kind = nullptr;
    double sets_build_ms = 0.0;
    // compute_adjEUMA on the device
    int32_t *d_euma_t = nullptr; int32_t nfl = 0; double *d_wf = nullptr, *d_adj = nullptr;
};

namespace {

#define HIPCHK(call)                                                                                   \
    do {                                                                                               \
        hipError_t e_ = (call);                                                                        \
        if (e_ != hipSuccess) {                                                                        \
            ctx->err = std::string(#call) + ": " + hipGetErrorString(e_);                              \
            return e_ == hipErrorOutOfMemory ? EMSAR_HIP_ERR_OOM : EMSAR_HIP_ERR_HIP;                   \
        }                                                                                              \
    } while (0)

inline void dfree(void *p) { if (p) (void)hipFree(p); }

inline int grid_for(int64_t n, int block) { return (int)((n + block - 1) / block); }

// The TILED layout may number the transcripts itself (renumber.hpp); every T-sized device vector is then in the LIBRARY's numbering and
// the ABI maps: lib[new_of_old[t]] = caller[t].  Empty map = the caller's numbering.
inline const std::vector<int32_t> &tid_map(const emsar_hip_ctx *ctx) { return ctx->TL.new_of_old; }
inline const double *to_lib(const emsar_hip_ctx *ctx, const double *caller, std::vector<double> &tmp) {
    const auto &m = tid_map(ctx);
    if (m.empty() || ctx->layout != EMSAR_LAYOUT_TILED) return caller;
    tmp.resize(m.size());
    for (size_t t = 0; t < m.size(); t++) tmp[(size_t)m[t]] = caller[t];
    return tmp.data();
}
inline void from_lib(const emsar_hip_ctx *ctx, double *v /* in place: library order -> caller order */) {
    const auto &m = tid_map(ctx);
    if (m.empty() || ctx->layout != EMSAR_LAYOUT_TILED) return;
    std::vector<double> tmp(v, v + m.size());
    for (size_t t = 0; t < m.size(); t++) v[t] = tmp[(size_t)m[t]];
}

// bytes one pass actually streams in the chosen layout: index arrays + row weights + the T-sized vectors
inline int64_t stored_bytes(const emsar_hip_ctx *ctx) {
    int64_t rows = ctx->layout == EMSAR_LAYOUT_TILED ? ctx->n_slots + ctx->n_left : ctx->n_rows;
    return ctx->bytes_stored + (ctx->weighted ? 4 * rows : 0) + (ctx->layout == EMSAR_LAYOUT_TILED ? 40 : 32) * (int64_t)ctx->n_tx;
}

void free_sets(emsar_hip_ctx *ctx) {
    for (auto &p : ctx->d_sdesc) { dfree(p); p = nullptr; }
    dfree(ctx->d_sstat); ctx->d_sstat = nullptr;
    if (ctx->h_sstat) { (void)hipHostFree(ctx->h_sstat); ctx->h_sstat = nullptr; }
    dfree(ctx->d_g_tid); dfree(ctx->d_g_u); dfree(ctx->d_row_w); dfree(ctx->d_usum);
    dfree(ctx->d_srp); dfree(ctx->d_sent); dfree(ctx->d_scp); dfree(ctx->d_scrow); dfree(ctx->d_kind);
    ctx->d_g_tid = nullptr; ctx->d_g_u = ctx->d_row_w = ctx->d_usum = nullptr;
    ctx->d_srp = ctx->d_sent = ctx->d_scp = ctx->d_scrow = nullptr; ctx->d_kind = nullptr;
    dfree(ctx->d_cdesc); dfree(ctx->d_cblk); dfree(ctx->d_crp); dfree(ctx->d_ccp); dfree(ctx->d_cpart); dfree(ctx->d_cent); dfree(ctx->d_ccrow);
    dfree(ctx->d_cg_tid); dfree(ctx->d_cg_u); dfree(ctx->d_crow_w); dfree(ctx->d_cscratch); dfree(ctx->d_cbar); dfree(ctx->d_cstat);
    if (ctx->h_cstat) { (void)hipHostFree(ctx->h_cstat); ctx->h_cstat = nullptr; }
    ctx->d_cdesc = nullptr; ctx->d_cblk = ctx->d_crp = ctx->d_ccp = ctx->d_cpart = nullptr; ctx->d_cent = ctx->d_ccrow = nullptr;
    ctx->d_cg_tid = nullptr; ctx->d_cg_u = ctx->d_crow_w = ctx->d_cscratch = nullptr; ctx->d_cbar = nullptr; ctx->d_cstat = nullptr; ctx->n_cstat = 0;
    ctx->RS = emsar::ResidentSets(); ctx->sets_ready = false; ctx->n_sstat = 0;
}

void free_structure(emsar_hip_ctx *ctx) {
    free_sets(ctx);
    dfree(ctx->d_euma_t); dfree(ctx->d_wf); dfree(ctx->d_adj); ctx->d_euma_t = nullptr; ctx->d_wf = ctx->d_adj = nullptr; ctx->nfl = 0;
    std::vector<uint64_t>().swap(ctx->h_row_ptr); std::vector<int32_t>().swap(ctx->h_col); std::vector<int32_t>().swap(ctx->h_wgt);
    dfree(ctx->d_row_ptr); dfree(ctx->d_col);
    ctx->d_row_ptr = nullptr; ctx->d_col = nullptr;
    dfree(ctx->d_wgt); dfree(ctx->d_rowval); ctx->d_wgt = nullptr; ctx->d_rowval = nullptr;
    dfree(ctx->d_units); ctx->d_units = nullptr; ctx->n_units = 0;
    dfree(ctx->d_utiles); ctx->d_utiles = nullptr;
    dfree(ctx->d_tiles); dfree(ctx->d_fwd); dfree(ctx->d_bwd); dfree(ctx->d_far);
    dfree(ctx->d_left_ptr); dfree(ctx->d_left_col); dfree(ctx->d_left_wgt); dfree(ctx->d_left_val); dfree(ctx->d_u);
    ctx->d_tiles = nullptr; ctx->d_fwd = ctx->d_bwd = nullptr; ctx->d_far = nullptr;
    ctx->d_left_ptr = nullptr; ctx->d_left_col = nullptr; ctx->d_left_wgt = nullptr; ctx->d_left_val = nullptr; ctx->d_u = nullptr;
    ctx->TL = emsar::TiledLayout(); ctx->n_left = ctx->n_tiles = ctx->n_slots = 0;
    dfree(ctx->d_den); dfree(ctx->d_acc); ctx->d_den = nullptr; ctx->d_acc = nullptr;
    for (auto &p : ctx->d_th) { dfree(p); p = nullptr; }
    for (auto &p : ctx->d_tmp) { dfree(p); p = nullptr; }
    dfree(ctx->d_itmp); ctx->d_itmp = nullptr;
    ctx->have_structure = ctx->have_sample = false;
}

// the fixed-point scales the EM kernels get (zeros = plain FP64 atomics; scatter passes always)
inline Fx fx_of(const emsar_hip_ctx *ctx, int mode = MODE_EM) { return (ctx->det && mode != MODE_SCATTER) ? Fx{ctx->fx_mass, ctx->fx_ll} : Fx{0.0, 0.0}; }

// one pass of the chosen layout.  mode: MODE_EM / MODE_EM_LL / MODE_SCATTER
int launch_pass(emsar_hip_ctx *ctx, int mode, const double *theta, double *acc, double *ll_out, bool rows_only = false /* the folded rows' likelihood terms are added by the caller */) {
    if (ctx->layout == EMSAR_LAYOUT_TILED) {
        const size_t lds = (size_t)kTiledLdsDoubles * sizeof(double);
        if (ctx->n_tiles > 0) {
            dim3 grid((unsigned)ctx->n_tiles), block(kTiledThreads);
#define LAUNCH_T(WT, MD)                                                                                          \
    hipLaunchKernelGGL((k_pass_tiled<WT, MD>), grid, block, lds, ctx->stream, ctx->d_tiles, ctx->d_fwd, ctx->d_bwd,   \
                       ctx->d_far, ctx->d_wgt, ctx->d_rowval, theta, acc, ll_out, fx_of(ctx, mode))
#define LAUNCH_PN(WT, MD, NN)                                                                                     \
    hipLaunchKernelGGL((k_pass_tiled_multi<WT, MD, NN>), dim3((unsigned)((ctx->n_tiles + NN - 1) / NN)), block, lds, ctx->stream, ctx->d_tiles,  \
                       (int)ctx->n_tiles, ctx->d_fwd, ctx->d_bwd, ctx->d_far, ctx->d_wgt, theta, acc, ll_out, fx_of(ctx, mode))
#define LAUNCH_P(WT, MD) LAUNCH_PN(WT, MD, 2)
            if (mode == MODE_SCATTER) LAUNCH_T(false, MODE_SCATTER);
            else if (!ctx->weighted && (ctx->tiled_multi >= 2 || (ctx->tiled_multi == 1 && ctx->n_tiles > kPairMinTiles))) {
                // more than one tile per workgroup.  Unweighted rows only: with the row weights in registers as well the body does not
                // fit 128 VGPRs (round 1, two tiles: 0.218 vs 0.179 ms; round 2, the unit kernel on merged rows, 72-92 B of scratch:
                // 0.124 vs 0.103 ms with one tile per workgroup; with the weights kept as integers its EM variant fits without
                // scratch and runs config 3's merged rows in 0.0959 ms against 0.0956 ms for one tile per workgroup: no gain,
                // and the likelihood variant -- twelve logs -- still spills).
                // Only when the tiles outnumber the chip's workgroup slots: below that a pass is one workgroup's latency, and
                // a pair takes twice as long as a tile (40 k reads: 47 -> 26 us per pass with one tile per workgroup)
                if (ctx->tiled_multi == 1 || ctx->tiled_multi == 5) {         // units: one dictionary for up to two tiles
#define LAUNCH_U(WT, MD) hipLaunchKernelGGL((k_pass_tiled_unit<WT, MD>), dim3((unsigned)ctx->n_units), block, lds, ctx->stream, ctx->d_utiles, ctx->unit_stride, \
                                           ctx->d_far, ctx->d_fwd, ctx->d_bwd, ctx->d_wgt, theta, acc, ll_out, fx_of(ctx, mode))
                    if (mode == MODE_EM_LL) LAUNCH_U(false, MODE_EM_LL); else LAUNCH_U(false, MODE_EM);
#undef LAUNCH_U
                }
                else if (ctx->tiled_multi == 3) { if (mode == MODE_EM_LL) LAUNCH_PN(false, MODE_EM_LL, 3); else LAUNCH_PN(false, MODE_EM, 3); }
                else if (ctx->tiled_multi == 4) { if (mode == MODE_EM_LL) LAUNCH_PN(false, MODE_EM_LL, 4); else LAUNCH_PN(false, MODE_EM, 4); }
                else if (mode == MODE_EM_LL) LAUNCH_P(false, MODE_EM_LL); else LAUNCH_P(false, MODE_EM);
            }
            else if (ctx->weighted && (ctx->weighted_unit == 2 || (ctx->weighted_unit == 1 && mode == MODE_EM)) &&
                     (ctx->tiled_multi == 5 || (ctx->tiled_multi == 1 && ctx->n_tiles > kPairMinTiles))) {
                // weighted rows (segments with read counts, merged rows) on the unit kernel: the weights are loaded as integers after the
                // forward batch is consumed; both variants fit 128 VGPRs without scratch (round 3).  Measured on the collapsed form of
                // config 3 (14.0 M segments of the family law / 5.4 M of the window law): plain pass 0.1273 -> 0.1221 / 0.0964 -> 0.0962 ms;
                // the likelihood variant takes its twelve logs per lane in one rolled loop (tile_e_step) and is SLOWER than the one-tile
                // kernel's unrolled logs (solve 0.161 against 0.150 ms per pass), so by default (1) only the plain EM pass of a SQUAREM
                // cycle runs here and the two likelihood passes stay with k_pass_tiled; 2 = both, 0 = neither (EMSAR_HIP_WEIGHTED_UNIT)
                hipLaunchKernelGGL((mode == MODE_EM_LL ? k_pass_tiled_unit<true, MODE_EM_LL> : k_pass_tiled_unit<true, MODE_EM>), dim3((unsigned)ctx->n_units), block, lds,
                                   ctx->stream, ctx->d_utiles, ctx->unit_stride, ctx->d_far, ctx->d_fwd, ctx->d_bwd, ctx->d_wgt, theta, acc, ll_out,
                                   fx_of(ctx, mode), (unsigned long long *)nullptr);
            }
            else if (ctx->weighted) { if (mode == MODE_EM_LL) LAUNCH_T(true, MODE_EM_LL); else LAUNCH_T(true, MODE_EM); }
            else { if (mode == MODE_EM_LL) LAUNCH_T(false, MODE_EM_LL); else LAUNCH_T(false, MODE_EM); }
#undef LAUNCH_P
#undef LAUNCH_PN
#undef LAUNCH_T
        }
        if (ctx->n_left > 0) {   // rows too long for a tile: generic CSR kernel on the leftover
            dim3 grid((unsigned)std::min<int64_t>((ctx->n_left + 255) / 256, 8192)), block(256);
#define LAUNCH_L(WT, MD)                                                                                          \
    hipLaunchKernelGGL((k_pass_csr<uint64_t, WT, MD>), grid, block, 0, ctx->stream, ctx->n_left, ctx->d_left_ptr,     \
                       ctx->d_left_col, ctx->d_left_wgt, ctx->d_left_val, theta, acc, ll_out, fx_of(ctx, mode))
            if (mode == MODE_SCATTER) LAUNCH_L(false, MODE_SCATTER);
            else if (ctx->weighted) { if (mode == MODE_EM_LL) LAUNCH_L(true, MODE_EM_LL); else LAUNCH_L(true, MODE_EM); }
            else { if (mode == MODE_EM_LL) LAUNCH_L(false, MODE_EM_LL); else LAUNCH_L(false, MODE_EM); }
#undef LAUNCH_L
        }
        if (mode == MODE_EM_LL && !rows_only)
            hipLaunchKernelGGL(k_single_ll, dim3(std::min(grid_for(ctx->n_tx, 256), 256)), dim3(256), 0, ctx->stream, ctx->n_tx,
                               ctx->d_u, theta, ll_out, fx_of(ctx).ll);
        HIPCHK(hipGetLastError());
        return EMSAR_HIP_OK;
    }
    if (ctx->n_rows == 0) return EMSAR_HIP_OK;
    int64_t blocks = (ctx->n_rows + 255) / 256;
    dim3 grid((unsigned)std::min<int64_t>(blocks, 256 * 32)), block(256);
#define LAUNCH_C(PT, WT, MD)                                                                                     \
    hipLaunchKernelGGL((k_pass_csr<PT, WT, MD>), grid, block, 0, ctx->stream, ctx->n_rows, (const PT *)ctx->d_row_ptr, \
                       ctx->d_col, ctx->d_wgt, ctx->d_rowval, theta, acc, ll_out, fx_of(ctx, mode))
#define LAUNCH_CP(WT, MD) do { if (ctx->ptr64) LAUNCH_C(uint64_t, WT, MD); else LAUNCH_C(uint32_t, WT, MD); } while (0)
    if (mode == MODE_SCATTER) LAUNCH_CP(false, MODE_SCATTER);
    else if (ctx->weighted) { if (mode == MODE_EM_LL) LAUNCH_CP(true, MODE_EM_LL); else LAUNCH_CP(true, MODE_EM); }
    else { if (mode == MODE_EM_LL) LAUNCH_CP(false, MODE_EM_LL); else LAUNCH_CP(false, MODE_EM); }
#undef LAUNCH_CP
#undef LAUNCH_C
    HIPCHK(hipGetLastError());
    return EMSAR_HIP_OK;
}

// a likelihood word of the host copy of the scalars (fixed point in deterministic mode)
inline double host_ll(const emsar_hip_ctx *ctx, int i) {
    const LlSum &L = ctx->h_scal->ll[i];             // the words of the sum, added in a fixed order (kernels_common.hpp)
    if (!ctx->det || ctx->fx_ll == 0.0) { double v = 0.0; for (int j = 0; j < kLlSlots; j++) v += L.s[j].v; return v; }
    long long b = 0;
    for (int j = 0; j < kLlSlots; j++) { long long x; memcpy(&x, &L.s[j].v, 8); b += x; }
    return (double)b / ctx->fx_ll;
}

// th_out = EM(th_in); ll slot receives sum R log S at th_in when want_ll
int em_pass(emsar_hip_ctx *ctx, const double *th_in, double *th_out, bool want_ll, int ll_slot, double abs_floor, int to_delta1 = 0) {
    int rc = launch_pass(ctx, want_ll ? MODE_EM_LL : MODE_EM, th_in, ctx->d_acc, &ctx->d_scal->ll[ll_slot].s[0].v);
    if (rc) return rc;
    hipLaunchKernelGGL(k_update, dim3(std::min(grid_for(ctx->n_tx, 256), ctx->update_grid)), dim3(256), 0, ctx->stream, ctx->n_tx, th_in, ctx->d_acc,
                       ctx->d_den, ctx->layout == EMSAR_LAYOUT_TILED ? ctx->d_u : nullptr, th_out, abs_floor, ctx->count_floor, ctx->zero_cut, ctx->d_scal,
                       ctx->delta_mask, to_delta1, fx_of(ctx).mass);
    HIPCHK(hipGetLastError());
    return EMSAR_HIP_OK;
}

// `cycles` cycles of the streaming solve on ctx->stream -- launched, or recorded when the stream is capturing.
// One cycle = one plain EM pass, or one SQUAREM cycle of three passes (8 launches, see k_update_p2).  The current point is
// ctx->d_th[0] before and after (plain EM swaps d_th[0]/d_th[1] on the host: record an even count).
int enqueue_cycles(emsar_hip_ctx *ctx, const emsar_em_params &p, double abs_step_base, int cycles) {
    const int n = ctx->n_tx, g = grid_for(n, 256);
    double **th = ctx->d_th;
    int rc;
    for (int c = 0; c < cycles; c++) {
        hipLaunchKernelGGL(k_cycle_begin, dim3(1), dim3(kLlSlots), 0, ctx->stream, ctx->d_scal, abs_step_base, p.accel ? 3 : 1);
        if (!p.accel) {
            if ((rc = em_pass(ctx, th[0], th[1], false, 0, p.abs_floor))) return rc;
            std::swap(th[0], th[1]);
            continue;
        }
        // the stopping rule is measured on the first (plain) step of the cycle only (delta1_bits)
        const double *u = ctx->layout == EMSAR_LAYOUT_TILED ? ctx->d_u : nullptr;
        const dim3 gv((unsigned)std::min(std::min(g, ctx->sq_grid), kSqPart)), bv(256);
        if ((rc = em_pass(ctx, th[0], th[1], false, 0, p.abs_floor, 1))) return rc;
        if ((rc = launch_pass(ctx, MODE_EM_LL, th[1], ctx->d_acc, &ctx->d_scal->ll[1].s[0].v, true))) return rc;
        hipLaunchKernelGGL(k_update_p2, gv, bv, 0, ctx->stream, n, th[0], th[1], ctx->d_acc, ctx->d_den, u, th[2], ctx->d_scal, ctx->d_sqpart, fx_of(ctx));
        hipLaunchKernelGGL(k_sq_extrap_ll, gv, bv, 0, ctx->stream, n, th[0], th[1], th[2], ctx->d_den, u, th[3], ctx->d_scal, ctx->d_sqpart, (int)gv.x, fx_of(ctx));
        if ((rc = launch_pass(ctx, MODE_EM_LL, th[3], ctx->d_acc, &ctx->d_scal->ll[2].s[0].v, true))) return rc;
        hipLaunchKernelGGL(k_update_p3, gv, bv, 0, ctx->stream, n, th[3], th[2], ctx->d_acc, ctx->d_den, u, th[0], ctx->d_scal, ctx->d_sqpart, (int)gv.x, fx_of(ctx));
        HIPCHK(hipGetLastError());
    }
    return EMSAR_HIP_OK;
}

struct CycleGraph {
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
    ~CycleGraph() {
        if (exec) (void)hipGraphExecDestroy(exec);
        if (graph) (void)hipGraphDestroy(graph);
    }
};

// scatter a per-row value (original row order, host) to its columns: out[t] = sum_c m_ct val[c]
int scatter_rows(emsar_hip_ctx *ctx, const double *val_host, double *d_out) {
    try {
        if (ctx->layout == EMSAR_LAYOUT_TILED) {
            const auto &L = ctx->TL;
            std::vector<double> slot((size_t)std::max<int64_t>(ctx->n_slots, 1), 0.0), left((size_t)std::max<int64_t>(ctx->n_left, 1), 0.0);
            std::vector<double> base((size_t)ctx->n_tx, 0.0);
            for (int64_t i = 0; i < ctx->n_slots; i++) {
                int64_t r = L.slot_row[(size_t)i];
                if (r < 0) continue;
                if (L.merged) { double v = 0; for (uint64_t q = L.mem_ptr[(size_t)r]; q < L.mem_ptr[(size_t)r + 1]; q++) v += val_host[L.mem_row[(size_t)q]]; slot[(size_t)i] = v; }
                else slot[(size_t)i] = val_host[r];
            }
            for (int64_t i = 0; i < ctx->n_left; i++) left[(size_t)i] = val_host[L.left_row[(size_t)i]];
            for (size_t i = 0; i < L.single_row.size(); i++) base[(size_t)L.single_tid[i]] += val_host[L.single_row[i]];
            if (!ctx->d_rowval) HIPCHK(hipMalloc(&ctx->d_rowval, slot.size() * sizeof(double)));
            if (!ctx->d_left_val) HIPCHK(hipMalloc(&ctx->d_left_val, left.size() * sizeof(double)));
            HIPCHK(hipMemcpyAsync(ctx->d_rowval, slot.data(), slot.size() * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
            HIPCHK(hipMemcpyAsync(ctx->d_left_val, left.data(), left.size() * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
            HIPCHK(hipMemcpyAsync(d_out, base.data(), base.size() * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
            int rc = launch_pass(ctx, MODE_SCATTER, nullptr, d_out, nullptr);
            if (rc) return rc;
            HIPCHK(hipStreamSynchronize(ctx->stream));
            return EMSAR_HIP_OK;
        }
        const double *src = val_host;
        size_t n = (size_t)ctx->n_rows;
        if (n == 0) return EMSAR_HIP_OK;
        if (!ctx->d_rowval) HIPCHK(hipMalloc(&ctx->d_rowval, std::max<size_t>(n, 1) * sizeof(double)));
        HIPCHK(hipMemcpyAsync(ctx->d_rowval, src, n * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
        HIPCHK(hipMemsetAsync(d_out, 0, (size_t)ctx->n_tx * sizeof(double), ctx->stream));
        int rc = launch_pass(ctx, MODE_SCATTER, nullptr, d_out, nullptr);
        if (rc) return rc;
        HIPCHK(hipStreamSynchronize(ctx->stream));
        return EMSAR_HIP_OK;
    } catch (const std::bad_alloc &) { ctx->err = "out of host memory"; return EMSAR_HIP_ERR_OOM; }
}

// find and pack the connected sets of the current sample (sets.hpp) and move the records to the device
int ensure_sets_impl(emsar_hip_ctx *ctx);
int ensure_sets(emsar_hip_ctx *ctx) {
    if (ctx->sets_ready) return EMSAR_HIP_OK;
    const int rc = ensure_sets_impl(ctx);
    if (rc != EMSAR_HIP_OK) free_sets(ctx);       // a half-uploaded record set is freed, the next solve starts over
    return rc;
}
int ensure_sets_impl(emsar_hip_ctx *ctx) {
    auto t0 = std::chrono::steady_clock::now();
    auto &S = ctx->RS;
    try {
        emsar::build_sets(ctx->n_rows, ctx->n_tx, ctx->h_row_ptr.data(), ctx->h_col.data(), ctx->h_wgt.data(), S);
    } catch (const std::bad_alloc &) { return EMSAR_HIP_ERR_OOM; }
    if (ctx->layout == EMSAR_LAYOUT_TILED && !tid_map(ctx).empty()) {
        // the sets were found on the caller's CSR; theta / den on the device are in the library's numbering
        const auto &m = tid_map(ctx);
        try {
            std::vector<uint8_t> kind(S.kind.size());
            std::vector<double> usum(S.usum.size());
            for (size_t t = 0; t < m.size(); t++) { kind[(size_t)m[t]] = S.kind[t]; usum[(size_t)m[t]] = S.usum[t]; }
            S.kind.swap(kind); S.usum.swap(usum);
        } catch (const std::bad_alloc &) { return EMSAR_HIP_ERR_OOM; }
        for (int32_t &t : S.g_tid) t = m[(size_t)t];
        for (int32_t &t : S.CL.g_tid) t = m[(size_t)t];
    }
    auto up = [&](void **dp, const void *src, size_t bytes) -> hipError_t {
        hipError_t e = hipMalloc(dp, std::max<size_t>(bytes, 16));
        if (e == hipSuccess && bytes) e = hipMemcpy(*dp, src, bytes, hipMemcpyHostToDevice);
        return e;
    };
    HIPCHK(up((void **)&ctx->d_kind, S.kind.data(), S.kind.size()));
    HIPCHK(up((void **)&ctx->d_usum, S.usum.data(), S.usum.size() * 8));
    const int64_t n = S.n_resident();
    if (n > 0) {
        HIPCHK(up((void **)&ctx->d_g_tid, S.g_tid.data(), S.g_tid.size() * 4));
        HIPCHK(up((void **)&ctx->d_g_u, S.g_u.data(), S.g_u.size() * 8));
        HIPCHK(up((void **)&ctx->d_row_w, S.row_w.data(), S.row_w.size() * 8));
        HIPCHK(up((void **)&ctx->d_srp, S.rp.data(), S.rp.size() * 2));
        HIPCHK(up((void **)&ctx->d_sent, S.ent.data(), S.ent.size() * 2));
        HIPCHK(up((void **)&ctx->d_scp, S.cp.data(), S.cp.size() * 2));
        HIPCHK(up((void **)&ctx->d_scrow, S.crow.data(), S.crow.size() * 2));
        for (int c = 0; c < emsar::kSetClasses; c++)
            if (!S.desc[c].empty()) HIPCHK(up((void **)&ctx->d_sdesc[c], S.desc[c].data(), S.desc[c].size() * sizeof(emsar::SetDesc)));
        HIPCHK(hipMalloc(&ctx->d_sstat, (size_t)n * sizeof(SetStat)));
        HIPCHK(hipHostMalloc((void **)&ctx->h_sstat, (size_t)n * sizeof(SetStat), hipHostMallocDefault));
        ctx->n_sstat = n;
        HIPCHK(hipFuncSetAttribute((const void *)k_solve_sets<64>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)emsar::kSetLdsCap[0]));
        HIPCHK(hipFuncSetAttribute((const void *)k_solve_sets<256>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)emsar::kSetLdsCap[1]));
        HIPCHK(hipFuncSetAttribute((const void *)k_solve_sets<512>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)emsar::kSetLdsCap[2]));
    }
    const int64_t nc = S.n_cluster_sets();
    if (nc > 0) {
        auto &CL = S.CL;
        HIPCHK(up((void **)&ctx->d_cdesc, CL.desc.data(), CL.desc.size() * sizeof(emsar::ClusterDesc)));
        HIPCHK(up((void **)&ctx->d_cblk, CL.blk_set.data(), CL.blk_set.size() * 4));
        HIPCHK(up((void **)&ctx->d_crp, CL.rp.data(), CL.rp.size() * 4));
        HIPCHK(up((void **)&ctx->d_ccp, CL.cp.data(), CL.cp.size() * 4));
        HIPCHK(up((void **)&ctx->d_cpart, CL.part.data(), CL.part.size() * 4));
        HIPCHK(up((void **)&ctx->d_cent, CL.ent.data(), CL.ent.size() * 2));
        HIPCHK(up((void **)&ctx->d_ccrow, CL.crow.data(), CL.crow.size() * 2));
        HIPCHK(up((void **)&ctx->d_cg_tid, CL.g_tid.data(), CL.g_tid.size() * 4));
        HIPCHK(up((void **)&ctx->d_cg_u, CL.g_u.data(), CL.g_u.size() * 8));
        HIPCHK(up((void **)&ctx->d_crow_w, CL.row_w.data(), CL.row_w.size() * 8));
        HIPCHK(hipMalloc(&ctx->d_cscratch, std::max<size_t>((size_t)CL.scratch_doubles, 2) * 8));
        HIPCHK(hipMalloc(&ctx->d_cbar, (size_t)nc * 2 * sizeof(unsigned)));
        HIPCHK(hipMalloc(&ctx->d_cstat, (size_t)nc * sizeof(ClusterStat)));
        HIPCHK(hipHostMalloc((void **)&ctx->h_cstat, (size_t)nc * sizeof(ClusterStat), hipHostMallocDefault));
        ctx->n_cstat = nc;
        HIPCHK(hipFuncSetAttribute((const void *)k_solve_cluster, hipFuncAttributeMaxDynamicSharedMemorySize, (int)emsar::kClusterLdsCap));
        // only the sizes are needed from here on
        std::vector<uint32_t>().swap(CL.rp); std::vector<uint32_t>().swap(CL.cp); std::vector<uint16_t>().swap(CL.ent); std::vector<uint16_t>().swap(CL.crow);
        std::vector<int32_t>().swap(CL.g_tid); std::vector<double>().swap(CL.g_u); std::vector<double>().swap(CL.row_w);
    }
    // the device copies are the only ones needed from here on (desc sizes and counters stay)
    std::vector<int32_t>().swap(S.g_tid); std::vector<double>().swap(S.g_u); std::vector<double>().swap(S.row_w);
    std::vector<uint16_t>().swap(S.rp); std::vector<uint16_t>().swap(S.ent); std::vector<uint16_t>().swap(S.cp); std::vector<uint16_t>().swap(S.crow);
    std::vector<double>().swap(S.usum);
    ctx->sets_build_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    ctx->sets_ready = true;
    return EMSAR_HIP_OK;
}

// closed-form transcripts and every LDS-resident set, written into theta (the streamed sets' entries are left alone)
int solve_resident_sets(emsar_hip_ctx *ctx, const SetSolveParams &P, const SetSolveParams &Pcluster, double *theta) {
    const auto &S = ctx->RS;
    hipLaunchKernelGGL(k_closed_form, dim3(grid_for(ctx->n_tx, 256)), dim3(256), 0, ctx->stream, ctx->n_tx, ctx->d_kind, ctx->d_usum,
                       ctx->d_den, theta);
    // The three size classes are independent (disjoint sets, disjoint theta entries): the larger two run on side streams
    // next to the 64-thread class, so the solve lasts as long as the slowest class, not as long as their sum.
    HIPCHK(hipEventRecord(ctx->ev_fork, ctx->stream));
    const size_t off[3] = {0, S.desc[0].size(), S.desc[0].size() + S.desc[1].size()};      // per-set results in class order
#define LAUNCH_S(C, TH, ST)                                                                                                \
    if (!S.desc[C].empty())                                                                                                \
        hipLaunchKernelGGL(k_solve_sets<TH>, dim3((unsigned)S.desc[C].size()), dim3(TH), S.max_lds[C], ST,                   \
                           ctx->d_sdesc[C], ctx->d_g_tid, ctx->d_g_u, ctx->d_row_w, ctx->d_srp, ctx->d_sent, ctx->d_scp,     \
                           ctx->d_scrow, ctx->d_den, theta, ctx->d_sstat + off[C], P);
    for (int i = 0; i < 3; i++) HIPCHK(hipStreamWaitEvent(ctx->side[i], ctx->ev_fork, 0));
    if (ctx->n_cstat > 0) {
        // The clusters, on a stream of their own.  Every workgroup of a launch must be resident at once (they wait for each other at
        // the cluster barriers): at most one workgroup per CU per launch -- each asks for most of a CU's LDS --, whole sets only.
        const int n_cu = ctx->n_cu;
        HIPCHK(hipMemsetAsync(ctx->d_cbar, 0, (size_t)ctx->n_cstat * 2 * sizeof(unsigned), ctx->side[2]));
        HIPCHK(hipEventRecord(ctx->ev_c0, ctx->side[2]));
        const auto &D = S.CL.desc;
        size_t first = 0;
        while (first < D.size()) {
            size_t last = first, wgs = 0;
            while (last < D.size() && (wgs == 0 || wgs + D[last].g <= (size_t)n_cu)) wgs += D[last++].g;
            hipLaunchKernelGGL(k_solve_cluster, dim3((unsigned)wgs), dim3(emsar::kClusterThreads), S.CL.max_lds, ctx->side[2], ctx->d_cdesc, ctx->d_cblk,
                               D[first].blk0, ctx->d_cg_tid, ctx->d_cg_u, ctx->d_crow_w, ctx->d_crp, ctx->d_cent, ctx->d_ccp, ctx->d_ccrow, ctx->d_cpart,
                               ctx->d_cscratch, ctx->d_cbar, ctx->d_cbar + ctx->n_cstat, ctx->d_den, theta, ctx->d_cstat, Pcluster);
            first = last;
        }
        HIPCHK(hipEventRecord(ctx->ev_c1, ctx->side[2]));
        // The workgroups of a cluster wait for each other inside the launch, so all of them must get a CU: nothing else may hold CUs while
        // the cluster batches run (k_solve_sets<512> asks for most of a CU's LDS too).  The size classes below start after the clusters.
        HIPCHK(hipStreamWaitEvent(ctx->side[0], ctx->ev_c1, 0));
        HIPCHK(hipStreamWaitEvent(ctx->side[1], ctx->ev_c1, 0));
        HIPCHK(hipStreamWaitEvent(ctx->stream, ctx->ev_c1, 0));
    }
    LAUNCH_S(2, 512, ctx->side[1])      // the big ones first: they are the fewest and the longest per pass
    LAUNCH_S(1, 256, ctx->side[0])
    LAUNCH_S(0, 64, ctx->stream)
#undef LAUNCH_S
    for (int i = 0; i < 3; i++) {
        HIPCHK(hipEventRecord(ctx->ev_join[i], ctx->side[i]));
        HIPCHK(hipStreamWaitEvent(ctx->stream, ctx->ev_join[i], 0));
    }
    HIPCHK(hipGetLastError());
    return EMSAR_HIP_OK;
}

}  // namespace

hipStream_t emsar_internal_stream(emsar_hip_ctx *ctx) { return ctx->stream; }
int emsar_internal_device(const emsar_hip_ctx *ctx) { return ctx->device; }
void emsar_internal_set_error(emsar_hip_ctx *ctx, const char *call, const char *what) { ctx->err = std::string(call) + ": " + what; }

// ==================================================================================================
// C ABI
// ==================================================================================================
extern "C" {

const char *emsar_hip_strerror(int status) {
    switch (status) {
        case EMSAR_HIP_OK: return "ok";
        case EMSAR_HIP_ERR_ARG: return "invalid argument or malformed CSR";
        case EMSAR_HIP_ERR_NO_DEVICE: return "no usable HIP device";
        case EMSAR_HIP_ERR_OOM: return "out of memory";
        case EMSAR_HIP_ERR_HIP: return "HIP runtime failure";
        case EMSAR_HIP_ERR_STATE: return "wrong call order (upload_structure -> upload_sample -> solve)";
        case EMSAR_HIP_ERR_NUMERIC: return "NaN/Inf in theta";
        default: return "unknown status";
    }
}

const char *emsar_hip_last_error(const emsar_hip_ctx *ctx) { return ctx ? ctx->err.c_str() : ""; }

int emsar_hip_create(emsar_hip_ctx **out, int device_id) {
    if (!out) return EMSAR_HIP_ERR_ARG;
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return EMSAR_HIP_ERR_NO_DEVICE;
    if (device_id < 0 || device_id >= n) return EMSAR_HIP_ERR_NO_DEVICE;
    emsar_hip_ctx *ctx = new (std::nothrow) emsar_hip_ctx();
    if (!ctx) return EMSAR_HIP_ERR_OOM;
    ctx->device = device_id;
    auto fail = [&](int rc) { emsar_hip_destroy(ctx); return rc; };
    if (hipSetDevice(device_id) != hipSuccess) return fail(EMSAR_HIP_ERR_NO_DEVICE);
    { int v = 0; if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, device_id) == hipSuccess && v > 0) ctx->n_cu = v; }
    if (const char *e = getenv("EMSAR_HIP_GRAPH")) ctx->use_graph = atoi(e) != 0;
    if (const char *e = getenv("EMSAR_HIP_DETERMINISTIC")) ctx->det = atoi(e) != 0;
    if (hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess) return fail(EMSAR_HIP_ERR_HIP);
    if (hipEventCreate(&ctx->ev0) != hipSuccess || hipEventCreate(&ctx->ev1) != hipSuccess || hipEventCreate(&ctx->ev2) != hipSuccess) return fail(EMSAR_HIP_ERR_HIP);
    if (hipEventCreate(&ctx->ev_c0) != hipSuccess || hipEventCreate(&ctx->ev_c1) != hipSuccess) return fail(EMSAR_HIP_ERR_HIP);
    for (int i = 0; i < 3; i++)
        if (hipStreamCreateWithFlags(&ctx->side[i], hipStreamNonBlocking) != hipSuccess ||
            hipEventCreateWithFlags(&ctx->ev_join[i], hipEventDisableTiming) != hipSuccess) return fail(EMSAR_HIP_ERR_HIP);
    if (hipEventCreateWithFlags(&ctx->ev_fork, hipEventDisableTiming) != hipSuccess) return fail(EMSAR_HIP_ERR_HIP);
    if (hipMalloc(&ctx->d_scal, sizeof(Scal)) != hipSuccess) return fail(EMSAR_HIP_ERR_OOM);
    if (hipMalloc(&ctx->d_sqpart, 4 * kSqPart * sizeof(double)) != hipSuccess) return fail(EMSAR_HIP_ERR_OOM);
    if (hipMemset(ctx->d_sqpart, 0, 4 * kSqPart * sizeof(double)) != hipSuccess) return fail(EMSAR_HIP_ERR_HIP);
    if (hipHostMalloc((void **)&ctx->h_scal, sizeof(Scal), hipHostMallocDefault) != hipSuccess) return fail(EMSAR_HIP_ERR_OOM);
    // both pass kernels may need more than the default dynamic-LDS limit
    *out = ctx;
    return EMSAR_HIP_OK;
}

int emsar_hip_set_deterministic(emsar_hip_ctx *ctx, int on) {
    if (!ctx) return EMSAR_HIP_ERR_ARG;
    ctx->det = on != 0;
    return EMSAR_HIP_OK;
}

void emsar_hip_destroy(emsar_hip_ctx *ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    free_structure(ctx);
    dfree(ctx->d_scal); dfree(ctx->d_sqpart);
    if (ctx->h_scal) (void)hipHostFree(ctx->h_scal);
    if (ctx->ev0) (void)hipEventDestroy(ctx->ev0);
    if (ctx->ev1) (void)hipEventDestroy(ctx->ev1);
    if (ctx->ev2) (void)hipEventDestroy(ctx->ev2);
    if (ctx->ev_fork) (void)hipEventDestroy(ctx->ev_fork);
    if (ctx->ev_c0) (void)hipEventDestroy(ctx->ev_c0);
    if (ctx->ev_c1) (void)hipEventDestroy(ctx->ev_c1);
    for (int i = 0; i < 3; i++) {
        if (ctx->side[i]) { (void)hipStreamSynchronize(ctx->side[i]); (void)hipStreamDestroy(ctx->side[i]); }
        if (ctx->ev_join[i]) (void)hipEventDestroy(ctx->ev_join[i]);
    }
    if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

int emsar_hip_upload_structure(emsar_hip_ctx *ctx, int64_t n_rows, int32_t n_tx, const uint64_t *row_ptr,
                               const int32_t *col_idx, int layout) {
    if (!ctx) return EMSAR_HIP_ERR_ARG;
    const bool merge_rows = (layout & EMSAR_LAYOUT_FLAG_MERGE_ROWS) != 0;
    layout &= ~EMSAR_LAYOUT_FLAG_MERGE_ROWS;
    if (layout != EMSAR_LAYOUT_AUTO && layout != EMSAR_LAYOUT_CSR && layout != EMSAR_LAYOUT_TILED) return EMSAR_HIP_ERR_ARG;
    if (merge_rows && layout != EMSAR_LAYOUT_AUTO && layout != EMSAR_LAYOUT_TILED) return EMSAR_HIP_ERR_ARG;
    if (emsar::validate_csr(n_rows, n_tx, row_ptr, col_idx) != 0) return EMSAR_HIP_ERR_ARG;
    HIPCHK(hipSetDevice(ctx->device));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    free_structure(ctx);
    ctx->n_rows = n_rows; ctx->n_tx = n_tx; ctx->nnz = (int64_t)row_ptr[n_rows];
    ctx->ptr64 = (uint64_t)ctx->nnz >= (1ull << 32);
    if (const char *e = getenv("EMSAR_HIP_FORCE_PTR64")) { if (atoi(e) != 0) ctx->ptr64 = true; }   // test hook: the 64-bit row_ptr kernels on small inputs
    if (layout == EMSAR_LAYOUT_AUTO) {
        layout = (n_rows < ((int64_t)1 << 32)) ? EMSAR_LAYOUT_TILED : EMSAR_LAYOUT_CSR;
        if (const char *e = getenv("EMSAR_HIP_LAYOUT")) { int v = atoi(e); if ((v == 1 || v == 3) && (v == 1 || n_rows < ((int64_t)1 << 32))) layout = v; }
        if (merge_rows && layout != EMSAR_LAYOUT_TILED) return EMSAR_HIP_ERR_ARG;
    }
    ctx->layout = layout;
    const size_t T = (size_t)n_tx;
    const bool dbg = getenv("EMSAR_HIP_DEBUG") != nullptr;
    const auto tu0 = std::chrono::steady_clock::now();
    auto since = [&](std::chrono::steady_clock::time_point a) { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - a).count(); };
    // the host copy of the CSR kept for the set-resident solver (built per sample: sets depend on which rows carry reads)
    // is made by a second thread while this one builds the device layout from the same arrays
    bool copy_failed = false;
    std::thread csr_copy([&] {
        try {
            ctx->h_row_ptr.assign(row_ptr, row_ptr + n_rows + 1);
            ctx->h_col.assign(col_idx, col_idx + ctx->nnz);
        } catch (const std::bad_alloc &) { copy_failed = true; }
    });
    struct Joiner { std::thread &t; ~Joiner() { if (t.joinable()) t.join(); } } joiner{csr_copy};
    try {
        if (layout == EMSAR_LAYOUT_TILED) {
            auto &L = ctx->TL;
            const int brc = emsar::build_tiled(n_rows, n_tx, row_ptr, col_idx, L, merge_rows);
            if (brc != 0) { ctx->err = "TILED layout builder: code " + std::to_string(brc); return EMSAR_HIP_ERR_ARG; }
            if (dbg) fprintf(stderr, "upload_structure: layout built after %.0f ms\n", since(tu0));
            ctx->n_tiles = (int64_t)L.tiles.size(); ctx->n_slots = L.n_slots(); ctx->n_left = (int64_t)L.left_row.size();
            auto up = [&](void **dp, const void *src, size_t bytes) -> hipError_t {
                hipError_t e = hipMalloc(dp, std::max<size_t>(bytes, 16));
                if (e == hipSuccess && bytes) e = hipMemcpy(*dp, src, bytes, hipMemcpyHostToDevice);
                return e;
            };
            HIPCHK(up((void **)&ctx->d_tiles, L.tiles.data(), L.tiles.size() * sizeof(Tile)));
            HIPCHK(up((void **)&ctx->d_units, L.unit_first.data(), L.unit_first.size() * 4));
            ctx->n_units = L.unit_first.empty() ? 0 : (int64_t)L.unit_first.size() - 1;
            {
                emsar::UnitTables U;
                emsar::build_unit_tables(L, U);
                ctx->unit_stride = U.stride;
                HIPCHK(up((void **)&ctx->d_utiles, U.utiles.data(), U.utiles.size() * sizeof(Tile)));
            }
            HIPCHK(up((void **)&ctx->d_fwd, L.fwd.data(), L.fwd.size() * 4));
            HIPCHK(up((void **)&ctx->d_bwd, L.bwd.data(), L.bwd.size() * 4));
            HIPCHK(up((void **)&ctx->d_far, L.far_tid.data(), L.far_tid.size() * 4));
            HIPCHK(up((void **)&ctx->d_left_ptr, L.left_ptr.data(), L.left_ptr.size() * 8));
            HIPCHK(up((void **)&ctx->d_left_col, L.left_col.data(), L.left_col.size() * 4));
            HIPCHK(hipMalloc(&ctx->d_u, T * 8));
            HIPCHK(hipMemset(ctx->d_u, 0, T * 8));
            ctx->bytes_stored = (int64_t)L.fwd.size() * 4 + (int64_t)L.bwd.size() * 4 + (int64_t)L.far_tid.size() * 4 +
                                (int64_t)L.tiles.size() * 64 + (int64_t)L.left_col.size() * 4 + (int64_t)L.left_ptr.size() * 8;
            ctx->tl_fwd_slots = L.padded_slots; ctx->tl_n_fslices = L.n_fslices;
            emsar::u32_vec().swap(L.fwd); emsar::u32_vec().swap(L.bwd);
            std::vector<int32_t>().swap(L.left_col);
            const size_t lds = (size_t)kTiledLdsDoubles * sizeof(double);
#define SETLDS_T(WT, MD) HIPCHK(hipFuncSetAttribute((const void *)k_pass_tiled<WT, MD>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds))
            SETLDS_T(false, MODE_EM); SETLDS_T(false, MODE_EM_LL); SETLDS_T(true, MODE_EM); SETLDS_T(true, MODE_EM_LL); SETLDS_T(false, MODE_SCATTER);
#undef SETLDS_T
#define SETLDS_P(WT, MD, NN) HIPCHK(hipFuncSetAttribute((const void *)k_pass_tiled_multi<WT, MD, NN>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds))
            SETLDS_P(false, MODE_EM, 2); SETLDS_P(false, MODE_EM_LL, 2);
#define SETLDS_U(WT, MD) HIPCHK(hipFuncSetAttribute((const void *)k_pass_tiled_unit<WT, MD>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds))
            SETLDS_U(false, MODE_EM); SETLDS_U(false, MODE_EM_LL); SETLDS_U(true, MODE_EM); SETLDS_U(true, MODE_EM_LL);
            { const char *pe = getenv("EMSAR_HIP_WEIGHTED_UNIT"); ctx->weighted_unit = pe ? atoi(pe) : 1; }
#undef SETLDS_U
            SETLDS_P(false, MODE_EM, 3); SETLDS_P(false, MODE_EM_LL, 3); SETLDS_P(false, MODE_EM, 4); SETLDS_P(false, MODE_EM_LL, 4);
#undef SETLDS_P
            { const char *pe = getenv("EMSAR_HIP_TILED_MULTI"); ctx->tiled_multi = pe ? atoi(pe) : 1; }
            { const char *pe = getenv("EMSAR_HIP_UPDATE_GRID"); if (pe && atoi(pe) >= 1) ctx->update_grid = atoi(pe); }
            { const char *pe = getenv("EMSAR_HIP_SQ_GRID"); if (pe && atoi(pe) >= 1) ctx->sq_grid = atoi(pe); }
        } else {
            if (ctx->ptr64) {
                HIPCHK(hipMalloc(&ctx->d_row_ptr, ((size_t)n_rows + 1) * 8));
                HIPCHK(hipMemcpy(ctx->d_row_ptr, row_ptr, ((size_t)n_rows + 1) * 8, hipMemcpyHostToDevice));
            } else {
                std::vector<uint32_t> rp((size_t)n_rows + 1);
                for (int64_t r = 0; r <= n_rows; r++) rp[(size_t)r] = (uint32_t)row_ptr[r];
                HIPCHK(hipMalloc(&ctx->d_row_ptr, rp.size() * 4));
                HIPCHK(hipMemcpy(ctx->d_row_ptr, rp.data(), rp.size() * 4, hipMemcpyHostToDevice));
            }
            HIPCHK(hipMalloc(&ctx->d_col, std::max<size_t>((size_t)ctx->nnz, 1) * 4));
            HIPCHK(hipMemcpy(ctx->d_col, col_idx, (size_t)ctx->nnz * 4, hipMemcpyHostToDevice));
            ctx->bytes_stored = ctx->nnz * 4 + (n_rows + 1) * (ctx->ptr64 ? 8 : 4);
        }
    } catch (const std::bad_alloc &) {
        csr_copy.join();
        free_structure(ctx);
        return EMSAR_HIP_ERR_OOM;
    }
    if (dbg) fprintf(stderr, "upload_structure: device copies done after %.0f ms\n", since(tu0));
    csr_copy.join();
    if (copy_failed) { free_structure(ctx); return EMSAR_HIP_ERR_OOM; }
    if (dbg) fprintf(stderr, "upload_structure: host CSR copy joined after %.0f ms\n", since(tu0));
    HIPCHK(hipMalloc(&ctx->d_den, T * 8));
    HIPCHK(hipMalloc(&ctx->d_acc, T * 8));
    for (auto &p : ctx->d_th) HIPCHK(hipMalloc(&p, T * 8));
    for (auto &p : ctx->d_tmp) HIPCHK(hipMalloc(&p, T * 8));
    HIPCHK(hipMalloc(&ctx->d_itmp, T * 4));
    HIPCHK(hipMemset(ctx->d_acc, 0, T * 8));
    ctx->have_structure = true;
    return EMSAR_HIP_OK;
}

int emsar_hip_upload_sample(emsar_hip_ctx *ctx, const int32_t *row_weight, const double *row_E, const double *den) {
    if (!ctx) return EMSAR_HIP_ERR_ARG;
    if (!ctx->have_structure) return EMSAR_HIP_ERR_STATE;
    HIPCHK(hipSetDevice(ctx->device));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    const int64_t n_rows = ctx->n_rows;
    // arguments first: a rejected call changes nothing
    if (row_weight || row_E) {
        for (int64_t r = 0; r < n_rows; r++) {
            if (row_weight && row_weight[r] < 0) return EMSAR_HIP_ERR_ARG;
            if (row_E && !(row_E[r] >= 0.0)) return EMSAR_HIP_ERR_ARG;  // negative or NaN
        }
    }
    if (den) for (int32_t t = 0; t < ctx->n_tx; t++) if (!(den[t] >= 0.0)) return EMSAR_HIP_ERR_ARG;
    // from here on the previous sample is gone: a call that fails half way (out of memory, HIP error) must not leave a
    // context that still says have_sample with its weight arrays freed (run_passes would launch kernels on null pointers)
    ctx->have_sample = false;
    // a row counts w = R (or 1) when it is inside the likelihood (E != 0), else 0
    ctx->weighted = (row_weight != nullptr) || (row_E != nullptr) || (ctx->layout == EMSAR_LAYOUT_TILED && ctx->TL.merged);
    ctx->loglik_const = 0.0;
    dfree(ctx->d_wgt); ctx->d_wgt = nullptr;
    auto weight_of = [&](int64_t r) -> int32_t {
        int32_t x = row_weight ? row_weight[r] : 1;
        if (row_E && row_E[r] == 0.0) x = 0;
        return x;
    };
    free_sets(ctx);
    try {
        ctx->h_wgt.resize((size_t)n_rows);
        int64_t total_w = 0;
        for (int64_t r = 0; r < n_rows; r++) { const int32_t x = weight_of(r); ctx->h_wgt[(size_t)r] = x; total_w += x; }
        // deterministic mode: no transcript is assigned more reads than the sample holds, |sum R log S| <= N * 745
        int e_mass = 0, e_ll = 0;
        (void)std::frexp((double)total_w + 1.0, &e_mass);
        (void)std::frexp(((double)total_w + 1.0) * 1024.0, &e_ll);
        ctx->fx_mass = std::ldexp(1.0, 61 - e_mass);
        ctx->fx_ll = std::ldexp(1.0, 61 - e_ll);
    } catch (const std::bad_alloc &) { return EMSAR_HIP_ERR_OOM; }
    if (ctx->layout == EMSAR_LAYOUT_TILED) {
        const auto &L = ctx->TL;
        dfree(ctx->d_left_wgt); ctx->d_left_wgt = nullptr;
        std::vector<double> u((size_t)ctx->n_tx, 0.0);
        for (size_t i = 0; i < L.single_row.size(); i++) {
            int32_t x = weight_of(L.single_row[i]);
            u[(size_t)L.single_tid[i]] += (double)x;
            if (x > 0 && row_E) ctx->loglik_const += (double)x * std::log(row_E[L.single_row[i]]);
        }
        HIPCHK(hipMemcpy(ctx->d_u, u.data(), u.size() * 8, hipMemcpyHostToDevice));
        if (ctx->weighted) {
            std::vector<int32_t> w((size_t)std::max<int64_t>(ctx->n_slots, 1), 0), wl((size_t)std::max<int64_t>(ctx->n_left, 1), 0);
            for (int64_t i = 0; i < ctx->n_slots; i++) {
                int64_t r = L.slot_row[(size_t)i];
                if (r < 0) continue;
                if (L.merged) {                                   // a slot stands for all rows with this tid multiset
                    int64_t sum = 0;
                    for (uint64_t q = L.mem_ptr[(size_t)r]; q < L.mem_ptr[(size_t)r + 1]; q++) {
                        int64_t o = L.mem_row[(size_t)q];
                        int32_t x = weight_of(o);
                        sum += x;
                        if (x > 0 && row_E) ctx->loglik_const += (double)x * std::log(row_E[o]);
                    }
                    if (sum > INT32_MAX) return EMSAR_HIP_ERR_ARG;
                    w[(size_t)i] = (int32_t)sum;
                    continue;
                }
                int32_t x = weight_of(r);
                w[(size_t)i] = x;
                if (x > 0 && row_E) ctx->loglik_const += (double)x * std::log(row_E[r]);
            }
            for (int64_t i = 0; i < ctx->n_left; i++) {
                int64_t r = L.left_row[(size_t)i];
                int32_t x = weight_of(r);
                wl[(size_t)i] = x;
                if (x > 0 && row_E) ctx->loglik_const += (double)x * std::log(row_E[r]);
            }
            HIPCHK(hipMalloc(&ctx->d_wgt, w.size() * 4));
            HIPCHK(hipMemcpy(ctx->d_wgt, w.data(), w.size() * 4, hipMemcpyHostToDevice));
            HIPCHK(hipMalloc(&ctx->d_left_wgt, wl.size() * 4));
            HIPCHK(hipMemcpy(ctx->d_left_wgt, wl.data(), wl.size() * 4, hipMemcpyHostToDevice));
        }
    } else if (ctx->weighted) {
        std::vector<int32_t> w(std::max<size_t>((size_t)n_rows, 1), 0);
        for (int64_t r = 0; r < n_rows; r++) {
            int32_t x = row_weight ? row_weight[r] : 1;
            if (row_E && row_E[r] == 0.0) x = 0;
            w[(size_t)r] = x;
            if (x > 0 && row_E) ctx->loglik_const += (double)x * std::log(row_E[r]);
        }
        HIPCHK(hipMalloc(&ctx->d_wgt, w.size() * 4));
        HIPCHK(hipMemcpy(ctx->d_wgt, w.data(), w.size() * 4, hipMemcpyHostToDevice));
    }
    if (den) {
        std::vector<double> tmp;
        try { den = to_lib(ctx, den, tmp); } catch (const std::bad_alloc &) { return EMSAR_HIP_ERR_OOM; }
        HIPCHK(hipMemcpy(ctx->d_den, den, (size_t)ctx->n_tx * 8, hipMemcpyHostToDevice));
    } else {
        std::vector<double> ones;
        const double *e = row_E;
        if (!e) { ones.assign((size_t)std::max<int64_t>(n_rows, 1), 1.0); e = ones.data(); }
        int rc = scatter_rows(ctx, e, ctx->d_den);
        if (rc) return rc;
    }
    ctx->bytes_formula = 4 * ctx->nnz + (ctx->ptr64 ? 8 : 4) * (ctx->n_rows + 1) + (row_weight ? 4 : 0) * ctx->n_rows + 32 * (int64_t)ctx->n_tx;
    ctx->have_sample = true;
    return emsar_hip_reset_theta(ctx);
}

int emsar_hip_reset_theta(emsar_hip_ctx *ctx) {
    if (!ctx) return EMSAR_HIP_ERR_ARG;
    if (!ctx->have_sample) return EMSAR_HIP_ERR_STATE;
    HIPCHK(hipSetDevice(ctx->device));
    hipLaunchKernelGGL(k_fill_start, dim3(grid_for(ctx->n_tx, 256)), dim3(256), 0, ctx->stream, ctx->n_tx, ctx->d_den, ctx->d_th[0]);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return EMSAR_HIP_OK;
}

int emsar_hip_set_theta(emsar_hip_ctx *ctx, const double *theta) {
    if (!ctx || !theta) return EMSAR_HIP_ERR_ARG;
    if (!ctx->have_sample) return EMSAR_HIP_ERR_STATE;
    HIPCHK(hipSetDevice(ctx->device));
    std::vector<double> tmp;
    try { theta = to_lib(ctx, theta, tmp); } catch (const std::bad_alloc &) { return EMSAR_HIP_ERR_OOM; }
    HIPCHK(hipMemcpyAsync(ctx->d_th[0], theta, (size_t)ctx->n_tx * 8, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return EMSAR_HIP_OK;
}

int emsar_hip_get_theta(emsar_hip_ctx *ctx, double *theta) {
    if (!ctx || !theta) return EMSAR_HIP_ERR_ARG;
    if (!ctx->have_sample) return EMSAR_HIP_ERR_STATE;
    HIPCHK(hipSetDevice(ctx->device));
    HIPCHK(hipMemcpyAsync(theta, ctx->d_th[0], (size_t)ctx->n_tx * 8, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    try { from_lib(ctx, theta); } catch (const std::bad_alloc &) { return EMSAR_HIP_ERR_OOM; }
    return EMSAR_HIP_OK;
}

int emsar_hip_run_passes(emsar_hip_ctx *ctx, int32_t n_passes, float *elapsed_ms, double *last_ll) {
    if (!ctx || n_passes < 0) return EMSAR_HIP_ERR_ARG;
    if (!ctx->have_sample) return EMSAR_HIP_ERR_STATE;
    HIPCHK(hipSetDevice(ctx->device));
    ctx->delta_mask = nullptr;
    hipLaunchKernelGGL(k_cycle_begin, dim3(1), dim3(kLlSlots), 0, ctx->stream, ctx->d_scal, 0.0, 0);
    HIPCHK(hipEventRecord(ctx->ev0, ctx->stream));
    int cur = 0;  // th[cur] holds the current point, th[cur^1] receives the next
    for (int i = 0; i < n_passes; i++) {
        bool ll = last_ll && i == n_passes - 1;
        int rc = em_pass(ctx, ctx->d_th[cur], ctx->d_th[cur ^ 1], ll, 0, 1e-6);
        if (rc) return rc;
        cur ^= 1;
    }
    HIPCHK(hipEventRecord(ctx->ev1, ctx->stream));
    if (cur == 1) HIPCHK(hipMemcpyAsync(ctx->d_th[0], ctx->d_th[1], (size_t)ctx->n_tx * 8, hipMemcpyDeviceToDevice, ctx->stream));
    HIPCHK(hipMemcpyAsync(ctx->h_scal, ctx->d_scal, sizeof(Scal), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    if (elapsed_ms) HIPCHK(hipEventElapsedTime(elapsed_ms, ctx->ev0, ctx->ev1));
    if (last_ll) *last_ll = host_ll(ctx, 0);
    return EMSAR_HIP_OK;
}

static int solve_impl(emsar_hip_ctx *ctx, const emsar_em_params *pp, double *fpkm_out, emsar_em_stats *stats);
int emsar_hip_solve(emsar_hip_ctx *ctx, const emsar_em_params *pp, double *fpkm_out, emsar_em_stats *stats) {
    if (!ctx || !fpkm_out) return EMSAR_HIP_ERR_ARG;
    if (!ctx->have_sample) return EMSAR_HIP_ERR_STATE;
    const int rc = solve_impl(ctx, pp, fpkm_out, stats);
    ctx->count_floor = 0.0; ctx->zero_cut = 0.0; ctx->delta_mask = nullptr;      // per-solve state, whatever the exit
    return rc;
}
static int solve_impl(emsar_hip_ctx *ctx, const emsar_em_params *pp, double *fpkm_out, emsar_em_stats *stats) {
    emsar_em_params p = pp ? *pp : emsar_em_params{0, 1, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    if (p.max_iter <= 0) p.max_iter = 100000;
    if (p.tol <= 0) p.tol = 1e-10;
    if (p.abs_floor <= 0) p.abs_floor = 1e-6;
    if (p.check_every <= 0) p.check_every = 8;
    if (!(p.count_floor >= 0.0)) return EMSAR_HIP_ERR_ARG;
    if (p.set_mode != 0 && p.set_mode != 1) return EMSAR_HIP_ERR_ARG;
    ctx->count_floor = p.count_floor;
    ctx->zero_cut = p.zero_cut > 0.0 ? p.zero_cut : 0.0;
    const double abs_step_base = p.abs_step > 0.0 ? p.abs_step : 0.0;
    ctx->delta_mask = nullptr;
    HIPCHK(hipSetDevice(ctx->device));
    int rc;
    bool use_sets = p.set_mode == 0;
    if (use_sets && (rc = ensure_sets(ctx))) return rc;
    if (use_sets && ctx->RS.giant) use_sets = false;      // one component holds most transcripts: plain streaming solve
    // the streaming passes run when asked for, or for the sets that do not fit a workgroup
    const bool need_stream = !use_sets || ctx->RS.n_streamed_sets > 0;
    if (use_sets && need_stream) ctx->delta_mask = ctx->d_kind;
    auto t0 = std::chrono::steady_clock::now();
    if ((rc = emsar_hip_reset_theta(ctx))) return rc;
    hipLaunchKernelGGL(k_scal_init, dim3(1), dim3(1), 0, ctx->stream, ctx->d_scal);
    HIPCHK(hipEventRecord(ctx->ev0, ctx->stream));
    const int n = ctx->n_tx, g = grid_for(n, 256);
    double **th = ctx->d_th;  // 0:th0 1:th1 2:th2 3:thx 4:thn (enqueue_cycles leaves the current point in th[0])
    int iters = 0, converged = need_stream ? 0 : 1, cycles = 0;
    double delta = need_stream ? INFINITY : 0.0;
    // The first 4 x check_every cycles are launched kernel by kernel (a quick solve never pays for a graph); after that
    // check_every cycles are recorded once into a hipGraph and replayed between the host's looks at the stopping rule.
    // Measured gain: 1-5 % on problems of 40 k .. 2 M rows (tools/graph_bench.py) -- the launches were already asynchronous,
    // and a pass of a small problem costs one workgroup's latency (12-26 us), not its launch.
    const int per_cycle = p.accel ? 3 : 1;
    const bool graph_ok = ctx->use_graph && (p.accel || p.check_every % 2 == 0);   // plain EM swaps th0/th1: an even count restores them
    CycleGraph G;
    ctx->graph_launches = 0;
    while (need_stream && iters < p.max_iter) {
        int todo = 1;
        if (graph_ok && cycles >= 4 * p.check_every && cycles % p.check_every == 0 &&
            (int64_t)iters + (int64_t)per_cycle * p.check_every <= (int64_t)p.max_iter) {
            todo = p.check_every;
            if (!G.exec) {
                HIPCHK(hipStreamBeginCapture(ctx->stream, hipStreamCaptureModeThreadLocal));
                rc = enqueue_cycles(ctx, p, abs_step_base, todo);
                hipError_t e = hipStreamEndCapture(ctx->stream, &G.graph);      // always closes the capture
                if (rc) return rc;
                HIPCHK(e);
                HIPCHK(hipGraphInstantiate(&G.exec, G.graph, nullptr, nullptr, 0));
            }
            HIPCHK(hipGraphLaunch(G.exec, ctx->stream));
            ctx->graph_launches++;
        } else if ((rc = enqueue_cycles(ctx, p, abs_step_base, 1))) return rc;
        cycles += todo;
        iters += todo * per_cycle;
        if (cycles % p.check_every == 0 || iters >= p.max_iter) {
            HIPCHK(hipMemcpyAsync(ctx->h_scal, ctx->d_scal, sizeof(Scal), hipMemcpyDeviceToHost, ctx->stream));
            HIPCHK(hipStreamSynchronize(ctx->stream));
            unsigned long long bits = p.accel ? ctx->h_scal->delta1_bits : ctx->h_scal->delta_bits;
            memcpy(&delta, &bits, 8);
            if (!std::isfinite(delta) || ctx->h_scal->bad) { ctx->err = "non-finite theta"; return EMSAR_HIP_ERR_NUMERIC; }
            if (delta < p.tol) { converged = 1; break; }
        }
    }
    ctx->delta_mask = nullptr;
    HIPCHK(hipEventRecord(ctx->ev1, ctx->stream));
    if (use_sets) {
        // zero_cut / abs_step exist because a boundary optimum is approached like 1/k by the EM; the sets that get Newton steps reach it
        // in a few steps and are held to the strict rule (same pass counts with and without the two rules on every problem measured,
        // and then nothing is printed differently); the rules stay in force for the streamed part and with newton_after < 0
        const bool strict_sets = p.newton_after >= 0;
        SetSolveParams P{p.tol, p.abs_floor, p.count_floor, (!strict_sets && p.zero_cut > 0.0) ? p.zero_cut : 0.0,
                         (!strict_sets && p.abs_step > 0.0) ? p.abs_step : 0.0, p.max_iter, p.accel, p.newton_after == 0 ? 60 : p.newton_after};
        // the cluster solver has no Newton step: its sets keep the two print-quantum rules whatever newton_after says (with the strict rule
        // alone a boundary optimum keeps a cluster going for 10^5 passes at ~32 us each)
        SetSolveParams Pc = P;
        Pc.zero_cut = p.zero_cut > 0.0 ? p.zero_cut : 0.0;
        Pc.abs_step = p.abs_step > 0.0 ? p.abs_step : 0.0;
        if ((rc = solve_resident_sets(ctx, P, Pc, th[0]))) return rc;
        if (ctx->n_sstat > 0)
            HIPCHK(hipMemcpyAsync(ctx->h_sstat, ctx->d_sstat, (size_t)ctx->n_sstat * sizeof(SetStat), hipMemcpyDeviceToHost, ctx->stream));
        if (ctx->n_cstat > 0)
            HIPCHK(hipMemcpyAsync(ctx->h_cstat, ctx->d_cstat, (size_t)ctx->n_cstat * sizeof(ClusterStat), hipMemcpyDeviceToHost, ctx->stream));
    }
    HIPCHK(hipEventRecord(ctx->ev2, ctx->stream));
    // F at the returned point: one likelihood-only pass (not counted in iters)
    hipLaunchKernelGGL(k_cycle_begin, dim3(1), dim3(kLlSlots), 0, ctx->stream, ctx->d_scal, 0.0, 0);
    if ((rc = launch_pass(ctx, MODE_EM_LL, th[0], ctx->d_acc, &ctx->d_scal->ll[0].s[0].v))) return rc;
    HIPCHK(hipMemsetAsync(ctx->d_acc, 0, (size_t)n * 8, ctx->stream));
    hipLaunchKernelGGL(k_dot, dim3(1), dim3(1024), 0, ctx->stream, n, th[0], ctx->d_den, &ctx->d_scal->ll[3].s[0].v);
    HIPCHK(hipMemcpyAsync(ctx->h_scal, ctx->d_scal, sizeof(Scal), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipMemcpyAsync(fpkm_out, th[0], (size_t)n * 8, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    try { from_lib(ctx, fpkm_out); } catch (const std::bad_alloc &) { return EMSAR_HIP_ERR_OOM; }
    ctx->count_floor = 0.0; ctx->zero_cut = 0.0;
    if (getenv("EMSAR_HIP_DEBUG"))
        fprintf(stderr, "emsar_hip_solve: %d streaming passes, %lld graph replays of %d cycles\n", iters, (long long)ctx->graph_launches, p.check_every);
    for (int32_t t = 0; t < n; t++)
        if (!std::isfinite(fpkm_out[t])) { ctx->err = "non-finite theta"; return EMSAR_HIP_ERR_NUMERIC; }
    int32_t set_max = 0, set_unconv = 0;
    int64_t set_sum = 0;
    if (use_sets)
        for (int64_t i = 0; i < ctx->n_sstat; i++) {
            const SetStat &q = ctx->h_sstat[i];
            set_max = std::max(set_max, q.passes); set_sum += q.passes;
            if (!q.converged) set_unconv++;
            if (!std::isfinite(q.delta)) { ctx->err = "non-finite theta in a connected set"; return EMSAR_HIP_ERR_NUMERIC; }
            if (q.delta > delta) delta = q.delta;
        }
    int32_t cl_max = 0;
    if (use_sets)
        for (int64_t i = 0; i < ctx->n_cstat; i++) {
            const ClusterStat &q = ctx->h_cstat[i];
            if (q.aborted) { ctx->err = "a workgroup cluster gave up waiting at its barrier"; return EMSAR_HIP_ERR_HIP; }
            cl_max = std::max(cl_max, q.passes); set_sum += q.passes;
            if (!q.converged) set_unconv++;
            if (!std::isfinite(q.delta)) { ctx->err = "non-finite theta in a connected set"; return EMSAR_HIP_ERR_NUMERIC; }
            if (q.delta > delta) delta = q.delta;
        }
    set_max = std::max(set_max, cl_max);
    if (set_unconv) converged = 0;
    if (stats) {
        float ms = 0, ms_sets = 0;
        HIPCHK(hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1));
        HIPCHK(hipEventElapsedTime(&ms_sets, ctx->ev1, ctx->ev2));
        memset(stats, 0, sizeof(*stats));
        stats->iters = iters + set_max;
        stats->converged = converged;
        stats->final_delta = delta;
        stats->loglik = host_ll(ctx, 0) + ctx->loglik_const - ctx->h_scal->ll[3].s[0].v;
        stats->kernel_ms = ms + (use_sets ? ms_sets : 0.0f);
        stats->solve_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        stats->bytes_per_pass = ctx->bytes_formula;
        stats->stored_bytes_per_pass = stored_bytes(ctx);
        if (p.set_mode == 0 && ctx->RS.giant) { stats->sets_streamed = 1; stats->sets_build_ms = ctx->sets_build_ms; }
        if (use_sets) {
            stats->sets_resident = (int32_t)ctx->RS.n_resident();
            stats->sets_streamed = (int32_t)ctx->RS.n_streamed_sets;
            stats->set_passes_max = set_max;
            stats->sets_unconverged = set_unconv;
            stats->set_passes_sum = set_sum;
            stats->sets_build_ms = ctx->sets_build_ms;
            stats->sets_kernel_ms = ms_sets;
            stats->sets_cluster = (int32_t)ctx->n_cstat;
            stats->cluster_passes_max = cl_max;
            if (ctx->n_cstat > 0) { float mc = 0; HIPCHK(hipEventElapsedTime(&mc, ctx->ev_c0, ctx->ev_c1)); stats->cluster_kernel_ms = mc; }
        }
    }
    return EMSAR_HIP_OK;
}

int emsar_hip_ieuma(emsar_hip_ctx *ctx, const double *row_L, double *ieuma_out) {
    if (!ctx || !row_L || !ieuma_out) return EMSAR_HIP_ERR_ARG;
    if (!ctx->have_structure) return EMSAR_HIP_ERR_STATE;
    HIPCHK(hipSetDevice(ctx->device));
    int rc = scatter_rows(ctx, row_L, ctx->d_tmp[0]);
    if (rc) return rc;
    HIPCHK(hipMemcpy(ieuma_out, ctx->d_tmp[0], (size_t)ctx->n_tx * 8, hipMemcpyDeviceToHost));
    try { from_lib(ctx, ieuma_out); } catch (const std::bad_alloc &) { return EMSAR_HIP_ERR_OOM; }
    return EMSAR_HIP_OK;
}

int emsar_hip_normalise(emsar_hip_ctx *ctx, const double *mean_fpkm, const double *ieuma, int64_t total_read_count,
                        double *tpm_out, double *ir_out, int32_t *iri_out) {
    if (!ctx || !mean_fpkm || !ieuma || !tpm_out || !ir_out || !iri_out) return EMSAR_HIP_ERR_ARG;
    if (!ctx->have_structure) return EMSAR_HIP_ERR_STATE;
    HIPCHK(hipSetDevice(ctx->device));
    const int n = ctx->n_tx, g = grid_for(n, 256);
    const size_t B = (size_t)n * 8;
    HIPCHK(hipMemcpyAsync(ctx->d_tmp[0], mean_fpkm, B, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipMemcpyAsync(ctx->d_tmp[1], ieuma, B, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipMemsetAsync(&ctx->d_scal->sum_b, 0, 8, ctx->stream));
    hipLaunchKernelGGL(k_sum, dim3(1), dim3(1024), 0, ctx->stream, n, ctx->d_tmp[0], &ctx->d_scal->sum_b);
    // tmp[2] <- tpm, acc <- iReadcount (acc is zero between passes and is cleared again below)
    hipLaunchKernelGGL(k_normalise, dim3(g), dim3(256), 0, ctx->stream, n, ctx->d_tmp[0], ctx->d_tmp[1],
                       (double)total_read_count / 1E6, &ctx->d_scal->sum_b, ctx->d_tmp[2], ctx->d_acc, ctx->d_itmp);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(tpm_out, ctx->d_tmp[2], B, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipMemcpyAsync(ir_out, ctx->d_acc, B, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipMemcpyAsync(iri_out, ctx->d_itmp, (size_t)n * 4, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipMemsetAsync(ctx->d_acc, 0, B, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return EMSAR_HIP_OK;
}

int emsar_hip_upload_euma(emsar_hip_ctx *ctx, const int32_t *euma, int32_t nfl) {
    if (!ctx || !euma || nfl <= 0) return EMSAR_HIP_ERR_ARG;
    if (!ctx->have_structure) return EMSAR_HIP_ERR_STATE;
    HIPCHK(hipSetDevice(ctx->device));
    dfree(ctx->d_euma_t); dfree(ctx->d_wf); dfree(ctx->d_adj); ctx->d_euma_t = nullptr; ctx->d_wf = ctx->d_adj = nullptr; ctx->nfl = 0;
    const size_t n = (size_t)ctx->n_rows * (size_t)nfl;
    int32_t *tmp = nullptr;
    HIPCHK(hipMalloc(&ctx->d_euma_t, std::max<size_t>(n, 1) * 4));
    HIPCHK(hipMalloc(&ctx->d_wf, (size_t)nfl * 8));
    HIPCHK(hipMalloc(&ctx->d_adj, std::max<size_t>((size_t)ctx->n_rows, 1) * 8));
    if (n) {
        HIPCHK(hipMalloc(&tmp, n * 4));
        hipError_t e = hipMemcpyAsync(tmp, euma, n * 4, hipMemcpyHostToDevice, ctx->stream);
        if (e == hipSuccess) {
            dim3 grid((unsigned)((ctx->n_rows + 63) / 64), (unsigned)((nfl + 63) / 64));
            hipLaunchKernelGGL(k_transpose_i32, grid, dim3(256), 0, ctx->stream, ctx->n_rows, (int)nfl, tmp, ctx->d_euma_t);
            e = hipGetLastError();
        }
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
        (void)hipFree(tmp);
        HIPCHK(e);
    }
    ctx->nfl = nfl;
    return EMSAR_HIP_OK;
}

int emsar_hip_adj_euma(emsar_hip_ctx *ctx, const double *wf, double *out) {
    if (!ctx || !wf || !out) return EMSAR_HIP_ERR_ARG;
    if (!ctx->have_structure || ctx->nfl <= 0) return EMSAR_HIP_ERR_STATE;
    HIPCHK(hipSetDevice(ctx->device));
    if (ctx->n_rows == 0) return EMSAR_HIP_OK;
    HIPCHK(hipMemcpyAsync(ctx->d_wf, wf, (size_t)ctx->nfl * 8, hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(k_adj_euma, dim3((unsigned)((ctx->n_rows + 255) / 256)), dim3(256), 0, ctx->stream, ctx->n_rows, (int)ctx->nfl,
                       ctx->d_euma_t, ctx->d_wf, ctx->d_adj);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(out, ctx->d_adj, (size_t)ctx->n_rows * 8, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return EMSAR_HIP_OK;
}

int emsar_hip_get_info(const emsar_hip_ctx *ctx, emsar_hip_info *o) {
    if (!ctx || !o) return EMSAR_HIP_ERR_ARG;
    if (!ctx->have_structure) return EMSAR_HIP_ERR_STATE;
    memset(o, 0, sizeof(*o));
    o->n_rows = ctx->n_rows; o->nnz = ctx->nnz; o->n_tx = ctx->n_tx; o->device_id = ctx->device;
    o->layout = ctx->layout | ((ctx->layout == EMSAR_LAYOUT_TILED && ctx->TL.merged) ? EMSAR_LAYOUT_FLAG_MERGE_ROWS : 0);
    if (ctx->layout == EMSAR_LAYOUT_TILED) {
        o->n_chunks = ctx->n_tiles; o->n_slices = ctx->tl_n_fslices; o->padded_entries = ctx->tl_fwd_slots;
        o->far_entries = ctx->TL.far_entries; o->window = emsar::kTileDict;
        o->tiled_entries = ctx->TL.tiled_entries; o->tiled_ids = ctx->TL.tiled_ids; o->renumbered = ctx->TL.renum.applied ? 1 : 0;
        o->n_units = ctx->n_units;
    }
    o->bytes_per_pass = ctx->bytes_formula;
    o->stored_bytes_per_pass = stored_bytes(ctx);
    return EMSAR_HIP_OK;
}

// Diagnostic only (not declared in the public header): one stamped pass of the TILED kernel on the current theta.
// out[0..6] = mean cycles per wave spent in: loads issued + dictionary, barrier, E-step, barrier, M-step, barrier, flush;
// out[7] = tiles.  The result vector theta is left untouched (acc is cleared again).
int emsar_hip_debug_tiled_stamps(emsar_hip_ctx *ctx, double *out) {
    if (!ctx || !out || ctx->layout != EMSAR_LAYOUT_TILED || !ctx->have_sample || ctx->weighted || ctx->n_tiles == 0) return EMSAR_HIP_ERR_STATE;
    HIPCHK(hipSetDevice(ctx->device));
    unsigned long long *d = nullptr;
    const size_t nw = (size_t)ctx->n_tiles * (kTiledThreads / 64), bytes = nw * 8 * sizeof(unsigned long long);
    HIPCHK(hipMalloc(&d, bytes));
    HIPCHK(hipMemsetAsync(d, 0, bytes, ctx->stream));
    const size_t lds = (size_t)kTiledLdsDoubles * sizeof(double);
    HIPCHK(hipFuncSetAttribute((const void *)k_pass_tiled<false, MODE_EM, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL((k_pass_tiled<false, MODE_EM, true>), dim3((unsigned)ctx->n_tiles), dim3(kTiledThreads), lds, ctx->stream, ctx->d_tiles,
                       ctx->d_fwd, ctx->d_bwd, ctx->d_far, ctx->d_wgt, ctx->d_rowval, ctx->d_th[0], ctx->d_acc, &ctx->d_scal->ll[3].s[0].v, Fx{0.0, 0.0}, d);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemsetAsync(ctx->d_acc, 0, (size_t)ctx->n_tx * 8, ctx->stream));
    std::vector<unsigned long long> h(nw * 8);
    HIPCHK(hipMemcpyAsync(h.data(), d, bytes, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    dfree(d);
    for (int i = 0; i < 7; i++) {
        double sum = 0;
        for (size_t w = 0; w < nw; w++) sum += (double)h[w * 8 + (size_t)i];
        out[i] = sum / (double)nw;   // mean cycles per wave
    }
    out[7] = (double)ctx->n_tiles;
    return EMSAR_HIP_OK;
}

// The same for the unit kernel (the one config 3 runs): out[0..5] = mean cycles per wave in: descriptor + dictionary + first loads,
// barrier, E-steps, M-steps, barrier, flush; out[6] = tiles per unit; out[7] = units.
int emsar_hip_debug_unit_stamps(emsar_hip_ctx *ctx, double *out, unsigned long long *timeline /* NULL or 4 words per unit: start, end (100 MHz ticks), place, tiles */) {
    if (!ctx || !out || ctx->layout != EMSAR_LAYOUT_TILED || !ctx->have_sample || ctx->weighted || ctx->n_units == 0) return EMSAR_HIP_ERR_STATE;
    HIPCHK(hipSetDevice(ctx->device));
    unsigned long long *d = nullptr;
    const size_t nw = (size_t)ctx->n_units * (kTiledThreads / 64), bytes = (nw * 8 + (size_t)ctx->n_units * 4) * sizeof(unsigned long long);
    HIPCHK(hipMalloc(&d, bytes));
    HIPCHK(hipMemsetAsync(d, 0, bytes, ctx->stream));
    const size_t lds = (size_t)kTiledLdsDoubles * sizeof(double);
    HIPCHK(hipFuncSetAttribute((const void *)k_pass_tiled_unit<false, MODE_EM, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL((k_pass_tiled_unit<false, MODE_EM, true>), dim3((unsigned)ctx->n_units), dim3(kTiledThreads), lds, ctx->stream, ctx->d_utiles, ctx->unit_stride,
                       ctx->d_far, ctx->d_fwd, ctx->d_bwd, ctx->d_wgt, ctx->d_th[0], ctx->d_acc, &ctx->d_scal->ll[3].s[0].v, Fx{0.0, 0.0}, d);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemsetAsync(ctx->d_acc, 0, (size_t)ctx->n_tx * 8, ctx->stream));
    std::vector<unsigned long long> h(nw * 8 + (size_t)ctx->n_units * 4);
    HIPCHK(hipMemcpyAsync(h.data(), d, bytes, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    dfree(d);
    if (timeline) std::copy(h.begin() + (std::ptrdiff_t)(nw * 8), h.end(), timeline);
    for (int i = 0; i < 7; i++) {
        double sum = 0;
        for (size_t w = 0; w < nw; w++) sum += (double)h[w * 8 + (size_t)i];
        out[i] = sum / (double)nw;
    }
    out[7] = (double)ctx->n_units;
    return EMSAR_HIP_OK;
}

int emsar_hip_layout_selfcheck_tiled(int64_t n_rows, int32_t n_tx, const uint64_t *row_ptr, const int32_t *col_idx,
                                     int merge_rows, emsar_hip_info *info_out) {
    try {
        if (emsar::validate_csr(n_rows, n_tx, row_ptr, col_idx) != 0) return EMSAR_HIP_ERR_ARG;
        emsar::TiledLayout L;
        if (emsar::build_tiled(n_rows, n_tx, row_ptr, col_idx, L, merge_rows != 0) != 0) return EMSAR_HIP_ERR_ARG;
        int rc = emsar::check_tiled(L, row_ptr, col_idx);
        if (rc == 0) { emsar::UnitTables U; emsar::build_unit_tables(L, U); rc = emsar::check_unit_tables(L, U); }     // what k_pass_tiled_unit reads first
        if (info_out) {
            memset(info_out, 0, sizeof(*info_out));
            info_out->n_rows = n_rows; info_out->nnz = L.nnz; info_out->n_tx = n_tx;
            info_out->layout = EMSAR_LAYOUT_TILED | (L.merged ? EMSAR_LAYOUT_FLAG_MERGE_ROWS : 0);
            info_out->n_chunks = (int64_t)L.tiles.size();
            info_out->n_slices = L.n_fslices;
            info_out->padded_entries = L.padded_slots; info_out->far_entries = L.far_entries; info_out->window = emsar::kTileDict;
            info_out->stored_bytes_per_pass = (int64_t)L.fwd.size() * 4 + (int64_t)L.bwd.size() * 4 +
                                              (int64_t)L.far_tid.size() * 4 + (int64_t)L.tiles.size() * 64 + (int64_t)L.left_col.size() * 4;
            info_out->bytes_per_pass = (int64_t)L.single_row.size();   /* diagnostic: number of folded single-tid rows */
            info_out->tiled_entries = L.tiled_entries; info_out->tiled_ids = L.tiled_ids; info_out->renumbered = L.renum.applied ? 1 : 0;
            info_out->n_units = L.unit_first.empty() ? 0 : (int64_t)L.unit_first.size() - 1;
        }
        return rc == 0 ? EMSAR_HIP_OK : EMSAR_HIP_ERR_ARG - 100 + rc;
    } catch (const std::bad_alloc &) { return EMSAR_HIP_ERR_OOM; }     // nothing may leave the C ABI as an exception
}

int emsar_hip_sets_selfcheck(int64_t n_rows, int32_t n_tx, const uint64_t *row_ptr, const int32_t *col_idx,
                             const int32_t *row_weight, emsar_hip_sets_info *o) {
    try {
        if (emsar::validate_csr(n_rows, n_tx, row_ptr, col_idx) != 0) return EMSAR_HIP_ERR_ARG;
        if (row_weight) for (int64_t r = 0; r < n_rows; r++) if (row_weight[r] < 0) return EMSAR_HIP_ERR_ARG;
        emsar::ResidentSets S;
        try {
            emsar::build_sets(n_rows, n_tx, row_ptr, col_idx, row_weight, S);
        } catch (const std::bad_alloc &) { return EMSAR_HIP_ERR_OOM; }
        int rc = emsar::check_sets(n_rows, n_tx, row_ptr, col_idx, row_weight, S);
        if (o) {
            memset(o, 0, sizeof(*o));
            o->n_components = S.n_components;
            for (int c = 0; c < emsar::kSetClasses; c++) { o->sets_resident[c] = (int64_t)S.desc[c].size(); o->max_lds_bytes[c] = (int64_t)S.max_lds[c]; }
            o->sets_streamed = S.n_streamed_sets;
            o->tids_closed = S.n_closed_tids; o->tids_resident = S.n_resident_tids; o->tids_streamed = S.n_streamed_tids;
            o->rows_in = S.rows_in; o->rows_stored = S.rows_stored + S.CL.rows_stored;
            o->sets_cluster = S.n_cluster_sets(); o->tids_cluster = S.CL.n_tids; o->max_lds_cluster = (int64_t)S.CL.max_lds;
        }
        return rc == 0 ? EMSAR_HIP_OK : EMSAR_HIP_ERR_ARG - 200 + rc;
    } catch (const std::bad_alloc &) { return EMSAR_HIP_ERR_OOM; }     // nothing may leave the C ABI as an exception
}

}  // extern "C"
