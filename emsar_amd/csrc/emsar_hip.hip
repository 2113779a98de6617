// emsar_hip.hip -- MI355X (gfx950 / CDNA4) abundance-estimation core behind include/emsar_hip.h.
//
// Replaces run_MLE_threads() (/root/reference/src/emsar_main.c:446; MLE/Fp/lambdap,
// emsar_functions.c:2946-3126) by an EM on the same segment Poisson likelihood:
//     E-step  w_c = R_c / S_c ,  S_c = sum_t m_ct theta_t        (rows with E_c == 0 are outside F)
//     M-step  theta_t <- theta_t * (sum_c m_ct w_c) / den_t ,    den_t = sum_c m_ct E_c
// and compute_iEUMA / the TPM + iReadcount arithmetic of print_FPKMfinal (emsar_functions.c:3176-3232).
//
// One pass is HBM-bound integer streaming plus FP64 adds: ~2 flop per nonzero -- no MFMA.
// Kernels:
//   k_pass_windowed   the hot one: one workgroup per chunk, theta/acc windows in LDS, one lane per row,
//                     column-major 64-row slices (256 contiguous bytes per wave load)
//   k_pass_csr        generic fallback on the caller's CSR, one lane per row, FP64 atomics to L2/HBM
//   k_update          theta' = theta*acc/den, clears acc, max-relative-change reduction
//   k_update_p2/p3, k_sq_extrap_ll   SQUAREM extrapolation / acceptance entirely on the device (no host round trip)
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

namespace {

using emsar::Chunk;
using emsar::Tile;
constexpr int kPassThreads = 512;     // 8 waves per workgroup
constexpr int kDefaultWindow = 4096;  // 2 x 32 KiB of LDS per workgroup -> 2 workgroups per CU
constexpr int64_t kChunkEntries = 65536;

// ------------------------------------------------------------------------------------------------
// device scalars of one solve (lives in HBM, polled by the host every check_every cycles)
// ------------------------------------------------------------------------------------------------
struct Scal {
    double ll[4];                 // sum_c R_c log S_c at the input of pass 0/1/2 of the cycle; [3] scratch
    double sr2, sv2, pen1, penx;  // SQUAREM norms, sum theta*den of th1 and of the extrapolated point
    double stepmax, s_used;
    unsigned long long delta_bits;  // max_t |dtheta|/(theta+floor) as IEEE bits (non-negative -> integer max)
    unsigned long long delta1_bits; // the same, frozen after the first (plain) pass of a SQUAREM cycle
    int32_t accepted, rejected;
    double sum_a, sum_b;          // generic reductions (normalise)
    long long passes;             // EM passes enqueued before the current cycle (k_cycle_begin keeps it: the cycle may replay from a hipGraph)
    double abs_step_cur;          // emsar_em_params.abs_step scaled to the pass count of the current cycle (0 = rule off)
};

__device__ __forceinline__ void atomic_add_f64(double *p, double v) {
    // gfx950: global_atomic_add_f64 / ds_add_f64 (no CAS loop; compiled with -munsafe-fp-atomics)
    __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void lds_add_f64(double *p, double v) {
    __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

__device__ __forceinline__ double wave_sum(double v) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

template <int THREADS>
__device__ __forceinline__ double block_sum(double v, double *red /* THREADS/64 doubles of LDS */) {
    v = wave_sum(v);
    int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) red[wave] = v;
    __syncthreads();
    double t = 0;
    if (threadIdx.x == 0)
        for (int i = 0; i < THREADS / 64; i++) t += red[i];
    return t;  // valid in thread 0
}

enum PassMode { MODE_EM = 0, MODE_EM_LL = 1, MODE_SCATTER = 2 };

// ------------------------------------------------------------------------------------------------
// k_pass_windowed: one EM pass (or a plain row-value scatter) over the WINDOWED layout.
//   chunks[blockIdx.x]  -> slices [slice_begin, +n_slices), LDS window [lo, lo+width)
//   a wave owns one 256-row slice at a time; lane l handles rows 4l..4l+3 of it; the j-th tids of those four
//   rows are ONE int4 at ent[slice_off + j*256 + 4l]  (1 KiB contiguous per wave load)
// Up to 8 loads (8 KiB per wave) are issued before the first use, the tids then stay in registers for both
// the E-step sums and the M-step adds; rows longer than 8 are streamed in segments of 8 and re-read (L2) for
// the adds.
// HBM traffic per pass: ent once (4 B per stored slot), slice_off (8 B per 256 rows), optional row weights;
// theta window loads and acc window flushes are O(n_tx + chunks*family) and stay in L2.
// ------------------------------------------------------------------------------------------------
constexpr int kSeg = 8;  // int4 loads in flight per lane

struct Window {
    const double *th_w; double *acc_w; const double *theta; double *acc; int lo; unsigned width;
    __device__ __forceinline__ double get(int t) const {
        unsigned d = (unsigned)(t - lo);
        return d < width ? th_w[d] : theta[t];
    }
    __device__ __forceinline__ void add(int t, double v) const {
        unsigned d = (unsigned)(t - lo);
        if (d < width) lds_add_f64(&acc_w[d], v);
        else atomic_add_f64(&acc[t], v);
    }
};

// The segment length n (1..8) is wave-uniform; each length gets its own straight-line code so that all n
// loads are issued back to back (n KiB in flight per wave) with no control flow between them.
template <int N>
__device__ __forceinline__ void load_n(int4 (&q)[N], const int4 *e) {
#pragma unroll
    for (int j = 0; j < N; j++) q[j] = e[(size_t)j * 64];
}

template <int N>
__device__ __forceinline__ void sum_n(const int4 (&q)[N], const Window &W, double (&S)[4]) {
#pragma unroll
    for (int j = 0; j < N; j++) {
        int4 t = q[j];
        if (t.x >= 0) S[0] += W.get(t.x);
        if (t.y >= 0) S[1] += W.get(t.y);
        if (t.z >= 0) S[2] += W.get(t.z);
        if (t.w >= 0) S[3] += W.get(t.w);
    }
}

// M-step adds of one segment.  The four rows of a lane are neighbours in the sorted order and often carry
// the same tid in column j: equal neighbours are merged in registers first, so that one LDS atomic carries
// up to four contributions (an LDS f64 atomic costs ~6 cycles per extra lane on the same address).
template <int N>
__device__ __forceinline__ void add_n(const int4 (&q)[N], const Window &W, const double (&w)[4]) {
#pragma unroll
    for (int j = 0; j < N; j++) {
        int4 t = q[j];
        double v0 = t.x >= 0 ? w[0] : 0.0, v1 = t.y >= 0 ? w[1] : 0.0;
        double v2 = t.z >= 0 ? w[2] : 0.0, v3 = t.w >= 0 ? w[3] : 0.0;
        if (t.y == t.x) { v1 += v0; v0 = 0.0; }
        if (t.z == t.y) { v2 += v1; v1 = 0.0; }
        if (t.w == t.z) { v3 += v2; v2 = 0.0; }
        if (v0 != 0.0) W.add(t.x, v0);
        if (v1 != 0.0) W.add(t.y, v1);
        if (v2 != 0.0) W.add(t.z, v2);
        if (v3 != 0.0) W.add(t.w, v3);
    }
}

template <bool WEIGHTED, int MODE>
__device__ __forceinline__ void row_weights(const double (&S)[4], const int32_t *wgt, uint64_t row0, double (&w)[4], double &ll) {
    double r[4] = {1.0, 1.0, 1.0, 1.0};
    if (WEIGHTED) {
        int4 rw = *reinterpret_cast<const int4 *>(wgt + row0);
        r[0] = rw.x; r[1] = rw.y; r[2] = rw.z; r[3] = rw.w;
    }
#pragma unroll
    for (int i = 0; i < 4; i++) {
        bool live = (S[i] > 0.0) && (r[i] > 0.0);
        w[i] = live ? r[i] / S[i] : 0.0;
        if (MODE == MODE_EM_LL && live) ll += r[i] * log(S[i]);
    }
}

__device__ __forceinline__ void scatter_weights(const double *rowval, uint64_t row0, double (&w)[4]) {
    const double2 *rv = reinterpret_cast<const double2 *>(rowval + row0);
    double2 a = rv[0], b = rv[1];
    w[0] = a.x; w[1] = a.y; w[2] = b.x; w[3] = b.y;
}

// A slice whose rows have at most 8 tids: all N loads (N KiB per wave) are issued back to back, the tids stay
// in registers for the E-step sums and the M-step adds.  One straight-line instance per N.
template <int N, bool WEIGHTED, int MODE>
__device__ __forceinline__ void slice_short(const int4 *e, uint64_t row0, const Window &W, const int32_t *wgt,
                                            const double *rowval, double &ll) {
    int4 q[N];
    double w[4];
    load_n<N>(q, e);
    if (MODE == MODE_SCATTER) {
        scatter_weights(rowval, row0, w);
    } else {
        double S[4] = {0.0, 0.0, 0.0, 0.0};
        sum_n<N>(q, W, S);
        row_weights<WEIGHTED, MODE>(S, wgt, row0, w, ll);
    }
    add_n<N>(q, W, w);
}

// Rows with more than 8 tids: streamed in segments of 8 loads for the sums, re-read (L2) for the adds.
template <bool WEIGHTED, int MODE>
__device__ __forceinline__ void slice_long(const int4 *e, int k, uint64_t row0, const Window &W, const int32_t *wgt,
                                           const double *rowval, double &ll) {
    double w[4];
    const int nfull = k / kSeg, rem = k % kSeg;
    if (MODE == MODE_SCATTER) {
        scatter_weights(rowval, row0, w);
    } else {
        double S[4] = {0.0, 0.0, 0.0, 0.0};
        for (int g = 0; g < nfull; g++) {
            int4 q[kSeg];
            load_n<kSeg>(q, e + (size_t)g * kSeg * 64);
            sum_n<kSeg>(q, W, S);
        }
        for (int j = nfull * kSeg; j < k; j++) {
            int4 q[1];
            load_n<1>(q, e + (size_t)j * 64);
            sum_n<1>(q, W, S);
        }
        row_weights<WEIGHTED, MODE>(S, wgt, row0, w, ll);
    }
    for (int g = 0; g < nfull; g++) {
        int4 q[kSeg];
        load_n<kSeg>(q, e + (size_t)g * kSeg * 64);
        add_n<kSeg>(q, W, w);
    }
    for (int j = nfull * kSeg; j < k; j++) {
        int4 q[1];
        load_n<1>(q, e + (size_t)j * 64);
        add_n<1>(q, W, w);
    }
    (void)rem;
}

template <int THREADS, bool WEIGHTED, int MODE>
__global__ __launch_bounds__(THREADS) void k_pass_windowed(const Chunk *__restrict__ chunks,
                                                           const uint64_t *__restrict__ slice_off,
                                                           const int32_t *__restrict__ ent,
                                                           const int32_t *__restrict__ wgt,   // sorted rows, padded
                                                           const double *__restrict__ rowval, // MODE_SCATTER
                                                           const double *__restrict__ theta,
                                                           double *__restrict__ acc, double *__restrict__ ll_out,
                                                           int window) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    double *th_w = lds;            // [window]
    double *acc_w = lds + window;  // [window]
    __shared__ double red[THREADS / 64];

    const Chunk c = chunks[blockIdx.x];
    const int lo = c.lo, width = c.width;
    for (int i = threadIdx.x; i < width; i += THREADS) {
        if (MODE != MODE_SCATTER) th_w[i] = theta[lo + i];
        acc_w[i] = 0.0;
    }
    __syncthreads();
    const Window W{th_w, acc_w, theta, acc, lo, (unsigned)width};

    const int lane = threadIdx.x & 63;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // keep slice bookkeeping in SGPRs
    double ll = 0.0;
    for (uint32_t s = wave; s < c.n_slices; s += THREADS / 64) {
        const uint32_t gs = c.slice_begin + s;
        const uint64_t off = slice_off[gs];
        const int k = (int)((slice_off[gs + 1] - off) >> 8);
        const int4 *e = reinterpret_cast<const int4 *>(ent + off) + lane;
        const uint64_t row0 = (uint64_t)gs * 256 + 4 * lane;
#define EMSAR_SHORT(NN) case NN: slice_short<NN, WEIGHTED, MODE>(e, row0, W, wgt, rowval, ll); break;
        switch (k) {
            EMSAR_SHORT(1) EMSAR_SHORT(2) EMSAR_SHORT(3) EMSAR_SHORT(4)
            EMSAR_SHORT(5) EMSAR_SHORT(6) EMSAR_SHORT(7) EMSAR_SHORT(8)
            default: slice_long<WEIGHTED, MODE>(e, k, row0, W, wgt, rowval, ll); break;
        }
#undef EMSAR_SHORT
    }
    __syncthreads();
    for (int i = threadIdx.x; i < width; i += THREADS) {
        double v = acc_w[i];
        if (v != 0.0) atomic_add_f64(&acc[lo + i], v);
    }
    if (MODE == MODE_EM_LL) {
        double t = block_sum<THREADS>(ll, red);
        if (threadIdx.x == 0 && t != 0.0) atomic_add_f64(ll_out, t);
    }
}

// ------------------------------------------------------------------------------------------------
// k_pass_tiled: one EM pass over the TILED layout (layout_tiled.hpp).  One workgroup = 4 waves = the 4 slices of a tile.
//   phase 0  every global load the wave needs first is issued at once (dictionary theta values, 8 forward columns,
//            8 backward segments); dictionary: th_w[d] = theta[tid(d)], acc_w[d] = 0, zero slots; barrier
//   phase E  the wave owns one slice (768 rows; lane l holds rows 64*i + l, i < 12): S_r = sum th_w[id]  (LDS reads only, 10-bit ids,
//            padding reads the zero slot: no branches), w_r = R_r / S_r -> the wave's own 6 KiB of LDS
//   phase M  the SAME wave walks the transposed index of its rows: a lane's segments (column id + 11 row ids) are
//            consecutive in column order; it gathers w_r from LDS into a register sum and adds it to acc_w when the
//            column changes; tiny columns via a COO list.  No barrier between E and M.
//   phase F  barrier; non-zero dictionary slots are flushed with one global FP64 atomic each
// HBM traffic: 10 bits per forward slot + 128 bits per 11 backward entries -- no row_ptr, no 32-bit tids.
// ------------------------------------------------------------------------------------------------
constexpr int kTiledThreads = 256;                       // 4 wavefronts = 4 slices
constexpr int kRPL = emsar::kRowsPerLane;                 // 12 rows per lane = twelve 10-bit ids per int4
constexpr int kTiledWr = emsar::kTileSliceRows + 8;       // w_r of one slice (768) + the zero padding row
constexpr int kTiledDictPad = emsar::kTileDict + 1;       // 960 slots incl. the zero slot
constexpr int kTiledLdsDoubles = 2 * kTiledDictPad + emsar::kTileSlices * kTiledWr;   // 40,192 B: 4 workgroups per CU

__device__ __forceinline__ double lds_at(const double *base, unsigned byte_off) {
    return *reinterpret_cast<const double *>(reinterpret_cast<const char *>(base) + byte_off);
}
// field f (0,1,2) of a packed dword -> LDS byte offset of the double it names
__device__ __forceinline__ unsigned id_off(unsigned dword, int f) { return ((dword >> (10 * f)) & 0x3FFu) << 3; }

// 8 independent 16-byte loads; positions beyond n repeat position n-1 (an L1 hit) so that there is no control flow
// between the loads and all of them are in flight together
__device__ __forceinline__ void load8_clamped(int4 (&q)[8], const int4 *e, int n) {
#pragma unroll
    for (int j = 0; j < 8; j++) q[j] = e[(size_t)(j < n ? j : n - 1) * 64];
}

// LDS gather of the double named by 10-bit field F of a packed dword: two VALU instructions per entry
// (v_bfe_u32 + v_lshl_add_u32 with the region's LDS byte address as the scalar addend) instead of the shift / and /
// add-base triple hipcc emits for the C expression -- the E- and M-steps are bound by instruction issue
// (4 cycles per wave64 instruction on a 16-lane SIMD), not by LDS bandwidth.
typedef __attribute__((address_space(3))) const double lds_cdouble;
__device__ __forceinline__ unsigned lds_byte_addr(const void *p) { return (unsigned)(uintptr_t)p; }   // low half of a flat LDS address
template <int F>
__device__ __forceinline__ unsigned lds_id_addr(unsigned dword, unsigned base /* wave-uniform */) {
    // inline asm: hipcc rewrites the C expression (and the ubfe intrinsic) back into shift + and + add
    unsigned a;   // one statement: hipcc pads a nop between two dependent asm statements
    asm("v_bfe_u32 %0, %1, %2, 10\n\tv_lshl_add_u32 %0, %0, 3, %3" : "=v"(a) : "v"(dword), "i"(10 * F), "s"(base));
    return a;
}
__device__ __forceinline__ double lds_ld(unsigned a) { return *reinterpret_cast<lds_cdouble *>(a); }
// the LDS byte addresses of 6 of the 12 ids of one int4 (H = 0: fields of .x .y, H = 1: of .z .w).  Addresses first,
// then the loads back to back, then the adds: the asm statements would otherwise serialise address -> load -> wait ->
// add per entry.  BATCH = 12 keeps a whole int4 in flight (36 temporaries), BATCH = 6 half of it (18).
template <int H>
__device__ __forceinline__ void lds_addr6(const int4 t, unsigned base, unsigned (&a)[6]) {
    const unsigned d0 = (unsigned)(H ? t.z : t.x), d1 = (unsigned)(H ? t.w : t.y);
    a[0] = lds_id_addr<0>(d0, base); a[1] = lds_id_addr<1>(d0, base); a[2] = lds_id_addr<2>(d0, base);
    a[3] = lds_id_addr<0>(d1, base); a[4] = lds_id_addr<1>(d1, base); a[5] = lds_id_addr<2>(d1, base);
}

// E-step sums of up to 8 forward columns held in registers (n is wave-uniform); one int4 = this lane's 12 rows
template <int BATCH = 12>
__device__ __forceinline__ void fwd_sum_regs(const int4 (&q)[8], int n, unsigned th_base, double (&S)[kRPL]) {
#pragma unroll
    for (int j = 0; j < 8; j++) {
        if (j < n) {
            if (BATCH == 12) {
                unsigned a0[6], a1[6];
                double v[12];
                lds_addr6<0>(q[j], th_base, a0); lds_addr6<1>(q[j], th_base, a1);
#pragma unroll
                for (int i = 0; i < 6; i++) { v[i] = lds_ld(a0[i]); }
#pragma unroll
                for (int i = 0; i < 6; i++) { v[6 + i] = lds_ld(a1[i]); }
#pragma unroll
                for (int i = 0; i < 12; i++) S[i] += v[i];
            } else {
                unsigned a[6];
                double v[6];
                lds_addr6<0>(q[j], th_base, a);
#pragma unroll
                for (int i = 0; i < 6; i++) v[i] = lds_ld(a[i]);
#pragma unroll
                for (int i = 0; i < 6; i++) S[i] += v[i];
                lds_addr6<1>(q[j], th_base, a);
#pragma unroll
                for (int i = 0; i < 6; i++) v[i] = lds_ld(a[i]);
#pragma unroll
                for (int i = 0; i < 6; i++) S[6 + i] += v[i];
            }
        }
    }
}

// M-step of up to 8 backward segments of one lane ({column, 11 row ids} each).  A lane's segments are consecutive
// in column order: the running sum stays in a register and goes to the LDS accumulator when the column changes.
template <int BATCH = 12>
__device__ __forceinline__ void bwd_sum_regs(const int4 (&q)[8], int n, unsigned ws_base, double *acc_w, unsigned &cur, double &part) {
#pragma unroll
    for (int j = 0; j < 8; j++) {
        if (j < n) {
            const unsigned col = id_off((unsigned)q[j].x, 0);
            double sum;
            if (BATCH == 12) {
                unsigned a0[6], a1[6];
                double v[12];
                lds_addr6<0>(q[j], ws_base, a0); lds_addr6<1>(q[j], ws_base, a1);
#pragma unroll
                for (int i = 1; i < 6; i++) v[i] = lds_ld(a0[i]);
#pragma unroll
                for (int i = 0; i < 6; i++) v[6 + i] = lds_ld(a1[i]);
                double s0 = v[1] + v[2], s1 = v[3] + v[4], s2 = v[5] + v[6], s3 = v[7] + v[8];
                s0 += v[9]; s1 += v[10]; s2 += v[11];
                sum = (s0 + s1) + (s2 + s3);
            } else {
                unsigned a[6];
                double v[6];
                lds_addr6<0>(q[j], ws_base, a);
#pragma unroll
                for (int i = 1; i < 6; i++) v[i] = lds_ld(a[i]);
                double s0 = v[1] + v[2], s1 = v[3] + v[4];
                s0 += v[5];
                lds_addr6<1>(q[j], ws_base, a);
#pragma unroll
                for (int i = 0; i < 6; i++) v[i] = lds_ld(a[i]);
                s0 += v[0]; s1 += v[1];
                double s2 = v[2] + v[3], s3 = v[4] + v[5];
                sum = (s0 + s1) + (s2 + s3);
            }
            if (col != cur) {
                if (part != 0.0) lds_add_f64(reinterpret_cast<double *>(reinterpret_cast<char *>(acc_w) + cur), part);
                cur = col; part = 0.0;
            }
            part += sum;
        }
    }
}

// In-kernel stamps (diagnostic instance only, STAMP=true; never the timed kernel): s_memtime per phase and wave,
// written to a slot of its own that no other code reads.
__device__ __forceinline__ unsigned long long stamp_now() {
    unsigned long long t;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    return t;
}

// sum_i log S_i over a lane's kRPL unweighted rows (S_i <= 0: row outside F or padding, no term) with two logs instead of
// twelve: log of the product of six row sums.  A product that leaves [1e-280, 1e280] (components decayed towards the
// boundary) falls back to the per-row logs.  The f64 log is ~40 VALU instructions: 19 % of a likelihood pass on config 3.
template <int N>
__device__ __forceinline__ double sum_log_rows(const double (&S)[N]) {
    static_assert(N % 6 == 0, "rows per lane");
    double ll = 0.0;
#pragma unroll
    for (int g = 0; g < N; g += 6) {
        double p = 1.0;
#pragma unroll
        for (int i = g; i < g + 6; i++) p *= S[i] > 0.0 ? S[i] : 1.0;
        if (p > 1e-280 && p < 1e280) ll += log(p);
        else {
#pragma unroll
            for (int i = g; i < g + 6; i++) if (S[i] > 0.0) ll += log(S[i]);
        }
    }
    return ll;
}

template <bool WEIGHTED, int MODE, bool STAMP = false>
__global__ __launch_bounds__(kTiledThreads, 4) void k_pass_tiled(const Tile *__restrict__ tiles, const uint32_t *__restrict__ fwd,
                                                              const uint32_t *__restrict__ bwd, const uint32_t *__restrict__ coo,
                                                              const int32_t *__restrict__ far_tid,
                                                              const int32_t *__restrict__ wgt,    // per row slot
                                                              const double *__restrict__ rowval,  // per row slot (MODE_SCATTER)
                                                              const double *__restrict__ theta, double *__restrict__ acc,
                                                              double *__restrict__ ll_out, unsigned long long *stamps = nullptr) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    double *th_w = lds;                     // [960]
    double *acc_w = lds + kTiledDictPad;    // [960]
    __shared__ double red[kTiledThreads / 64];
    unsigned long long ts[6];
    if (STAMP) ts[0] = stamp_now();

    const Tile T = tiles[blockIdx.x];
    const int nd = (int)T.near_n + (int)T.far_n;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const bool has_slice = wave < (int)T.n_slices;
    double *w_s = lds + 2 * kTiledDictPad + wave * kTiledWr;     // this wave's row weights [768] + zero row
    const unsigned th_base = __builtin_amdgcn_readfirstlane(lds_byte_addr(th_w));
    const unsigned ws_base = __builtin_amdgcn_readfirstlane(lds_byte_addr(w_s));

    // ---- issue every global load this wave needs first, in the order of use: dictionary values, 8 forward columns,
    //      8 backward segments.  One HBM round trip per tile; the rest of the pass touches LDS only.
    double thv[4];
    int tid_d[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int d = threadIdx.x + i * kTiledThreads;
        thv[i] = 0.0; tid_d[i] = -1;
        if (d < nd) {
            tid_d[i] = d < (int)T.near_n ? T.lo + d : far_tid[T.far_off + (d - (int)T.near_n)];
            if (MODE != MODE_SCATTER) thv[i] = theta[tid_d[i]];
        }
    }
    int4 A[8], B[8];
    const int4 *e = nullptr, *b = nullptr;
    int k = 0, m = 0;
    unsigned coo_base = T.coo_off, coo_n = 0;
    if (has_slice) {
        unsigned foff = 0, boff = 0;                 // KiB units (256 dwords) from the tile's bases
#pragma unroll
        for (int s = 0; s < emsar::kTileSlices; s++) {
            if (s < wave) { foff += T.k[s]; boff += T.m[s]; coo_base += T.coo_n[s]; }
            if (s == wave) { k = T.k[s]; m = T.m[s]; coo_n = T.coo_n[s]; }
        }
        // everything above is wave-uniform; say so, or the loops below are compiled as divergent code
        k = __builtin_amdgcn_readfirstlane(k); m = __builtin_amdgcn_readfirstlane(m);
        coo_n = __builtin_amdgcn_readfirstlane(coo_n); coo_base = __builtin_amdgcn_readfirstlane(coo_base);
        foff = __builtin_amdgcn_readfirstlane(foff); boff = __builtin_amdgcn_readfirstlane(boff);
        e = reinterpret_cast<const int4 *>(fwd + T.fwd_off / 4 + (size_t)foff * 256) + lane;
        b = reinterpret_cast<const int4 *>(bwd + T.bwd_off / 4 + (size_t)boff * 256) + lane;
        if (MODE != MODE_SCATTER) load8_clamped(A, e, k < 8 ? k : 8);
        if (m > 0) load8_clamped(B, b, m < 8 ? m : 8);
    }
    // ---- phase 0: dictionary into LDS (slot nd is the zero slot) ----
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int d = threadIdx.x + i * kTiledThreads;
        if (d <= nd) { th_w[d] = thv[i]; acc_w[d] = 0.0; }
    }
    if (lane < 8) w_s[emsar::kTileSliceRows + lane] = 0.0;      // padding row of this wave's slice
    if (STAMP) ts[1] = stamp_now();
    __syncthreads();
    if (STAMP) ts[2] = stamp_now();

    double ll = 0.0;
    if (has_slice) {
        // ---- E: row sums of this wave's 768 rows ----
        // row i of lane l is slot 64*i + l of the slice: the lanes of one gather hold consecutive sorted rows
        const size_t slot0 = (size_t)T.row_base + (size_t)wave * emsar::kTileSliceRows + lane;
        double w[kRPL];
        if (MODE == MODE_SCATTER) {
#pragma unroll
            for (int i = 0; i < kRPL; i++) w[i] = rowval[slot0 + 64 * i];
        } else {
            double S[kRPL];
#pragma unroll
            for (int i = 0; i < kRPL; i++) S[i] = 0.0;
            for (int j0 = 0; j0 < k; j0 += 8) {
                const int n0 = k - j0 < 8 ? k - j0 : 8;
                if (j0) load8_clamped(A, e + (size_t)j0 * 64, n0);
                fwd_sum_regs(A, n0, th_base, S);
            }
            double r[kRPL];
#pragma unroll
            for (int i = 0; i < kRPL; i++) r[i] = 1.0;
            if (WEIGHTED) {
#pragma unroll
                for (int i = 0; i < kRPL; i++) r[i] = (double)wgt[slot0 + 64 * i];
            }
#pragma unroll
            for (int i = 0; i < kRPL; i++) {
                bool live = (S[i] > 0.0) && (r[i] > 0.0);
                w[i] = live ? r[i] / S[i] : 0.0;
                if (MODE == MODE_EM_LL && WEIGHTED && live) ll += r[i] * log(S[i]);
            }
            if (MODE == MODE_EM_LL && !WEIGHTED) ll += sum_log_rows(S);
        }
#pragma unroll
        for (int i = 0; i < kRPL; i++) w_s[64 * i + lane] = w[i];
        // the M-step below reads rows written by OTHER lanes of this same wave: DS operations of one wave execute in
        // order, so only the compiler has to be kept from moving the reads up
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (STAMP) ts[3] = stamp_now();
        // ---- M: column sums over the same 768 rows, through the slice's transposed index ----
        unsigned cur = 0xFFFFFFFFu;
        double part = 0.0;
        for (int j0 = 0; j0 < m; j0 += 8) {
            const int n0 = m - j0 < 8 ? m - j0 : 8;
            if (j0) load8_clamped(B, b + (size_t)j0 * 64, n0);
            bwd_sum_regs(B, n0, ws_base, acc_w, cur, part);
        }
        if (part != 0.0) lds_add_f64(reinterpret_cast<double *>(reinterpret_cast<char *>(acc_w) + cur), part);
        for (unsigned q = lane; q < coo_n; q += 64) {
            const unsigned p = coo[coo_base + q];
            const double v = lds_at(w_s, (p & 0xFFFFu) << 3);
            if (v != 0.0) lds_add_f64(reinterpret_cast<double *>(reinterpret_cast<char *>(acc_w) + ((p >> 16) << 3)), v);
        }
    } else if (STAMP) ts[3] = stamp_now();
    if (STAMP) ts[4] = stamp_now();
    __syncthreads();
    if (STAMP) ts[5] = stamp_now();
    // ---- F: flush the dictionary ----
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int d = threadIdx.x + i * kTiledThreads;
        if (d < nd) {
            const double v = acc_w[d];
            if (v != 0.0) atomic_add_f64(&acc[tid_d[i]], v);
        }
    }
    if (STAMP && lane == 0) {   // [tile][wave][5 phases]: issue+dictionary, barrier, E, M, barrier
        for (int i = 0; i < 5; i++) stamps[((size_t)blockIdx.x * (kTiledThreads / 64) + wave) * 8 + i] = ts[i + 1] - ts[i];
    }
    if (MODE == MODE_EM_LL) {
        double t = block_sum<kTiledThreads>(ll, red);
        if (threadIdx.x == 0 && t != 0.0) atomic_add_f64(ll_out, t);
    }
}

// ------------------------------------------------------------------------------------------------
// k_pass_tiled_multi: the same pass, N tiles per workgroup, software-pipelined by hand.  In k_pass_tiled a wave has
// loads in flight only at its start (~20 % of its life); with four workgroups per CU there is often nobody loading and
// the CU's share of HBM idles.  Here the forward registers are refilled with the next tile's columns as soon as the
// E-step has consumed them, the backward registers after the M-step, and the next dictionary is requested before the
// M-step: the next tile's HBM round trip hides behind this tile's LDS work.  Measured on config 3: N = 2 0.218 ms,
// N = 3 0.224, N = 4 0.242 (fewer, longer workgroups: the tail grows), a persistent loop 0.251 (hipcc spills the
// loop-carried register arrays); one tile per workgroup 0.225.
// ------------------------------------------------------------------------------------------------
struct TileWave {           // what one wave needs to know about its slice of a tile (all wave-uniform but e/b)
    const int4 *e, *b;
    int k, m, nd;
    unsigned coo_base, coo_n;
    bool has_slice;
};
__device__ __forceinline__ TileWave tile_wave(const Tile &T, int wave, int lane, const uint32_t *fwd, const uint32_t *bwd) {
    TileWave W;
    W.nd = (int)T.near_n + (int)T.far_n;
    W.has_slice = wave < (int)T.n_slices;
    W.e = nullptr; W.b = nullptr; W.k = 0; W.m = 0; W.coo_base = T.coo_off; W.coo_n = 0;
    if (W.has_slice) {
        unsigned foff = 0, boff = 0;
        int k = 0, m = 0; unsigned cn = 0, cb = T.coo_off;
#pragma unroll
        for (int s = 0; s < emsar::kTileSlices; s++) {
            if (s < wave) { foff += T.k[s]; boff += T.m[s]; cb += T.coo_n[s]; }
            if (s == wave) { k = T.k[s]; m = T.m[s]; cn = T.coo_n[s]; }
        }
        W.k = __builtin_amdgcn_readfirstlane(k); W.m = __builtin_amdgcn_readfirstlane(m);
        W.coo_n = __builtin_amdgcn_readfirstlane(cn); W.coo_base = __builtin_amdgcn_readfirstlane(cb);
        foff = __builtin_amdgcn_readfirstlane(foff); boff = __builtin_amdgcn_readfirstlane(boff);
        W.e = reinterpret_cast<const int4 *>(fwd + T.fwd_off / 4 + (size_t)foff * 256) + lane;
        W.b = reinterpret_cast<const int4 *>(bwd + T.bwd_off / 4 + (size_t)boff * 256) + lane;
    }
    return W;
}
__device__ __forceinline__ void tile_dict_issue(const Tile &T, int nd, const int32_t *far_tid, const double *theta, double (&thv)[4], int (&tid_d)[4]) {
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int d = threadIdx.x + i * kTiledThreads;
        thv[i] = 0.0; tid_d[i] = -1;
        if (d < nd) {
            tid_d[i] = d < (int)T.near_n ? T.lo + d : far_tid[T.far_off + (d - (int)T.near_n)];
            thv[i] = theta[tid_d[i]];
        }
    }
}
__device__ __forceinline__ void tile_dict_store(int nd, const double (&thv)[4], double *th_w, double *acc_w) {
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int d = threadIdx.x + i * kTiledThreads;
        if (d <= nd) { th_w[d] = thv[i]; acc_w[d] = 0.0; }
    }
}
template <bool WEIGHTED, int MODE>
__device__ __forceinline__ void tile_e_step(const TileWave &W, int4 (&A)[8], size_t slot0, const int32_t *wgt, const double *th_w, double *w_s,
                                            int lane, double &ll) {
    const unsigned th_base = __builtin_amdgcn_readfirstlane(lds_byte_addr(th_w));
    double S[kRPL], w[kRPL], r[kRPL];
#pragma unroll
    for (int i = 0; i < kRPL; i++) { S[i] = 0.0; r[i] = 1.0; }
    for (int j0 = 0; j0 < W.k; j0 += 8) {
        const int n0 = W.k - j0 < 8 ? W.k - j0 : 8;
        if (j0) load8_clamped(A, W.e + (size_t)j0 * 64, n0);
        fwd_sum_regs<6>(A, n0, th_base, S);
    }
    if (WEIGHTED) {
#pragma unroll
        for (int i = 0; i < kRPL; i++) r[i] = (double)wgt[slot0 + 64 * i];
    }
#pragma unroll
    for (int i = 0; i < kRPL; i++) {
        bool live = (S[i] > 0.0) && (r[i] > 0.0);
        w[i] = live ? r[i] / S[i] : 0.0;
        if (MODE == MODE_EM_LL && WEIGHTED && live) ll += r[i] * log(S[i]);
    }
    if (MODE == MODE_EM_LL && !WEIGHTED) ll += sum_log_rows(S);
#pragma unroll
    for (int i = 0; i < kRPL; i++) w_s[64 * i + lane] = w[i];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
__device__ __forceinline__ void tile_m_step(const TileWave &W, int4 (&B)[8], const uint32_t *coo, const double *w_s, double *acc_w, int lane) {
    const unsigned ws_base = __builtin_amdgcn_readfirstlane(lds_byte_addr(w_s));
    unsigned cur = 0xFFFFFFFFu;
    double part = 0.0;
    for (int j0 = 0; j0 < W.m; j0 += 8) {
        const int n0 = W.m - j0 < 8 ? W.m - j0 : 8;
        if (j0) load8_clamped(B, W.b + (size_t)j0 * 64, n0);
        bwd_sum_regs<6>(B, n0, ws_base, acc_w, cur, part);
    }
    if (part != 0.0) lds_add_f64(reinterpret_cast<double *>(reinterpret_cast<char *>(acc_w) + cur), part);
    for (unsigned q = lane; q < W.coo_n; q += 64) {
        const unsigned p = coo[W.coo_base + q];
        const double v = lds_at(w_s, (p & 0xFFFFu) << 3);
        if (v != 0.0) lds_add_f64(reinterpret_cast<double *>(reinterpret_cast<char *>(acc_w) + ((p >> 16) << 3)), v);
    }
}
__device__ __forceinline__ void tile_flush(int nd, const int (&tid_d)[4], const double *acc_w, double *acc) {
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int d = threadIdx.x + i * kTiledThreads;
        if (d < nd) {
            const double v = acc_w[d];
            if (v != 0.0) atomic_add_f64(&acc[tid_d[i]], v);
        }
    }
}

struct TileEnv {            // per-launch constants of the multi-tile kernel
    const Tile *tiles; int n_tiles, stride;
    const uint32_t *fwd, *bwd, *coo; const int32_t *far_tid, *wgt; const double *theta; double *acc;
    double *th_w, *acc_w, *w_s; int lane, wave;
};
// stage I of N: tile `it` is in the registers (A, B in flight or landed, dictionary values in thv); while it is being
// worked on, tile it + stride is requested into the registers as they fall free.  Straight-line code, no loop: hipcc
// keeps loop-carried register arrays of this size in scratch.
template <bool WEIGHTED, int MODE, int I, int N>
__device__ __forceinline__ void tiled_stage(const TileEnv &V, int it, const Tile &T, const TileWave &W, int4 (&A)[8], int4 (&B)[8],
                                            double (&thv)[4], const int (&tid)[4], double &ll) {
    // a thread rewrites only the dictionary slots it flushed at the end of the previous stage
    tile_dict_store(W.nd, thv, V.th_w, V.acc_w);
    const int in = it + V.stride;
    const bool has_next = (I + 1 < N) && in < V.n_tiles;
    const Tile Tn = V.tiles[has_next ? in : it];
    __syncthreads();
    if (W.has_slice)
        tile_e_step<WEIGHTED, MODE>(W, A, (size_t)T.row_base + (size_t)V.wave * emsar::kTileSliceRows + V.lane, V.wgt, V.th_w, V.w_s, V.lane, ll);
    const TileWave Wn = tile_wave(Tn, V.wave, V.lane, V.fwd, V.bwd);
    int tidn[4] = {-1, -1, -1, -1};
    if (has_next) {
        if (Wn.has_slice) load8_clamped(A, Wn.e, Wn.k < 8 ? Wn.k : 8);
        tile_dict_issue(Tn, Wn.nd, V.far_tid, V.theta, thv, tidn);
    }
    if (W.has_slice) tile_m_step(W, B, V.coo, V.w_s, V.acc_w, V.lane);
    if (has_next && Wn.has_slice && Wn.m > 0) load8_clamped(B, Wn.b, Wn.m < 8 ? Wn.m : 8);
    __syncthreads();
    tile_flush(W.nd, tid, V.acc_w, V.acc);
    if constexpr (I + 1 < N) {
        if (has_next) tiled_stage<WEIGHTED, MODE, I + 1, N>(V, in, Tn, Wn, A, B, thv, tidn, ll);
    }
}

template <bool WEIGHTED, int MODE, int N>
__global__ __launch_bounds__(kTiledThreads, 4) void k_pass_tiled_multi(const Tile *__restrict__ tiles, int n_tiles, const uint32_t *__restrict__ fwd,
                                                                    const uint32_t *__restrict__ bwd, const uint32_t *__restrict__ coo,
                                                                    const int32_t *__restrict__ far_tid, const int32_t *__restrict__ wgt,
                                                                    const double *__restrict__ theta, double *__restrict__ acc,
                                                                    double *__restrict__ ll_out) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    __shared__ double red[kTiledThreads / 64];
    TileEnv V;
    V.tiles = tiles; V.n_tiles = n_tiles; V.stride = (int)gridDim.x; V.fwd = fwd; V.bwd = bwd; V.coo = coo; V.far_tid = far_tid; V.wgt = wgt;
    V.theta = theta; V.acc = acc;
    V.lane = threadIdx.x & 63;
    V.wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    V.th_w = lds; V.acc_w = lds + kTiledDictPad; V.w_s = lds + 2 * kTiledDictPad + V.wave * kTiledWr;
    const Tile T = tiles[blockIdx.x];
    const TileWave W = tile_wave(T, V.wave, V.lane, fwd, bwd);
    double thv[4]; int tid[4];
    int4 A[8], B[8];
    tile_dict_issue(T, W.nd, far_tid, theta, thv, tid);
    if (W.has_slice) {
        load8_clamped(A, W.e, W.k < 8 ? W.k : 8);
        if (W.m > 0) load8_clamped(B, W.b, W.m < 8 ? W.m : 8);
    }
    if (V.lane < 8) V.w_s[emsar::kTileSliceRows + V.lane] = 0.0;
    double ll = 0.0;
    tiled_stage<WEIGHTED, MODE, 0, N>(V, (int)blockIdx.x, T, W, A, B, thv, tid, ll);
    if (MODE == MODE_EM_LL) {
        double t = block_sum<kTiledThreads>(ll, red);
        if (threadIdx.x == 0 && t != 0.0) atomic_add_f64(ll_out, t);
    }
}

// likelihood terms of the folded single-tid rows: sum_t u_t log theta_t
__global__ __launch_bounds__(256) void k_single_ll(int n, const double *__restrict__ u, const double *__restrict__ theta, double *ll_out) {
    __shared__ double red[4];
    double s = 0.0;
    for (int t = blockIdx.x * 256 + threadIdx.x; t < n; t += gridDim.x * 256) {
        double x = theta[t], c = u[t];
        if (c > 0.0 && x > 0.0) s += c * log(x);
    }
    double tot = block_sum<256>(s, red);
    if (threadIdx.x == 0 && tot != 0.0) atomic_add_f64(ll_out, tot);
}

// ------------------------------------------------------------------------------------------------
// k_pass_csr: the same pass on the caller's CSR (any row order), one lane per row.
// ------------------------------------------------------------------------------------------------
template <typename PTR, bool WEIGHTED, int MODE>
__global__ __launch_bounds__(256) void k_pass_csr(int64_t n_rows, const PTR *__restrict__ row_ptr,
                                                  const int32_t *__restrict__ col, const int32_t *__restrict__ wgt,
                                                  const double *__restrict__ rowval, const double *__restrict__ theta,
                                                  double *__restrict__ acc, double *__restrict__ ll_out) {
    __shared__ double red[4];
    double ll = 0.0;
    for (int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x; r < n_rows; r += (int64_t)gridDim.x * 256) {
        const uint64_t b = row_ptr[r], e = row_ptr[r + 1];
        double w;
        if (MODE == MODE_SCATTER) {
            w = rowval[r];
        } else {
            double S = 0.0;
            for (uint64_t k = b; k < e; k++) S += theta[col[k]];
            double rw = WEIGHTED ? (double)wgt[r] : 1.0;
            bool live = (S > 0.0) && (rw > 0.0);
            w = live ? rw / S : 0.0;
            if (MODE == MODE_EM_LL && live) ll += rw * log(S);
        }
        if (w != 0.0)
            for (uint64_t k = b; k < e; k++) atomic_add_f64(&acc[col[k]], w);
    }
    if (MODE == MODE_EM_LL) {
        double t = block_sum<256>(ll, red);
        if (threadIdx.x == 0 && t != 0.0) atomic_add_f64(ll_out, t);
    }
}

// ------------------------------------------------------------------------------------------------
// T-sized vector kernels
// ------------------------------------------------------------------------------------------------
__global__ void k_fill_start(int n, const double *__restrict__ den, double *__restrict__ theta) {
    int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n) theta[t] = den[t] > 0.0 ? 1.0 : 0.0;  // uniform interior start; tids outside F are defined 0
}

// theta_out = theta_in * acc / den ; acc <- 0 ; scal.delta = max |dtheta| / (theta_out + floor)
// grid-stride, one atomicMax per workgroup (hundreds of same-address atomics cost ~12 ns each)
__global__ __launch_bounds__(256) void k_update(int n, const double *__restrict__ th_in, double *__restrict__ acc,
                                                const double *__restrict__ den, const double *__restrict__ u /* folded single-tid rows, may be null */,
                                                double *__restrict__ th_out, double abs_floor, double count_floor, double zero_cut, Scal *scal,
                                                const uint8_t *__restrict__ kind /* non-null: only KIND_STREAMED transcripts enter the stopping rule */,
                                                int to_delta1 /* the first (plain) step of a SQUAREM cycle: the cycle's stopping rule */) {
    __shared__ double red[4];
    double d = 0.0;
    const double abs_step = scal->abs_step_cur;      // written by k_cycle_begin, nobody writes it during a pass
    for (int t = blockIdx.x * 256 + threadIdx.x; t < n; t += gridDim.x * 256) {
        double a = acc[t], dn = den[t], x = th_in[t];
        // a row {t} contributes R/theta_t to acc_t, i.e. R to theta_t*acc_t: added analytically (TILED layout)
        double y = dn > 0.0 ? (u ? (x > 0.0 ? (x * a + u[t]) / dn : 0.0) : x * a / dn) : 0.0;
        th_out[t] = y;
        acc[t] = 0.0;
        double fl = abs_floor;
        if (count_floor > 0.0 && dn > 0.0) fl = fmax(fl, count_floor / dn);    // floor expressed in inferred reads
        double dd = fabs(y - x) / (fabs(y) + fl);
        if (!(dd == dd)) dd = __builtin_huge_val();  // NaN -> +inf so that the host sees it
        if (y < zero_cut && y <= x) dd = 0.0;        // below the print quantum and still falling: prints as 0.000000 either way
        if (fabs(y - x) < abs_step) dd = 0.0;         // moves by less than abs_step per pass (emsar_em_params.abs_step)
        if (kind && kind[t] != emsar::KIND_STREAMED) dd = 0.0;
        d = fmax(d, dd);
    }
    for (int o = 32; o > 0; o >>= 1) d = fmax(d, __shfl_xor(d, o, 64));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = d;
    __syncthreads();
    if (threadIdx.x == 0) {
        d = fmax(fmax(red[0], red[1]), fmax(red[2], red[3]));
        // the word only grows during a kernel: workgroups whose maximum is already covered skip the same-address atomic
        unsigned long long *dst = to_delta1 ? &scal->delta1_bits : &scal->delta_bits;
        const unsigned long long bits = (unsigned long long)__double_as_longlong(d);
        if (d > 0.0 && bits > __hip_atomic_load(dst, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(dst, bits);
    }
}

// abs_step_base > 0: the projected-drift bound of emsar_em_params.abs_step, |dtheta| < base * 2e5 / K at pass K >= 1000.  K is
// counted here, on the device, so that a cycle recorded once in a hipGraph carries the right bound at every replay.
__global__ void k_cycle_begin(Scal *s, double abs_step_base, int passes_in_cycle) {
    s->ll[0] = s->ll[1] = s->ll[2] = s->ll[3] = 0.0;
    s->sr2 = s->sv2 = s->pen1 = s->penx = 0.0;
    s->delta_bits = 0ull; s->delta1_bits = 0ull;
    const long long done = s->passes;
    s->abs_step_cur = abs_step_base > 0.0 ? abs_step_base * 2e5 / (double)(done + 1 > 1000 ? done + 1 : 1000) : 0.0;
    s->passes = done + passes_in_cycle;
}
__global__ void k_scal_init(Scal *s) {
    s->stepmax = 1.0; s->s_used = 1.0; s->accepted = 0; s->rejected = 0; s->sum_a = s->sum_b = 0.0;
    s->passes = 0; s->abs_step_cur = 0.0;
}

// ---- the SQUAREM cycle of the streaming solve with the O(T) work folded into the three update kernels ----
// (8 launches per cycle instead of 13: on a problem of a few thousand rows the cycle is pure launch latency)
//   k_update_p1   th1 = EM(th0); stopping rule of the cycle -> delta1_bits
//   k_update_p2   th2 = EM(th1); F(th1) terms: sum u log th1 -> ll[1], sum th1*den -> pen1; |r|^2, |v|^2
//   k_sq_extrap_ll thx = th0 + 2 s r + s^2 v  (Varadhan & Roland 2008, S3: s = |r|/|v| clamped to [1, stepmax]); components that
//                 would leave the interior keep the plain EM value th2; s <= 1.01 -> thx = th2; + sum u log thx -> ll[2]
//   k_update_p3   th0 = accepted ? EM(thx) : th2, accepted iff F(thx) >= F(th1), F = ll - sum theta*den; step bounds x4 / :4
__device__ __forceinline__ double em_new_theta(double x, double a, double dn, const double *u, int t) {
    return dn > 0.0 ? (u ? (x > 0.0 ? (x * a + u[t]) / dn : 0.0) : x * a / dn) : 0.0;
}
__global__ __launch_bounds__(256) void k_update_p2(int n, const double *__restrict__ th0, const double *__restrict__ th1, double *__restrict__ acc,
                                                   const double *__restrict__ den, const double *__restrict__ u, double *__restrict__ th2, Scal *scal) {
    __shared__ double red[4];
    double r2 = 0, v2 = 0, p1 = 0, l1 = 0;
    for (int t = blockIdx.x * 256 + threadIdx.x; t < n; t += gridDim.x * 256) {
        const double x = th1[t], dn = den[t];
        const double y = em_new_theta(x, acc[t], dn, u, t);
        th2[t] = y;
        acc[t] = 0.0;
        const double r = x - th0[t], v = (y - x) - r;
        r2 += r * r; v2 += v * v; p1 += x * dn;
        if (u) { const double c = u[t]; if (c > 0.0 && x > 0.0) l1 += c * log(x); }
    }
    double a = block_sum<256>(r2, red); __syncthreads();
    double b = block_sum<256>(v2, red); __syncthreads();
    double c = block_sum<256>(p1, red); __syncthreads();
    double d = block_sum<256>(l1, red);
    if (threadIdx.x == 0) {
        atomic_add_f64(&scal->sr2, a); atomic_add_f64(&scal->sv2, b); atomic_add_f64(&scal->pen1, c);
        if (d != 0.0) atomic_add_f64(&scal->ll[1], d);
    }
}
__global__ __launch_bounds__(256) void k_sq_extrap_ll(int n, const double *__restrict__ th0, const double *__restrict__ th1,
                                                      const double *__restrict__ th2, const double *__restrict__ den, const double *__restrict__ u,
                                                      double *__restrict__ thx, Scal *scal) {
    __shared__ double red[4];
    double s = scal->sv2 > 0.0 ? sqrt(scal->sr2 / scal->sv2) : 1.0;
    s = fmin(fmax(s, 1.0), scal->stepmax);
    const bool extrap = s > 1.01;
    double px = 0, lx = 0;
    for (int t = blockIdx.x * 256 + threadIdx.x; t < n; t += gridDim.x * 256) {
        double x2 = th2[t], x = x2;
        if (extrap) {
            double r = th1[t] - th0[t], v = (x2 - th1[t]) - r;
            double y = th0[t] + 2.0 * s * r + s * s * v;
            x = (y > 0.0 && x2 > 0.0) ? y : x2;
        }
        thx[t] = x;
        px += x * den[t];
        if (u) { const double c = u[t]; if (c > 0.0 && x > 0.0) lx += c * log(x); }
    }
    double p = block_sum<256>(px, red); __syncthreads();
    double l = block_sum<256>(lx, red);
    if (threadIdx.x == 0) {
        atomic_add_f64(&scal->penx, p);
        if (l != 0.0) atomic_add_f64(&scal->ll[2], l);
        if (blockIdx.x == 0) scal->s_used = extrap ? s : 1.0;
    }
}
__global__ __launch_bounds__(256) void k_update_p3(int n, const double *__restrict__ thx, const double *__restrict__ th2, double *__restrict__ acc,
                                                   const double *__restrict__ den, const double *__restrict__ u, double *__restrict__ th0, Scal *scal) {
    const double s = scal->s_used;
    const bool extrap = s > 1.0;
    const bool ok = !extrap || (scal->ll[2] - scal->penx >= scal->ll[1] - scal->pen1);
    for (int t = blockIdx.x * 256 + threadIdx.x; t < n; t += gridDim.x * 256) {
        const double y = em_new_theta(thx[t], acc[t], den[t], u, t);
        acc[t] = 0.0;
        th0[t] = ok ? y : th2[t];
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {      // nobody reads these three during this kernel
        double sm = scal->stepmax;
        if (!ok) { scal->rejected++; if (s >= sm) sm = fmax(1.0, sm / 4.0); }
        else { scal->accepted++; }
        if ((ok ? s : 1.0) >= sm) sm *= 4.0;
        scal->stepmax = sm;
    }
}

// ------------------------------------------------------------------------------------------------
// k_solve_sets: the SET-RESIDENT solver.  One workgroup owns one connected set (sets.hpp) and runs the whole
// SQUAREM-accelerated EM on it out of LDS: theta vectors, den, the folded single-row counts, the row weights and
// the 16-bit CSR/CSC indices are loaded once, then every pass is two LDS sweeps (E-step over rows, M-step over
// transcripts through the CSC -- no atomics, so the result is reproducible bit for bit) and a workgroup
// reduction.  No global synchronisation, no kernel launch per pass: a pass costs ~1 us instead of the ~30 us
// launch-latency floor of the streaming kernels, and every set stops at its own convergence.
// Same update, same start (theta = 1 where den > 0), same stopping rule and the same likelihood-safeguarded S3
// step as the streaming solve below, so both reach the same fixed point.
// ------------------------------------------------------------------------------------------------
struct SetStat { int32_t passes, converged; double delta; };
struct SetSolveParams { double tol, abs_floor, count_floor, zero_cut, abs_step; int32_t max_iter, accel; };

// Wave-wide reductions on the DPP path (row shifts inside rows of 16 lanes, then row broadcasts; the total lands in lane
// 63 and is read back as a scalar): ~6 cross-lane moves per value instead of the twelve ds_bpermute round trips of a
// shuffle butterfly.  The per-set solver is a chain of dependent steps, so the latency of its reductions is pass time.
template <int CTRL>
__device__ __forceinline__ double dpp_move(double v, double identity) {
    const unsigned long long b = (unsigned long long)__double_as_longlong(v), id = (unsigned long long)__double_as_longlong(identity);
    const int lo = __builtin_amdgcn_update_dpp((int)(unsigned)id, (int)(unsigned)b, CTRL, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp((int)(unsigned)(id >> 32), (int)(unsigned)(b >> 32), CTRL, 0xf, 0xf, false);
    return __longlong_as_double((long long)(((unsigned long long)(unsigned)hi << 32) | (unsigned long long)(unsigned)lo));
}
__device__ __forceinline__ double wave_bcast63(double v) {
    const unsigned long long b = (unsigned long long)__double_as_longlong(v);
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)b, 63), hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(b >> 32), 63);
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}
__device__ __forceinline__ double wave_sum_dpp(double v) {
    v += dpp_move<0x111>(v, 0.0);      // row_shr:1
    v += dpp_move<0x112>(v, 0.0);      // row_shr:2
    v += dpp_move<0x114>(v, 0.0);      // row_shr:4
    v += dpp_move<0x118>(v, 0.0);      // row_shr:8   -> lane 15 of every row holds its row's sum
    v += dpp_move<0x142>(v, 0.0);      // row_bcast:15 -> lane 31 / 63 hold the sums of rows 0-1 / 2-3 (plus their own rows)
    v += dpp_move<0x143>(v, 0.0);      // row_bcast:31 -> lane 63 holds the wave's sum
    return wave_bcast63(v);
}
__device__ __forceinline__ double wave_max_dpp(double v) {      // v >= 0
    v = fmax(v, dpp_move<0x111>(v, 0.0));
    v = fmax(v, dpp_move<0x112>(v, 0.0));
    v = fmax(v, dpp_move<0x114>(v, 0.0));
    v = fmax(v, dpp_move<0x118>(v, 0.0));
    v = fmax(v, dpp_move<0x142>(v, 0.0));
    v = fmax(v, dpp_move<0x143>(v, 0.0));
    return wave_bcast63(v);
}

template <int THREADS, int N>
__device__ __forceinline__ void set_reduce_sum(double (&v)[N], double *red) {
#pragma unroll
    for (int i = 0; i < N; i++) v[i] = wave_sum_dpp(v[i]);
    if (THREADS > 64) {
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        __syncthreads();                       // red may still be read from the previous reduction
        if (lane == 0)
#pragma unroll
            for (int i = 0; i < N; i++) red[wave * N + i] = v[i];
        __syncthreads();
#pragma unroll
        for (int i = 0; i < N; i++) {
            double t = 0;
            for (int w = 0; w < THREADS / 64; w++) t += red[w * N + i];
            v[i] = t;
        }
    }
}
template <int THREADS>
__device__ __forceinline__ double set_reduce_max(double v, double *red) {
    v = wave_max_dpp(v);
    if (THREADS > 64) {
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        __syncthreads();
        if (lane == 0) red[wave] = v;
        __syncthreads();
        double t = 0;
        for (int w = 0; w < THREADS / 64; w++) t = fmax(t, red[w]);
        v = t;
    }
    return v;
}

// 1/x for normal positive x: v_rcp_f64 (about half the mantissa) refined by two Newton steps -- five dependent
// instructions instead of the dozen of the IEEE division sequence (scaling, fix-up).  The per-set solver is a chain of
// dependent steps; theta, den and the row sums it divides by are far from the exponent range where the fix-ups matter.
__device__ __forceinline__ double fast_rcp(double x) {
    double r = __builtin_amdgcn_rcp(x);
    double e = __builtin_fma(-x, r, 1.0);
    r = __builtin_fma(r, e, r);
    e = __builtin_fma(-x, r, 1.0);
    return __builtin_fma(r, e, r);
}

struct SetLds {
    double *den, *u, *w, *rw, *red;
    const uint16_t *rp, *ent, *cp, *crow;
    int nt, nr;
};

// E-step over the rows at x, then M-step: y = (x*acc + u)/den.  Returns this thread's share of sum R log S (+ the
// folded single rows' u log x) when LL.
template <int THREADS, bool LL>
__device__ __forceinline__ double set_em_estep(const SetLds &L, const double *x) {
    double ll = 0.0;
    for (int j = threadIdx.x; j < L.nr; j += THREADS) {
        double S = 0.0;
        const int b = L.rp[j], e = L.rp[j + 1];
        for (int k = b; k < e; k += 4) {  // four independent index -> value chains in flight, also for the last 1..3 entries
            const int l = e - 1;
            const int i0 = L.ent[k], i1 = L.ent[k + 1 < e ? k + 1 : l], i2 = L.ent[k + 2 < e ? k + 2 : l], i3 = L.ent[k + 3 < e ? k + 3 : l];
            const double v0 = x[i0], v1 = x[i1], v2 = x[i2], v3 = x[i3];
            S += (v0 + (k + 1 < e ? v1 : 0.0)) + ((k + 2 < e ? v2 : 0.0) + (k + 3 < e ? v3 : 0.0));
        }
        const double r = L.rw[j];
        const bool live = S > 0.0;
        L.w[j] = live ? r * fast_rcp(S) : 0.0;
        if (LL && live) ll += r * log(S);
    }
    __syncthreads();
    return ll;
}
__device__ __forceinline__ double set_em_acc(const SetLds &L, int i) {
    double a = 0.0;
    const int b = L.cp[i], e = L.cp[i + 1];
    for (int k = b; k < e; k += 4) {
        const int l = e - 1;
        const int j0 = L.crow[k], j1 = L.crow[k + 1 < e ? k + 1 : l], j2 = L.crow[k + 2 < e ? k + 2 : l], j3 = L.crow[k + 3 < e ? k + 3 : l];
        const double v0 = L.w[j0], v1 = L.w[j1], v2 = L.w[j2], v3 = L.w[j3];
        a += (v0 + (k + 1 < e ? v1 : 0.0)) + ((k + 2 < e ? v2 : 0.0) + (k + 3 < e ? v3 : 0.0));
    }
    return a;
}
__device__ __forceinline__ double set_em_update(double x, double a, double u, double dn) {
    return dn > 0.0 ? (x > 0.0 ? (x * a + u) * fast_rcp(dn) : 0.0) : 0.0;
}

template <int THREADS>
__global__ __launch_bounds__(THREADS) void k_solve_sets(const emsar::SetDesc *__restrict__ desc, const int32_t *__restrict__ g_tid,
                                                        const double *__restrict__ g_u, const double *__restrict__ row_w,
                                                        const uint16_t *__restrict__ rp_g, const uint16_t *__restrict__ ent_g,
                                                        const uint16_t *__restrict__ cp_g, const uint16_t *__restrict__ crow_g,
                                                        const double *__restrict__ den_g, double *__restrict__ theta_g,
                                                        SetStat *__restrict__ stat, SetSolveParams P) {
    extern __shared__ double smem[];
    const emsar::SetDesc d = desc[blockIdx.x];
    const int nt = (int)d.n_t, nr = (int)d.n_r, nnz = (int)d.nnz;
    double *A = smem, *B = A + nt, *Cc = B + nt;
    SetLds L;
    L.den = Cc + nt; L.u = L.den + nt; L.w = L.u + nt; L.rw = L.w + nr; L.red = L.rw + nr;
    uint16_t *rp = (uint16_t *)(L.red + emsar::kSetRedDoubles), *ent = rp + (nr + 1), *cp = ent + nnz, *crow = cp + (nt + 1);
    L.rp = rp; L.ent = ent; L.cp = cp; L.crow = crow; L.nt = nt; L.nr = nr;
    for (int i = threadIdx.x; i < nt; i += THREADS) {
        const double dn = den_g[g_tid[d.tid_off + i]];
        L.den[i] = dn; L.u[i] = g_u[d.tid_off + i];
        A[i] = dn > 0.0 ? 1.0 : 0.0;
    }
    for (int j = threadIdx.x; j < nr; j += THREADS) L.rw[j] = row_w[d.row_off + j];
    for (int j = threadIdx.x; j <= nr; j += THREADS) rp[j] = rp_g[d.rp_off + j];
    for (int i = threadIdx.x; i <= nt; i += THREADS) cp[i] = cp_g[d.cp_off + i];
    for (int k = threadIdx.x; k < nnz; k += THREADS) { ent[k] = ent_g[d.ent_off + k]; crow[k] = crow_g[d.ent_off + k]; }
    __syncthreads();

    double stepmax = 1.0, delta = __builtin_huge_val();
    int passes = 0, converged = 0;
    double *res = A;
    for (;;) {
        // pass 1 (plain): B = EM(A); the stopping rule is measured on this step only
        (void)set_em_estep<THREADS, false>(L, A);
        double dloc = 0.0;
        for (int i = threadIdx.x; i < nt; i += THREADS) {
            const double x = A[i], dn = L.den[i];
            const double y = set_em_update(x, set_em_acc(L, i), L.u[i], dn);
            B[i] = y;
            double fl = P.abs_floor;
            if (P.count_floor > 0.0 && dn > 0.0) fl = fmax(fl, P.count_floor / dn);
            double dd = fabs(y - x) * fast_rcp(fabs(y) + fl);
            if (!(dd == dd)) dd = __builtin_huge_val();
            if (y < P.zero_cut && y <= x) dd = 0.0;
            if (fabs(y - x) * (double)(passes + 1 > 1000 ? passes + 1 : 1000) < P.abs_step * 2e5) dd = 0.0;   // projected drift, see emsar_em_params.abs_step
            dloc = fmax(dloc, dd);
        }
        delta = set_reduce_max<THREADS>(dloc, L.red);
        __syncthreads();
        passes++;
        res = B;
        if (delta < P.tol) { converged = 1; break; }
        if (passes >= P.max_iter || delta == __builtin_huge_val()) break;
        if (!P.accel) { double *t = A; A = B; B = t; continue; }
        // pass 2: C = EM(B) with F(B); r = B-A, v = (C-B)-r
        double s4[4];
        s4[0] = set_em_estep<THREADS, true>(L, B);
        s4[1] = s4[2] = s4[3] = 0.0;
        for (int i = threadIdx.x; i < nt; i += THREADS) {
            const double x = B[i], dn = L.den[i], u = L.u[i];
            const double y = set_em_update(x, set_em_acc(L, i), u, dn);
            Cc[i] = y;
            if (u > 0.0 && x > 0.0) s4[0] += u * log(x);
            s4[1] += x * dn;
            const double r = x - A[i], v = (y - x) - r;
            s4[2] += r * r; s4[3] += v * v;
        }
        set_reduce_sum<THREADS, 4>(s4, L.red);
        const double F1 = s4[0] - s4[1];
        double s = s4[3] > 0.0 ? sqrt(s4[2] / s4[3]) : 1.0;
        s = fmin(fmax(s, 1.0), stepmax);
        const bool extrap = s > 1.01;
        // extrapolated point, in place of B
        double s2[2] = {0.0, 0.0};
        __syncthreads();
        for (int i = threadIdx.x; i < nt; i += THREADS) {
            const double x2 = Cc[i];
            double x = x2;
            if (extrap) {
                const double r = B[i] - A[i], v = (x2 - B[i]) - r;
                const double y = A[i] + 2.0 * s * r + s * s * v;
                x = (y > 0.0 && x2 > 0.0) ? y : x2;
            }
            B[i] = x;
            s2[1] += x * L.den[i];
        }
        __syncthreads();
        // pass 3: A = EM(B) with F(B)
        s2[0] = set_em_estep<THREADS, true>(L, B);
        for (int i = threadIdx.x; i < nt; i += THREADS) {
            const double x = B[i], u = L.u[i];
            A[i] = set_em_update(x, set_em_acc(L, i), u, L.den[i]);
            if (u > 0.0 && x > 0.0) s2[0] += u * log(x);
        }
        set_reduce_sum<THREADS, 2>(s2, L.red);
        const bool ok = !extrap || (s2[0] - s2[1] >= F1);
        __syncthreads();
        if (!ok) {
            for (int i = threadIdx.x; i < nt; i += THREADS) A[i] = Cc[i];
            if (s >= stepmax) stepmax = fmax(1.0, stepmax / 4.0);
        }
        if ((ok ? s : 1.0) >= stepmax) stepmax *= 4.0;
        __syncthreads();
        passes += 2;
        res = A;
        if (passes >= P.max_iter) break;
    }
    for (int i = threadIdx.x; i < nt; i += THREADS) theta_g[g_tid[d.tid_off + i]] = res[i];
    if (threadIdx.x == 0) { stat[blockIdx.x].passes = passes; stat[blockIdx.x].converged = converged; stat[blockIdx.x].delta = delta; }
}

// EUMA [rows][nfl] -> [nfl][rows] through a 64 x 64 LDS tile (once per rsh)
__global__ __launch_bounds__(256) void k_transpose_i32(int64_t n_rows, int nfl, const int32_t *__restrict__ in, int32_t *__restrict__ out) {
    __shared__ int32_t tile[64][65];
    const int64_t r0 = (int64_t)blockIdx.x * 64;
    const int c0 = (int)blockIdx.y * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int j = ty; j < 64; j += 4) {
        const int64_t r = r0 + j; const int c = c0 + tx;
        tile[j][tx] = (r < n_rows && c < nfl) ? in[(size_t)r * (size_t)nfl + (size_t)c] : 0;
    }
    __syncthreads();
    for (int j = ty; j < 64; j += 4) {
        const int c = c0 + j; const int64_t r = r0 + tx;
        if (r < n_rows && c < nfl) out[(size_t)c * (size_t)n_rows + (size_t)r] = tile[tx][j];
    }
}
// compute_adjEUMA (emsar_functions.c:2517-2523): one lane per row, fragment lengths in ascending order, product and sum
// rounded separately (no FMA) -- bit-identical to the reference's scalar loop; every load is a coalesced 256 B per wave
__global__ __launch_bounds__(256) void k_adj_euma(int64_t n_rows, int nfl, const int32_t *__restrict__ euma_t, const double *__restrict__ wf,
                                                  double *__restrict__ out) {
#pragma clang fp contract(off)   // hipcc fuses a + x*y into an FMA by default (one rounding instead of the reference's two);
                                 // plain operators: the __dmul_rn / __dadd_rn wrappers carry their own contraction flag
    const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (r >= n_rows) return;
    double a = 0.0;
    int i = 0;
    for (; i + 8 <= nfl; i += 8) {
        int32_t e[8];
#pragma unroll
        for (int j = 0; j < 8; j++) e[j] = euma_t[(size_t)(i + j) * (size_t)n_rows + (size_t)r];
#pragma unroll
        for (int j = 0; j < 8; j++) { const double p = wf[i + j] * (double)e[j]; a = a + p; }
    }
    for (; i < nfl; i++) { const double p = wf[i] * (double)euma_t[(size_t)i * (size_t)n_rows + (size_t)r]; a = a + p; }
    out[r] = a;
}

// transcripts outside every multi-transcript set: theta = (reads of its single-transcript rows) / den
__global__ void k_closed_form(int n, const uint8_t *__restrict__ kind, const double *__restrict__ usum,
                              const double *__restrict__ den, double *__restrict__ theta) {
    int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n && kind[t] == emsar::KIND_CLOSED) theta[t] = den[t] > 0.0 ? usum[t] / den[t] : 0.0;
}

// sum of the mean FPKM (the TPM denominator, emsar_functions.c:3176-3181): ONE workgroup, fixed order -- the printed TPM
// column must not depend on the arrival order of atomics (the per-set solver is bit-reproducible, its output should be too)
__global__ __launch_bounds__(1024) void k_sum(int n, const double *__restrict__ x, double *out) {
    __shared__ double red[16];
    double s = 0.0;
    for (int t = threadIdx.x; t < n; t += 1024) s += x[t];
    double tot = block_sum<1024>(s, red);
    if (threadIdx.x == 0) *out = *out + tot;
}
__global__ __launch_bounds__(256) void k_dot(int n, const double *__restrict__ x, const double *__restrict__ y, double *out) {
    __shared__ double red[4];
    int t = blockIdx.x * 256 + threadIdx.x;
    double s = block_sum<256>(t < n ? x[t] * y[t] : 0.0, red);
    if (threadIdx.x == 0) atomic_add_f64(out, s);
}
// print_FPKMfinal arithmetic (emsar_functions.c:3203-3207): TPM, iReadcount, Round_off
__global__ void k_normalise(int n, const double *__restrict__ mean, const double *__restrict__ ieuma, double nreads_m,
                            const double *total, double *__restrict__ tpm, double *__restrict__ ir, int32_t *__restrict__ iri) {
    int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    double m = mean[t];
    tpm[t] = m * 1E6 / *total;
    double x = (ieuma[t] / 1E3) * m * nreads_m;
    ir[t] = x;
    int xi = (int)x;
    iri[t] = (x - xi >= 0.5) ? xi + 1 : xi;
}

}  // namespace

// ==================================================================================================
// context
// ==================================================================================================
constexpr int64_t kPairMinTiles = 2048;   // 256 CUs x 4 resident workgroups x 2 tiles

struct emsar_hip_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr, ev2 = nullptr;
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
    // WINDOWED layout
    emsar::WindowedLayout L;     // host copy keeps perm / slice_off / chunks (ent freed after upload)
    Chunk *d_chunks = nullptr;
    uint64_t *d_slice_off = nullptr;
    int32_t *d_ent = nullptr;
    int64_t padded_rows = 0;
    // TILED layout
    emsar::TiledLayout TL;       // host copy keeps slot_row / single_* / left_row (index arrays freed after upload)
    Tile *d_tiles = nullptr;
    uint32_t *d_fwd = nullptr, *d_bwd = nullptr;
    uint32_t *d_coo = nullptr;
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
    int update_grid = 256;       // workgroups of k_update (EMSAR_HIP_UPDATE_GRID)
    int sq_grid = 256;           // workgroups of the SQUAREM vector kernels (EMSAR_HIP_SQ_GRID)
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

// bytes one pass actually streams in the chosen layout: index arrays + row weights + the T-sized vectors
inline int64_t stored_bytes(const emsar_hip_ctx *ctx) {
    int64_t rows = ctx->layout == EMSAR_LAYOUT_WINDOWED ? ctx->padded_rows : ctx->layout == EMSAR_LAYOUT_TILED ? ctx->n_slots + ctx->n_left : ctx->n_rows;
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
    ctx->RS = emsar::ResidentSets(); ctx->sets_ready = false; ctx->n_sstat = 0;
}

void free_structure(emsar_hip_ctx *ctx) {
    free_sets(ctx);
    dfree(ctx->d_euma_t); dfree(ctx->d_wf); dfree(ctx->d_adj); ctx->d_euma_t = nullptr; ctx->d_wf = ctx->d_adj = nullptr; ctx->nfl = 0;
    std::vector<uint64_t>().swap(ctx->h_row_ptr); std::vector<int32_t>().swap(ctx->h_col); std::vector<int32_t>().swap(ctx->h_wgt);
    dfree(ctx->d_row_ptr); dfree(ctx->d_col); dfree(ctx->d_chunks); dfree(ctx->d_slice_off); dfree(ctx->d_ent);
    ctx->d_row_ptr = nullptr; ctx->d_col = nullptr; ctx->d_chunks = nullptr; ctx->d_slice_off = nullptr; ctx->d_ent = nullptr;
    dfree(ctx->d_wgt); dfree(ctx->d_rowval); ctx->d_wgt = nullptr; ctx->d_rowval = nullptr;
    dfree(ctx->d_tiles); dfree(ctx->d_fwd); dfree(ctx->d_bwd); dfree(ctx->d_coo); dfree(ctx->d_far);
    dfree(ctx->d_left_ptr); dfree(ctx->d_left_col); dfree(ctx->d_left_wgt); dfree(ctx->d_left_val); dfree(ctx->d_u);
    ctx->d_tiles = nullptr; ctx->d_fwd = ctx->d_bwd = nullptr; ctx->d_coo = nullptr; ctx->d_far = nullptr;
    ctx->d_left_ptr = nullptr; ctx->d_left_col = nullptr; ctx->d_left_wgt = nullptr; ctx->d_left_val = nullptr; ctx->d_u = nullptr;
    ctx->TL = emsar::TiledLayout(); ctx->n_left = ctx->n_tiles = ctx->n_slots = 0;
    dfree(ctx->d_den); dfree(ctx->d_acc); ctx->d_den = nullptr; ctx->d_acc = nullptr;
    for (auto &p : ctx->d_th) { dfree(p); p = nullptr; }
    for (auto &p : ctx->d_tmp) { dfree(p); p = nullptr; }
    dfree(ctx->d_itmp); ctx->d_itmp = nullptr;
    ctx->L = emsar::WindowedLayout();
    ctx->have_structure = ctx->have_sample = false;
}

// one pass of the chosen layout.  mode: MODE_EM / MODE_EM_LL / MODE_SCATTER
int launch_pass(emsar_hip_ctx *ctx, int mode, const double *theta, double *acc, double *ll_out, bool rows_only = false /* the folded rows' likelihood terms are added by the caller */) {
    if (ctx->layout == EMSAR_LAYOUT_TILED) {
        const size_t lds = (size_t)kTiledLdsDoubles * sizeof(double);
        if (ctx->n_tiles > 0) {
            dim3 grid((unsigned)ctx->n_tiles), block(kTiledThreads);
#define LAUNCH_T(WT, MD)                                                                                          \
    hipLaunchKernelGGL((k_pass_tiled<WT, MD>), grid, block, lds, ctx->stream, ctx->d_tiles, ctx->d_fwd, ctx->d_bwd,   \
                       ctx->d_coo, ctx->d_far, ctx->d_wgt, ctx->d_rowval, theta, acc, ll_out)
#define LAUNCH_PN(WT, MD, NN)                                                                                     \
    hipLaunchKernelGGL((k_pass_tiled_multi<WT, MD, NN>), dim3((unsigned)((ctx->n_tiles + NN - 1) / NN)), block, lds, ctx->stream, ctx->d_tiles,  \
                       (int)ctx->n_tiles, ctx->d_fwd, ctx->d_bwd, ctx->d_coo, ctx->d_far, ctx->d_wgt, theta, acc, ll_out)
#define LAUNCH_P(WT, MD) LAUNCH_PN(WT, MD, 2)
            if (mode == MODE_SCATTER) LAUNCH_T(false, MODE_SCATTER);
            else if (!ctx->weighted && (ctx->tiled_multi >= 2 || (ctx->tiled_multi == 1 && ctx->n_tiles > kPairMinTiles))) {
                // two tiles per workgroup, software-pipelined: +3 % on config 3.  Unweighted rows only: with the row
                // weights in registers as well the two-tile body does not fit 128 VGPRs (0.218 vs 0.179 ms measured).
                // Only when the tiles outnumber the chip's workgroup slots: below that a pass is one workgroup's latency, and
                // a pair takes twice as long as a tile (40 k reads: 47 -> 26 us per pass with one tile per workgroup)
                if (mode == MODE_EM_LL) LAUNCH_P(false, MODE_EM_LL); else LAUNCH_P(false, MODE_EM);
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
                       ctx->d_left_col, ctx->d_left_wgt, ctx->d_left_val, theta, acc, ll_out)
            if (mode == MODE_SCATTER) LAUNCH_L(false, MODE_SCATTER);
            else if (ctx->weighted) { if (mode == MODE_EM_LL) LAUNCH_L(true, MODE_EM_LL); else LAUNCH_L(true, MODE_EM); }
            else { if (mode == MODE_EM_LL) LAUNCH_L(false, MODE_EM_LL); else LAUNCH_L(false, MODE_EM); }
#undef LAUNCH_L
        }
        if (mode == MODE_EM_LL && !rows_only)
            hipLaunchKernelGGL(k_single_ll, dim3(std::min(grid_for(ctx->n_tx, 256), 256)), dim3(256), 0, ctx->stream, ctx->n_tx,
                               ctx->d_u, theta, ll_out);
        HIPCHK(hipGetLastError());
        return EMSAR_HIP_OK;
    }
    if (ctx->layout == EMSAR_LAYOUT_WINDOWED) {
        const int W = ctx->L.window;
        const size_t lds = (size_t)W * 2 * sizeof(double);
        dim3 grid((unsigned)ctx->L.chunks.size()), block(kPassThreads);
        if (grid.x == 0) return EMSAR_HIP_OK;
#define LAUNCH_W(WT, MD)                                                                                          \
    hipLaunchKernelGGL((k_pass_windowed<kPassThreads, WT, MD>), grid, block, lds, ctx->stream, ctx->d_chunks,      \
                       ctx->d_slice_off, ctx->d_ent, ctx->d_wgt, ctx->d_rowval, theta, acc, ll_out, W)
        if (mode == MODE_SCATTER) LAUNCH_W(false, MODE_SCATTER);
        else if (ctx->weighted) { if (mode == MODE_EM_LL) LAUNCH_W(true, MODE_EM_LL); else LAUNCH_W(true, MODE_EM); }
        else { if (mode == MODE_EM_LL) LAUNCH_W(false, MODE_EM_LL); else LAUNCH_W(false, MODE_EM); }
#undef LAUNCH_W
    } else {
        if (ctx->n_rows == 0) return EMSAR_HIP_OK;
        int64_t blocks = (ctx->n_rows + 255) / 256;
        dim3 grid((unsigned)std::min<int64_t>(blocks, 256 * 32)), block(256);
#define LAUNCH_C(PT, WT, MD)                                                                                     \
    hipLaunchKernelGGL((k_pass_csr<PT, WT, MD>), grid, block, 0, ctx->stream, ctx->n_rows, (const PT *)ctx->d_row_ptr, \
                       ctx->d_col, ctx->d_wgt, ctx->d_rowval, theta, acc, ll_out)
#define LAUNCH_CP(WT, MD) do { if (ctx->ptr64) LAUNCH_C(uint64_t, WT, MD); else LAUNCH_C(uint32_t, WT, MD); } while (0)
        if (mode == MODE_SCATTER) LAUNCH_CP(false, MODE_SCATTER);
        else if (ctx->weighted) { if (mode == MODE_EM_LL) LAUNCH_CP(true, MODE_EM_LL); else LAUNCH_CP(true, MODE_EM); }
        else { if (mode == MODE_EM_LL) LAUNCH_CP(false, MODE_EM_LL); else LAUNCH_CP(false, MODE_EM); }
#undef LAUNCH_CP
#undef LAUNCH_C
    }
    HIPCHK(hipGetLastError());
    return EMSAR_HIP_OK;
}

// th_out = EM(th_in); ll slot receives sum R log S at th_in when want_ll
int em_pass(emsar_hip_ctx *ctx, const double *th_in, double *th_out, bool want_ll, int ll_slot, double abs_floor, int to_delta1 = 0) {
    int rc = launch_pass(ctx, want_ll ? MODE_EM_LL : MODE_EM, th_in, ctx->d_acc, &ctx->d_scal->ll[ll_slot]);
    if (rc) return rc;
    hipLaunchKernelGGL(k_update, dim3(std::min(grid_for(ctx->n_tx, 256), ctx->update_grid)), dim3(256), 0, ctx->stream, ctx->n_tx, th_in, ctx->d_acc,
                       ctx->d_den, ctx->layout == EMSAR_LAYOUT_TILED ? ctx->d_u : nullptr, th_out, abs_floor, ctx->count_floor, ctx->zero_cut, ctx->d_scal,
                       ctx->delta_mask, to_delta1);
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
        hipLaunchKernelGGL(k_cycle_begin, dim3(1), dim3(1), 0, ctx->stream, ctx->d_scal, abs_step_base, p.accel ? 3 : 1);
        if (!p.accel) {
            if ((rc = em_pass(ctx, th[0], th[1], false, 0, p.abs_floor))) return rc;
            std::swap(th[0], th[1]);
            continue;
        }
        // the stopping rule is measured on the first (plain) step of the cycle only (delta1_bits)
        const double *u = ctx->layout == EMSAR_LAYOUT_TILED ? ctx->d_u : nullptr;
        const dim3 gv((unsigned)std::min(g, ctx->sq_grid)), bv(256);
        if ((rc = em_pass(ctx, th[0], th[1], false, 0, p.abs_floor, 1))) return rc;
        if ((rc = launch_pass(ctx, MODE_EM_LL, th[1], ctx->d_acc, &ctx->d_scal->ll[1], true))) return rc;
        hipLaunchKernelGGL(k_update_p2, gv, bv, 0, ctx->stream, n, th[0], th[1], ctx->d_acc, ctx->d_den, u, th[2], ctx->d_scal);
        hipLaunchKernelGGL(k_sq_extrap_ll, gv, bv, 0, ctx->stream, n, th[0], th[1], th[2], ctx->d_den, u, th[3], ctx->d_scal);
        if ((rc = launch_pass(ctx, MODE_EM_LL, th[3], ctx->d_acc, &ctx->d_scal->ll[2], true))) return rc;
        hipLaunchKernelGGL(k_update_p3, gv, bv, 0, ctx->stream, n, th[3], th[2], ctx->d_acc, ctx->d_den, u, th[0], ctx->d_scal);
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
    std::vector<double> tmp;
    const double *src = val_host;
    size_t n = (size_t)ctx->n_rows;
    if (ctx->layout == EMSAR_LAYOUT_WINDOWED) {
        n = (size_t)ctx->padded_rows;
        tmp.assign(n, 0.0);
        for (int64_t i = 0; i < ctx->L.n_sorted_rows; i++) tmp[(size_t)i] = val_host[ctx->L.perm[(size_t)i]];
        src = tmp.data();
    }
    if (n == 0) return EMSAR_HIP_OK;
    if (!ctx->d_rowval) HIPCHK(hipMalloc(&ctx->d_rowval, std::max<size_t>(n, 1) * sizeof(double)));
    HIPCHK(hipMemcpyAsync(ctx->d_rowval, src, n * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipMemsetAsync(d_out, 0, (size_t)ctx->n_tx * sizeof(double), ctx->stream));
    int rc = launch_pass(ctx, MODE_SCATTER, nullptr, d_out, nullptr);
    if (rc) return rc;
    HIPCHK(hipStreamSynchronize(ctx->stream));  // tmp must outlive the copy
    return EMSAR_HIP_OK;
}

// find and pack the connected sets of the current sample (sets.hpp) and move the records to the device
int ensure_sets(emsar_hip_ctx *ctx) {
    if (ctx->sets_ready) return EMSAR_HIP_OK;
    auto t0 = std::chrono::steady_clock::now();
    auto &S = ctx->RS;
    try {
        emsar::build_sets(ctx->n_rows, ctx->n_tx, ctx->h_row_ptr.data(), ctx->h_col.data(), ctx->h_wgt.data(), S);
    } catch (const std::bad_alloc &) { free_sets(ctx); return EMSAR_HIP_ERR_OOM; }
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
    // the device copies are the only ones needed from here on (desc sizes and counters stay)
    std::vector<int32_t>().swap(S.g_tid); std::vector<double>().swap(S.g_u); std::vector<double>().swap(S.row_w);
    std::vector<uint16_t>().swap(S.rp); std::vector<uint16_t>().swap(S.ent); std::vector<uint16_t>().swap(S.cp); std::vector<uint16_t>().swap(S.crow);
    std::vector<double>().swap(S.usum);
    ctx->sets_build_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    ctx->sets_ready = true;
    return EMSAR_HIP_OK;
}

// closed-form transcripts and every LDS-resident set, written into theta (the streamed sets' entries are left alone)
int solve_resident_sets(emsar_hip_ctx *ctx, const SetSolveParams &P, double *theta) {
    const auto &S = ctx->RS;
    hipLaunchKernelGGL(k_closed_form, dim3(grid_for(ctx->n_tx, 256)), dim3(256), 0, ctx->stream, ctx->n_tx, ctx->d_kind, ctx->d_usum,
                       ctx->d_den, theta);
    size_t off = 0;
#define LAUNCH_S(C, TH)                                                                                                    \
    if (!S.desc[C].empty()) {                                                                                              \
        hipLaunchKernelGGL(k_solve_sets<TH>, dim3((unsigned)S.desc[C].size()), dim3(TH), S.max_lds[C], ctx->stream,          \
                           ctx->d_sdesc[C], ctx->d_g_tid, ctx->d_g_u, ctx->d_row_w, ctx->d_srp, ctx->d_sent, ctx->d_scp,     \
                           ctx->d_scrow, ctx->d_den, theta, ctx->d_sstat + off, P);                                        \
        off += S.desc[C].size();                                                                                           \
    }
    LAUNCH_S(0, 64) LAUNCH_S(1, 256) LAUNCH_S(2, 512)
#undef LAUNCH_S
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
    if (const char *e = getenv("EMSAR_HIP_GRAPH")) ctx->use_graph = atoi(e) != 0;
    if (hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess) return fail(EMSAR_HIP_ERR_HIP);
    if (hipEventCreate(&ctx->ev0) != hipSuccess || hipEventCreate(&ctx->ev1) != hipSuccess || hipEventCreate(&ctx->ev2) != hipSuccess) return fail(EMSAR_HIP_ERR_HIP);
    if (hipMalloc(&ctx->d_scal, sizeof(Scal)) != hipSuccess) return fail(EMSAR_HIP_ERR_OOM);
    if (hipHostMalloc((void **)&ctx->h_scal, sizeof(Scal), hipHostMallocDefault) != hipSuccess) return fail(EMSAR_HIP_ERR_OOM);
    // both pass kernels may need more than the default dynamic-LDS limit
    *out = ctx;
    return EMSAR_HIP_OK;
}

void emsar_hip_destroy(emsar_hip_ctx *ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    free_structure(ctx);
    dfree(ctx->d_scal);
    if (ctx->h_scal) (void)hipHostFree(ctx->h_scal);
    if (ctx->ev0) (void)hipEventDestroy(ctx->ev0);
    if (ctx->ev1) (void)hipEventDestroy(ctx->ev1);
    if (ctx->ev2) (void)hipEventDestroy(ctx->ev2);
    if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

int emsar_hip_upload_structure(emsar_hip_ctx *ctx, int64_t n_rows, int32_t n_tx, const uint64_t *row_ptr,
                               const int32_t *col_idx, int layout) {
    if (!ctx) return EMSAR_HIP_ERR_ARG;
    const bool merge_rows = (layout & EMSAR_LAYOUT_FLAG_MERGE_ROWS) != 0;
    layout &= ~EMSAR_LAYOUT_FLAG_MERGE_ROWS;
    if (layout != EMSAR_LAYOUT_AUTO && layout != EMSAR_LAYOUT_CSR && layout != EMSAR_LAYOUT_WINDOWED && layout != EMSAR_LAYOUT_TILED) return EMSAR_HIP_ERR_ARG;
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
        if (const char *e = getenv("EMSAR_HIP_LAYOUT")) { int v = atoi(e); if (v >= 1 && v <= 3 && (v == 1 || n_rows < ((int64_t)1 << 32))) layout = v; }
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
            if (emsar::build_tiled(n_rows, n_tx, row_ptr, col_idx, L, merge_rows) != 0) return EMSAR_HIP_ERR_ARG;
            if (dbg) fprintf(stderr, "upload_structure: layout built after %.0f ms\n", since(tu0));
            ctx->n_tiles = (int64_t)L.tiles.size(); ctx->n_slots = L.n_slots(); ctx->n_left = (int64_t)L.left_row.size();
            auto up = [&](void **dp, const void *src, size_t bytes) -> hipError_t {
                hipError_t e = hipMalloc(dp, std::max<size_t>(bytes, 16));
                if (e == hipSuccess && bytes) e = hipMemcpy(*dp, src, bytes, hipMemcpyHostToDevice);
                return e;
            };
            HIPCHK(up((void **)&ctx->d_tiles, L.tiles.data(), L.tiles.size() * sizeof(Tile)));
            HIPCHK(up((void **)&ctx->d_fwd, L.fwd.data(), L.fwd.size() * 4));
            HIPCHK(up((void **)&ctx->d_bwd, L.bwd.data(), L.bwd.size() * 4));
            HIPCHK(up((void **)&ctx->d_coo, L.coo.data(), L.coo.size() * 4));
            HIPCHK(up((void **)&ctx->d_far, L.far_tid.data(), L.far_tid.size() * 4));
            HIPCHK(up((void **)&ctx->d_left_ptr, L.left_ptr.data(), L.left_ptr.size() * 8));
            HIPCHK(up((void **)&ctx->d_left_col, L.left_col.data(), L.left_col.size() * 4));
            HIPCHK(hipMalloc(&ctx->d_u, T * 8));
            HIPCHK(hipMemset(ctx->d_u, 0, T * 8));
            ctx->bytes_stored = (int64_t)L.fwd.size() * 4 + (int64_t)L.bwd.size() * 4 + (int64_t)L.coo.size() * 4 + (int64_t)L.far_tid.size() * 4 +
                                (int64_t)L.tiles.size() * 64 + (int64_t)L.left_col.size() * 4 + (int64_t)L.left_ptr.size() * 8;
            ctx->tl_fwd_slots = L.padded_slots; ctx->tl_n_fslices = L.n_fslices;
            std::vector<uint32_t>().swap(L.fwd); std::vector<uint32_t>().swap(L.bwd); std::vector<uint32_t>().swap(L.coo);
            std::vector<int32_t>().swap(L.left_col);
            const size_t lds = (size_t)kTiledLdsDoubles * sizeof(double);
#define SETLDS_T(WT, MD) HIPCHK(hipFuncSetAttribute((const void *)k_pass_tiled<WT, MD>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds))
            SETLDS_T(false, MODE_EM); SETLDS_T(false, MODE_EM_LL); SETLDS_T(true, MODE_EM); SETLDS_T(true, MODE_EM_LL); SETLDS_T(false, MODE_SCATTER);
#undef SETLDS_T
#define SETLDS_P(WT, MD, NN) HIPCHK(hipFuncSetAttribute((const void *)k_pass_tiled_multi<WT, MD, NN>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds))
            SETLDS_P(false, MODE_EM, 2); SETLDS_P(false, MODE_EM_LL, 2);
#undef SETLDS_P
            { const char *pe = getenv("EMSAR_HIP_TILED_MULTI"); ctx->tiled_multi = pe ? atoi(pe) : 1; }
            { const char *pe = getenv("EMSAR_HIP_UPDATE_GRID"); if (pe && atoi(pe) >= 1) ctx->update_grid = atoi(pe); }
            { const char *pe = getenv("EMSAR_HIP_SQ_GRID"); if (pe && atoi(pe) >= 1) ctx->sq_grid = atoi(pe); }
        } else if (layout == EMSAR_LAYOUT_WINDOWED) {
            const char *wenv = getenv("EMSAR_HIP_WINDOW");
            int window = wenv ? atoi(wenv) : kDefaultWindow;
            if (window < emsar::kMinBlockTids || window > 8192) window = kDefaultWindow;
            const char *cenv = getenv("EMSAR_HIP_CHUNK_ENTRIES");
            int64_t chunk_entries = cenv ? atoll(cenv) : kChunkEntries;
            if (chunk_entries < 1024) chunk_entries = kChunkEntries;
            if (emsar::build_windowed(n_rows, n_tx, row_ptr, col_idx, window, chunk_entries, ctx->L) != 0) return EMSAR_HIP_ERR_ARG;
            auto &L = ctx->L;
            ctx->padded_rows = L.n_slices() * emsar::kSliceRows;
            HIPCHK(hipMalloc(&ctx->d_chunks, std::max<size_t>(L.chunks.size(), 1) * sizeof(Chunk)));
            HIPCHK(hipMalloc(&ctx->d_slice_off, L.slice_off.size() * sizeof(uint64_t)));
            HIPCHK(hipMalloc(&ctx->d_ent, std::max<size_t>(L.ent.size(), 1) * sizeof(int32_t)));
            HIPCHK(hipMemcpy(ctx->d_chunks, L.chunks.data(), L.chunks.size() * sizeof(Chunk), hipMemcpyHostToDevice));
            HIPCHK(hipMemcpy(ctx->d_slice_off, L.slice_off.data(), L.slice_off.size() * sizeof(uint64_t), hipMemcpyHostToDevice));
            HIPCHK(hipMemcpy(ctx->d_ent, L.ent.data(), L.ent.size() * sizeof(int32_t), hipMemcpyHostToDevice));
            ctx->bytes_stored = (int64_t)L.ent.size() * 4 + (int64_t)L.slice_off.size() * 8 + (int64_t)L.chunks.size() * 16;
            std::vector<int32_t>().swap(L.ent);  // the device copy is the only one needed from here on
            const size_t lds = (size_t)window * 2 * sizeof(double);
#define SETLDS(WT, MD) HIPCHK(hipFuncSetAttribute((const void *)k_pass_windowed<kPassThreads, WT, MD>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds))
            SETLDS(false, MODE_EM); SETLDS(false, MODE_EM_LL); SETLDS(true, MODE_EM); SETLDS(true, MODE_EM_LL); SETLDS(false, MODE_SCATTER);
#undef SETLDS
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
    // a row counts w = R (or 1) when it is inside the likelihood (E != 0), else 0
    ctx->weighted = (row_weight != nullptr) || (row_E != nullptr) || (ctx->layout == EMSAR_LAYOUT_TILED && ctx->TL.merged);
    ctx->loglik_const = 0.0;
    dfree(ctx->d_wgt); ctx->d_wgt = nullptr;
    if (row_weight || row_E) {
        for (int64_t r = 0; r < n_rows; r++) {
            if (row_weight && row_weight[r] < 0) return EMSAR_HIP_ERR_ARG;
            if (row_E && !(row_E[r] >= 0.0)) return EMSAR_HIP_ERR_ARG;  // negative or NaN
        }
    }
    auto weight_of = [&](int64_t r) -> int32_t {
        int32_t x = row_weight ? row_weight[r] : 1;
        if (row_E && row_E[r] == 0.0) x = 0;
        return x;
    };
    free_sets(ctx);
    try {
        ctx->h_wgt.resize((size_t)n_rows);
        for (int64_t r = 0; r < n_rows; r++) ctx->h_wgt[(size_t)r] = weight_of(r);
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
        const bool win = ctx->layout == EMSAR_LAYOUT_WINDOWED;
        size_t n = win ? (size_t)ctx->padded_rows : (size_t)n_rows;
        std::vector<int32_t> w(std::max<size_t>(n, 1), 0);
        int64_t cnt = win ? ctx->L.n_sorted_rows : n_rows;
        for (int64_t i = 0; i < cnt; i++) {
            int64_t r = win ? (int64_t)ctx->L.perm[(size_t)i] : i;
            int32_t x = row_weight ? row_weight[r] : 1;
            if (row_E && row_E[r] == 0.0) x = 0;
            w[(size_t)i] = x;
            if (x > 0 && row_E) ctx->loglik_const += (double)x * std::log(row_E[r]);
        }
        HIPCHK(hipMalloc(&ctx->d_wgt, w.size() * 4));
        HIPCHK(hipMemcpy(ctx->d_wgt, w.data(), w.size() * 4, hipMemcpyHostToDevice));
    }
    if (den) {
        for (int32_t t = 0; t < ctx->n_tx; t++) if (!(den[t] >= 0.0)) return EMSAR_HIP_ERR_ARG;
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
    return EMSAR_HIP_OK;
}

int emsar_hip_run_passes(emsar_hip_ctx *ctx, int32_t n_passes, float *elapsed_ms, double *last_ll) {
    if (!ctx || n_passes < 0) return EMSAR_HIP_ERR_ARG;
    if (!ctx->have_sample) return EMSAR_HIP_ERR_STATE;
    HIPCHK(hipSetDevice(ctx->device));
    ctx->delta_mask = nullptr;
    hipLaunchKernelGGL(k_cycle_begin, dim3(1), dim3(1), 0, ctx->stream, ctx->d_scal, 0.0, 0);
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
    if (last_ll) *last_ll = ctx->h_scal->ll[0];
    return EMSAR_HIP_OK;
}

int emsar_hip_solve(emsar_hip_ctx *ctx, const emsar_em_params *pp, double *fpkm_out, emsar_em_stats *stats) {
    if (!ctx || !fpkm_out) return EMSAR_HIP_ERR_ARG;
    if (!ctx->have_sample) return EMSAR_HIP_ERR_STATE;
    emsar_em_params p = pp ? *pp : emsar_em_params{0, 1, 0, 0, 0, 0, 0, 0, 0};
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
            if (!std::isfinite(delta)) { ctx->err = "non-finite theta"; return EMSAR_HIP_ERR_NUMERIC; }
            if (delta < p.tol) { converged = 1; break; }
        }
    }
    ctx->delta_mask = nullptr;
    HIPCHK(hipEventRecord(ctx->ev1, ctx->stream));
    if (use_sets) {
        SetSolveParams P{p.tol, p.abs_floor, p.count_floor, p.zero_cut > 0.0 ? p.zero_cut : 0.0, p.abs_step > 0.0 ? p.abs_step : 0.0, p.max_iter, p.accel};
        if ((rc = solve_resident_sets(ctx, P, th[0]))) return rc;
        if (ctx->n_sstat > 0)
            HIPCHK(hipMemcpyAsync(ctx->h_sstat, ctx->d_sstat, (size_t)ctx->n_sstat * sizeof(SetStat), hipMemcpyDeviceToHost, ctx->stream));
    }
    HIPCHK(hipEventRecord(ctx->ev2, ctx->stream));
    // F at the returned point: one likelihood-only pass (not counted in iters)
    hipLaunchKernelGGL(k_cycle_begin, dim3(1), dim3(1), 0, ctx->stream, ctx->d_scal, 0.0, 0);
    if ((rc = launch_pass(ctx, MODE_EM_LL, th[0], ctx->d_acc, &ctx->d_scal->ll[0]))) return rc;
    HIPCHK(hipMemsetAsync(ctx->d_acc, 0, (size_t)n * 8, ctx->stream));
    hipLaunchKernelGGL(k_dot, dim3(g), dim3(256), 0, ctx->stream, n, th[0], ctx->d_den, &ctx->d_scal->ll[3]);
    HIPCHK(hipMemcpyAsync(ctx->h_scal, ctx->d_scal, sizeof(Scal), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipMemcpyAsync(fpkm_out, th[0], (size_t)n * 8, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
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
            if (q.delta > delta) delta = q.delta;
        }
    if (set_unconv) converged = 0;
    if (stats) {
        float ms = 0, ms_sets = 0;
        HIPCHK(hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1));
        HIPCHK(hipEventElapsedTime(&ms_sets, ctx->ev1, ctx->ev2));
        memset(stats, 0, sizeof(*stats));
        stats->iters = iters + set_max;
        stats->converged = converged;
        stats->final_delta = delta;
        stats->loglik = ctx->h_scal->ll[0] + ctx->loglik_const - ctx->h_scal->ll[3];
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
    }
    if (ctx->layout == EMSAR_LAYOUT_WINDOWED) {
        o->n_chunks = (int64_t)ctx->L.chunks.size();
        o->n_slices = ctx->L.n_slices();
        o->padded_entries = (int64_t)ctx->L.slice_off.back();
        o->far_entries = ctx->L.far_entries;
        o->window = ctx->L.window;
    }
    o->bytes_per_pass = ctx->bytes_formula;
    o->stored_bytes_per_pass = stored_bytes(ctx);
    return EMSAR_HIP_OK;
}

// Host-only self check of the WINDOWED layout builder (no HIP call: usable on a machine without a GPU).
// Builds the layout for the given CSR, decodes it again and compares; fills *info_out (may be NULL).
int emsar_hip_layout_selfcheck(int64_t n_rows, int32_t n_tx, const uint64_t *row_ptr, const int32_t *col_idx,
                               int32_t window, int64_t chunk_entries, emsar_hip_info *info_out) {
    if (emsar::validate_csr(n_rows, n_tx, row_ptr, col_idx) != 0) return EMSAR_HIP_ERR_ARG;
    emsar::WindowedLayout L;
    if (window <= 0) window = kDefaultWindow;
    if (chunk_entries <= 0) chunk_entries = kChunkEntries;
    if (emsar::build_windowed(n_rows, n_tx, row_ptr, col_idx, window, chunk_entries, L) != 0) return EMSAR_HIP_ERR_ARG;
    int rc = emsar::check_windowed(L, row_ptr, col_idx);
    if (info_out) {
        memset(info_out, 0, sizeof(*info_out));
        info_out->n_rows = n_rows; info_out->nnz = L.nnz; info_out->n_tx = n_tx; info_out->layout = EMSAR_LAYOUT_WINDOWED;
        info_out->n_chunks = (int64_t)L.chunks.size(); info_out->n_slices = L.n_slices();
        info_out->padded_entries = (int64_t)L.slice_off.back(); info_out->far_entries = L.far_entries; info_out->window = window;
    }
    return rc == 0 ? EMSAR_HIP_OK : EMSAR_HIP_ERR_ARG - 100 + rc;
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
                       ctx->d_fwd, ctx->d_bwd, ctx->d_coo, ctx->d_far, ctx->d_wgt, ctx->d_rowval, ctx->d_th[0], ctx->d_acc, &ctx->d_scal->ll[3], d);
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

int emsar_hip_layout_selfcheck_tiled(int64_t n_rows, int32_t n_tx, const uint64_t *row_ptr, const int32_t *col_idx,
                                     int merge_rows, emsar_hip_info *info_out) {
    if (emsar::validate_csr(n_rows, n_tx, row_ptr, col_idx) != 0) return EMSAR_HIP_ERR_ARG;
    emsar::TiledLayout L;
    if (emsar::build_tiled(n_rows, n_tx, row_ptr, col_idx, L, merge_rows != 0) != 0) return EMSAR_HIP_ERR_ARG;
    int rc = emsar::check_tiled(L, row_ptr, col_idx);
    if (info_out) {
        memset(info_out, 0, sizeof(*info_out));
        info_out->n_rows = n_rows; info_out->nnz = L.nnz; info_out->n_tx = n_tx;
        info_out->layout = EMSAR_LAYOUT_TILED | (L.merged ? EMSAR_LAYOUT_FLAG_MERGE_ROWS : 0);
        info_out->n_chunks = (int64_t)L.tiles.size();
        info_out->n_slices = L.n_fslices;
        info_out->padded_entries = L.padded_slots; info_out->far_entries = L.far_entries; info_out->window = emsar::kTileDict;
        info_out->stored_bytes_per_pass = (int64_t)L.fwd.size() * 4 + (int64_t)L.bwd.size() * 4 + (int64_t)L.coo.size() * 4 +
                                          (int64_t)L.far_tid.size() * 4 + (int64_t)L.tiles.size() * 64 + (int64_t)L.left_col.size() * 4;
        info_out->bytes_per_pass = (int64_t)L.single_row.size();   /* diagnostic: number of folded single-tid rows */
    }
    return rc == 0 ? EMSAR_HIP_OK : EMSAR_HIP_ERR_ARG - 100 + rc;
}

int emsar_hip_sets_selfcheck(int64_t n_rows, int32_t n_tx, const uint64_t *row_ptr, const int32_t *col_idx,
                             const int32_t *row_weight, emsar_hip_sets_info *o) {
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
        o->rows_in = S.rows_in; o->rows_stored = S.rows_stored;
    }
    return rc == 0 ? EMSAR_HIP_OK : EMSAR_HIP_ERR_ARG - 200 + rc;
}

}  // extern "C"
