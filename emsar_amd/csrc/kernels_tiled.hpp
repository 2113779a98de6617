// kernels_tiled.hpp -- k_pass_tiled: the EM pass over the TILED layout (layout_tiled.hpp), the default
#pragma once
// included by emsar_hip.hip only (one translation unit: the kernels live in its anonymous namespace)

namespace {

// LDS reads in flight per gather batch (experiment knobs; 12 = a whole int4 of ids, 6 = half: 18 fewer VGPRs)
#ifndef EMSAR_E_BATCH
#define EMSAR_E_BATCH 6
#endif
#ifndef EMSAR_M_BATCH
#define EMSAR_M_BATCH 12
#endif
constexpr int kTiledThreads = 64 * emsar::kTileWaves;     // 4 wavefronts, each working on one slice at a time
constexpr int kRPL = emsar::kRowsPerLane;                 // 12 rows per lane = twelve 10-bit ids per int4
constexpr int kTiledWr = emsar::kTileSliceRows + 8;       // w_r of one slice (768) + the zero padding row
constexpr int kTiledDictPad = emsar::kTileDict + 1;       // 960 slots incl. the zero slot
constexpr int kTiledLdsDoubles = 2 * kTiledDictPad + emsar::kTileWaves * kTiledWr;   // 40,192 B: 4 workgroups per CU

__device__ __forceinline__ double lds_at(const double *base, unsigned byte_off) {
    return *reinterpret_cast<const double *>(reinterpret_cast<const char *>(base) + byte_off);
}
// field f (0,1,2) of a packed dword -> LDS byte offset of the double it names
__device__ __forceinline__ unsigned id_off(unsigned dword, int f) { return ((dword >> (10 * f)) & 0x3FFu) << 3; }

// 8 independent 16-byte loads; positions beyond n repeat position n-1 (an L1 hit) so that there is no control flow
// between the loads and all of them are in flight together
// The index stream is read exactly once per pass: non-temporal loads (global_load_dwordx4 ... nt) keep it from evicting
// theta and the acc lines from the XCD's L2 -- 0.1805 -> 0.1745 ms per pass on config 3.
typedef int v4i_t __attribute__((ext_vector_type(4)));
// base is wave-uniform (it lives in scalar registers), the lane adds its own 16 bytes: one VGPR of address per stream
__device__ __forceinline__ void load8_clamped(int4 (&q)[8], const int4 *base, unsigned lane, int n) {
#pragma unroll
    for (int j = 0; j < 8; j++) {
        const int4 *col = base + (size_t)(unsigned)(j < n ? j : n - 1) * 64;          // uniform
        const v4i_t t = __builtin_nontemporal_load(reinterpret_cast<const v4i_t *>(col + lane));
        q[j] = make_int4(t.x, t.y, t.z, t.w);
    }
}

// LDS gather of the double named by 10-bit field F of a packed dword: two VALU instructions per entry
// (v_bfe_u32 + v_lshl_add_u32 with the region's LDS byte address as the scalar addend) instead of the shift / and /
// add-base triple hipcc emits for the C expression -- the E- and M-steps are bound by instruction issue
// (4 cycles per wave64 instruction on a 16-lane SIMD), not by LDS bandwidth.
typedef __attribute__((address_space(3))) const double lds_cdouble;
__device__ __forceinline__ unsigned lds_byte_addr(const void *p) { return (unsigned)(uintptr_t)p; }   // low half of a flat LDS address
// timing-only ablations (never in a shipped build; results are wrong): with 5-bit ids two lanes that hit one LDS bank hit one
// address, i.e. the gather runs without bank conflicts -- EMSAR_ABL_E5 / EMSAR_ABL_M5 bound what a conflict-free E / M gather could win
#ifdef EMSAR_ABL_E5
constexpr int kIdBitsE = 5;
#else
constexpr int kIdBitsE = 10;
#endif
#ifdef EMSAR_ABL_M5
constexpr int kIdBitsM = 5;
#else
constexpr int kIdBitsM = 10;
#endif
template <int F, int BITS = 10>
__device__ __forceinline__ unsigned lds_id_addr(unsigned dword, unsigned base /* wave-uniform */) {
    // inline asm: hipcc rewrites the C expression (and the ubfe intrinsic) back into shift + and + add
    unsigned a;   // one statement: hipcc pads a nop between two dependent asm statements
    asm("v_bfe_u32 %0, %1, %2, %4\n\tv_lshl_add_u32 %0, %0, 3, %3" : "=v"(a) : "v"(dword), "i"(10 * F), "s"(base), "i"(BITS));
    return a;
}
__device__ __forceinline__ double lds_ld(unsigned a) { return *reinterpret_cast<lds_cdouble *>(a); }
// the LDS byte addresses of 6 of the 12 ids of one int4 (H = 0: fields of .x .y, H = 1: of .z .w).  Addresses first,
// then the loads back to back, then the adds: the asm statements would otherwise serialise address -> load -> wait ->
// add per entry.  BATCH = 12 keeps a whole int4 in flight (36 temporaries), BATCH = 6 half of it (18).
template <int H, int BITS = 10>
__device__ __forceinline__ void lds_addr6(const int4 t, unsigned base, unsigned (&a)[6]) {
    const unsigned d0 = (unsigned)(H ? t.z : t.x), d1 = (unsigned)(H ? t.w : t.y);
    a[0] = lds_id_addr<0, BITS>(d0, base); a[1] = lds_id_addr<1, BITS>(d0, base); a[2] = lds_id_addr<2, BITS>(d0, base);
    a[3] = lds_id_addr<0, BITS>(d1, base); a[4] = lds_id_addr<1, BITS>(d1, base); a[5] = lds_id_addr<2, BITS>(d1, base);
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
                lds_addr6<0, kIdBitsE>(q[j], th_base, a0); lds_addr6<1, kIdBitsE>(q[j], th_base, a1);
#pragma unroll
                for (int i = 0; i < 6; i++) { v[i] = lds_ld(a0[i]); }
#pragma unroll
                for (int i = 0; i < 6; i++) { v[6 + i] = lds_ld(a1[i]); }
#pragma unroll
                for (int i = 0; i < 12; i++) S[i] += v[i];
            } else {
                unsigned a[6];
                double v[6];
                lds_addr6<0, kIdBitsE>(q[j], th_base, a);
#pragma unroll
                for (int i = 0; i < 6; i++) v[i] = lds_ld(a[i]);
#pragma unroll
                for (int i = 0; i < 6; i++) S[i] += v[i];
                lds_addr6<1, kIdBitsE>(q[j], th_base, a);
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
                lds_addr6<0, kIdBitsM>(q[j], ws_base, a0); lds_addr6<1, kIdBitsM>(q[j], ws_base, a1);
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
                lds_addr6<0, kIdBitsM>(q[j], ws_base, a);
#pragma unroll
                for (int i = 1; i < 6; i++) v[i] = lds_ld(a[i]);
                double s0 = v[1] + v[2], s1 = v[3] + v[4];
                s0 += v[5];
                lds_addr6<1, kIdBitsM>(q[j], ws_base, a);
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

// ------------------------------------------------------------------------------------------------
// k_pass_tiled: one EM pass over the TILED layout (layout_tiled.hpp).  One workgroup (4 waves) = one CHUNK.
//   per GROUP of the chunk (slices that share a dictionary):
//     dictionary   th_w[d] = theta[tid(d)], acc_w[d] = 0, zero slot; barrier
//     slices       every wave takes slices of the group one after another (the first four statically, the rest from a
//                  counter in LDS: the waves stay busy until the group runs out, whatever the slices' lengths).  Per slice:
//       loads      far tids (they come back first), 8 forward columns, 8 backward segments -- all in flight together
//       E          the wave owns the slice's 768 rows (lane l holds rows 64*i + l, i < 12): S_r = sum th_w[id] (LDS reads only,
//                  10-bit ids, padding reads the zero slot: no branches); the rows of the last one or two fields may carry an
//                  EXPORTED far entry: its theta is gathered from global memory straight into that field's sum
//                  w_r = R_r / S_r -> the wave's own 6 KiB of LDS; the far fields' w_r also go to far_w, each to its entry's place in
//                  transcript order (the update kernels add the contiguous run of a transcript: no dictionary slot, no atomic)
//       M          the SAME wave walks the transposed index of its rows: a lane's segments (column id + 11 row ids) are
//                  consecutive in column order; it gathers w_r from LDS into a register sum and adds it to acc_w when the
//                  column changes; tiny columns via a COO list.  No barrier between E and M, none between slices.
//     flush        barrier; non-zero dictionary slots are flushed with one global FP64 atomic each
// HBM traffic: 10 bits per forward slot + 128 bits per 11 backward entries -- no row_ptr, no 32-bit tids; a dictionary is
// loaded and flushed once per group (tens of slices), not once per 4 slices.
// ------------------------------------------------------------------------------------------------
struct TiledArgs {
    const emsar::ChunkDesc *chunks; const emsar::GroupDesc *groups; const emsar::SliceDesc *slices;
    const uint32_t *fwd, *bwd, *coo;
    const int32_t *far_dict;        // explicit dictionary far lists
    const int32_t *far_blk_tid;     // exported far entries: [block][64] far tid (-1 = none)
    const uint32_t *far_blk_dst;    //                       [block][64] place of the row's weight in far_w (0xFFFFFFFF = none)
    double *far_w;                  // weights of the rows with an exported entry, one per exported ENTRY, in transcript order
    const int32_t *wgt;             // per row slot (WEIGHTED)
    const double *rowval;           // per row slot (MODE_SCATTER)
    unsigned long long *stamps;     // diagnostic runs only (emsar_hip_debug_chunk_times), else null: per wave {start, end (100 MHz
                                    // clock), cycles spent inside slices, slices processed}, then one cycle count per slice;
                                    // read by nobody else
};
__device__ __forceinline__ unsigned long long stamp_now() {
    unsigned long long t;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    return t;
}
__device__ __forceinline__ unsigned long long stamp_real() {
    unsigned long long t;
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    return t;
}

// the first loads of a slice: its far tids (they come back first) and its first 8 forward columns
template <int MODE>
__device__ __forceinline__ void slice_head(const TiledArgs &P, const emsar::SliceDesc &D, unsigned lane, int4 (&A)[8], int &ft0, int &ft1) {
    if (MODE == MODE_SCATTER) return;
    const int k = __builtin_amdgcn_readfirstlane((int)D.k), nf = __builtin_amdgcn_readfirstlane((int)D.nf);
    const unsigned far_blk = __builtin_amdgcn_readfirstlane(D.far_blk);
    const int32_t *fb = P.far_blk_tid + (size_t)far_blk * 64;            // wave-uniform base, the lane adds 4 bytes of its own
    if (nf > 0) ft0 = __builtin_nontemporal_load(fb + lane);
    if (nf > 1) ft1 = __builtin_nontemporal_load(fb + 64 + lane);
    const int4 *e = reinterpret_cast<const int4 *>(P.fwd) + (size_t)__builtin_amdgcn_readfirstlane(D.fwd_kib) * 64;
    load8_clamped(A, e, lane, k < 8 ? k : 8);
}

template <bool WEIGHTED, int MODE>
__global__ __launch_bounds__(kTiledThreads, 4) void k_pass_tiled(const TiledArgs P, const double *__restrict__ theta, double *__restrict__ acc,
                                                              double *__restrict__ ll_out) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    double *th_w = lds;                     // [960]
    double *acc_w = lds + kTiledDictPad;    // [960]
    __shared__ double red[kTiledThreads / 64];
    __shared__ unsigned next_slice;
    const int lane = threadIdx.x & 63;
    const unsigned ulane = (unsigned)lane;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    double *w_s = lds + 2 * kTiledDictPad + wave * kTiledWr;     // this wave's row weights [768] + zero row
    const unsigned th_base = __builtin_amdgcn_readfirstlane(lds_byte_addr(th_w));
    const unsigned ws_base = __builtin_amdgcn_readfirstlane(lds_byte_addr(w_s));
    if (lane < 8) w_s[emsar::kTileSliceRows + lane] = 0.0;      // padding row of this wave's slices
    const emsar::ChunkDesc C = P.chunks[blockIdx.x];
    double ll = 0.0;
    const bool stamped = P.stamps != nullptr;                     // uniform; the timed runs never take these branches
    unsigned long long t_begin = 0, t_slices = 0, n_done = 0, t_ph[4] = {0, 0, 0, 0};   // phases: issue, E, w + next-slice issue, M
    if (stamped) t_begin = stamp_real();
    // (Tried and dropped: slices handed out by a counter in GLOBAL memory so that workgroups that finish early can help the
    // others -- all 1024 workgroups are resident from the first cycle to the last and the hardware issues the oldest waves
    // first, so the first workgroup of a CU finishes an equal share 20 % before the fourth.  A returning global atomic per slice
    // sits in the wave's in-order memory queue in front of every later load: 187 -> 292 us per pass.)
    for (unsigned g = C.group_begin; g < C.group_end; g++) {
        const emsar::GroupDesc G = P.groups[g];
        const int near_n = (int)G.near_n, nd = near_n + (int)G.far_n;
        // the wave's first slice (the first four slices are dealt, the rest come from the group's counter in LDS): its loads leave
        // before the dictionary is fetched, not after the barrier behind it
        unsigned s = G.slice_begin + (unsigned)wave;                       // wave-uniform
        emsar::SliceDesc D;
        int4 A[8];
        int ft0 = -1, ft1 = -1;
        if (s < G.slice_end) { D = P.slices[s]; slice_head<MODE>(P, D, ulane, A, ft0, ft1); }
        // ---- dictionary into LDS (slot nd is the zero slot) ----
        for (int d = threadIdx.x; d <= nd; d += kTiledThreads) {
            double v = 0.0;
            if (MODE != MODE_SCATTER && d < nd) {
                const int t = d < near_n ? G.lo + d : __builtin_nontemporal_load(&P.far_dict[G.far_off + (unsigned)(d - near_n)]);
                v = theta[t];
            }
            th_w[d] = v; acc_w[d] = 0.0;
        }
        if (threadIdx.x == 0) next_slice = G.slice_begin + emsar::kTileWaves;
        __syncthreads();
        // The wave's slices, software-pipelined: while slice s is in its M-step (LDS only), the forward columns and far tids of
        // the wave's NEXT slice are already on their way into the registers the E-step has just released -- a wave that asked for
        // its loads only when it needed them spent a third of its life waiting for HBM with the LDS pipe idle.
        while (s < G.slice_end) {
            const unsigned long long ts0 = stamped ? stamp_now() : 0ull;
#ifdef EMSAR_TILED_PRIO
            // the hardware issues by priority, then age: a wave's priority changes from slice to slice (a hash of the slice number),
            // so that no workgroup of a CU is served first for the whole kernel
            switch ((s * 2654435761u >> 13) & 3u) {
                case 0: __builtin_amdgcn_s_setprio(0); break;
                case 1: __builtin_amdgcn_s_setprio(1); break;
                case 2: __builtin_amdgcn_s_setprio(2); break;
                default: __builtin_amdgcn_s_setprio(3); break;
            }
#endif
            const int k = __builtin_amdgcn_readfirstlane((int)D.k), m = __builtin_amdgcn_readfirstlane((int)D.m);
            const int nf = __builtin_amdgcn_readfirstlane((int)D.nf);
            const unsigned coo_n = __builtin_amdgcn_readfirstlane((unsigned)D.coo_n), coo_base = __builtin_amdgcn_readfirstlane(D.coo_off);
            const unsigned far_blk = __builtin_amdgcn_readfirstlane(D.far_blk);
            const int4 *e = reinterpret_cast<const int4 *>(P.fwd) + (size_t)__builtin_amdgcn_readfirstlane(D.fwd_kib) * 64;   // wave-uniform
            const int4 *b = reinterpret_cast<const int4 *>(P.bwd) + (size_t)__builtin_amdgcn_readfirstlane(D.bwd_kib) * 64;
            // ---- the rest of this slice's loads: 8 backward segments, theta of the far entries, where their weights go ----
            int4 B[8];
            if (m > 0) load8_clamped(B, b, ulane, m < 8 ? m : 8);
            double fv0 = 0.0, fv1 = 0.0;                                   // theta of the exported far entries of fields 11 / 10
            if (MODE != MODE_SCATTER) {
                if (ft0 >= 0) fv0 = theta[ft0];
                if (ft1 >= 0) fv1 = theta[ft1];
            }
            // where the far rows' weights go (their entries' places in transcript order): requested now, in the registers the far
            // tids have just left, needed right after the E-step
            unsigned dst0 = 0xFFFFFFFFu, dst1 = 0xFFFFFFFFu;
            const uint32_t *fd = P.far_blk_dst + (size_t)far_blk * 64;     // wave-uniform base
            if (nf > 0) dst0 = __builtin_nontemporal_load(fd + ulane);
            if (nf > 1) dst1 = __builtin_nontemporal_load(fd + 64 + ulane);
            // the wave's next slice: its number from the group's counter and its descriptor (a scalar load, a trip to L2) are asked
            // for now and used after the E-step
            unsigned nx = 0;
            if (lane == 0) nx = __hip_atomic_fetch_add(&next_slice, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            const unsigned sn = (unsigned)__builtin_amdgcn_readfirstlane((int)nx);
            emsar::SliceDesc Dn = D;
            if (sn < G.slice_end) Dn = P.slices[sn];
            const unsigned long long tp1 = stamped ? stamp_now() : 0ull;
            // ---- E: row sums of the slice's 768 rows ----
            // row i of lane l is slot 64*i + l of the slice: the lanes of one gather hold consecutive sorted rows
            const size_t slot_base = (size_t)s * emsar::kTileSliceRows;    // wave-uniform; row i of this lane is slot slot_base + 64 i + lane
            double w[kRPL];
            if (MODE == MODE_SCATTER) {
                const double *rv = P.rowval + slot_base;
#pragma unroll
                for (int i = 0; i < kRPL; i++) w[i] = rv[64u * (unsigned)i + ulane];
            } else {
                double S[kRPL];
#pragma unroll
                for (int i = 0; i < kRPL; i++) S[i] = 0.0;
                for (int j0 = 0; j0 < k; j0 += 8) {
                    const int n0 = k - j0 < 8 ? k - j0 : 8;
                    if (j0) load8_clamped(A, e + (size_t)j0 * 64, ulane, n0);
                    fwd_sum_regs<EMSAR_E_BATCH>(A, n0, th_base, S);
                }
                S[kRPL - 1] += fv0; S[kRPL - 2] += fv1;
                double r[kRPL];
#pragma unroll
                for (int i = 0; i < kRPL; i++) r[i] = 1.0;
                if (WEIGHTED) {
                    const int32_t *wg = P.wgt + slot_base;
#pragma unroll
                    for (int i = 0; i < kRPL; i++) r[i] = (double)__builtin_nontemporal_load(wg + (64u * (unsigned)i + ulane));
                }
#pragma unroll
                for (int i = 0; i < kRPL; i++) {
                    bool live = (S[i] > 0.0) && (r[i] > 0.0);
                    w[i] = live ? r[i] * fast_rcp(S[i]) : 0.0;      // v_rcp_f64 + two Newton steps: 5 instructions instead of the IEEE division's dozen
                    if (MODE == MODE_EM_LL && WEIGHTED && live) ll += r[i] * log(S[i]);
                }
                if (MODE == MODE_EM_LL && !WEIGHTED) ll += sum_log_rows(S);
            }
            const unsigned long long tp2 = stamped ? stamp_now() : 0ull;
#pragma unroll
            for (int i = 0; i < kRPL; i++) w_s[64 * i + lane] = w[i];
            // The far fields' weights go straight to their entries' places in transcript order (the update kernels then add up
            // contiguous runs).  The place of these scattered stores in the program matters: a wave's loads and stores complete in
            // issue order (one vmcnt counter), so they stand in front of every load issued after them.  Here, the next loads
            // waited for are the M-step's reloads of backward segments, eight segments of LDS work away.  At the END of the slice
            // they stood in front of the next slice's first wait (4.8 k cycles per slice); coalesced stores to a [block][64] array
            // that the update kernels gather from cost nothing here but 35 us per pass there.
            if (dst0 != 0xFFFFFFFFu) P.far_w[dst0] = w[kRPL - 1];
            if (dst1 != 0xFFFFFFFFu) P.far_w[dst1] = w[kRPL - 2];
            // the M-step below reads rows written by OTHER lanes of this same wave: DS operations of one wave execute in
            // order, so only the compiler has to be kept from moving the reads up
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            // ---- the wave's next slice: its number from the group's counter, its first loads into the forward registers (the row weights are
            // in LDS by now: the registers of both are free) ----
            int ftn0 = -1, ftn1 = -1;
            if (sn < G.slice_end) slice_head<MODE>(P, Dn, ulane, A, ftn0, ftn1);
            const unsigned long long tp3 = stamped ? stamp_now() : 0ull;
            // ---- M: column sums over the same 768 rows, through the slice's transposed index ----
            unsigned cur = 0xFFFFFFFFu;
            double part = 0.0;
            for (int j0 = 0; j0 < m; j0 += 8) {
                const int n0 = m - j0 < 8 ? m - j0 : 8;
                if (j0) load8_clamped(B, b + (size_t)j0 * 64, ulane, n0);
                bwd_sum_regs<EMSAR_M_BATCH>(B, n0, ws_base, acc_w, cur, part);
            }
            if (part != 0.0) lds_add_f64(reinterpret_cast<double *>(reinterpret_cast<char *>(acc_w) + cur), part);
            const uint32_t *cb = P.coo + coo_base;                         // wave-uniform base
            for (unsigned q = ulane; q < coo_n; q += 64) {
                const unsigned p = __builtin_nontemporal_load(cb + q);
                const double v = lds_at(w_s, (p & 0xFFFFu) << 3);
                if (v != 0.0) lds_add_f64(reinterpret_cast<double *>(reinterpret_cast<char *>(acc_w) + ((p >> 16) << 3)), v);
            }
            // the next slice's E-step overwrites w_s: again only the compiler needs telling
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            if (stamped) {
                const unsigned long long te = stamp_now();
                const unsigned long long dt = te - ts0;
                t_slices += dt; n_done++;
                t_ph[0] += tp1 - ts0; t_ph[1] += tp2 - tp1; t_ph[2] += tp3 - tp2; t_ph[3] += te - tp3;
                if (lane == 0) P.stamps[(size_t)gridDim.x * emsar::kTileWaves * 8 + s] = dt | ((unsigned long long)blockIdx.x << 40);
            }
            s = sn; D = Dn; ft0 = ftn0; ft1 = ftn1;
        }
        __syncthreads();
        // ---- flush the dictionary ----
        for (int d = threadIdx.x; d < nd; d += kTiledThreads) {
            const double v = acc_w[d];
            if (v != 0.0) {
                const int t = d < near_n ? G.lo + d : __builtin_nontemporal_load(&P.far_dict[G.far_off + (unsigned)(d - near_n)]);
                atomic_add_f64(&acc[t], v);
            }
        }
        // a thread rewrites only the dictionary slots it has just flushed (same d -> thread mapping), next_slice is rewritten
        // by thread 0 and read by nobody before the next barrier: no barrier needed here
    }
    if (stamped && lane == 0) {
        unsigned long long *o = P.stamps + ((size_t)blockIdx.x * emsar::kTileWaves + wave) * 8;
        o[0] = t_begin; o[1] = stamp_real(); o[2] = t_slices; o[3] = n_done;
        o[4] = t_ph[0]; o[5] = t_ph[1]; o[6] = t_ph[2]; o[7] = t_ph[3];
    }
    if (MODE == MODE_EM_LL) {
        double t = block_sum<kTiledThreads>(ll, red);
        if (threadIdx.x == 0 && t != 0.0) atomic_add_f64(ll_out, t);
    }
}

// ------------------------------------------------------------------------------------------------
// k_pass_pairs: the rows of two transcripts far from each other (layout_tiled.hpp) -- two theta gathers, one reciprocal and two
// stores per row, w = R / (theta_a + theta_b) to the places of the row's two entries in far_w.  Runs on a side stream next to
// k_pass_tiled (a chain of dependent trips to memory per thread: inside the pass kernel it held a workgroup's slot for 5 us).
// ------------------------------------------------------------------------------------------------
template <bool WEIGHTED, int MODE>
__global__ __launch_bounds__(256) void k_pass_pairs(int64_t n_pairs, const int32_t *__restrict__ pair_tid, const uint32_t *__restrict__ pair_dst,
                                                    const int32_t *__restrict__ pair_wgt, const double *__restrict__ pair_val,
                                                    const double *__restrict__ theta, double *__restrict__ far_w, double *__restrict__ ll_out) {
    __shared__ double red[4];
    double ll = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n_pairs; i += (int64_t)gridDim.x * 256) {
        double w;
        if (MODE == MODE_SCATTER) w = pair_val[i];
        else {
            const double S = theta[pair_tid[2 * i]] + theta[pair_tid[2 * i + 1]];
            const double r = WEIGHTED ? (double)pair_wgt[i] : 1.0;
            const bool live = (S > 0.0) && (r > 0.0);
            w = live ? r * fast_rcp(S) : 0.0;
            if (MODE == MODE_EM_LL && live) ll += r * log(S);
        }
        far_w[pair_dst[2 * i]] = w;
        far_w[pair_dst[2 * i + 1]] = w;
    }
    if (MODE == MODE_EM_LL) {
        double t = block_sum<256>(ll, red);
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

}  // namespace
