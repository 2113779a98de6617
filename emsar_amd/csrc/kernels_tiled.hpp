// kernels_tiled.hpp -- k_pass_tiled / k_pass_tiled_unit / k_pass_tiled_multi: the EM pass over the TILED layout (layout_tiled.hpp), the default
#pragma once
// included by emsar_hip.hip only (one translation unit: the kernels live in its anonymous namespace)

namespace {

// ------------------------------------------------------------------------------------------------
// k_pass_tiled: one EM pass over the TILED layout (layout_tiled.hpp).  One workgroup = 4 waves = the 4 slices of a tile.
//   phase 0  every global load the wave needs first is issued at once (the three theta of the thread's dictionary block, 8
//            forward columns, 8 backward segments); table: T[8b + m] = sum of the theta of block b's slots in subset m, the
//            per-entry accumulators W[8b + m] = 0; barrier
//   phase E  the wave owns one slice (768 rows; lane l holds rows 64*i + l, i < 12): S_r = sum T[entry]  (LDS reads only, 10-bit
//            entries = block and subset, padding is entry 0 = the empty subset: no branches), w_r = R_r / S_r -> the wave's own
//            6 KiB of LDS
//   phase M  the SAME wave walks the transposed index of its rows: a lane's segments (entry value + 11 row ids) are
//            consecutive in entry order; it gathers w_r from LDS into a register sum and adds it to W[entry] when the
//            entry changes.  No barrier between E and M.
//   phase F  barrier; thread b folds the 8 accumulators of block b into its 3 transcripts (transcript i collects the subsets
//            that hold it) and flushes them with one global FP64 atomic each
// k_pass_tiled_unit (the default above 2048 tiles): the same for a UNIT of up to two tiles that share one dictionary.
// HBM traffic: 10 bits per forward slot + 128 bits per 11 backward entries -- no row_ptr, no 32-bit tids.
// ------------------------------------------------------------------------------------------------
#ifndef EMSAR_UE_BATCH          // LDS gathers in flight per step of the E / M loops of the unit and multi kernels (tile_e_step,
#define EMSAR_UE_BATCH 6        // tile_m_step); 6 / 12 in any combination measured 0.1159 - 0.1165 ms: no difference
#endif
#ifndef EMSAR_UM_BATCH
#define EMSAR_UM_BATCH 6
#endif
constexpr int kTiledThreads = 256;                       // 4 wavefronts = 4 slices
constexpr int kRPL = emsar::kRowsPerLane;                 // 12 rows per lane = twelve 10-bit ids per int4
constexpr int kTiledWr = emsar::kTileSliceRows + 8;       // w_r of one slice (768) + the zero padding row
constexpr int kTiledDictPad = emsar::kDictEntries;        // 960 table entries: 120 blocks x 8 subset sums (entry 0 = the empty subset = 0)
constexpr int kBlk = emsar::kBlk;
constexpr int kTiledLdsDoubles = 2 * kTiledDictPad + emsar::kTileSlices * kTiledWr;   // 40,192 B: 4 workgroups per CU

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
__device__ __forceinline__ void load8_clamped(int4 (&q)[8], const int4 *e, int n) {
#pragma unroll
    for (int j = 0; j < 8; j++) {
        const v4i_t t = __builtin_nontemporal_load(reinterpret_cast<const v4i_t *>(e + (size_t)(j < n ? j : n - 1) * 64));
        q[j] = make_int4(t.x, t.y, t.z, t.w);
    }
}

// half of a batch: positions LO .. LO + 3 of the n (>= 1) columns / segments that start at e
template <int LO>
__device__ __forceinline__ void load4_clamped(int4 (&q)[8], const int4 *e, int n) {
#pragma unroll
    for (int j = LO; j < LO + 4; j++) {
        const v4i_t t = __builtin_nontemporal_load(reinterpret_cast<const v4i_t *>(e + (size_t)(j < n ? j : n - 1) * 64));
        q[j] = make_int4(t.x, t.y, t.z, t.w);
    }
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
template <int BATCH = 12, int LO = 0, int HI = 8>
__device__ __forceinline__ void fwd_sum_regs(const int4 (&q)[8], int n, unsigned th_base, double (&S)[kRPL]) {
#pragma unroll
    for (int j = LO; j < HI; j++) {
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

// a lane's finished column sum -> the tile's LDS accumulator.  Deterministic mode (fx != 0, wave-uniform): the sum times theta of
// the slot (the LDS dictionary sits kTiledDictPad doubles below the accumulators) is the mass of the column's rows, added as a
// fixed-point integer -- the four waves of the tile reach a slot in any order, the bits are the same.
__device__ __forceinline__ void tile_acc_add(double *acc_w, unsigned off, double part, double fx) {
    double *p = reinterpret_cast<double *>(reinterpret_cast<char *>(acc_w) + off);
    if (fx != 0.0) lds_add_i64(p, __double2ll_rn(part * *(p - kTiledDictPad) * fx));
    else lds_add_f64(p, part);
}
// Every vector-memory request of this wave has landed.  Placed where a wave-uniform branch may have skipped the code that consumed a load
// (a wave without dictionary work skips block_dict_store): hipcc's s_waitcnt pass then carries the load as PENDING into the tile loop and
// puts s_waitcnt vmcnt(0) in front of the first instruction that reuses its register -- in the middle of the E-step, where it waits for
// the index loads issued to run ahead.
__device__ __forceinline__ void vm_loads_landed() { __builtin_amdgcn_s_waitcnt(0x0F70); }      // vmcnt(0), expcnt and lgkmcnt untouched

// A tile descriptor as the kernels hold it: every field a register of its own, decoded from SIXTEEN DWORDS read at a wave-uniform address
// (scalar loads).  Reading emsar::Tile's 8- and 16-bit fields directly makes hipcc fetch each of them with a VECTOR load (global_load_ushort /
// _ubyte + s_waitcnt vmcnt(0) + v_readfirstlane: there is no scalar sub-dword load) -- a memory round trip in the middle of the tile loop that
// also waits for every index load in flight.
struct DTile {
    uint64_t fwd_off, bwd_off;
    uint32_t row_base, far_off;
    int32_t lo;
    uint32_t near_n, far_n, n_slices, follows, wave_of;
    uint32_t k[4], m[4];
};
struct TileWords { uint32_t w[16]; };
static_assert(sizeof(TileWords) == sizeof(Tile), "tile_load reads a Tile as 16 dwords");
static_assert(offsetof(Tile, row_base) == 16 && offsetof(Tile, lo) == 28 && offsetof(Tile, near_n) == 32 && offsetof(Tile, n_slices) == 36 && offsetof(Tile, follows) == 38 &&
              offsetof(Tile, wave_of) == 39 && offsetof(Tile, k) == 40 && offsetof(Tile, m) == 48 && offsetof(Tile, coo_n) == 56, "tile_load's field positions");
__device__ __forceinline__ DTile tile_load(const Tile *p /* wave-uniform */) {
    const TileWords r = *reinterpret_cast<const TileWords *>(p);
    DTile T;
    T.fwd_off = (uint64_t)r.w[0] | (uint64_t)r.w[1] << 32;
    T.bwd_off = (uint64_t)r.w[2] | (uint64_t)r.w[3] << 32;
    T.row_base = r.w[4]; T.far_off = r.w[5]; T.lo = (int32_t)r.w[7];
    T.near_n = r.w[8] & 0xFFFFu; T.far_n = r.w[8] >> 16;
    T.n_slices = r.w[9] & 0xFFFFu; T.follows = (r.w[9] >> 16) & 0xFFu; T.wave_of = r.w[9] >> 24;
#pragma unroll
    for (int s = 0; s < 4; s++) {
        T.k[s] = (r.w[10 + s / 2] >> (16 * (s & 1))) & 0xFFFFu;
        T.m[s] = (r.w[12 + s / 2] >> (16 * (s & 1))) & 0xFFFFu;
    }
    return T;
}

// ---- the dictionary of a tile (layout_tiled.hpp): near blocks of three transcripts with eight subset sums each, then one entry
// per far transcript ----
// Thread b < nb (the tile's near blocks, <= 120) fetches the three theta of block b and writes the eight subset sums T[8b + m];
// the other threads fetch three far transcripts each (far entry i lives at T[8 nb + i] = its theta).  Everybody clears the
// accumulators of what it wrote.  At the end of the tile the near part is folded BY SLOT: thread s (and s + 256) adds up the
// subsets of its block that hold slot s and sends the sum to its transcript -- consecutive lanes add to consecutive transcripts,
// 512 contiguous bytes per wave instruction (device-scope float atomics run at full rate only on contiguous addresses); a far
// entry is flushed by the thread that fetched it.
struct BlockDict { double th[kBlk]; int tid[kBlk]; int stid[2]; };     // stid: the transcripts of near SLOTS threadIdx.x and threadIdx.x + 256
constexpr int kSlotsPerThread = (emsar::kTileDict + kTiledThreads - 1) / kTiledThreads;
static_assert(kSlotsPerThread <= 2, "BlockDict::stid");
static_assert(emsar::kFarMax <= kBlk * (kTiledThreads - emsar::kDictBlocks), "kBlk far entries per thread that owns no near block");
// the far list of a unit (n transcripts at far), three per thread that owns no near block
__device__ __forceinline__ void block_dict_far_issue(const int32_t *far, int n, BlockDict &D) {
#pragma unroll
    for (int i = 0; i < kBlk; i++) {
        D.tid[i] = -1;
        const int f = ((int)threadIdx.x - emsar::kDictBlocks) + i * (kTiledThreads - emsar::kDictBlocks);
        if ((int)threadIdx.x >= emsar::kDictBlocks && f < n) D.tid[i] = __builtin_nontemporal_load(&far[f]);
    }
}
template <int MODE, bool FAR_ISSUED = false>
__device__ __forceinline__ void block_dict_issue(const DTile &T, int nd, const int32_t *far_tid, const double *theta, BlockDict &D) {
    const int near_n = (int)T.near_n, far_n = nd - near_n, nb = (near_n + kBlk - 1) / kBlk;
    const bool near_thread = (int)threadIdx.x < emsar::kDictBlocks;
#pragma unroll
    for (int i = 0; i < kBlk; i++) {
        D.th[i] = 0.0;
        if (!FAR_ISSUED || near_thread) D.tid[i] = -1;
        if (near_thread) {
            const int d = (int)threadIdx.x * kBlk + i;
            if ((int)threadIdx.x < nb && d < near_n) D.tid[i] = T.lo + d;
        } else if (!FAR_ISSUED) {
            const int f = ((int)threadIdx.x - emsar::kDictBlocks) + i * (kTiledThreads - emsar::kDictBlocks);      // far entry: strided, coalesced
            if (f < far_n) D.tid[i] = __builtin_nontemporal_load(&far_tid[T.far_off + f]);
        }
        if (MODE != MODE_SCATTER && D.tid[i] >= 0) D.th[i] = theta[D.tid[i]];
    }
#pragma unroll
    for (int j = 0; j < 2; j++) {
        const int sl = (int)threadIdx.x + j * kTiledThreads;
        D.stid[j] = sl < near_n ? T.lo + sl : -1;
    }
}
__device__ __forceinline__ void block_dict_store(const DTile &T, const BlockDict &D, double *th_w, double *acc_w) {
    constexpr int NE = emsar::kBlkEntries;
    const int nb = ((int)T.near_n + kBlk - 1) / kBlk;
    if ((int)threadIdx.x < emsar::kDictBlocks) {
        if ((int)threadIdx.x < nb) {
            double *t = th_w + threadIdx.x * NE, *a = acc_w + threadIdx.x * NE;
            double v[NE];
            v[0] = 0.0;
#pragma unroll
            for (int m = 1; m < NE; m++) {             // subset sum of m = that of m without its highest slot + theta of that slot
                const int hi = 31 - __builtin_clz((unsigned)m);
                v[m] = v[m & ~(1 << hi)] + D.th[hi];
            }
            const int sw = EMSAR_SWZ ? (int)(threadIdx.x & (NE - 1)) : 0;     // bank swizzle of the block (layout_tiled.hpp: entry_code)
#pragma unroll
            for (int m = 0; m < NE; m++) { t[m ^ sw] = v[m]; a[m] = 0.0; }
        }
    } else {
#pragma unroll
        for (int i = 0; i < kBlk; i++) {
            if (D.tid[i] < 0) continue;
            const int e = NE * nb + ((int)threadIdx.x - emsar::kDictBlocks) + i * (kTiledThreads - emsar::kDictBlocks);
            th_w[e] = D.th[i]; acc_w[e] = 0.0;
        }
    }
}
__device__ __forceinline__ void block_dict_flush(const DTile &T, const BlockDict &D, const double *th_w, const double *acc_w, double *acc, double fx) {
    constexpr int NE = emsar::kBlkEntries;
#pragma unroll
    for (int j = 0; j < 2; j++) {
        const int tid = D.stid[j];
        if (tid < 0) continue;
        const int sl = (int)threadIdx.x + j * kTiledThreads, b = sl / kBlk, i = sl - b * kBlk;
        const double *a = acc_w + b * NE;
        const int sw = EMSAR_SWZ ? (b & (NE - 1)) : 0;
        if (fx != 0.0) {
            // deterministic mode: W[e] holds, as an integer, the MASS of entry e (sum of w_r T[e] fx over its rows: tile_acc_add);
            // the column sum of the entry is that over T[e], the slot's transcript gets its theta times the sum over the subsets that hold it
            const double *t = th_w + b * NE;
            double si = 0.0;
#pragma unroll
            for (int m = 1; m < NE; m++) {
                const double tm = t[m ^ sw];
                if ((m >> i & 1) && tm > 0.0) si += (double)__double_as_longlong(a[m ^ sw]) / tm;
            }
            const long long iv = __double2ll_rn(t[(1 << i) ^ sw] * si);
            if (iv != 0) atomic_add_i64(&acc[tid], iv);
        } else {
            double si = 0.0;
#pragma unroll
            for (int m = 1; m < NE; m++) if (m >> i & 1) si += a[m ^ sw];
            if (si != 0.0) atomic_add_f64(&acc[tid], si);
        }
    }
#ifndef EMSAR_EXP_NO_FAR_FLUSH
    if ((int)threadIdx.x >= emsar::kDictBlocks) {      // the far entries this thread fetched: the accumulator IS the transcript's sum (its mass, in deterministic mode)
        const int nb = ((int)T.near_n + kBlk - 1) / kBlk;
#pragma unroll
        for (int i = 0; i < kBlk; i++) {
            if (D.tid[i] < 0) continue;
            const int f = ((int)threadIdx.x - emsar::kDictBlocks) + i * (kTiledThreads - emsar::kDictBlocks), e = NE * nb + f;
            const double v = acc_w[e];
            if (fx != 0.0) { const long long iv = __double_as_longlong(v); if (iv != 0) atomic_add_i64(&acc[D.tid[i]], iv); }
            else if (v != 0.0) atomic_add_f64(&acc[D.tid[i]], v);
        }
    }
#endif
}

// M-step of up to 8 backward segments of one lane ({column, 11 row ids} each).  A lane's segments are consecutive
// in column order: the running sum stays in a register and goes to the LDS accumulator when the column changes.
template <int BATCH = 12, int LO = 0, int HI = 8>
__device__ __forceinline__ void bwd_sum_regs(const int4 (&q)[8], int n, unsigned ws_base, double *acc_w, unsigned &cur, double &part, double fx) {
#pragma unroll
    for (int j = LO; j < HI; j++) {
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
                if (part != 0.0) tile_acc_add(acc_w, cur, part, fx);
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

// chip-wide 100 MHz clock and the place a wave runs at (timeline of the stamped instance)
__device__ __forceinline__ unsigned long long stamp_real() {
    unsigned long long t;
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    return t;
}
__device__ __forceinline__ unsigned stamp_place() {        // XCC_ID << 16 | HW_ID[15:0] (wave, simd, pipe, cu, sh, se)
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)\n\ts_getreg_b32 %1, hwreg(HW_REG_XCC_ID)" : "=s"(hw), "=s"(xcc));
    return (xcc & 0xFu) << 16 | (hw & 0xFFFFu);
}

// sum_i log S_i over a lane's kRPL unweighted rows (S_i <= 0: row outside F or padding, no term) with two logs instead of
// twelve: log of the product of the twelve row sums of a lane (of six, of one, as the range allows).  A product that leaves [1e-280, 1e280] (components decayed towards the
// boundary) falls back to the per-row logs.  The f64 log is ~40 VALU instructions: 19 % of a likelihood pass on config 3.
// (the fallback is a rolled loop over LDS: fifteen inlined copies of the f64 log made the likelihood variant of the unit kernel
// 65 KB of code, more than the instruction cache two CUs share; an out-of-line function cost scratch for its call frame)
// w_s: the wave's own row-weight region of LDS, free at this point of the E-step (written right afterwards): scratch of the fallback
// the likelihood terms a lane collects in the unit and multi kernels: sum of r log S (weighted rows), or the product of the row sums
// as mantissa and exponent (unweighted rows; tile_e_step)
struct LlAcc { double v = 0.0, p = 1.0; int e = 0; };
__device__ __forceinline__ double ll_value(const LlAcc &a) { return a.v + (log(a.p) + (double)a.e * 0.6931471805599453094); }
template <int N>
__device__ __forceinline__ double sum_log_rows(const double (&S)[N], double *w_s, int lane) {
    static_assert(N == 12, "rows per lane");
    double h[2];
#pragma unroll
    for (int g = 0; g < 2; g++) {
        double p = 1.0;
#pragma unroll
        for (int i = 6 * g; i < 6 * g + 6; i++) p *= S[i] > 0.0 ? S[i] : 1.0;
        h[g] = p;
    }
    // one log for all twelve rows when the product stays inside the range (the usual case: row sums are inferred read
    // densities, 1e-3 .. 1e5); else one per row
    if (h[0] > 1e-140 && h[0] < 1e140 && h[1] > 1e-140 && h[1] < 1e140) return log(h[0] * h[1]);
#pragma unroll
    for (int i = 0; i < N; i++) w_s[64 * i + lane] = S[i];
    double ll = 0.0;
#pragma unroll 1
    for (int i = 0; i < N; i++) { const double v = w_s[64 * i + lane]; if (v > 0.0) ll += log(v); }     // ONE copy of the log in the code
    return ll;
}

template <bool WEIGHTED, int MODE, bool STAMP = false>
__global__ __launch_bounds__(kTiledThreads, 4) void k_pass_tiled(const Tile *__restrict__ tiles, const uint32_t *__restrict__ fwd,
                                                              const uint32_t *__restrict__ bwd,
                                                              const int32_t *__restrict__ far_tid,
                                                              const int32_t *__restrict__ wgt,    // per row slot
                                                              const double *__restrict__ rowval,  // per row slot (MODE_SCATTER)
                                                              const double *__restrict__ theta, double *__restrict__ acc,
                                                              double *__restrict__ ll_out, Fx fx, unsigned long long *stamps = nullptr) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    double *th_w = lds;                     // [960]
    double *acc_w = lds + kTiledDictPad;    // [960]
    __shared__ double red[kTiledThreads / 64];
    unsigned long long ts[6];
    if (STAMP) ts[0] = stamp_now();

    const DTile T = tile_load(tiles + blockIdx.x);
    const int nd = (int)T.near_n + (int)T.far_n;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const bool has_slice = wave < (int)T.n_slices;
    double *w_s = lds + 2 * kTiledDictPad + wave * kTiledWr;     // this wave's row weights [768] + zero row
    const unsigned th_base = __builtin_amdgcn_readfirstlane(lds_byte_addr(th_w));
    const unsigned ws_base = __builtin_amdgcn_readfirstlane(lds_byte_addr(w_s));

    // ---- issue every global load this wave needs first, in the order of use: dictionary values, 8 forward columns,
    //      8 backward segments.  One HBM round trip per tile; the rest of the pass touches LDS only.
    BlockDict D;
    block_dict_far_issue(far_tid + T.far_off, (int)T.far_n, D);       // the far list first, all of it: one round trip, then every theta in one more
    int4 A[8], B[8];
    const int4 *e = reinterpret_cast<const int4 *>(fwd), *b = reinterpret_cast<const int4 *>(bwd);       // (a wave without a slice: one line, eight times)
    int k = 1, m = 1;
    if (has_slice) {
        unsigned foff = 0, boff = 0;                 // KiB units (256 dwords) from the tile's bases
#pragma unroll
        for (int s = 0; s < emsar::kTileSlices; s++) {
            if (s < wave) { foff += T.k[s]; boff += T.m[s]; }
            if (s == wave) { k = T.k[s]; m = T.m[s]; }
        }
        // everything above is wave-uniform; say so, or the loops below are compiled as divergent code
        k = __builtin_amdgcn_readfirstlane(k); m = __builtin_amdgcn_readfirstlane(m);
        foff = __builtin_amdgcn_readfirstlane(foff); boff = __builtin_amdgcn_readfirstlane(boff);
        e = reinterpret_cast<const int4 *>(fwd + T.fwd_off / 4 + (size_t)foff * 256) + lane;
        b = reinterpret_cast<const int4 *>(bwd + T.bwd_off / 4 + (size_t)boff * 256) + lane;
    }
    // the index requests are unconditional and stand BEFORE the dictionary's theta, whose addresses wait for the far list: that wait is then
    // vmcnt(16) on every path (hipcc merges paths with different numbers of requests into vmcnt(0))
    if (MODE != MODE_SCATTER) load8_clamped(A, e, k < 8 ? k : 8);
    load8_clamped(B, b, m < 8 ? m : 8);
    block_dict_issue<MODE, true>(T, nd, nullptr, theta, D);
    if (!has_slice) { k = 0; m = 0; }
    // ---- phase 0: the table of subset sums into LDS, the accumulators cleared ----
    block_dict_store(T, D, th_w, acc_w);
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
                for (int i = 0; i < kRPL; i++) r[i] = (double)__builtin_nontemporal_load(&wgt[slot0 + 64 * i]);
            }
#pragma unroll
            for (int i = 0; i < kRPL; i++) {
                bool live = (S[i] > 0.0) && (r[i] > 0.0);
                w[i] = live ? r[i] / S[i] : 0.0;
                if (MODE == MODE_EM_LL && WEIGHTED && live) ll += r[i] * log(S[i]);
            }
            if (MODE == MODE_EM_LL && !WEIGHTED) ll += sum_log_rows(S, w_s, lane);
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
            bwd_sum_regs(B, n0, ws_base, acc_w, cur, part, fx.mass);
        }
        if (part != 0.0) tile_acc_add(acc_w, cur, part, fx.mass);
    } else if (STAMP) ts[3] = stamp_now();
    if (STAMP) ts[4] = stamp_now();
    __syncthreads();
    if (STAMP) ts[5] = stamp_now();
    // ---- F: the accumulators of every block folded into its transcripts and flushed ----
    block_dict_flush(T, D, th_w, acc_w, acc, fx.mass);
    if (STAMP && lane == 0) {   // [tile][wave][5 phases]: issue+dictionary, barrier, E, M, barrier
        for (int i = 0; i < 5; i++) stamps[((size_t)blockIdx.x * (kTiledThreads / 64) + wave) * 8 + i] = ts[i + 1] - ts[i];
    }
    if (MODE == MODE_EM_LL) {
        double t = block_sum<kTiledThreads>(ll, red);
        if (threadIdx.x == 0 && t != 0.0) ll_add(ll_out, t, fx.ll);
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
    int k, m, nd, slice;
    bool has_slice;
};
__device__ __forceinline__ TileWave tile_wave(const DTile &T, int wave, int lane, const uint32_t *fwd, const uint32_t *bwd) {
    TileWave W;
    W.nd = (int)T.near_n + (int)T.far_n;
    W.has_slice = wave >= 0 && wave < (int)T.n_slices;          // (wave = the slice index; -1: this wave has none in the tile)
    W.slice = wave;
    W.e = nullptr; W.b = nullptr; W.k = 0; W.m = 0;
    if (W.has_slice) {
        unsigned foff = 0, boff = 0;
        int k = 0, m = 0;
#pragma unroll
        for (int s = 0; s < emsar::kTileSlices; s++) {
            if (s < wave) { foff += T.k[s]; boff += T.m[s]; }
            if (s == wave) { k = T.k[s]; m = T.m[s]; }
        }
        W.k = __builtin_amdgcn_readfirstlane(k); W.m = __builtin_amdgcn_readfirstlane(m);
        foff = __builtin_amdgcn_readfirstlane(foff); boff = __builtin_amdgcn_readfirstlane(boff);
        W.e = reinterpret_cast<const int4 *>(fwd + T.fwd_off / 4 + (size_t)foff * 256) + lane;
        W.b = reinterpret_cast<const int4 *>(bwd + T.bwd_off / 4 + (size_t)boff * 256) + lane;
    }
    return W;
}
// the slice of tile T that wave `wave` of a unit's workgroup takes, -1 if none (Tile::wave_of)
__device__ __forceinline__ int unit_slice(const DTile &T, int wave) {
    int slice = -1;
#pragma unroll
    for (int s = 0; s < emsar::kTileSlices; s++)
        if (s < (int)T.n_slices && (((unsigned)T.wave_of >> (2 * s)) & 3u) == (unsigned)wave) slice = s;
    return slice;
}
template <bool WEIGHTED, int MODE>
__device__ __forceinline__ void tile_e_step(const TileWave &W, int4 (&A)[8], size_t slot0, const int32_t *wgt, const double *th_w, double *w_s,
                                            int lane, LlAcc &ll) {
    const unsigned th_base = __builtin_amdgcn_readfirstlane(lds_byte_addr(th_w));
    double S[kRPL];
    int r[kRPL];                 // row weights (read counts) stay integers until they are used: 12 registers, not 24
#pragma unroll
    for (int i = 0; i < kRPL; i++) { S[i] = 0.0; r[i] = 1; }
    for (int j0 = 0; j0 < W.k; j0 += 8) {
        const int n0 = W.k - j0 < 8 ? W.k - j0 : 8;
        if (j0) load8_clamped(A, W.e + (size_t)j0 * 64, n0);
        fwd_sum_regs<EMSAR_UE_BATCH>(A, n0, th_base, S);
    }
    if (WEIGHTED && MODE == MODE_EM_LL) {
        // Weighted likelihood: sum r log S needs a log per row.  Twelve inlined f64 logs next to the weights and both index batches
        // spilt registers (and made the unit kernel's likelihood variant slower than the one-tile kernel); here the row sums go through
        // the wave's own w_s region (free until the weights are written) and ONE rolled loop holds the only copy of the log.
#pragma unroll
        for (int i = 0; i < kRPL; i++) w_s[64 * i + lane] = S[i];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        double acc = 0.0;
#pragma unroll 1
        for (int i = 0; i < kRPL; i++) {
            const double s = w_s[64 * i + lane];
            const int ri = __builtin_nontemporal_load(&wgt[slot0 + 64 * i]);
            const bool live = s > 0.0 && ri > 0;
            const double rd = (double)ri;
            w_s[64 * i + lane] = live ? rd / s : 0.0;
            if (live) acc += rd * log(s);
        }
        ll.v += acc;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        return;
    }
    if (WEIGHTED) {
#pragma unroll
        for (int i = 0; i < kRPL; i++) r[i] = __builtin_nontemporal_load(&wgt[slot0 + 64 * i]);
    }
    // Unweighted likelihood: sum log S = log(product of the mantissas) + ln 2 * (sum of the exponents).  Four VALU instructions per row
    // and NO log here -- one per lane at the end of the kernel (ll_value): the f64 log (about 40 instructions, a dozen temporaries)
    // inside the E-step, with the weights and both index batches in registers, spilt 5-13 registers in the likelihood variants;
    // no product can leave the double range either, whatever theta (the old two-product form needed a per-row fallback).
    if (MODE == MODE_EM_LL && !WEIGHTED) {
#pragma unroll
        for (int i = 0; i < kRPL; i++) {
            const bool pos = S[i] > 0.0;
            ll.p *= pos ? __builtin_amdgcn_frexp_mant(S[i]) : 1.0;         // [0.5, 1): twelve of them stay above 2^-12
            ll.e += pos ? __builtin_amdgcn_frexp_exp(S[i]) : 0;
        }
        ll.e += __builtin_amdgcn_frexp_exp(ll.p);
        ll.p = __builtin_amdgcn_frexp_mant(ll.p);
    }
#pragma unroll
    for (int i = 0; i < kRPL; i++) {
        const bool live = (S[i] > 0.0) && (r[i] > 0);
        const double ri = WEIGHTED ? (double)r[i] : 1.0;
        w_s[64 * i + lane] = live ? ri / S[i] : 0.0;
        if (MODE == MODE_EM_LL && WEIGHTED && live) ll.v += ri * log(S[i]);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
__device__ __forceinline__ void tile_m_step(const TileWave &W, int4 (&B)[8], const double *w_s, double *acc_w, int lane, double fx) {
    const unsigned ws_base = __builtin_amdgcn_readfirstlane(lds_byte_addr(w_s));
    unsigned cur = 0xFFFFFFFFu;
    double part = 0.0;
    // (the second batch of segments -- half of config 3's slices have one -- is requested in halves while the first is worked on: tile_e_step)
    bwd_sum_regs<EMSAR_UM_BATCH, 0, 4>(B, W.m, ws_base, acc_w, cur, part, fx);
    if (W.m > 8) {
        const int n1 = W.m - 8 < 8 ? W.m - 8 : 8;
        const int4 *b1 = W.b + (size_t)8 * 64;
        __builtin_amdgcn_sched_barrier(0);          // (the refill must not move up across the sums that still read these registers: it would need new ones)
        load4_clamped<0>(B, b1, n1);
        __builtin_amdgcn_sched_barrier(0);
        bwd_sum_regs<EMSAR_UM_BATCH, 4, 8>(B, 8, ws_base, acc_w, cur, part, fx);
        __builtin_amdgcn_sched_barrier(0);
        load4_clamped<4>(B, b1, n1);
        __builtin_amdgcn_sched_barrier(0);
        bwd_sum_regs<EMSAR_UM_BATCH, 0, 4>(B, n1, ws_base, acc_w, cur, part, fx);
        bwd_sum_regs<EMSAR_UM_BATCH, 4, 8>(B, n1, ws_base, acc_w, cur, part, fx);
        for (int j0 = 16; j0 < W.m; j0 += 8) {
            const int n0 = W.m - j0 < 8 ? W.m - j0 : 8;
            load8_clamped(B, W.b + (size_t)j0 * 64, n0);
            bwd_sum_regs<EMSAR_UM_BATCH>(B, n0, ws_base, acc_w, cur, part, fx);
        }
    } else bwd_sum_regs<EMSAR_UM_BATCH, 4, 8>(B, W.m, ws_base, acc_w, cur, part, fx);
    if (part != 0.0) tile_acc_add(acc_w, cur, part, fx);
    (void)lane;
}
struct TileEnv {            // per-launch constants of the multi-tile kernel
    const Tile *tiles; int n_tiles, stride;
    const uint32_t *fwd, *bwd; const int32_t *far_tid, *wgt; const double *theta; double *acc;
    double *th_w, *acc_w, *w_s; int lane, wave; double fx;
};
// stage I of N: tile `it` is in the registers (A, B in flight or landed, dictionary values in thv); while it is being
// worked on, tile it + stride is requested into the registers as they fall free.  Straight-line code, no loop: hipcc
// keeps loop-carried register arrays of this size in scratch.
template <bool WEIGHTED, int MODE, int I, int N>
__device__ __forceinline__ void tiled_stage(const TileEnv &V, int it, const DTile &T, const TileWave &W, int4 (&A)[8], int4 (&B)[8],
                                            BlockDict &D, LlAcc &ll) {
    block_dict_store(T, D, V.th_w, V.acc_w);
    const BlockDict Dcur = D;                     // tids of THIS tile's block, for its flush; D is refilled for the next tile below
    const int in = it + V.stride;
    const bool has_next = (I + 1 < N) && in < V.n_tiles;
    const DTile Tn = tile_load(V.tiles + (has_next ? in : it));
    __syncthreads();
    if (W.has_slice)
        tile_e_step<WEIGHTED, MODE>(W, A, (size_t)T.row_base + (size_t)V.wave * emsar::kTileSliceRows + V.lane, V.wgt, V.th_w, V.w_s, V.lane, ll);
    const TileWave Wn = tile_wave(Tn, V.wave, V.lane, V.fwd, V.bwd);
    if (has_next) {
        if (Wn.has_slice) load8_clamped(A, Wn.e, Wn.k < 8 ? Wn.k : 8);
        block_dict_far_issue(V.far_tid + Tn.far_off, (int)Tn.far_n, D);
        block_dict_issue<MODE, true>(Tn, Wn.nd, nullptr, V.theta, D);
    }
    if (W.has_slice) tile_m_step(W, B, V.w_s, V.acc_w, V.lane, V.fx);
    if (has_next && Wn.has_slice && Wn.m > 0) load8_clamped(B, Wn.b, Wn.m < 8 ? Wn.m : 8);
    __syncthreads();
    block_dict_flush(T, Dcur, V.th_w, V.acc_w, V.acc, V.fx);
    if constexpr (I + 1 < N) {
        if (has_next) __syncthreads();        // the flush reads by slot what the next stage's store writes by block: other threads
        if (has_next) tiled_stage<WEIGHTED, MODE, I + 1, N>(V, in, Tn, Wn, A, B, D, ll);
    }
}

template <bool WEIGHTED, int MODE, int N>
__global__ __launch_bounds__(kTiledThreads, 4) void k_pass_tiled_multi(const Tile *__restrict__ tiles, int n_tiles, const uint32_t *__restrict__ fwd,
                                                                    const uint32_t *__restrict__ bwd,
                                                                    const int32_t *__restrict__ far_tid, const int32_t *__restrict__ wgt,
                                                                    const double *__restrict__ theta, double *__restrict__ acc,
                                                                    double *__restrict__ ll_out, Fx fx) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    __shared__ double red[kTiledThreads / 64];
    TileEnv V;
    V.tiles = tiles; V.n_tiles = n_tiles; V.stride = (int)gridDim.x; V.fwd = fwd; V.bwd = bwd; V.far_tid = far_tid; V.wgt = wgt;
    V.theta = theta; V.acc = acc; V.fx = fx.mass;
    V.lane = threadIdx.x & 63;
    V.wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    V.th_w = lds; V.acc_w = lds + kTiledDictPad; V.w_s = lds + 2 * kTiledDictPad + V.wave * kTiledWr;
    const DTile T = tile_load(tiles + blockIdx.x);
    const TileWave W = tile_wave(T, V.wave, V.lane, fwd, bwd);
    BlockDict D;
    int4 A[8], B[8];
    block_dict_far_issue(far_tid + T.far_off, (int)T.far_n, D);
    block_dict_issue<MODE, true>(T, W.nd, nullptr, theta, D);
    if (W.has_slice) {
        load8_clamped(A, W.e, W.k < 8 ? W.k : 8);
        if (W.m > 0) load8_clamped(B, W.b, W.m < 8 ? W.m : 8);
    }
    if (V.lane < 8) V.w_s[emsar::kTileSliceRows + V.lane] = 0.0;
    LlAcc ll;
    tiled_stage<WEIGHTED, MODE, 0, N>(V, (int)blockIdx.x, T, W, A, B, D, ll);
    if (MODE == MODE_EM_LL) {
        double t = block_sum<kTiledThreads>(ll_value(ll), red);
        if (threadIdx.x == 0 && t != 0.0) ll_add(ll_out, t, fx.ll);
    }
}

// ------------------------------------------------------------------------------------------------
// k_pass_tiled_unit: one workgroup per UNIT -- one or two tiles that share a dictionary (the second `follows` the first:
// layout_tiled.hpp).  With block entries a slice is half the work it was, and the per-tile chain (dictionary fetch -> table ->
// barrier ... barrier -> flush) had grown to 44 % of a wave's life in barrier waits: a unit pays it once for up to eight slices.
// Tile 2's forward columns are requested while tile 1's M-step runs, its backward segments afterwards (as in k_pass_tiled_multi).
// ------------------------------------------------------------------------------------------------
template <bool WEIGHTED, int MODE, bool STAMP = false>
__global__ __launch_bounds__(kTiledThreads, 4) void k_pass_tiled_unit(const Tile *__restrict__ utiles, int stride, const int32_t *__restrict__ far_tid,
                                                                   const uint32_t *__restrict__ fwd, const uint32_t *__restrict__ bwd,
                                                                   
                                                                   const int32_t *__restrict__ wgt, const double *__restrict__ theta,
                                                                   double *__restrict__ acc, double *__restrict__ ll_out, Fx fx,
                                                                   unsigned long long *stamps = nullptr) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    __shared__ double red[kTiledThreads / 64];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    double *th_w = lds, *acc_w = lds + kTiledDictPad, *w_s = lds + 2 * kTiledDictPad + wave * kTiledWr;
    unsigned long long ts0 = 0, ts1 = 0, ts2 = 0, te = 0, tm = 0, tr0 = 0;      // STAMP: shader-clock stamps of the phases (diagnostic instance only)
    if (STAMP) { ts0 = stamp_now(); tr0 = stamp_real(); }
    const Tile *tiles = utiles + (size_t)blockIdx.x * (size_t)stride;          // the unit's tiles: address known from the start
    BlockDict D;
    DTile T = tile_load(tiles);
    // (a copy of the far list at a fixed stride per unit, requested together with the descriptor, was measured: 1 % SLOWER than
    // this dependent load -- the round trip it saves is not what the waves wait for, the unused tail of the list it reads is traffic)
    block_dict_far_issue(far_tid + T.far_off, (int)T.far_n, D);
    const int dict_near_n = (int)T.near_n, dict_lo = T.lo, dict_far_n = (int)T.far_n;
    const uint32_t dict_far_off = T.far_off;                       // the dictionary is the unit's, not the tile's
    // which slice of a tile a wave takes is the layout's choice (Tile::wave_of: slices dealt to the waves by work)
    TileWave W = tile_wave(T, unit_slice(T, wave), lane, fwd, bwd);
    int4 A[8], B[8];
    // the first forward columns go out BEFORE the dictionary's theta (whose addresses wait for the far list): unconditionally -- a wave without a
    // slice asks for one line eight times -- so that the wait for the far list is vmcnt(8) on every path, not vmcnt(0)
    load8_clamped(A, W.has_slice ? W.e : reinterpret_cast<const int4 *>(fwd), W.has_slice ? (W.k < 8 ? W.k : 8) : 1);
    block_dict_issue<MODE, true>(T, W.nd, nullptr, theta, D);
    if (lane < 8) w_s[emsar::kTileSliceRows + lane] = 0.0;
    block_dict_store(T, D, th_w, acc_w);
    vm_loads_landed();
    if (STAMP) ts1 = stamp_now();
    __syncthreads();
    if (STAMP) ts2 = stamp_now();
    LlAcc ll;
    int n_done = 0;
    for (int t = 0;; t++) {
        unsigned long long ta = 0, tb = 0;
        if (STAMP) ta = stamp_now();              // the tiles of the unit, one after the other, on the same table
        // Order of the requests, and why they are unconditional inside a wave's branch: hipcc counts outstanding loads exactly (s_waitcnt
        // vmcnt(n)) only along straight-line code; where two paths meet that issued different numbers of loads it waits for the larger
        // share.  So this tile's backward segments are requested right before its E-step (in flight during it) and the next tile's forward
        // columns right before its M-step (in flight during it), both inside the has_slice branch; when there is no next tile the wave
        // requests eight copies of one 16-byte line instead (every lane the same address) so that the count the M-step waits on is the same.
        // (Requests by EVERY wave, also those without a slice, were measured: 0.1049 against 0.1028 ms -- idle waves should stay idle.)
        if (W.has_slice) {
            load8_clamped(B, W.b, W.m < 8 ? W.m : 8);          // (a slice has rows, a row has entries: m >= 1, check_tiled_extents)
            tile_e_step<WEIGHTED, MODE>(W, A, (size_t)T.row_base + (size_t)W.slice * emsar::kTileSliceRows + lane, wgt, th_w, w_s, lane, ll);
        }
        const DTile Tn = tile_load(tiles + (t + 1 < stride ? t + 1 : t));
        const bool more = t + 1 < stride && Tn.n_slices > 0;
        TileWave Wn = tile_wave(Tn, unit_slice(Tn, wave), lane, fwd, bwd);
        if (!more) Wn.has_slice = false;
        if (STAMP) tb = stamp_now();
        if (W.has_slice) {
            load8_clamped(A, Wn.has_slice ? Wn.e : reinterpret_cast<const int4 *>(fwd), Wn.has_slice ? (Wn.k < 8 ? Wn.k : 8) : 1);
            tile_m_step(W, B, w_s, acc_w, lane, fx.mass);
        } else if (Wn.has_slice) load8_clamped(A, Wn.e, Wn.k < 8 ? Wn.k : 8);
        if (STAMP) { const unsigned long long tc = stamp_now(); te += tb - ta; tm += tc - tb; }
        T = Tn; W = Wn;
        n_done = t + 1;
        if (!more) break;
    }
    unsigned long long ts3 = 0, ts4 = 0;
    if (STAMP) ts3 = stamp_now();
    // The transcripts of the dictionary are fetched / derived again for the flush instead of being kept: five registers held across
    // the E- and M-steps were five registers spilt (128 VGPRs at four workgroups per CU); the far list is an L2 hit by now and its
    // latency is covered by the barrier.
    block_dict_far_issue(far_tid + dict_far_off, dict_far_n, D);
    __syncthreads();
    if (STAMP) ts4 = stamp_now();
#pragma unroll
    for (int j = 0; j < 2; j++) {
        const int sl = (int)threadIdx.x + j * kTiledThreads;
        D.stid[j] = sl < dict_near_n ? dict_lo + sl : -1;
    }
    T.near_n = (uint32_t)dict_near_n;            // (T is the last tile read by now: an absent one when the unit has fewer tiles than the stride)
    block_dict_flush(T, D, th_w, acc_w, acc, fx.mass);
    if (STAMP && lane == 0) {      // [unit][wave]: descriptor + dictionary + first loads, barrier, E-steps, M-steps, barrier, flush, tiles
        unsigned long long *o = stamps + ((size_t)blockIdx.x * (kTiledThreads / 64) + wave) * 8;
        o[0] = ts1 - ts0; o[1] = ts2 - ts1; o[2] = te; o[3] = tm; o[4] = ts4 - ts3; o[5] = stamp_now() - ts4; o[6] = n_done; o[7] = ts3 - ts2 - te - tm;
        if (wave == 0) {           // the workgroup's timeline record, behind the per-wave records: start, end (100 MHz), place
            unsigned long long *g = stamps + (size_t)gridDim.x * (kTiledThreads / 64) * 8 + (size_t)blockIdx.x * 4;
            g[0] = tr0; g[1] = stamp_real(); g[2] = stamp_place(); g[3] = (unsigned long long)n_done;
        }
    }
    if (MODE == MODE_EM_LL) {
        double t = block_sum<kTiledThreads>(ll_value(ll), red);
        if (threadIdx.x == 0 && t != 0.0) ll_add(ll_out, t, fx.ll);
    }
}

// likelihood terms of the folded single-tid rows: sum_t u_t log theta_t
__global__ __launch_bounds__(256) void k_single_ll(int n, const double *__restrict__ u, const double *__restrict__ theta, double *ll_out, double fx_ll) {
    __shared__ double red[4];
    double s = 0.0;
    for (int t = blockIdx.x * 256 + threadIdx.x; t < n; t += gridDim.x * 256) {
        double x = theta[t], c = u[t];
        if (c > 0.0 && x > 0.0) s += c * log(x);
    }
    double tot = block_sum<256>(s, red);
    if (threadIdx.x == 0 && tot != 0.0) ll_add(ll_out, tot, fx_ll);
}

}  // namespace
