// layout_tiled.hpp -- host-side construction of the TILED HBM layout (pure C++, no HIP calls).
//
// Goal: a pass whose inner loops touch only LDS with 10-bit operands, with no atomics in the inner loops, no workgroup
// barrier between the E-step and the M-step, and a dictionary (theta / accumulator window in LDS) that is loaded and flushed
// once for MANY slices.
//
//   * rows with ONE tid never reach the kernel: they are folded into a per-transcript count vector u
//     (acc_t += u_t / theta_t is applied analytically in k_update);
//   * rows with 2..kMaxRowLen tids are sorted by (block(anchor tid), length class, anchor tid), anchor = median tid, and cut
//     into SLICES of at most 768 rows; one wavefront processes one slice at a time:
//       forward index (E-step, row sums): column-major [k][768]; row p of the slice is field p/64 of lane p%64, so column j
//         of a lane's 12 rows is ONE int4 of twelve 10-bit dictionary ids; padding points at a zero slot (no branches);
//       backward index (M-step, column sums) OF THE SAME 768 ROWS: for every dictionary column with >= 4 entries in the
//         slice, its slice-local rows cut into segments of 11 row ids headed by the column id (12 x 10 bit = one int4).
//         Segments are dealt to lanes in contiguous column order (a lane keeps the running sum of a column in a register),
//         but stored interleaved so that wave loads stay 1 KiB contiguous.  Columns with < 4 entries go to a COO list;
//   * FAR ENTRIES.  An entry more than kFarReach tids away from its row's anchor (a read that also hits a transcript of another
//     gene family) would need a dictionary slot of its own, a scattered theta gather and a scattered flush atomic -- on
//     BASELINE config 3 such entries were 0.6 % of the entries but 19 % of all dictionary slots and two thirds of the flush
//     atomics.  The first far entry of a row is therefore EXPORTED: the row is placed in one of the last fields of its slice
//     (field 11, then 10: at most 128 such rows per slice), the far tid goes to a per-slice FAR BLOCK of 64 tids that the
//     lanes of the wave gather straight from theta into the register that holds that field's row sum, and the row's weight
//     w_r is stored to far_w[block][lane] (one coalesced 512-byte store per block); the update kernels add, for transcript t,
//     far_w[far_src[q]] over its exported entries q in [far_ptr[t], far_ptr[t+1]) -- a CSR by transcript whose values are
//     gathered through far_src: no dictionary slot, no atomic, fixed summation order.
//     Further far entries of the same row (rare) keep explicit dictionary slots (the group's far list);
//     A row of exactly TWO transcripts that are far from each other (a read with one hit here and one in another family) is a
//     PAIR: it never enters a slice; the pass kernel computes w = R / (theta_a + theta_b) from two gathers and stores it to far_w
//     twice, once for each transcript (slices full of such rows would need a far entry for every row);
//   * consecutive slices whose near entries fit one window of <= 959 transcripts form a GROUP: one dictionary, loaded
//     once, flushed once; the waves of the workgroup take the group's slices one after another (no barrier in between);
//   * the slices are cut into CHUNKS of equal work, one workgroup each (as many chunks as the device holds workgroups, so
//     that every workgroup is resident from the start and all of them finish together); a chunk is one or more groups;
//   * rows longer than kMaxRowLen go to a leftover CSR processed by the generic kernel.
#pragma once
#include <algorithm>
#include <thread>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstddef>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <memory>
#include <new>
#include <system_error>
#include <utility>
#include <vector>

#include "layout.hpp"

namespace emsar {

constexpr int kTileWaves = 4;          // wavefronts per workgroup
constexpr int kRowsPerLane = 12;       // twelve 10-bit ids per int4
constexpr int kTileSliceRows = 64 * kRowsPerLane;   // 768
constexpr int kTileDict = 959;         // theta + acc windows in LDS: 2 x 7.5 KiB; +1 zero slot
constexpr int64_t kFragRows = 1 << 21;   // sorted rows per independently sliced fragment (build_tiled)
constexpr int kMaxRowLen = 768;        // longer rows -> leftover CSR (a row must fit one dictionary)
constexpr int kSegRows = 11;           // row ids per backward segment (plus 1 header = 12 x 10 bit = one int4)
constexpr int kSliceDwords = kTileSliceRows / 3;    // dwords per forward column of a slice (256 = 1 KiB)
constexpr int kFarReach = 200;         // |tid - anchor| beyond this: a far entry (block 512 + 2 x 200 < 959)
constexpr int kMaxFarBlocks = 2;       // far blocks (64 exported rows each) per slice
constexpr int64_t kSliceEntries = 65535;   // entries of one slice (16-bit COO count)
constexpr uint32_t kFarHot = 4096;     // a transcript that is the far hit of more rows than this keeps dictionary slots (see export_index)

// field i (0..11) of a packed int4: dword i/3, bits 10*(i%3) .. +10
inline void pack10(uint32_t *q, int i, uint32_t id) { q[i / 3] |= (id & 0x3FFu) << (10 * (i % 3)); }
inline uint32_t unpack10(const uint32_t *q, int i) { return (q[i / 3] >> (10 * (i % 3))) & 0x3FFu; }
constexpr int kDenseMin = 4;           // columns with fewer entries in a slice use the COO list

// vectors whose resize(n) leaves the new elements uninitialised (resize(n, v) still fills): the big index arrays are
// sized once and filled by several threads, a zero fill by one thread first would cost as much as the copy
template <class T> struct no_init_alloc : std::allocator<T> {
    template <class U> struct rebind { using other = no_init_alloc<U>; };
    template <class U, class... A> void construct(U *p, A &&...a) {
        if constexpr (sizeof...(A) == 0) ::new ((void *)p) U; else ::new ((void *)p) U(std::forward<A>(a)...);
    }
};
using u32_vec = std::vector<uint32_t, no_init_alloc<uint32_t>>;
using i64_vec = std::vector<int64_t, no_init_alloc<int64_t>>;

struct SliceDesc {           // 32 bytes; slice i owns row slots [768 i, 768 i + 768): slot = 768 i + 64 field + lane
    uint32_t fwd_kib;        // forward block: int4 index = 64 * fwd_kib   (1 KiB = one column of the slice)
    uint32_t bwd_kib;        // backward block, same unit (1 KiB = one segment per lane)
    uint32_t coo_off;        // first pair in coo[]
    uint32_t far_blk;        // first far block of the slice: block j serves field 11 - j
    uint16_t k;              // forward columns (padded row length, the exported entry not counted)
    uint16_t m;              // backward segments (int4) per lane
    uint16_t coo_n;          // COO pairs
    uint16_t nf;             // far blocks (0..kMaxFarBlocks)
    uint32_t n_rows;         // rows stored (<= 768)
    uint32_t pad0;
};
static_assert(sizeof(SliceDesc) == 32, "SliceDesc must stay 32 bytes");

struct GroupDesc {           // 32 bytes: consecutive slices that share one dictionary
    int32_t lo;              // dictionary slot d < near_n  <->  tid lo + d
    uint16_t near_n, far_n;  // slot near_n + i <-> far_tid[far_off + i]; zero slot = near_n + far_n
    uint32_t far_off;
    uint32_t slice_begin, slice_end;
    uint32_t pad0[3];
};
static_assert(sizeof(GroupDesc) == 32, "GroupDesc must stay 32 bytes");

struct ChunkDesc { uint32_t group_begin, group_end; };   // one workgroup

struct TiledLayout {
    int64_t n_rows = 0, nnz = 0;
    int32_t n_tx = 0;
    // singletons: folded rows
    std::vector<uint32_t> single_row;   // original row index
    std::vector<int32_t> single_tid;
    // slices, groups, chunks
    std::vector<SliceDesc> slices;
    std::vector<GroupDesc> groups;
    std::vector<ChunkDesc> chunks;
    i64_vec slot_row;                   // row slot -> original row, or merged-row id when `merged` (-1 = padding)
    bool merged = false;                // identical rows were merged: a slot stands for mem_row[mem_ptr[id] .. mem_ptr[id+1])
    std::vector<uint64_t> mem_ptr;
    std::vector<uint32_t> mem_row;
    u32_vec fwd, bwd;                   // packed 10-bit ids
    std::vector<uint32_t> coo;          // (col_id << 16) | row_id
    std::vector<int32_t> far_tid;       // explicit dictionary far lists of the groups
    // exported far entries
    std::vector<int32_t> far_blk_tid;   // [n_far_blocks][64]: far tid of the row in (block, lane), -1 = none
    std::vector<uint32_t> far_ptr;      // [n_tx + 1]: exported entries by transcript: far_src[far_ptr[t] .. far_ptr[t+1])
    std::vector<uint32_t> far_src;      // -> index into far_w: 64 block + lane for a slice's row, 64 n_far_blocks + i for pair i
    int64_t n_exported = 0;             // = far_ptr[n_tx] = far_src.size()
    // pairs: rows of two transcripts far from each other, both entries exported
    std::vector<int64_t> pair_row;      // original row, or merged-row id when `merged`
    std::vector<int32_t> pair_tid;      // [2 n_pairs]
    // leftover rows (too long for a slice): plain CSR + original row ids
    std::vector<uint64_t> left_ptr;
    std::vector<int32_t> left_col;
    std::vector<uint32_t> left_row;
    int64_t tiled_entries = 0, far_entries = 0 /* explicit dictionary far slots used, summed over entries */, exported_entries = 0,
            coo_entries = 0, padded_slots = 0;
    int64_t n_slots() const { return (int64_t)slot_row.size(); }
    int64_t n_far_blocks() const { return (int64_t)far_blk_tid.size() / 64; }
};

// fn(0) on the calling thread, fn(1..nt-1) on threads of their own.  An allocation failure inside a thread must not end the
// process (an exception that leaves a std::thread calls std::terminate): it is caught there and rethrown here after the
// joins, where emsar_hip_upload_structure turns it into EMSAR_HIP_ERR_OOM.  A thread that cannot be started runs inline.
template <class F>
inline void run_on_threads(int nt, F fn) {
    std::atomic<bool> oom{false};
    auto guarded = [&](int t) { try { fn(t); } catch (const std::bad_alloc &) { oom.store(true); } };
    std::vector<std::thread> pool;
    std::vector<int> inline_ids;
    for (int t = 1; t < nt; t++) {
        try { pool.emplace_back(guarded, t); } catch (const std::system_error &) { inline_ids.push_back(t); }
    }
    guarded(0);
    for (int t : inline_ids) guarded(t);
    for (auto &th : pool) th.join();
    if (oom.load()) throw std::bad_alloc();
}

// Every range a kernel derives from a descriptor must lie inside the arrays it indexes -- k_pass_tiled has no bounds checks, an
// out-of-range read is a GPU memory fault.  Checked on the host for every layout before it is uploaded (build_tiled's last
// step) and again by check_tiled.  As the kernels compute them (kernels_tiled.hpp):
//   chunk     groups [group_begin, group_end) inside groups[]
//   group     slices [slice_begin, slice_end) inside slices[]; dictionary near_n + far_n <= kTileDict, tids lo .. lo + near_n - 1
//             and the far list [far_off, far_off + far_n) inside [0, n_tx)
//   slice i   forward   int4 reads [64 fwd_kib, 64 (fwd_kib + k))        k >= 1 (load8_clamped reads position n - 1)
//             backward  int4 reads [64 bwd_kib, 64 (bwd_kib + m))
//             COO       dword reads [coo_off, coo_off + coo_n)
//             far blocks [far_blk, far_blk + nf), nf <= kMaxFarBlocks, every tid in it -1 or inside [0, n_tx)
//             row slots [768 i, 768 i + 768)   (weights, scatter values, slot_row)
//   pairs     two tids inside [0, n_tx)
//   far_src   every entry < 64 n_far_blocks + n_pairs (the length of far_w)
// 0 = fine, else a negative code naming the first violated rule.
inline int check_tiled_extents(const TiledLayout &L) {
    const uint64_t n_fwd = (uint64_t)L.fwd.size() / 4, n_bwd = (uint64_t)L.bwd.size() / 4;     // int4 units
    const uint64_t n_blk = (uint64_t)L.far_blk_tid.size() / 64;
    if (L.far_blk_tid.size() % 64) return -20;
    for (const ChunkDesc &C : L.chunks) if (C.group_begin > C.group_end || C.group_end > L.groups.size()) return -21;
    for (const GroupDesc &G : L.groups) {
        if (G.slice_begin >= G.slice_end || G.slice_end > L.slices.size()) return -21;
        if ((int)G.near_n + (int)G.far_n > kTileDict || G.near_n < 1) return -22;
        if (G.lo < 0 || (int64_t)G.lo + G.near_n > (int64_t)L.n_tx) return -23;
        if ((uint64_t)G.far_off + G.far_n > (uint64_t)L.far_tid.size()) return -24;
        for (uint32_t i = 0; i < G.far_n; i++) { const int32_t t = L.far_tid[(size_t)G.far_off + i]; if (t < 0 || t >= L.n_tx) return -24; }
    }
    for (const SliceDesc &D : L.slices) {
        if (D.k < 1 || D.k > kMaxRowLen || D.nf > kMaxFarBlocks || D.n_rows < 1 || D.n_rows > (uint32_t)kTileSliceRows) return -26;
        if (64 * ((uint64_t)D.fwd_kib + D.k) > n_fwd) return -27;
        if (64 * ((uint64_t)D.bwd_kib + D.m) > n_bwd) return -28;
        if ((uint64_t)D.coo_off + D.coo_n > (uint64_t)L.coo.size()) return -29;
        if ((uint64_t)D.far_blk + D.nf > n_blk) return -33;
    }
    for (int32_t t : L.far_blk_tid) if (t < -1 || t >= L.n_tx) return -33;
    if ((uint64_t)L.slices.size() * kTileSliceRows != (uint64_t)L.slot_row.size()) return -30;
    if (L.far_ptr.size() != (size_t)L.n_tx + 1 || L.far_ptr.front() != 0 || (int64_t)L.far_ptr.back() != L.n_exported) return -34;
    for (size_t t = 0; t + 1 < L.far_ptr.size(); t++) if (L.far_ptr[t] > L.far_ptr[t + 1]) return -34;
    if ((int64_t)L.far_src.size() != L.n_exported) return -34;
    for (uint32_t x : L.far_src) if ((uint64_t)x >= (uint64_t)L.far_blk_tid.size() + L.pair_row.size()) return -34;
    if (L.left_ptr.size() != L.left_row.size() + 1 || (L.left_ptr.empty() ? 0 : L.left_ptr.back()) != (uint64_t)L.left_col.size()) return -31;
    if (L.single_row.size() != L.single_tid.size()) return -32;
    if (L.pair_tid.size() != 2 * L.pair_row.size()) return -35;
    for (size_t i = 0; i < L.pair_tid.size(); i++) if (L.pair_tid[i] < 0 || L.pair_tid[i] >= L.n_tx) return -35;
    return 0;
}

// With merge_rows, rows with the same tid multiset (2..kMaxRowLen tids) are stored once and weighted by the sum of
// their members' weights -- what the reference's update_ReadCounts does when it counts reads per segment
// (emsar_functions.c:838-943).  Every quantity the library computes is a sum over rows of a function of the row's tid
// set times a per-row weight, so the merge is exact up to summation order.
// n_wg_slots: workgroups the device holds at once (4 per CU): the number of equal-work chunks.
inline int build_tiled(int64_t n_rows, int32_t n_tx, const uint64_t *row_ptr_in, const int32_t *col_idx_in, TiledLayout &out,
                       bool merge_rows = false, int n_wg_slots = 1024) {
    if (n_rows >= (int64_t)1 << 32) return -1;
    out = TiledLayout();
    const uint64_t *row_ptr = row_ptr_in;
    const int32_t *col_idx = col_idx_in;
    // merged view of the matrix (only built when asked for): unique rows with sorted tids
    std::vector<uint64_t> m_ptr;
    std::vector<int32_t> m_col;
    std::vector<uint32_t> orig_of_merged;   // first member, for rows that are not merged (singles, long rows)
    if (merge_rows) {
        const int64_t nnz_in = (int64_t)row_ptr_in[n_rows];
        std::vector<int32_t> scol((size_t)nnz_in);
        std::vector<uint64_t> hash((size_t)n_rows, 0);
        for (int64_t r = 0; r < n_rows; r++) {
            uint64_t b = row_ptr_in[r], e = row_ptr_in[r + 1];
            std::copy(col_idx_in + b, col_idx_in + e, scol.begin() + (int64_t)b);
            std::sort(scol.begin() + (int64_t)b, scol.begin() + (int64_t)e);
            uint64_t h = 1469598103934665603ull ^ (e - b);
            for (uint64_t k = b; k < e; k++) { h ^= (uint64_t)(uint32_t)scol[(size_t)k]; h *= 1099511628211ull; }
            hash[(size_t)r] = h;
        }
        uint64_t cap = 16;
        while (cap < (uint64_t)n_rows * 2 + 2) cap <<= 1;
        std::vector<int64_t> table((size_t)cap, -1);       // -> merged id
        std::vector<int64_t> merged_of((size_t)n_rows, -1);
        std::vector<uint32_t> cnt;                          // members per merged row
        m_ptr.push_back(0);
        for (int64_t r = 0; r < n_rows; r++) {
            uint64_t b = row_ptr_in[r], e = row_ptr_in[r + 1], len = e - b;
            bool mergeable = len >= 2 && len <= (uint64_t)kMaxRowLen;
            int64_t id = -1;
            if (mergeable) {
                uint64_t h = hash[(size_t)r] & (cap - 1);
                while (table[(size_t)h] >= 0) {
                    int64_t o = table[(size_t)h];
                    uint64_t ob = m_ptr[(size_t)o], oe = m_ptr[(size_t)o + 1];
                    if (oe - ob == len && std::memcmp(m_col.data() + ob, scol.data() + b, len * 4) == 0) { id = o; break; }
                    h = (h + 1) & (cap - 1);
                }
                if (id < 0) table[(size_t)h] = (int64_t)cnt.size();
            }
            if (id < 0) {
                id = (int64_t)cnt.size();
                m_col.insert(m_col.end(), scol.begin() + (int64_t)b, scol.begin() + (int64_t)e);
                m_ptr.push_back((uint64_t)m_col.size());
                cnt.push_back(0);
                orig_of_merged.push_back((uint32_t)r);
            }
            merged_of[(size_t)r] = id;
            cnt[(size_t)id]++;
        }
        const int64_t n_m = (int64_t)cnt.size();
        out.mem_ptr.assign((size_t)n_m + 1, 0);
        for (int64_t i = 0; i < n_m; i++) out.mem_ptr[(size_t)i + 1] = out.mem_ptr[(size_t)i] + cnt[(size_t)i];
        out.mem_row.resize((size_t)n_rows);
        std::vector<uint64_t> fillp(out.mem_ptr.begin(), out.mem_ptr.end() - 1);
        for (int64_t r = 0; r < n_rows; r++) out.mem_row[(size_t)fillp[(size_t)merged_of[(size_t)r]]++] = (uint32_t)r;
        out.merged = true;
        row_ptr = m_ptr.data();
        col_idx = m_col.data();
        const int64_t n_rows_orig = n_rows;
        n_rows = n_m;
        out.n_rows = n_rows_orig; out.n_tx = n_tx; out.nnz = nnz_in;
    }
    if (!merge_rows) { out.n_rows = n_rows; out.n_tx = n_tx; out.nnz = (int64_t)row_ptr[n_rows]; }
    out.left_ptr.push_back(0);

    const bool dbg_t = getenv("EMSAR_HIP_DEBUG") != nullptr;
    auto t_now = [] { return std::chrono::steady_clock::now(); };
    auto t_ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
    const auto tp0 = t_now();
    // ---- classify rows; keys of the tiled ones ----
    // host threads of the builder (fragments below use the same rule); every parallel step gives the result of the
    // sequential one, so the layout does not depend on the number of threads
    int n_host = 1;
    int64_t min_chunk = (int64_t)1 << 16;                 // rows per thread at least
    {
        unsigned hw = std::thread::hardware_concurrency();
        n_host = (int)(hw ? std::min(hw, 16u) : 1u);
        if (n_rows < (int64_t)1 << 16) n_host = 1;
        if (const char *e = getenv("EMSAR_HOST_THREADS")) { int v = atoi(e); if (v >= 1) { n_host = std::min(v, 64); min_chunk = 16; } }   // tests: threads on small inputs
    }
    auto par_ranges = [&](int64_t n, const std::function<void(int, int64_t, int64_t)> &fn) {
        const int nt = (int)std::min<int64_t>(n_host, std::max<int64_t>(1, n / min_chunk));
        run_on_threads(nt, [&](int t) { fn(t, n * t / nt, n * (t + 1) / nt); });
        return nt;
    };
    int far_reach = kFarReach;            // entries further than this from the row's anchor are far
    if (const char *e = getenv("EMSAR_HIP_FAR_REACH")) { int v = atoi(e); if (v >= 1 && v <= kFarReach) far_reach = v; }
    bool far_export = true;               // EMSAR_HIP_FAR_EXPORT=0: no entry is exported, every far entry keeps a dictionary slot
    if (const char *e = getenv("EMSAR_HIP_FAR_EXPORT")) far_export = atoi(e) != 0;
    uint32_t far_hot = kFarHot;
    if (const char *e = getenv("EMSAR_HIP_FAR_HOT")) { long v = atol(e); if (v >= 1) far_hot = (uint32_t)std::min<long>(v, 1 << 30); }
    std::vector<int32_t> mintid((size_t)n_rows, -1);      // the anchor tid of every tiled row (see below)
    bool anchor_median = true;
    if (const char *e = getenv("EMSAR_HIP_TILE_ANCHOR")) anchor_median = atoi(e) != 0;
    // the row's anchor in tid space decides which tile it joins: the MEDIAN id, not the smallest -- a read that also
    // hits one transcript of another family stays with its own family, and only that one entry is far from the
    // tile's window (anchored at the minimum, half of such rows landed in the other family's tile with ALL their
    // in-family ids far: 6.5 M far entries on config 3 instead of 1.6 M)
    par_ranges(n_rows, [&](int, int64_t lo, int64_t hi) {
        std::vector<int32_t> tmp;
        for (int64_t r = lo; r < hi; r++) {
            const uint64_t b = row_ptr[r], e = row_ptr[r + 1], len = e - b;
            if (len < 2 || len > (uint64_t)kMaxRowLen) continue;
            int32_t m;
            if (anchor_median) {
                tmp.assign(col_idx + b, col_idx + e);
                std::nth_element(tmp.begin(), tmp.begin() + (std::ptrdiff_t)(len / 2), tmp.end());
                m = tmp[(size_t)(len / 2)];
            } else {
                m = col_idx[b];
                for (uint64_t k = b + 1; k < e; k++) m = std::min(m, col_idx[k]);
            }
            mintid[(size_t)r] = m;
        }
    });
    std::vector<uint32_t> act;                             // the tiled rows, ascending
    {
        // every thread classifies a contiguous range of rows into lists of its own; the lists are joined in range order
        struct Part { std::vector<uint32_t> single_row, left_row, act; std::vector<int32_t> single_tid, left_col, pair_tid; std::vector<uint64_t> left_len; std::vector<int64_t> pair_row; };
        std::vector<Part> part((size_t)std::max(1, n_host));
        const int np = par_ranges(n_rows, [&](int t, int64_t lo, int64_t hi) {
            Part &P = part[(size_t)t];
            for (int64_t r = lo; r < hi; r++) {
                const uint64_t b = row_ptr[r], e = row_ptr[r + 1], len = e - b;
                if (len == 0) continue;
                const uint32_t r_orig = merge_rows ? orig_of_merged[(size_t)r] : (uint32_t)r;   // singles / long rows are never merged
                if (len == 1) { P.single_row.push_back(r_orig); P.single_tid.push_back(col_idx[b]); continue; }
                if (len == 2 && far_export) {
                    const int32_t ta = col_idx[b], tb = col_idx[b + 1];
                    if ((ta > tb ? ta - tb : tb - ta) > far_reach) { P.pair_row.push_back(r); P.pair_tid.push_back(ta); P.pair_tid.push_back(tb); continue; }
                }
                bool too_wide = false;                        // near span + far entries beyond one dictionary (only possible for len > 550)
                if (len > (uint64_t)(kTileDict - 2 * kFarReach - 1) && len <= (uint64_t)kMaxRowLen) {
                    const int32_t a = mintid[(size_t)r];
                    int32_t lo = a, hi = a; int64_t nfar = 0;
                    for (uint64_t k = b; k < e; k++) {
                        const int32_t t = col_idx[k], dist = t > a ? t - a : a - t;
                        if (dist > far_reach) nfar++; else { lo = std::min(lo, t); hi = std::max(hi, t); }
                    }
                    too_wide = (int64_t)hi - lo + 1 + nfar > kTileDict;
                }
                if (len > (uint64_t)kMaxRowLen || too_wide) {
                    P.left_row.push_back(r_orig);
                    P.left_col.insert(P.left_col.end(), col_idx + b, col_idx + e);
                    P.left_len.push_back(len);
                    continue;
                }
                P.act.push_back((uint32_t)r);
            }
        });
        size_t ns = 0, nl = 0, nlc = 0, na = 0;
        for (int t = 0; t < np; t++) { ns += part[(size_t)t].single_row.size(); nl += part[(size_t)t].left_row.size(); nlc += part[(size_t)t].left_col.size(); na += part[(size_t)t].act.size(); }
        out.single_row.reserve(ns); out.single_tid.reserve(ns); out.left_row.reserve(nl); out.left_col.reserve(nlc); out.left_ptr.reserve(nl + 1); act.reserve(na);
        for (int t = 0; t < np; t++) {
            Part &P = part[(size_t)t];
            out.single_row.insert(out.single_row.end(), P.single_row.begin(), P.single_row.end());
            out.single_tid.insert(out.single_tid.end(), P.single_tid.begin(), P.single_tid.end());
            out.left_row.insert(out.left_row.end(), P.left_row.begin(), P.left_row.end());
            out.left_col.insert(out.left_col.end(), P.left_col.begin(), P.left_col.end());
            for (uint64_t len : P.left_len) out.left_ptr.push_back(out.left_ptr.back() + len);
            act.insert(act.end(), P.act.begin(), P.act.end());
            out.pair_row.insert(out.pair_row.end(), P.pair_row.begin(), P.pair_row.end());
            out.pair_tid.insert(out.pair_tid.end(), P.pair_tid.begin(), P.pair_tid.end());
            P = Part();
        }
    }
    const int64_t n_act = (int64_t)act.size();
    // Sort granularity in tid space.  A tile's dictionary must hold a block's tid range plus the rows' reach, so
    // blocks stay small; wide blocks keep the (block, length) buckets large, i.e. the slices uniform.
    int32_t block = 256;                  // two neighbouring blocks plus the rows' reach fit one dictionary (2 x 256 + 2 x kFarReach < kTileDict)
    if (const char *e = getenv("EMSAR_HIP_TILE_BLOCK")) { int v = atoi(e); if (v >= 64 && v <= 900) block = v; }
    int dense_min = kDenseMin;
    if (const char *e = getenv("EMSAR_HIP_TILE_DENSE")) { int v = atoi(e); if (v >= 1 && v <= 64) dense_min = v; }
    if (const char *e = getenv("EMSAR_HIP_CHUNKS")) { int v = atoi(e); if (v >= 1 && v <= (1 << 20)) n_wg_slots = v; }
    const auto tp1 = t_now();
    // ---- sort: pass A by anchor tid, pass B by (block, length class); both stable ----
    // A stable counting sort over chunks of the input: one histogram per chunk, offsets ordered (key, chunk), then every
    // chunk scatters its own rows in order -- the permutation of the sequential sort.
    auto counting_sort = [&](const std::vector<uint32_t> &in, std::vector<uint32_t> &dst, size_t n_keys, auto key) {
        const int64_t n = (int64_t)in.size();
        dst.resize((size_t)n);
        const int nc = (int)std::min<int64_t>(n_host, std::max<int64_t>(1, n / min_chunk));
        std::vector<std::vector<uint64_t>> hist((size_t)nc);
        auto chunk = [&](int c) { return std::make_pair(n * c / nc, n * (c + 1) / nc); };
        auto count = [&](int c) {
            hist[(size_t)c].assign(n_keys, 0);
            auto [lo, hi] = chunk(c);
            for (int64_t i = lo; i < hi; i++) hist[(size_t)c][key(in[(size_t)i])]++;
        };
        run_on_threads(nc, count);
        uint64_t run = 0;
        for (size_t k = 0; k < n_keys; k++)
            for (int c = 0; c < nc; c++) { const uint64_t h = hist[(size_t)c][k]; hist[(size_t)c][k] = run; run += h; }
        auto scatter = [&](int c) {
            auto [lo, hi] = chunk(c);
            uint64_t *off = hist[(size_t)c].data();
            for (int64_t i = lo; i < hi; i++) { const uint32_t r = in[(size_t)i]; dst[(size_t)off[key(r)]++] = r; }
        };
        run_on_threads(nc, scatter);
    };
    std::vector<uint32_t> pa, perm;
    {
        counting_sort(act, pa, (size_t)n_tx, [&](uint32_t r) { return (size_t)mintid[r]; });
        std::vector<uint32_t>().swap(act);
        const int64_t n_blocks = ((int64_t)n_tx + block - 1) / block;
        counting_sort(pa, perm, (size_t)(n_blocks * kLenClasses), [&](uint32_t r) {
            return (size_t)(mintid[r] / block) * kLenClasses + (size_t)len_class((int64_t)(row_ptr[r + 1] - row_ptr[r]));
        });
    }
    std::vector<uint32_t>().swap(pa);
    const auto tp2 = t_now();


    // ---- P1: slices ----
    // The sorted rows are cut into fragments of kFragRows rows; every fragment is cut into slices on its own.  The cut
    // points depend on the data only, so the layout is the same whatever the number of host threads that build it.
    // A row is FAR when it has an entry more than far_reach tids from its anchor and its slice still has room in a far
    // block; such rows take the first lanes of the top fields (block 0 = field 11, block 1 = field 10), the others fill the rest.
    struct SliceTmp {
        int64_t begin = 0, end = 0;          // sorted-row range
        int32_t nmin = 0, nmax = -1;         // tid range of the entries that need dictionary slots by position (near entries)
        uint32_t n_norm = 0, n_far = 0, k = 0;
        uint32_t m = 0, coo_n = 0;           // backward units (int4 per lane) and COO pairs the encoder will produce
        int64_t ents = 0;                    // stored entries (exported ones not counted)
        std::vector<int32_t> expl;           // distinct un-exported far tids (explicit dictionary slots), ascending
    };
    auto is_far = [&](int32_t t, int32_t anchor) { return (t > anchor ? t - anchor : anchor - t) > far_reach; };
    // index (within the row) of the first far entry, or -1
    auto first_far = [&](uint32_t r) -> int64_t {
        const int32_t a = mintid[r];
        for (uint64_t q = row_ptr[r]; q < row_ptr[r + 1]; q++) if (is_far(col_idx[q], a)) return (int64_t)(q - row_ptr[r]);
        return -1;
    };
    // Exporting pays for COLD far transcripts: one row here, one there.  A transcript that is the far hit of many rows (a highly
    // expressed paralog) is better served by a dictionary slot -- the rows of a group that hit it share one theta gather and one
    // flush atomic -- and its run in far_w is summed by one workgroup of the update kernel, one run after the other.  So the far
    // hits are counted per transcript first, and only transcripts with at most kFarHot of them are exported.
    std::vector<uint32_t> far_refs;
    if (far_export) {
        far_refs.assign((size_t)n_tx, 0);
        std::vector<std::vector<uint32_t>> part((size_t)std::max(1, n_host));
        const int np = par_ranges(n_act, [&](int t, int64_t lo, int64_t hi) {
            auto &h = part[(size_t)t];
            h.assign((size_t)n_tx, 0);
            for (int64_t i = lo; i < hi; i++) { const uint32_t r = perm[(size_t)i]; const int64_t ex = first_far(r); if (ex >= 0) h[(size_t)col_idx[row_ptr[r] + (uint64_t)ex]]++; }
        });
        for (int t = 0; t < np; t++) for (int32_t x = 0; x < n_tx; x++) far_refs[(size_t)x] += part[(size_t)t][(size_t)x];
    }
    // index (within the row) of the entry that would be exported, or -1
    auto export_index = [&](uint32_t r) -> int64_t {
        if (!far_export) return -1;
        const int64_t ex = first_far(r);
        if (ex >= 0 && far_refs[(size_t)col_idx[row_ptr[r] + (uint64_t)ex]] > far_hot) return -1;
        return ex;
    };
    auto cut_slices = [&](int64_t range_begin, int64_t range_end, std::vector<SliceTmp> &dst) {
        std::vector<int32_t> stamp((size_t)n_tx, -1);
        std::vector<uint32_t> colcnt((size_t)n_tx, 0);       // entries per transcript inside the current slice
        std::vector<int32_t> touched;
        int32_t sid = 0;
        int64_t i = range_begin;
        while (i < range_end) {
            SliceTmp S;
            S.begin = i;
            touched.clear();
            int32_t nmin = INT32_MAX, nmax = -1;
            std::vector<int32_t> row_expl;
            for (; i < range_end; i++) {
                const uint32_t r = perm[(size_t)i];
                const uint64_t b = row_ptr[r], e = row_ptr[r + 1];
                const int32_t a = mintid[r];
                const int64_t ex = export_index(r);
                const bool far_row = ex >= 0 && S.n_far < (uint32_t)(64 * kMaxFarBlocks);
                const uint32_t n_norm1 = S.n_norm + (far_row ? 0 : 1), n_far1 = S.n_far + (far_row ? 1 : 0);
                const int64_t stored = (int64_t)(e - b) - (far_row ? 1 : 0);
                int32_t lo1 = nmin, hi1 = nmax;
                row_expl.clear();
                for (uint64_t q = b; q < e; q++) {
                    if (far_row && (int64_t)(q - b) == ex) continue;
                    const int32_t t = col_idx[q];
                    if (is_far(t, a)) { if (stamp[(size_t)t] != sid && std::find(row_expl.begin(), row_expl.end(), t) == row_expl.end()) row_expl.push_back(t); }
                    else { lo1 = std::min(lo1, t); hi1 = std::max(hi1, t); }
                }
                if (S.n_norm + S.n_far > 0) {
                    const bool rows_full = n_norm1 + n_far1 > (uint32_t)kTileSliceRows;
                    const bool ents_full = S.ents + stored > kSliceEntries;
                    const int64_t span = hi1 >= lo1 ? (int64_t)hi1 - lo1 + 1 : 0;
                    const bool dict_full = span + (int64_t)S.expl.size() + (int64_t)row_expl.size() > kTileDict;
                    if (rows_full || ents_full || dict_full) break;
                }
                S.n_norm = n_norm1; S.n_far = n_far1; S.ents += stored;
                S.k = std::max<uint32_t>(S.k, (uint32_t)stored);
                nmin = lo1; nmax = hi1;
                for (int32_t t : row_expl) { stamp[(size_t)t] = sid; S.expl.push_back(t); }
                for (uint64_t q = b; q < e; q++) {
                    if (far_row && (int64_t)(q - b) == ex) continue;
                    const int32_t t = col_idx[q];
                    if (colcnt[(size_t)t]++ == 0) touched.push_back(t);
                }
            }
            S.end = i;
            {   // what the encoder will make of the slice's columns: segments of kSegRows rows, or COO pairs below dense_min
                uint64_t nseg = 0, ncoo = 0;
                for (int32_t t : touched) {
                    const uint32_t c = colcnt[(size_t)t];
                    if (c < (uint32_t)dense_min) ncoo += c; else nseg += (c + kSegRows - 1) / kSegRows;
                    colcnt[(size_t)t] = 0;
                }
                S.m = (uint32_t)((nseg + 63) / 64); S.coo_n = (uint32_t)ncoo;
            }
            S.nmin = nmin == INT32_MAX ? 0 : nmin; S.nmax = nmax;
            std::sort(S.expl.begin(), S.expl.end());
            dst.push_back(std::move(S));
            sid++;
        }
    };
    std::vector<SliceTmp> st;
    {
        int64_t frag_rows = kFragRows;
        if (const char *e = getenv("EMSAR_HIP_FRAG_ROWS")) { long long v = atoll(e); if (v >= kTileSliceRows) frag_rows = v; }   // tests: many fragments on small inputs
        const int64_t n_frag = std::max<int64_t>(1, (n_act + frag_rows - 1) / frag_rows);
        std::vector<std::vector<SliceTmp>> frag((size_t)n_frag);
        const int nthr = (int)std::min<int64_t>(n_frag, std::max(1, n_host));
        std::atomic<int64_t> next{0};
        run_on_threads(nthr, [&](int) {
            for (;;) {
                const int64_t g = next.fetch_add(1);
                if (g >= n_frag) break;
                cut_slices(g * frag_rows, std::min(n_act, (g + 1) * frag_rows), frag[(size_t)g]);
            }
        });
        size_t ns = 0;
        for (auto &f : frag) ns += f.size();
        st.reserve(ns);
        for (auto &f : frag) { for (auto &x : f) st.push_back(std::move(x)); std::vector<SliceTmp>().swap(f); }
    }
    const int64_t n_slices = (int64_t)st.size();
    if (n_slices * kTileSliceRows >= ((int64_t)1 << 32)) return -1;
    const auto tp3 = t_now();

    // ---- P2: chunks of equal work (one workgroup each), groups inside a chunk, heavy slices first inside a group ----
    // work of a slice in shader cycles, fitted to per-slice stamps of a pass over BASELINE config 3 (tools/chunk_times.py):
    // c0 + ck per forward column + cm per backward unit (one int4 per lane) + cc per COO pair + cf per far block
    {
        double c0 = 3500, ck = 700, cm = 1300, cc = 25, cf = 3000;
        if (const char *e = getenv("EMSAR_HIP_COST")) { double v[5]; if (sscanf(e, "%lf,%lf,%lf,%lf,%lf", v, v + 1, v + 2, v + 3, v + 4) == 5) { c0 = v[0]; ck = v[1]; cm = v[2]; cc = v[3]; cf = v[4]; } }
        auto work_of = [&](const SliceTmp &S) { return c0 + ck * S.k + cm * S.m + cc * S.coo_n + cf * (double)((S.n_far + 63) / 64); };
        double total = 0;
        for (const auto &S : st) total += work_of(S);
        const int64_t n_chunks = std::max<int64_t>(1, std::min<int64_t>(n_slices, n_wg_slots));
        std::vector<int32_t> stamp((size_t)n_tx, -1);
        int32_t gid = 0;
        // natural breaks: where a dictionary that started at the previous break is full (greedy, over all slices).  A chunk cut that
        // falls just behind such a break would leave a sliver of the old dictionary at the head of the chunk -- a group of one or
        // two slices on which most of the workgroup's waves wait -- so cuts within 10 % of a chunk's work of a break move onto it.
        std::vector<uint8_t> is_break((size_t)n_slices + 1, 0);
        {
            int64_t g0 = 0;
            while (g0 < n_slices) {
                int32_t lo = INT32_MAX, hi = -1;
                int64_t n_expl = 0, g1 = g0;
                gid++;
                for (; g1 < n_slices; g1++) {
                    const SliceTmp &S = st[(size_t)g1];
                    int32_t lo1 = lo, hi1 = hi;
                    if (S.nmax >= S.nmin) { lo1 = std::min(lo1, S.nmin); hi1 = std::max(hi1, S.nmax); }
                    int64_t add = 0;
                    for (int32_t t : S.expl) if (stamp[(size_t)t] != gid) add++;
                    if (g1 > g0 && (hi1 >= lo1 ? (int64_t)hi1 - lo1 + 1 : 0) + n_expl + add > kTileDict) break;
                    lo = lo1; hi = hi1;
                    for (int32_t t : S.expl) if (stamp[(size_t)t] != gid) { stamp[(size_t)t] = gid; n_expl++; }
                }
                is_break[(size_t)g1] = 1;
                g0 = g1;
            }
        }
        std::vector<double> prefix((size_t)n_slices + 1, 0.0);
        for (int64_t i = 0; i < n_slices; i++) prefix[(size_t)i + 1] = prefix[(size_t)i] + work_of(st[(size_t)i]);
        const double snap = 0.10 * total / (double)n_chunks;
        // EMSAR_HIP_AGE_SKEW "a,b,c,d" (experiment): relative work of the chunks in each quarter of the chunk order.  All workgroups are
        // resident at once and the hardware serves the oldest waves first: with equal shares the first quarter finishes well before
        // the last (143 / 153 / 165 / 175 us measured on config 3).
        double skew[4] = {1, 1, 1, 1};
        if (const char *e = getenv("EMSAR_HIP_AGE_SKEW")) { double v[4]; if (sscanf(e, "%lf,%lf,%lf,%lf", v, v + 1, v + 2, v + 3) == 4 && v[0] > 0 && v[1] > 0 && v[2] > 0 && v[3] > 0) for (int i = 0; i < 4; i++) skew[i] = v[i]; }
        std::vector<double> cum((size_t)n_chunks + 1, 0.0);
        for (int64_t c = 0; c < n_chunks; c++) cum[(size_t)c + 1] = cum[(size_t)c] + skew[(size_t)(c * 4 / n_chunks)];
        int64_t s = 0;
        for (int64_t c = 0; c < n_chunks && s < n_slices; c++) {
            // slices [s, e): up to the point where the running work reaches its share of the total; at least one slice,
            // and enough left for the chunks to come
            const double target = total * cum[(size_t)c + 1] / cum[(size_t)n_chunks];
            const int64_t e_max = n_slices - (n_chunks - 1 - c);            // leave one slice for each chunk to come
            int64_t e = s + 1;
            while (e < e_max && prefix[(size_t)e] + work_of(st[(size_t)e]) / 2 <= target) e++;
            if (c == n_chunks - 1) e = n_slices;
            else {
                int64_t best = -1;
                for (int64_t b = e; b > s && prefix[(size_t)e] - prefix[(size_t)b] <= snap; b--) if (is_break[(size_t)b]) { best = b; break; }
                for (int64_t b = e + 1; b <= e_max && prefix[(size_t)b] - prefix[(size_t)e] <= snap; b++)
                    if (is_break[(size_t)b]) { if (best < 0 || prefix[(size_t)b] - prefix[(size_t)e] < prefix[(size_t)e] - prefix[(size_t)best]) best = b; break; }
                if (best > s) e = best;
            }
            ChunkDesc C;
            C.group_begin = (uint32_t)out.groups.size();
            // groups: consecutive slices whose near range plus explicit far tids fit one dictionary.  With sort blocks of 256
            // transcripts a chunk that touches two neighbouring blocks usually still fits one (2 x 256 + 2 x kFarReach < kTileDict):
            // most chunks are ONE group and their four waves never wait for each other before the chunk's end.  A chunk that does
            // not fit is cut where its two parts carry equal work if both parts fit (a greedy cut can leave ONE long-row slice
            // in a group of its own: one wave works for 70 us, three wait at the group's barrier), else greedily.
            auto dict_need = [&](int64_t a0, int64_t a1) -> int64_t {          // slots the slices [a0, a1) need in one dictionary
                int32_t lo = INT32_MAX, hi = -1;
                int64_t n_expl = 0;
                gid++;
                for (int64_t i = a0; i < a1; i++) {
                    const SliceTmp &S = st[(size_t)i];
                    if (S.nmax >= S.nmin) { lo = std::min(lo, S.nmin); hi = std::max(hi, S.nmax); }
                    for (int32_t t : S.expl) if (stamp[(size_t)t] != gid) { stamp[(size_t)t] = gid; n_expl++; }
                }
                return (hi >= lo ? (int64_t)hi - lo + 1 : 0) + n_expl;      // conservative: explicit tids inside the window counted too
            };
            std::vector<int64_t> cuts;                                        // group boundaries inside [s, e)
            cuts.push_back(s);
            if (dict_need(s, e) > kTileDict && e - s >= 2) {
                double half = 0, acc = 0;
                for (int64_t i = s; i < e; i++) half += work_of(st[(size_t)i]);
                half /= 2;
                int64_t mid = s + 1;
                for (int64_t i = s; i < e - 1; i++) { acc += work_of(st[(size_t)i]); mid = i + 1; if (acc >= half) break; }
                if (dict_need(s, mid) <= kTileDict && dict_need(mid, e) <= kTileDict) cuts.push_back(mid);
                else {                                                        // greedy: as many slices as fit, again and again
                    int64_t g0 = s;
                    while (g0 < e) {
                        int64_t g1 = g0 + 1;
                        while (g1 < e && dict_need(g0, g1 + 1) <= kTileDict) g1++;
                        if (g1 < e) cuts.push_back(g1);
                        g0 = g1;
                    }
                }
            }
            cuts.push_back(e);
            for (size_t ci = 0; ci + 1 < cuts.size(); ci++) {
                const int64_t g0 = cuts[ci], g1 = cuts[ci + 1];
                GroupDesc G;
                std::memset(&G, 0, sizeof G);
                int32_t lo = INT32_MAX, hi = -1;
                std::vector<int32_t> expl;
                gid++;
                for (int64_t i = g0; i < g1; i++) {
                    const SliceTmp &S = st[(size_t)i];
                    if (S.nmax >= S.nmin) { lo = std::min(lo, S.nmin); hi = std::max(hi, S.nmax); }
                    for (int32_t t : S.expl) if (stamp[(size_t)t] != gid) { stamp[(size_t)t] = gid; expl.push_back(t); }
                }
                // explicit far tids that fall inside the window after all need no slot of their own
                std::sort(expl.begin(), expl.end());
                if (hi < lo) { lo = expl.empty() ? 0 : expl[0]; hi = lo; }      // cannot happen (every row has its anchor), kept harmless
                G.lo = lo;
                G.near_n = (uint16_t)(hi - lo + 1);
                G.far_off = (uint32_t)out.far_tid.size();
                for (int32_t t : expl) if (t < lo || t > hi) out.far_tid.push_back(t);
                G.far_n = (uint16_t)(out.far_tid.size() - G.far_off);
                G.slice_begin = (uint32_t)g0; G.slice_end = (uint32_t)g1;
                out.groups.push_back(G);
                // the waves take a group's slices in descriptor order: heaviest first, so that the group ends on its lightest slices
                std::stable_sort(st.begin() + g0, st.begin() + g1, [&](const SliceTmp &x, const SliceTmp &y) { return work_of(x) > work_of(y); });
            }
            C.group_end = (uint32_t)out.groups.size();
            out.chunks.push_back(C);
            s = e;
        }
        if (out.far_tid.size() >= ((size_t)1 << 32)) return -1;
    }
    const auto tp4 = t_now();

    // ---- P3: encode ----
    // forward blocks, far blocks and row slots have known sizes (prefix sums over the slices): written in place by the pool.
    // Backward blocks and COO lists are built per range of groups in private buffers and joined in order.
    out.slices.resize((size_t)n_slices);
    {
        uint64_t fk = 0, fb = 0;
        for (int64_t i = 0; i < n_slices; i++) {
            SliceDesc &D = out.slices[(size_t)i];
            std::memset(&D, 0, sizeof D);
            const SliceTmp &S = st[(size_t)i];
            D.fwd_kib = (uint32_t)fk; D.far_blk = (uint32_t)fb;
            D.k = (uint16_t)S.k; D.nf = (uint16_t)((S.n_far + 63) / 64); D.n_rows = S.n_norm + S.n_far;
            fk += S.k; fb += D.nf;
            if (fk >= ((uint64_t)1 << 32)) return -1;
        }
        out.fwd.resize((size_t)fk * kSliceDwords);
        out.far_blk_tid.assign((size_t)fb * 64, -1);
        out.slot_row.resize((size_t)n_slices * kTileSliceRows);
        out.padded_slots = (int64_t)fk * kTileSliceRows;
    }
    struct EncPart { u32_vec bwd; std::vector<uint32_t> coo; std::vector<std::pair<int32_t, uint32_t>> farp; int64_t tiled = 0, far = 0, exported = 0; bool bad = false; };
    const int64_t n_groups = (int64_t)out.groups.size();
    const int n_parts = (int)std::max<int64_t>(1, std::min<int64_t>(n_groups, (int64_t)std::max(1, n_host) * 4));
    std::vector<EncPart> parts((size_t)n_parts);
    auto encode_groups = [&](int64_t gb, int64_t ge, EncPart &P) {
        std::vector<int32_t> loc((size_t)n_tx, 0);
        std::vector<uint32_t> pairs, sorted, ccount, fill, segs;
        for (int64_t g = gb; g < ge; g++) {
            const GroupDesc &G = out.groups[(size_t)g];
            const int32_t lo = G.lo, near_n = G.near_n;
            const int nd = (int)G.near_n + (int)G.far_n;
            for (uint32_t i = 0; i < G.far_n; i++) loc[(size_t)out.far_tid[(size_t)G.far_off + i]] = near_n + (int32_t)i;
            auto loc_of = [&](int32_t t) -> uint32_t { return (uint32_t)((t >= lo && t - lo < near_n) ? t - lo : loc[(size_t)t]); };
            const uint32_t zero_id = (uint32_t)nd, pad_row = (uint32_t)kTileSliceRows;
            const uint32_t zero_dword = zero_id | (zero_id << 10) | (zero_id << 20);
            for (uint32_t si = G.slice_begin; si < G.slice_end; si++) {
                SliceDesc &D = out.slices[(size_t)si];
                const SliceTmp &S = st[(size_t)si];
                uint32_t *fw = out.fwd.data() + (size_t)D.fwd_kib * kSliceDwords;
                std::fill(fw, fw + (size_t)D.k * kSliceDwords, zero_dword);
                int64_t *slots = out.slot_row.data() + (size_t)si * kTileSliceRows;
                std::fill(slots, slots + kTileSliceRows, (int64_t)-1);
                pairs.clear();
                uint32_t n_norm = 0, n_far = 0;
                // Rows with an exported entry sit in the first lanes of the top fields (block 0 = field 11, block 1 = field 10);
                // the other rows fill the remaining positions in sorted order, the free lanes of a far field included.
                const uint32_t far_total = S.n_far, far_b1 = far_total > 64 ? far_total - 64 : 0, far_b0 = far_total - far_b1;
                auto far_base = [&](uint32_t blk) { return (uint32_t)kTileSliceRows - 64 * (blk + 1); };
                auto norm_pos = [&](uint32_t i) {          // i-th row without an exported entry -> slice position
                    if (far_b1 > 0 && i >= far_base(1)) i += far_b1;                     // skip the taken lanes of field 10 ...
                    if (i >= far_base(0)) i += far_b0;                                   // ... and of field 11
                    return i;
                };
                for (int64_t i = S.begin; i < S.end; i++) {
                    const uint32_t r = perm[(size_t)i];
                    const uint64_t b = row_ptr[r], e = row_ptr[r + 1];
                    const int64_t ex = export_index(r);
                    const bool far_row = ex >= 0 && n_far < (uint32_t)(64 * kMaxFarBlocks);      // the rule of cut_slices
                    uint32_t p;
                    if (far_row) {
                        const uint32_t blk = n_far / 64, ln = n_far % 64;
                        p = far_base(blk) + ln;
                        const int32_t ft = col_idx[b + (uint64_t)ex];
                        out.far_blk_tid[((size_t)D.far_blk + blk) * 64 + ln] = ft;
                        P.farp.emplace_back(ft, (uint32_t)(((size_t)D.far_blk + blk) * 64 + ln));
                        n_far++; P.exported++;
                    } else p = norm_pos(n_norm++);
                    slots[p] = (int64_t)r;
                    const uint32_t fl = p & 63u, fi = p >> 6;
                    uint32_t c = 0;
                    for (uint64_t q = b; q < e; q++) {
                        if (far_row && (int64_t)(q - b) == ex) continue;
                        const uint32_t d = loc_of(col_idx[q]);
                        uint32_t *dw = &fw[(size_t)c * kSliceDwords + fl * 4 + fi / 3];
                        const int sh = 10 * (int)(fi % 3);
                        *dw = (*dw & ~(0x3FFu << sh)) | (d << sh);
                        pairs.push_back((d << 16) | p);
                        if ((int32_t)d >= near_n) P.far++;
                        c++;
                    }
                    P.tiled += (int64_t)(e - b);
                }
                // backward index: (column, row) pairs sorted by column
                ccount.assign((size_t)nd + 1, 0);
                for (uint32_t pr : pairs) ccount[(pr >> 16) + 1]++;
                for (int d = 0; d < nd; d++) ccount[(size_t)d + 1] += ccount[(size_t)d];
                sorted.resize(pairs.size());
                fill.assign(ccount.begin(), ccount.end() - 1);
                for (uint32_t pr : pairs) sorted[fill[pr >> 16]++] = pr;
                segs.clear();
                const size_t coo_before = P.coo.size();
                for (int d = 0; d < nd; d++) {
                    const uint32_t b = ccount[(size_t)d], e = ccount[(size_t)d + 1];
                    if (e - b < (uint32_t)dense_min) {
                        for (uint32_t q = b; q < e; q++) P.coo.push_back(((uint32_t)d << 16) | (sorted[q] & 0xFFFF));
                        continue;
                    }
                    for (uint32_t q = b; q < e; q += kSegRows) {
                        uint32_t seg[4] = {0, 0, 0, 0};
                        pack10(seg, 0, (uint32_t)d);
                        for (uint32_t j = 0; j < (uint32_t)kSegRows; j++) pack10(seg, 1 + (int)j, q + j < e ? (sorted[q + j] & 0xFFFF) : pad_row);
                        segs.insert(segs.end(), seg, seg + 4);
                    }
                }
                D.coo_off = (uint32_t)coo_before;                         // part-local; rebased when the parts are joined
                D.coo_n = (uint16_t)(P.coo.size() - coo_before);
                const int64_t nseg = (int64_t)segs.size() / 4;
                const int m = (int)((nseg + 63) / 64);
                D.m = (uint16_t)m;
                if ((uint32_t)m != S.m || D.coo_n != S.coo_n) P.bad = true;      // cut_slices counted the same columns: a mismatch is a builder bug
                D.bwd_kib = (uint32_t)(P.bwd.size() / kSliceDwords);      // part-local
                uint32_t empty[4] = {0, 0, 0, 0};                          // unused segment: zero column, padding rows
                pack10(empty, 0, zero_id);
                for (int j = 1; j < 12; j++) pack10(empty, j, pad_row);
                const size_t base = P.bwd.size();
                P.bwd.resize(base + (size_t)m * 64 * 4);
                for (int64_t gg = 0; gg < (int64_t)m * 64; gg++) {
                    // logical segment gg -> lane gg / m, unit gg % m ; physical int4 index (unit*64 + lane)
                    const int64_t lane = gg / m, unit = gg % m;
                    const size_t u0 = base + (size_t)((unit * 64 + lane) * 4);
                    const uint32_t *src = gg < nseg ? &segs[(size_t)gg * 4] : empty;
                    for (int w = 0; w < 4; w++) P.bwd[u0 + (size_t)w] = src[w];
                }
            }
        }
    };
    {
        std::atomic<int> next{0};
        run_on_threads((int)std::min<int64_t>(n_parts, std::max(1, n_host)), [&](int) {
            for (;;) {
                const int pi = next.fetch_add(1);
                if (pi >= n_parts) break;
                encode_groups(n_groups * pi / n_parts, n_groups * (pi + 1) / n_parts, parts[(size_t)pi]);
            }
        });
    }
    // join the parts: rebase the part-local backward / COO offsets, concatenate
    {
        std::vector<size_t> bb((size_t)n_parts + 1, 0), cb((size_t)n_parts + 1, 0), fp((size_t)n_parts + 1, 0);
        for (int pi = 0; pi < n_parts; pi++) {
            bb[(size_t)pi + 1] = bb[(size_t)pi] + parts[(size_t)pi].bwd.size();
            cb[(size_t)pi + 1] = cb[(size_t)pi] + parts[(size_t)pi].coo.size();
            fp[(size_t)pi + 1] = fp[(size_t)pi] + parts[(size_t)pi].farp.size();
        }
        for (const auto &P : parts) if (P.bad) return -5;
        if (bb[(size_t)n_parts] / kSliceDwords >= ((size_t)1 << 32) || cb[(size_t)n_parts] >= ((size_t)1 << 32) || fp[(size_t)n_parts] >= ((size_t)1 << 32)) return -1;
        out.bwd.resize(bb[(size_t)n_parts]);
        out.coo.resize(cb[(size_t)n_parts]);
        for (int pi = 0; pi < n_parts; pi++) {
            const int64_t gb = n_groups * pi / n_parts, ge = n_groups * (pi + 1) / n_parts;
            if (gb < ge)
                for (uint32_t si = out.groups[(size_t)gb].slice_begin; si < out.groups[(size_t)ge - 1].slice_end; si++) {
                    out.slices[(size_t)si].bwd_kib += (uint32_t)(bb[(size_t)pi] / kSliceDwords);
                    out.slices[(size_t)si].coo_off += (uint32_t)cb[(size_t)pi];
                }
            out.tiled_entries += parts[(size_t)pi].tiled; out.far_entries += parts[(size_t)pi].far; out.exported_entries += parts[(size_t)pi].exported;
        }
        std::atomic<int> next{0};
        run_on_threads((int)std::min<int64_t>(n_parts, std::max(1, n_host)), [&](int) {
            for (;;) {
                const int pi = next.fetch_add(1);
                if (pi >= n_parts) break;
                EncPart &P = parts[(size_t)pi];
                if (!P.bwd.empty()) memcpy(out.bwd.data() + bb[(size_t)pi], P.bwd.data(), P.bwd.size() * 4);
                if (!P.coo.empty()) memcpy(out.coo.data() + cb[(size_t)pi], P.coo.data(), P.coo.size() * 4);
                u32_vec().swap(P.bwd); std::vector<uint32_t>().swap(P.coo);
            }
        });
        out.coo_entries = (int64_t)out.coo.size();
        // exported entries by transcript: the parts in order are in slice order, a stable counting sort by tid keeps it
        out.far_ptr.assign((size_t)n_tx + 1, 0);
        for (const auto &P : parts) for (const auto &fe : P.farp) out.far_ptr[(size_t)fe.first + 1]++;
        for (int32_t t : out.pair_tid) out.far_ptr[(size_t)t + 1]++;
        if (fp[(size_t)n_parts] + out.pair_tid.size() >= ((size_t)1 << 32)) return -1;
        for (int32_t t = 0; t < n_tx; t++) out.far_ptr[(size_t)t + 1] += out.far_ptr[(size_t)t];
        out.n_exported = (int64_t)(fp[(size_t)n_parts] + out.pair_tid.size());
        if (out.far_blk_tid.size() + out.pair_row.size() >= ((size_t)1 << 32)) return -1;
        out.far_src.resize((size_t)out.n_exported);                 // a transcript's run: its slices' entries in slice order, then its pairs
        std::vector<uint32_t> fillp(out.far_ptr.begin(), out.far_ptr.end() - 1);
        for (const auto &P : parts) for (const auto &fe : P.farp) out.far_src[fillp[(size_t)fe.first]++] = fe.second;
        const uint32_t pair_base = (uint32_t)out.far_blk_tid.size();
        for (size_t i = 0; i < out.pair_tid.size(); i++) out.far_src[fillp[(size_t)out.pair_tid[i]]++] = pair_base + (uint32_t)(i / 2);
    }
    if (const char *e = getenv("EMSAR_HIP_CHUNK_REVERSE")) if (atoi(e)) std::reverse(out.chunks.begin(), out.chunks.end());   // experiment: does a workgroup's speed follow its index or its chunk?
    if (dbg_t) fprintf(stderr, "build_tiled: classify %.0f ms, sort %.0f ms, slices %.0f ms, chunks + groups %.0f ms, encode %.0f ms on %d thread(s); %lld slices, %lld groups, %lld chunks, %lld exported far entries, %lld explicit\n",
                       t_ms(tp0, tp1), t_ms(tp1, tp2), t_ms(tp2, tp3), t_ms(tp3, tp4), t_ms(tp4, t_now()), n_host, (long long)n_slices, (long long)out.groups.size(),
                       (long long)out.chunks.size(), (long long)out.exported_entries, (long long)out.far_entries);
    return check_tiled_extents(out);       // O(slices): no descriptor may point outside the arrays that are uploaded next
}

// Decode and compare with the input (host self-check, used by the CPU tests). 0 = identical.
inline int check_tiled(const TiledLayout &L, const uint64_t *row_ptr, const int32_t *col_idx) {
    if (int rc = check_tiled_extents(L)) return rc;
    std::vector<uint8_t> seen((size_t)L.n_rows, 0);
    for (size_t i = 0; i < L.single_row.size(); i++) {
        uint32_t r = L.single_row[i];
        if (seen[r] || row_ptr[r + 1] - row_ptr[r] != 1 || col_idx[row_ptr[r]] != L.single_tid[i]) return -1;
        seen[r] = 1;
    }
    for (size_t i = 0; i < L.left_row.size(); i++) {
        uint32_t r = L.left_row[i];
        uint64_t n = L.left_ptr[i + 1] - L.left_ptr[i];
        if (seen[r] || row_ptr[r + 1] - row_ptr[r] != n) return -2;
        std::vector<int32_t> x(col_idx + row_ptr[r], col_idx + row_ptr[r + 1]), y(L.left_col.begin() + (int64_t)L.left_ptr[i], L.left_col.begin() + (int64_t)L.left_ptr[i + 1]);
        if (L.merged) std::sort(x.begin(), x.end());          // the merged view keeps every row's tids sorted
        if (x != y) return -2;
        seen[r] = 1;
    }
    // exported entries: far_ptr / far_src is the transpose of the far blocks and the pairs -- every (block, lane) that holds a tid is
    // named exactly once, in the run of that tid; every pair exactly twice, once in the run of each of its transcripts
    {
        const size_t nb = L.far_blk_tid.size();
        std::vector<uint8_t> hit(nb + L.pair_row.size(), 0);
        for (int32_t t = 0; t < L.n_tx; t++)
            for (uint32_t q = L.far_ptr[(size_t)t]; q < L.far_ptr[(size_t)t + 1]; q++) {
                const uint32_t x = L.far_src[q];
                if (x < nb) { if (hit[x] || L.far_blk_tid[x] != t) return -11; hit[x] = 1; }
                else {
                    const size_t i = x - nb;
                    if (hit[x] >= 2 || (L.pair_tid[2 * i] != t && L.pair_tid[2 * i + 1] != t)) return -11;
                    hit[x]++;
                }
            }
        for (size_t i = 0; i < nb; i++) if ((hit[i] != 0) != (L.far_blk_tid[i] >= 0)) return -11;
        for (size_t i = 0; i < L.pair_row.size(); i++) if (hit[nb + i] != 2) return -11;
    }
    for (size_t i = 0; i < L.pair_row.size(); i++) {
        const int64_t r = L.pair_row[i];
        int32_t pa = L.pair_tid[2 * i], pb = L.pair_tid[2 * i + 1];
        if (pa > pb) std::swap(pa, pb);
        auto same = [&](uint32_t o) {
            if (row_ptr[o + 1] - row_ptr[o] != 2) return false;
            int32_t x = col_idx[row_ptr[o]], y = col_idx[row_ptr[o] + 1];
            if (x > y) std::swap(x, y);
            return x == pa && y == pb;
        };
        if (L.merged) {
            if (L.mem_ptr[(size_t)r + 1] == L.mem_ptr[(size_t)r]) return -13;
            for (uint64_t q = L.mem_ptr[(size_t)r]; q < L.mem_ptr[(size_t)r + 1]; q++) { const uint32_t o = L.mem_row[(size_t)q]; if (seen[o] || !same(o)) return -13; seen[o] = 1; }
        } else { if (seen[(size_t)r] || !same((uint32_t)r)) return -13; seen[(size_t)r] = 1; }
    }
    std::vector<int32_t> a, b;
    std::vector<uint32_t> pf, pb;
    std::vector<uint8_t> slice_seen(L.slices.size(), 0);
    for (size_t ci = 0; ci < L.chunks.size(); ci++) {
        if (L.chunks[ci].group_begin != (ci ? L.chunks[ci - 1].group_end : 0u)) return -12;          // chunks tile the groups
        for (uint32_t g = L.chunks[ci].group_begin; g < L.chunks[ci].group_end; g++) {
            const GroupDesc &G = L.groups[g];
            if (G.slice_begin != (g ? L.groups[g - 1].slice_end : 0u)) return -12;                     // groups tile the slices
            const int nd = G.near_n + G.far_n;
            auto tid_of = [&](int d) { return d < G.near_n ? G.lo + d : L.far_tid[(size_t)G.far_off + (size_t)(d - G.near_n)]; };
            for (uint32_t si = G.slice_begin; si < G.slice_end; si++) {
                const SliceDesc &D = L.slices[si];
                slice_seen[si] = 1;
                pf.clear(); pb.clear();
                const size_t foff = (size_t)D.fwd_kib * kSliceDwords;
                uint32_t rows_found = 0;
                for (int i = 0; i < kTileSliceRows; i++) {
                    const int64_t r = L.slot_row[(size_t)si * kTileSliceRows + (size_t)i];
                    a.clear();
                    const int fl = i & 63, fi = i >> 6;
                    for (int j = 0; j < D.k; j++) {
                        const int d = (int)((L.fwd[foff + (size_t)j * kSliceDwords + (size_t)(fl * 4 + fi / 3)] >> (10 * (fi % 3))) & 0x3FFu);
                        if (d > nd) return -4;
                        if (d == nd) continue;                        // zero slot = padding
                        a.push_back(tid_of(d));
                        pf.push_back(((uint32_t)d << 16) | (uint32_t)i);
                    }
                    // the exported entry of a row in a far field
                    const int blk = 11 - fi;
                    if (blk < (int)D.nf) {
                        const int32_t ft = L.far_blk_tid[((size_t)D.far_blk + (size_t)blk) * 64 + (size_t)fl];
                        if (ft >= 0) { if (r < 0) return -5; a.push_back(ft); }
                    }
                    if (r < 0) { if (!a.empty()) return -5; continue; }
                    rows_found++;
                    std::sort(a.begin(), a.end());
                    if (L.merged) {                                   // every member row has this tid multiset
                        if (L.mem_ptr[(size_t)r + 1] == L.mem_ptr[(size_t)r]) return -6;
                        for (uint64_t q = L.mem_ptr[(size_t)r]; q < L.mem_ptr[(size_t)r + 1]; q++) {
                            uint32_t o = L.mem_row[(size_t)q];
                            if (seen[o]) return -6;
                            seen[o] = 1;
                            b.assign(col_idx + row_ptr[o], col_idx + row_ptr[o + 1]);
                            std::sort(b.begin(), b.end());
                            if (a != b) return -7;
                        }
                        continue;
                    }
                    if (seen[(size_t)r]) return -6;
                    seen[(size_t)r] = 1;
                    b.assign(col_idx + row_ptr[r], col_idx + row_ptr[r + 1]);
                    std::sort(b.begin(), b.end());
                    if (a != b) return -7;
                }
                if (rows_found != D.n_rows) return -5;
                const size_t boff = (size_t)D.bwd_kib * kSliceDwords;
                const int m = D.m;
                for (int64_t gg = 0; gg < (int64_t)m * 64; gg++) {
                    const int64_t lane = gg / m, unit = gg % m;
                    const uint32_t *q = &L.bwd[boff + (size_t)((unit * 64 + lane) * 4)];
                    const uint32_t d = unpack10(q, 0);
                    for (int w = 1; w < 12; w++) {
                        const uint32_t rl = unpack10(q, w);
                        if (rl == (uint32_t)kTileSliceRows) continue;
                        if (rl > (uint32_t)kTileSliceRows || d >= (uint32_t)nd) return -8;
                        pb.push_back((d << 16) | rl);
                    }
                }
                for (uint32_t q = 0; q < D.coo_n; q++) pb.push_back(L.coo[(size_t)D.coo_off + q]);
                std::sort(pf.begin(), pf.end()); std::sort(pb.begin(), pb.end());
                if (pf != pb) return -9;                                // the backward index is the transpose of the forward one
            }
        }
    }
    for (uint8_t x : slice_seen) if (!x) return -12;
    for (int64_t r = 0; r < L.n_rows; r++)
        if (!seen[(size_t)r] && row_ptr[r + 1] != row_ptr[r]) return -10;
    return 0;
}

}  // namespace emsar
