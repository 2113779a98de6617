// layout_tiled.hpp -- host-side construction of the TILED HBM layout (pure C++, no HIP calls).
//
// Goal: a pass whose inner loops touch only LDS with 16-bit operands, with no atomics on the hot path and no
// workgroup barrier between the E-step and the M-step.
//
//   * rows with ONE tid never reach the kernel: they are folded into a per-transcript count vector u
//     (acc_t += u_t / theta_t is applied analytically in k_update);
//   * rows with 2..kMaxRowLen tids are sorted by (block(anchor tid), class of their number of entries, anchor tid), anchor = median
//     tid, and cut into TILES of at most 3072 rows whose distinct tids fit a 360-slot tile-local DICTIONARY: the contiguous range
//     [lo, lo+near_n) that covers most of the tile's tids plus an explicit list of far tids.  The dictionary is organised in blocks
//     of three slots and every stored operand is a 10-BIT ENTRY = block * 8 + subset (see kBlk below), three to a dword; a far
//     tid has an entry of its own after the near blocks (kFarMax);
//   * a tile has up to 4 SLICES of 768 rows; one wavefront owns one slice for the whole pass; two consecutive tiles may share a
//     dictionary (a unit, Tile::follows):
//       forward index (E-step, row sums): column-major [k][768]; lane l owns rows l, l+64, ..., so column j of its
//         rows is ONE int4 of twelve 10-bit entries; padding is entry 0, the empty subset (no branches);
//       backward index (M-step, column sums) OF THE SAME 768 ROWS: for every entry value that occurs in the slice, its
//         slice-local rows cut into segments of 11 row ids headed by the entry value (12 x 10 bit = one int4).  Segments are
//         dealt to lanes in contiguous entry order (a lane keeps the running sum of a column in a register), but stored
//         interleaved so that wave loads stay 1 KiB contiguous.  (kDenseMin > 1 sends rarer entries to a COO list of
//         (entry, row) pairs: off, it costs more than a mostly empty segment.)
//     Because the transposed index is per slice, the wave that computed w_r for its 768 rows is the only consumer
//     of them: E-step and M-step need no workgroup barrier in between;
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
#include "renumber.hpp"

namespace emsar {

constexpr int kTileSlices = 4;         // wavefronts per workgroup
constexpr int kRowsPerLane = 12;       // twelve 10-bit ids per int4
constexpr int kTileSliceRows = 64 * kRowsPerLane;   // 768
constexpr int kTileRows = kTileSlices * kTileSliceRows;
// The dictionary of a tile is organised in BLOCKS of kBlk consecutive slots; a stored operand does not name one slot but a block
// and a SUBSET of it: entry = block * 8 + mask.  The kernel's LDS table holds, for every block, the 8 subset sums of its three
// theta values (T[block*8 + mask]), so ONE gather serves every transcript of the block that a row hits -- isoforms of a gene are
// neighbours in tid order and reads hit several of them: 2.5 ids per entry on config 3, i.e. 2.5x fewer index bytes and LDS
// gathers for the same matrix.  The M-step accumulates per entry value and folds the 8 words of a block into its 3 transcripts
// when the tile is flushed.  Entry 0 (block 0, empty subset) is the padding: T[0] = 0.
#ifndef EMSAR_BLK
#define EMSAR_BLK 4
#endif
constexpr int kBlk = EMSAR_BLK;                  // slots per block (3: 120 blocks x 8 sums; 4: 60 blocks x 16 sums -- both 960 table entries).
                                                 // Round 2 (sort block 128): 4 packs 3.1 tids per entry instead of 2.5, but a tile then holds 240
                                                 // transcripts instead of 360 and there were 10.9 k tiles instead of 8.8 k: 0.140 against 0.117 ms.
                                                 // Round 3: with the narrow sort block (32-48 tids) a unit's rows fit 240 transcripts, and 4 wins on
                                                 // everything large -- config 3 family law 0.1150 -> 0.1060 ms (1.59 -> 1.82 ids per entry, 4616 ->
                                                 // 3517 units, 486 -> 443 MB), window law 0.1017 -> 0.0969 (2.51 -> 2.95, 382 -> 340 MB), config 5
                                                 // x 0.25 0.2463 -> 0.2195 (1147 -> 926 MB); config 2 (705 units, latency-bound) 0.0190 -> 0.0208.
                                                 // 5 slots (32 sums, 150 transcripts per dictionary): far entries x 6, 51 MB instead of 44 at 1/10 size
constexpr int kBlkEntries = 1 << kBlk;           // subset sums per block
constexpr int kDictBlocks = 960 / kBlkEntries;
constexpr int kDictEntries = kDictBlocks * kBlkEntries;   // 960 doubles of LDS for T, 960 for the per-entry accumulators
constexpr int kTileDict = kBlk * kDictBlocks;    // 360 transcripts per tile
constexpr int64_t kFragRows = 1 << 21;   // sorted rows per independently tiled fragment (build_tiled)
// Far transcripts (outside the tile's contiguous near range: cross-family hits, one or two rows each) do not get a slot of a block:
// each has ONE table entry of its own after the near blocks (entry 8 nb + i names far transcript i; T = its theta).  A tile's
// dictionary fits when 8 * ceil(near_n / 3) + far_n <= 960 and far_n <= kFarMax.
constexpr int kFarMax = 3 * (256 - kDictBlocks);   // far entries are fetched and flushed by the threads that own no near block, 3 each
constexpr int kTileDistinct = 700;               // distinct transcripts gathered before the dictionary is laid out (rows are given back if it does not fit)
inline int near_blocks(int near_n) { return (near_n + kBlk - 1) / kBlk; }
inline bool dict_fits(int64_t near_n, int64_t far_n) {
    return far_n <= kFarMax && (int64_t)kBlkEntries * ((near_n + kBlk - 1) / kBlk) + far_n <= kDictEntries;
}
constexpr int kTileTids = kTileDict - (kBlk - 1);   // distinct transcripts of a tile: its near range starts at a multiple of kBlk, which may cost kBlk - 1 slots
constexpr int kMaxRowLen = kTileTids;  // longer rows -> leftover CSR (a row must fit one dictionary)
constexpr int kSegRows = 11;           // row ids per backward segment (plus 1 header = 12 x 10 bit = one int4)
constexpr int kSliceDwords = kTileSliceRows / 3;    // dwords per forward column of a slice (256 = 1 KiB)

// field i (0..11) of a packed int4: dword i/3, bits 10*(i%3) .. +10
inline void pack10(uint32_t *q, int i, uint32_t id) { q[i / 3] |= (id & 0x3FFu) << (10 * (i % 3)); }
inline uint32_t unpack10(const uint32_t *q, int i) { return (q[i / 3] >> (10 * (i % 3))) & 0x3FFu; }
#ifndef EMSAR_SWZ
#define EMSAR_SWZ 0
#endif
// Optional bank swizzle of the table: the subset of block b is stored as mask ^ (b & 7), so that the same subset of blocks four
// apart (which share LDS banks) lands on different banks.  Entry 0 stays the padding (block 0 is not swizzled).
// Measured on config 3 (-DEMSAR_SWZ=1): 0.1164 ms with and without -- the gathers are not what the waves wait for.  Off.
inline uint32_t entry_swz(uint32_t b) { return EMSAR_SWZ ? (b & (uint32_t)(kBlkEntries - 1)) : 0u; }
inline uint32_t entry_code(uint32_t b, uint32_t m) { return b * (uint32_t)kBlkEntries + (m ^ entry_swz(b)); }
inline uint32_t entry_mask(uint32_t e) { return (e & (uint32_t)(kBlkEntries - 1)) ^ entry_swz(e >> kBlk); }
constexpr int kUnitMaxTiles = 4;       // tiles that may share one dictionary
constexpr int kDenseMin = 1;           // columns with fewer entries in a slice would use a COO list; MUST stay 1, the kernels read none (with block entries the COO path --
                                       // a load, an LDS read and an LDS atomic per entry -- costs more than a mostly empty segment: 4 / 2 / 1 -> 0.139 / 0.133 / 0.127 ms)
// the entries (block * 8 + mask) of one row from its dictionary slots (any order; a slot that occurs twice -- an internal repeat
// of the transcript -- opens a second entry of the same block: a subset holds a transcript once)
inline void slots_to_entries(std::vector<uint32_t> &slots, std::vector<uint32_t> &ent, uint32_t near_n) {
    std::sort(slots.begin(), slots.end());
    ent.clear();
    int cur_b = -1;
    uint32_t cur_m = 0;
    const uint32_t far_base = (uint32_t)kBlkEntries * (uint32_t)near_blocks((int)near_n);
    for (uint32_t sl : slots) {
        if (sl >= near_n) {                       // a far transcript: its own entry (slots are sorted: the far ones come last)
            if (cur_b >= 0) { ent.push_back(entry_code((uint32_t)cur_b, cur_m)); cur_b = -1; cur_m = 0; }
            ent.push_back(far_base + (sl - near_n));
            continue;
        }
        const int b = (int)(sl / (uint32_t)kBlk);
        const uint32_t bit = 1u << (sl % (uint32_t)kBlk);
        if (b != cur_b || (cur_m & bit)) {
            if (cur_b >= 0) ent.push_back(entry_code((uint32_t)cur_b, cur_m));
            cur_b = b; cur_m = bit;
        } else cur_m |= bit;
    }
    if (cur_b >= 0) ent.push_back(entry_code((uint32_t)cur_b, cur_m));
}
constexpr int64_t kTileEntries = 65536;

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

struct Tile {                // 64 bytes
    uint64_t fwd_off;        // byte offset into fwd (multiple of 1024)
    uint64_t bwd_off;        // byte offset into bwd (multiple of 1024)
    uint32_t row_base;       // first row slot of the tile (multiple of 768); slot = row_base + slice*768 + 64*i + lane
    uint32_t far_off;        // index of the tile's first far tid in far_tid[]
    uint32_t coo_off;        // index of the tile's first pair in coo[]
    int32_t lo;              // dictionary slot d < near_n  <->  tid lo + d
    uint16_t near_n, far_n;  // slot near_n + i <-> far_tid[far_off + i]; slots near_n + far_n .. 359 are empty (theta 0)
    uint16_t n_slices;       // <= 4
    uint8_t follows;         // 1: this tile uses the dictionary of the tile before it (a unit of up to 8 slices)
    uint8_t wave_of;         // in a unit: slice s belongs to wave (wave_of >> 2s) & 3 of the workgroup (k_pass_tiled_unit; distinct waves)
    uint16_t k[4];           // padded row length of each slice's forward index
    uint16_t m[4];           // backward segments (int4) per lane of each slice
    uint16_t coo_n[4];       // COO pairs of each slice
};
static_assert(sizeof(Tile) == 64, "Tile must stay 64 bytes");

struct TiledLayout {
    int64_t n_rows = 0, nnz = 0;
    int32_t n_tx = 0;
    // singletons: folded rows
    std::vector<uint32_t> single_row;   // original row index
    std::vector<int32_t> single_tid;
    // tiles
    std::vector<Tile> tiles;
    std::vector<uint32_t> unit_first;   // n_units + 1: the tiles of unit u are unit_first[u] .. unit_first[u+1] (one or two; the second follows)
    i64_vec slot_row;                   // row slot -> original row, or merged-row id when `merged` (-1 = padding)
    bool merged = false;                // identical rows were merged: a slot stands for mem_row[mem_ptr[id] .. mem_ptr[id+1])
    std::vector<uint64_t> mem_ptr;
    std::vector<uint32_t> mem_row;
    u32_vec fwd, bwd;                   // packed 10-bit ids
    std::vector<uint32_t> coo;          // (col_id << 16) | row_id
    std::vector<int32_t> far_tid;
    // leftover rows (too long for a tile): plain CSR + original row ids
    std::vector<uint64_t> left_ptr;
    std::vector<int32_t> left_col;
    std::vector<uint32_t> left_row;
    int64_t tiled_entries = 0, far_entries = 0, coo_entries = 0, n_fslices = 0, padded_slots = 0, tiled_ids = 0;
    // the library's transcript numbering (renumber.hpp): new_of_old[t] = id under which the caller's transcript t is stored; empty = the
    // caller's numbering.  Everything in this structure (single_tid, far_tid, Tile::lo, left_col) is in the LIBRARY's numbering.
    std::vector<int32_t> new_of_old;
    RenumberStats renum;
    int64_t n_slots() const { return (int64_t)slot_row.size(); }
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

// Every address the pass kernels form from a tile descriptor, checked against the arrays as they are uploaded (the device arrays
// have exactly these sizes).  The kernels read, per slice s < n_slices of a tile: k[s] forward columns of 256 dwords, m[s] backward
// units of 64 int4, coo_n[s] COO words; far_n far tids; theta / acc at lo .. lo + near_n and at the far tids; the row weights and row
// values at the 768 slots of the slice.  Every stored 10-bit entry must name a block and a subset whose slots exist (< near_n + far_n), every
// stored row id a row of the slice (<= 768, the padding row).  0 = fine, -20 .. -29 = the first descriptor that is not.
// Runs at the end of build_tiled (an upload never happens with a bad descriptor) and again in emsar_hip_layout_selfcheck_tiled.
inline int check_tiled_extents(const TiledLayout &L) {
    const size_t nf = L.fwd.size(), nb = L.bwd.size(), nc = L.coo.size(), nfar = L.far_tid.size(), nslot = L.slot_row.size();
    if (nslot % (size_t)kTileSliceRows) return -20;
    for (int32_t t : L.far_tid) if (t < 0 || t >= L.n_tx) return -21;
    std::vector<uint8_t> slice_used(nslot / (size_t)kTileSliceRows, 0);
    for (const Tile &T : L.tiles) {
        const int nd = (int)T.near_n + (int)T.far_n;
        if (T.n_slices < 1 || T.n_slices > kTileSlices || T.near_n < 1 || !dict_fits(T.near_n, T.far_n)) return -22;
        if (T.lo < 0 || (int64_t)T.lo + T.near_n > (int64_t)L.n_tx) return -23;
        if ((size_t)T.far_off + T.far_n > nfar) return -24;
        if (T.row_base % (uint32_t)kTileSliceRows || (size_t)T.row_base + (size_t)T.n_slices * kTileSliceRows > nslot) return -25;
        if (T.fwd_off % 1024 || T.bwd_off % 1024) return -26;
        size_t foff = (size_t)(T.fwd_off / 4), boff = (size_t)(T.bwd_off / 4), coff = T.coo_off;
        for (int s = 0; s < kTileSlices; s++) {
            if (s >= T.n_slices) { if (T.k[s] || T.m[s] || T.coo_n[s]) return -27; continue; }
            if (T.coo_n[s]) return -27;                          // the kernels no longer read a COO list (kDenseMin = 1: the builder makes none)
            uint8_t &u = slice_used[(size_t)T.row_base / (size_t)kTileSliceRows + (size_t)s];
            if (u) return -25;                                                  // two tiles on one slice of row slots
            u = 1;
            if (T.k[s] > kMaxRowLen) return -27;
            if (T.k[s] < 1 || T.m[s] < 1) return -27;            // a slice has rows and a row has entries: the kernels request column 0 and segment 0 of every slice without asking
            const size_t fw = (size_t)T.k[s] * kSliceDwords, bw = (size_t)T.m[s] * 64 * 4;
            if (foff + fw > nf || boff + bw > nb || coff + T.coo_n[s] > nc) return -28;
            // an entry names a block and a subset of it: every slot of the subset must exist in the tile's dictionary
            const uint32_t far_base = (uint32_t)kBlkEntries * (uint32_t)near_blocks(T.near_n);
            auto entry_ok = [&](uint32_t e) {
                if (e >= far_base) return e - far_base < (uint32_t)T.far_n;        // a far transcript's own entry
                const uint32_t m = entry_mask(e), b = e >> kBlk;
                for (int i = 0; i < kBlk; i++) if ((m >> i & 1u) && b * (uint32_t)kBlk + (uint32_t)i >= (uint32_t)T.near_n) return false;
                return true;
            };
            auto entry_empty = [&](uint32_t e) { return e < far_base && entry_mask(e) == 0; };
            for (size_t i = 0; i < fw; i++) {
                const uint32_t d = L.fwd[foff + i];
                if (!entry_ok(d & 0x3FFu) || !entry_ok((d >> 10) & 0x3FFu) || !entry_ok((d >> 20) & 0x3FFu)) return -29;
            }
            for (size_t i = 0; i < bw; i += 4) {
                const uint32_t *q = &L.bwd[boff + i];
                bool any = false;
                for (int w = 1; w < 12; w++) { const uint32_t rl = unpack10(q, w); if (rl > (uint32_t)kTileSliceRows) return -29; any |= rl != (uint32_t)kTileSliceRows; }
                const uint32_t e = unpack10(q, 0);
                if (!entry_ok(e) || (any && entry_empty(e))) return -29;
            }
            for (size_t i = 0; i < T.coo_n[s]; i++) {
                const uint32_t p = L.coo[coff + i];
                if (!entry_ok(p >> 16) || entry_empty(p >> 16) || (p & 0xFFFFu) >= (uint32_t)kTileSliceRows) return -29;
            }
            foff += fw; boff += bw; coff += T.coo_n[s];
        }
    }
    for (size_t i = 0; i < L.tiles.size(); i++) {            // a tile that follows shares the dictionary of the tile before it
        const Tile &T = L.tiles[i];
        if (T.follows > 1 || (T.follows && i == 0)) return -30;
        for (int a = 0; a < T.n_slices; a++)
            for (int b = a + 1; b < T.n_slices; b++)
                if (((T.wave_of >> (2 * a)) & 3) == ((T.wave_of >> (2 * b)) & 3)) return -32;      // two slices of a tile on one wave
        if (T.follows) {
            const Tile &P = L.tiles[i - 1];
            if (P.lo != T.lo || P.near_n != T.near_n || P.far_n != T.far_n || P.far_off != T.far_off) return -30;
        }
    }
    if (!L.tiles.empty()) {
        if (L.unit_first.size() < 2 || L.unit_first.front() != 0 || L.unit_first.back() != L.tiles.size()) return -31;
        for (size_t u = 0; u + 1 < L.unit_first.size(); u++) {
            const uint32_t a = L.unit_first[u], b = L.unit_first[u + 1];
            if (b <= a || b - a > (uint32_t)kUnitMaxTiles || b > L.tiles.size() || L.tiles[a].follows) return -31;
            for (uint32_t q = a + 1; q < b; q++) if (!L.tiles[q].follows) return -31;
        }
    }
    for (int64_t r : L.slot_row) if (r < -1 || r >= (L.merged ? (int64_t)L.mem_ptr.size() - 1 : L.n_rows)) return -20;
    if (L.left_ptr.size() != L.left_row.size() + 1 || (!L.left_ptr.empty() && L.left_ptr.back() != L.left_col.size())) return -20;
    for (int32_t t : L.left_col) if (t < 0 || t >= L.n_tx) return -21;
    return 0;
}

// With merge_rows, rows with the same tid multiset (2..kMaxRowLen tids) are stored once and weighted by the sum of
// their members' weights -- what the reference's update_ReadCounts does when it counts reads per segment
// (emsar_functions.c:838-943).  Every quantity the library computes is a sum over rows of a function of the row's tid
// set times a per-row weight, so the merge is exact up to summation order.
inline int build_tiled(int64_t n_rows, int32_t n_tx, const uint64_t *row_ptr_in, const int32_t *col_idx_in, TiledLayout &out,
                       bool merge_rows = false, bool renumber = true) {
    if (n_rows >= (int64_t)1 << 32) return -1;
    out = TiledLayout();
    const uint64_t *row_ptr = row_ptr_in;
    const int32_t *col_idx = col_idx_in;
    if (renumber) cooccurrence_order(n_rows, n_tx, row_ptr_in, col_idx_in, kMaxRowLen, kBlk, out.new_of_old, out.renum);
    // the id under which entry k of the caller's col_idx is stored
    const int32_t *tid_map = out.new_of_old.empty() ? nullptr : out.new_of_old.data();
    auto tid_at = [&](uint64_t k) -> int32_t { return tid_map ? tid_map[col_idx[k]] : col_idx[k]; };
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
            for (uint64_t k = b; k < e; k++) scol[(size_t)k] = tid_at(k);
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
        tid_map = nullptr;                      // the merged view is in the library's numbering already
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
    std::vector<int32_t> mintid((size_t)n_rows, -1);      // the anchor tid of every tiled row (see below)
    std::vector<uint16_t> ecnt((size_t)n_rows, 0);        // its number of block entries (dictionaries start at multiples of kBlk, so
                                                          // a near tid t lies in block t / kBlk whatever the tile): the sort's length
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
            tmp.resize((size_t)len);
            for (uint64_t k = b; k < e; k++) tmp[(size_t)(k - b)] = tid_at(k);
            std::sort(tmp.begin(), tmp.end());
            mintid[(size_t)r] = anchor_median ? tmp[(size_t)(len / 2)] : tmp[0];
            int n_ent = 0, cur_b = -1;
            uint32_t cur_m = 0;
            for (int32_t t : tmp) {                        // the rule of slots_to_entries, on tids
                const int bb = t / kBlk;
                const uint32_t bit = 1u << (t % kBlk);
                if (bb != cur_b || (cur_m & bit)) { n_ent++; cur_b = bb; cur_m = bit; } else cur_m |= bit;
            }
            ecnt[(size_t)r] = (uint16_t)n_ent;
        }
    });
    std::vector<uint32_t> act;                             // the tiled rows, ascending
    {
        // every thread classifies a contiguous range of rows into lists of its own; the lists are joined in range order
        struct Part { std::vector<uint32_t> single_row, left_row, act; std::vector<int32_t> single_tid, left_col; std::vector<uint64_t> left_len; };
        std::vector<Part> part((size_t)std::max(1, n_host));
        const int np = par_ranges(n_rows, [&](int t, int64_t lo, int64_t hi) {
            Part &P = part[(size_t)t];
            for (int64_t r = lo; r < hi; r++) {
                const uint64_t b = row_ptr[r], e = row_ptr[r + 1], len = e - b;
                if (len == 0) continue;
                const uint32_t r_orig = merge_rows ? orig_of_merged[(size_t)r] : (uint32_t)r;   // singles / long rows are never merged
                if (len == 1) { P.single_row.push_back(r_orig); P.single_tid.push_back(tid_at(b)); continue; }
                if (len > (uint64_t)kMaxRowLen) {
                    P.left_row.push_back(r_orig);
                    for (uint64_t k = b; k < e; k++) P.left_col.push_back(tid_at(k));
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
            P = Part();
        }
    }
    const int64_t n_act = (int64_t)act.size();
    // Sort granularity in tid space.  A tile's dictionary must hold a block's tid range plus the rows' reach, so
    // blocks stay small; wide blocks keep the (block, length) buckets large, i.e. the slices uniform.
    int32_t block = 32;        // with 4-slot blocks (kBlk): 24 / 32 / 48 / 64 / 96 -> family law 0.1063 / 0.1060 / 0.1071 / 0.1089 / 0.1146, window law 0.0967 /
                               // 0.0969 / 0.0980 / 0.1030 / 0.1014, config 5 x 0.25 - / 0.2195 / 0.2204 / 0.2192 / 0.2216 ms (gpurun_out/sweep_blk4).
                               // With 3-slot blocks: 16 / 24 / 32 / 48 / 64 / 96 / 128 -> family law 0.1122 / 0.1126 / 0.1128 / 0.1141 / 0.1152 / 0.1205 / 0.1236,
                               // window law 0.1017 / 0.1012 / 0.1015 / 0.1015 / 0.1024 / 0.1042 / 0.1059, config 5 x 0.25 - / - / 0.2469 / 0.2376 / 0.2439 /
                               // 0.2442 / 0.2456 ms per pass (gpurun_out/sweep2,3): narrow blocks close FEWER units (4.4 k instead of 5.2-5.4 k on config 3 --
                               // the rows of a unit share more of their transcripts) although the slices are less uniform (more padding, more stored
                               // bytes): per-unit latency, not bytes, is what the pass pays for.  Before (rounds 1-2, 128):
                               // the rows of one block of the sort fit one dictionary (360 transcripts) with room for their far hits;
                               // config 3, rows sorted by entry count inside a block: 96 / 128 / 160 / 192 tids -> 0.1178 / 0.1169 / 0.1187 / 0.124 ms per pass
                               // end of round 2 (spill-free unit kernel): 96 / 104 / 112 / 120 / 128 / 144 / 160 -> 0.1052 / 0.1056 / 0.1055 / 0.1066 / 0.1063 /
                               // 0.1072 / 0.1067 on config 3, but config 5 (20 transcripts per read) 0.954 ms at 112 against 0.940 at 128: stays 128
    if (const char *e = getenv("EMSAR_HIP_TILE_BLOCK")) { int v = atoi(e); if (v >= 16 && v <= 900) block = v; }
    int short_ecnt = 0; int32_t short_block = 512;      // rows of <= short_ecnt entries: sort block short_block (0 = no such class)
    if (const char *e = getenv("EMSAR_HIP_SHORT_ECNT")) { int v = atoi(e); if (v >= 0 && v <= 32) short_ecnt = v; }
    if (const char *e = getenv("EMSAR_HIP_SHORT_BLOCK")) { int v = atoi(e); if (v >= block && v <= (1 << 20)) short_block = v; }
    if (short_block < block) short_block = block;
    int64_t tile_rows = kTileRows;
    int unit_tiles = 2;                 // tiles that may share one dictionary (a unit: one workgroup, one dictionary load, one flush)
    if (const char *e = getenv("EMSAR_HIP_UNIT_TILES")) { int v = atoi(e); if (v >= 1 && v <= kUnitMaxTiles) unit_tiles = v; }
    if (const char *e = getenv("EMSAR_HIP_TILE_ROWS")) { int v = atoi(e); if (v >= kTileSliceRows && v <= kTileRows) tile_rows = v / kTileSliceRows * kTileSliceRows; }
    // A unit may grow beyond unit_tiles tiles' worth of rows (up to unit_tiles_max) as long as that does not fill its dictionary with far
    // entries: rows of many neighbouring transcripts (config 5: 20 per read) pack three tiles under one dictionary with ~150 far slots,
    // rows of gene families (config 3) would pay 2.6 x the far entries for the third tile.  Measured with a fixed 3: config 5 x 0.25
    // 0.2121 -> 0.1896 ms, config 3 family law 0.1054 -> 0.1106, window law 0.0966 -> 0.1068 (gpurun_out/sweep4).
    int unit_tiles_max = 3, far_soft = 160;
    if (const char *e = getenv("EMSAR_HIP_UNIT_TILES_MAX")) { int v = atoi(e); if (v >= 1 && v <= kUnitMaxTiles) unit_tiles_max = v; }
    if (const char *e = getenv("EMSAR_HIP_UNIT_FAR_SOFT")) { int v = atoi(e); if (v >= 0 && v <= kFarMax) far_soft = v; }
    if (unit_tiles_max < unit_tiles) unit_tiles_max = unit_tiles;
    const int64_t base_rows = tile_rows * unit_tiles;
    const int64_t unit_rows = tile_rows * unit_tiles_max;
    int dense_min = kDenseMin;
    if (const char *e = getenv("EMSAR_HIP_TILE_DENSE")) { int v = atoi(e); if (v >= 1 && v <= 64) dense_min = v; }
    bool unit_sort = true;
    if (const char *e = getenv("EMSAR_HIP_UNIT_SORT")) unit_sort = atoi(e) != 0;
    bool unit_lpt = true;               // slices dealt to the waves by work (0: in order, alternate tiles mirrored)
    if (const char *e = getenv("EMSAR_HIP_UNIT_LPT")) unit_lpt = atoi(e) != 0;
    bool cut_at_slices = true;
    if (const char *e = getenv("EMSAR_HIP_TILE_CUT")) cut_at_slices = atoi(e) != 0;
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
        // Rows of few entries touch few transcripts: they are sorted in WIDER blocks (their (block, count) buckets then hold several
        // slices of equal rows -- no padding -- and a unit of them still spans a narrow range of anchors); they come after the others.
        counting_sort(pa, perm, (size_t)(2 * n_blocks * kLenClasses), [&](uint32_t r) {
            const int e = (int)ecnt[r];
            if (e <= short_ecnt) return (size_t)(n_blocks + mintid[r] / short_block) * kLenClasses + (size_t)len_class((int64_t)e);
            return (size_t)(mintid[r] / block) * kLenClasses + (size_t)len_class((int64_t)e);
        });
    }
    std::vector<uint32_t>().swap(pa);
    const auto tp2 = t_now();

    // ---- tiles ----
    // The sorted rows are cut into fragments of kFragRows rows; every fragment is tiled on its own (into a private
    // TiledLayout) and the fragments are concatenated.  The cut points depend on the data only, so the layout is the same
    // whatever the number of host threads that happen to build it.
    auto form_tiles = [&](int64_t range_begin, int64_t range_end, TiledLayout &out) -> int {
    std::vector<int32_t> stamp((size_t)n_tx, -1), loc((size_t)n_tx, 0);
        std::vector<int32_t> distinct;
        std::vector<uint32_t> pairs, sorted;   // (col_local << 16) | row_in_slice
        std::vector<uint32_t> ccount, fill;
        std::vector<uint32_t> segs;            // 4 dwords per segment
        std::vector<uint32_t> uord;            // the unit's rows in the order they are laid out
        std::vector<uint32_t> rslots, rents, rent, rent_ptr;   // one row's slots / entries; all rows' entries of the tile
        int64_t i0 = range_begin;
        int32_t tile_id = 0;
        const int64_t n_act = range_end;      // rows beyond the range belong to another fragment
        while (i0 < n_act) {
            // 1. how many rows fit: row cap, entry cap, distinct-tid cap
            distinct.clear();
            int64_t ents = 0, i1 = i0;
            while (i1 < n_act && i1 - i0 < unit_rows) {
                uint32_t r = perm[(size_t)i1];
                uint64_t b = row_ptr[r], e = row_ptr[r + 1];
                if (i1 > i0 && ents + (int64_t)(e - b) > (int64_t)unit_tiles_max * kTileEntries) break;
                size_t before = distinct.size();
                for (uint64_t k = b; k < e; k++) {
                    int32_t t = tid_at(k);
                    if (stamp[(size_t)t] != tile_id) { stamp[(size_t)t] = tile_id; distinct.push_back(t); }
                }
                if (i1 > i0 && (int64_t)distinct.size() > kTileDistinct) {   // undo this row, close the tile
                    for (size_t q = before; q < distinct.size(); q++) stamp[(size_t)distinct[q]] = -1;
                    distinct.resize(before);
                    break;
                }
                ents += (int64_t)(e - b);
                i1++;
            }
            if ((int64_t)distinct.size() > kTileDistinct) return -3;   // a single row with too many tids: excluded by kMaxRowLen
            // a tile closed by the dictionary or entry cap in the middle of a slice would pad that slice with empty rows
            // (forward bytes and gathers for nothing): give the rows of the started slice to the next tile instead
            if (cut_at_slices && i1 < n_act && i1 - i0 > kTileSliceRows && (i1 - i0) % kTileSliceRows != 0) {
                const int64_t keep = (i1 - i0) / kTileSliceRows * kTileSliceRows;
                for (int32_t t : distinct) stamp[(size_t)t] = -1;
                distinct.clear();
                i1 = i0 + keep;
                for (int64_t i = i0; i < i1; i++) {
                    uint32_t r = perm[(size_t)i];
                    for (uint64_t k = row_ptr[r]; k < row_ptr[r + 1]; k++) {
                        int32_t t = tid_at(k);
                        if (stamp[(size_t)t] != tile_id) { stamp[(size_t)t] = tile_id; distinct.push_back(t); }
                    }
                }
            }
            // 2. dictionary: the contiguous tid range [distinct[a], distinct[c]] that covers the MOST of the tile's tids while the
            //    near blocks (8 table entries per 3 slots of the range, tids present or not) plus one entry per tid outside the range
            //    (cross-family hits on either side: the far list) still fit the table.  The cost falls as a grows and rises as c grows,
            //    so the smallest feasible a for every c gives the widest cover.  If not even the best window fits, the unit gives
            //    back its last slice of rows and tries again.
            size_t best_a = 0, best_c = 0;
            for (;;) {
                std::sort(distinct.begin(), distinct.end());
                const size_t n = distinct.size();
                size_t a = 0;
                bool any = false;
                for (size_t c = 0; c < n; c++) {
                    auto fits = [&](size_t aa) {
                        const int64_t lo_a = distinct[aa] - distinct[aa] % kBlk;
                        return dict_fits((int64_t)distinct[c] - lo_a + 1, (int64_t)n - (int64_t)(c - aa + 1));
                    };
                    while (a < c && !fits(a)) a++;
                    if (!fits(a)) continue;
                    if (!any || c - a > best_c - best_a) { best_a = a; best_c = c; any = true; }
                }
                // a unit larger than the base size keeps its extra rows only if they did not flood the dictionary with far entries
                const bool too_far = any && i1 - i0 > base_rows && (int64_t)n - (int64_t)(best_c - best_a + 1) > (int64_t)far_soft;
                if (any && !too_far) break;
                if (i1 - i0 <= 1) return -3;
                const int64_t keep = too_far ? base_rows : i1 - i0 > kTileSliceRows ? (i1 - i0 - 1) / kTileSliceRows * kTileSliceRows : (i1 - i0) / 2;
                for (int32_t t : distinct) stamp[(size_t)t] = -1;
                distinct.clear();
                i1 = i0 + std::max<int64_t>(keep, 1);
                for (int64_t i = i0; i < i1; i++) {
                    uint32_t r = perm[(size_t)i];
                    for (uint64_t k = row_ptr[r]; k < row_ptr[r + 1]; k++) {
                        int32_t t = tid_at(k);
                        if (stamp[(size_t)t] != tile_id) { stamp[(size_t)t] = tile_id; distinct.push_back(t); }
                    }
                }
            }
            const int32_t lo = distinct[best_a] - distinct[best_a] % kBlk;       // blocks of the near range = tid / kBlk (see ecnt)
            const int32_t near_n = distinct[best_c] - lo + 1;
            Tile T;
            std::memset(&T, 0, sizeof T);
            T.lo = lo; T.near_n = (uint16_t)near_n;
            T.far_off = (uint32_t)out.far_tid.size();
            for (size_t q = 0; q < distinct.size(); q++) {
                int32_t t = distinct[q];
                if (t >= lo && t - lo < near_n) loc[(size_t)t] = t - lo;
                else { loc[(size_t)t] = near_n + (int32_t)(out.far_tid.size() - T.far_off); out.far_tid.push_back(t); }
            }
            const int32_t far_n = (int32_t)(out.far_tid.size() - T.far_off);     // (rounding lo down may have taken in a tid or two)
            T.far_n = (uint16_t)far_n;
            const int nd = near_n + far_n;
            const uint32_t pad_row = (uint32_t)kTileSliceRows;             // w_r[768] of every slice = 0
            const Tile Tdict = T;                                          // what the tiles of the unit share: the dictionary
            // The rows of the unit in descending order of their entry count (stable: rows of one count keep the order of the
            // sort above).  A slice's forward width is its longest row, so slices of equal rows carry the least padding, and
            // the slices come out in descending order of work -- what the unit kernel's wave assignment expects.
            uord.assign(perm.begin() + i0, perm.begin() + i1);
            if (unit_sort) std::stable_sort(uord.begin(), uord.end(), [&](uint32_t a, uint32_t b) { return ecnt[a] > ecnt[b]; });
            // the entries of every row of the tile, once (used by the forward and by the backward index)
            rent_ptr.assign(1, 0u); rent.clear();
            for (int64_t i = i0; i < i1; i++) {
                const uint32_t r = uord[(size_t)(i - i0)];
                rslots.clear();
                for (uint64_t q = row_ptr[r]; q < row_ptr[r + 1]; q++) {
                    const int32_t d = loc[(size_t)tid_at(q)];
                    rslots.push_back((uint32_t)d);
                    if (d >= near_n) out.far_entries++;
                }
                slots_to_entries(rslots, rents, (uint32_t)near_n);
                rent.insert(rent.end(), rents.begin(), rents.end());
                rent_ptr.push_back((uint32_t)rent.size());
                out.tiled_entries += (int64_t)rents.size();
                out.tiled_ids += (int64_t)rslots.size();
            }
            // The rows of the unit (longest first) are cut into slices of 768; the slices are dealt to the four waves of the workgroup,
            // heaviest first, each to the wave with the least work so far (work = forward columns + entries: the first slice of a unit
            // holds its long-row tail -- 18 columns on config 3 where the others have 2-7 -- and used to share a wave with the last one).
            // Tile t of the unit holds the t-th slice of every wave that has one; Tile::follows says which wave takes which slice.
            const int64_t n_chunks_u = (i1 - i0 + kTileSliceRows - 1) / kTileSliceRows;
            std::vector<int> wave_slices[kTileSlices];
            if (unit_lpt) {
                std::vector<std::pair<int64_t, int>> cost((size_t)n_chunks_u);
                for (int64_t c = 0; c < n_chunks_u; c++) {
                    const int64_t a0 = i0 + c * kTileSliceRows, bnd = std::min(i1, a0 + kTileSliceRows);
                    int64_t k = 0;
                    for (int64_t i = a0; i < bnd; i++) k = std::max<int64_t>(k, (int64_t)(rent_ptr[(size_t)(i - i0) + 1] - rent_ptr[(size_t)(i - i0)]));
                    cost[(size_t)c] = {k * kTileSliceRows + (int64_t)(rent_ptr[(size_t)(bnd - i0)] - rent_ptr[(size_t)(a0 - i0)]), (int)c};
                }
                std::stable_sort(cost.begin(), cost.end(), [](const std::pair<int64_t, int> &a, const std::pair<int64_t, int> &b) { return a.first > b.first; });
                int64_t load[kTileSlices] = {0, 0, 0, 0};
                for (const auto &cc : cost) {
                    int w = -1;
                    for (int q = 0; q < kTileSlices; q++)
                        if ((int)wave_slices[q].size() < kUnitMaxTiles && (w < 0 || load[q] < load[w])) w = q;
                    wave_slices[w].push_back(cc.second); load[w] += cc.first;
                }
            } else {                                      // in order, alternate tiles mirrored (the scheme before)
                for (int64_t c = 0; c < n_chunks_u; c++) {
                    const int64_t per = tile_rows / kTileSliceRows, t = c / per, s = c % per;
                    wave_slices[(t & 1) ? (int)(kTileSlices - 1 - s) : (int)s].push_back((int)c);
                }
            }
            size_t n_tiles_u = 0;
            for (int q = 0; q < kTileSlices; q++) n_tiles_u = std::max(n_tiles_u, wave_slices[q].size());
            for (size_t tu = 0; tu < n_tiles_u; tu++) {
            int chunk_of[kTileSlices]; unsigned wave_map = 0; int ns = 0;
            for (int q = 0; q < kTileSlices; q++)
                if (wave_slices[q].size() > tu) { chunk_of[ns] = wave_slices[q][tu]; wave_map |= (unsigned)q << (2 * ns); ns++; }
            T = Tdict;
            T.follows = tu > 0 ? 1 : 0;
            T.wave_of = (uint8_t)wave_map;
            T.n_slices = (uint16_t)ns;
            T.row_base = (uint32_t)out.slot_row.size();
            out.slot_row.resize(out.slot_row.size() + (size_t)T.n_slices * kTileSliceRows, -1);
            T.fwd_off = (uint64_t)out.fwd.size() * 4;
            T.bwd_off = (uint64_t)out.bwd.size() * 4;
            T.coo_off = (uint32_t)out.coo.size();
            out.n_fslices += T.n_slices;
            // 3. forward slices (all of them first: the tile's forward block is contiguous).  Column j of a slice is 256
            //    dwords; row p of the slice (p = position in sorted order) is field p/64 of the int4 of lane p%64: the 64
            //    lanes of one E-step gather read 64 CONSECUTIVE sorted rows, i.e. mostly one family -- the same few
            //    table entries (LDS broadcast) or neighbouring ones (distinct banks) instead of a random spread
            for (int s = 0; s < T.n_slices; s++) {
                int64_t a0 = i0 + (int64_t)chunk_of[s] * kTileSliceRows, bnd = std::min(i1, a0 + kTileSliceRows);
                int64_t k = 0;
                for (int64_t i = a0; i < bnd; i++) k = std::max<int64_t>(k, (int64_t)(rent_ptr[(size_t)(i - i0) + 1] - rent_ptr[(size_t)(i - i0)]));
                T.k[s] = (uint16_t)k;
                size_t base = out.fwd.size();
                out.fwd.resize(base + (size_t)k * kSliceDwords, 0u);            // entry 0 = the empty subset: padding
                out.padded_slots += k * kTileSliceRows;
                for (int64_t i = a0; i < bnd; i++) {
                    uint32_t r = uord[(size_t)(i - i0)];
                    uint32_t in_slice = (uint32_t)(i - a0);
                    out.slot_row[(size_t)T.row_base + (size_t)s * kTileSliceRows + in_slice] = (int64_t)r;
                    const uint32_t b = rent_ptr[(size_t)(i - i0)], e = rent_ptr[(size_t)(i - i0) + 1];
                    for (uint32_t q = b; q < e; q++) {
                        const uint32_t d = rent[q];
                        const uint32_t fl = in_slice & 63u, fi = in_slice >> 6;      // lane, field: see slot numbering above
                        uint32_t *dw = &out.fwd[base + (size_t)(q - b) * kSliceDwords + fl * 4 + fi / 3];
                        const int sh = 10 * (int)(fi % 3);
                        *dw = (*dw & ~(0x3FFu << sh)) | (d << sh);
                    }
                }
            }
            // 4. backward index of each slice: its (entry value, row) pairs sorted by entry value
            for (int s = 0; s < T.n_slices; s++) {
                int64_t a0 = i0 + (int64_t)chunk_of[s] * kTileSliceRows, bnd = std::min(i1, a0 + kTileSliceRows);
                pairs.clear();
                for (int64_t i = a0; i < bnd; i++) {
                    uint32_t in_slice = (uint32_t)(i - a0);
                    for (uint32_t q = rent_ptr[(size_t)(i - i0)]; q < rent_ptr[(size_t)(i - i0) + 1]; q++) pairs.push_back((rent[q] << 16) | in_slice);
                }
                ccount.assign((size_t)kDictEntries + 1, 0);
                for (uint32_t p : pairs) ccount[(p >> 16) + 1]++;
                for (int d = 0; d < kDictEntries; d++) ccount[(size_t)d + 1] += ccount[(size_t)d];
                sorted.resize(pairs.size());
                fill.assign(ccount.begin(), ccount.end() - 1);
                for (uint32_t p : pairs) sorted[fill[p >> 16]++] = p;
                segs.clear();
                size_t coo_before = out.coo.size();
                for (int d = 0; d < kDictEntries; d++) {
                    uint32_t b = ccount[(size_t)d], e = ccount[(size_t)d + 1];
                    if (e - b < (uint32_t)dense_min) {
                        for (uint32_t q = b; q < e; q++) out.coo.push_back(((uint32_t)d << 16) | (sorted[q] & 0xFFFF));
                        continue;
                    }
                    for (uint32_t q = b; q < e; q += kSegRows) {
                        uint32_t seg[4] = {0, 0, 0, 0};
                        pack10(seg, 0, (uint32_t)d);
                        for (uint32_t j = 0; j < (uint32_t)kSegRows; j++) pack10(seg, 1 + (int)j, q + j < e ? (sorted[q + j] & 0xFFFF) : pad_row);
                        segs.insert(segs.end(), seg, seg + 4);
                    }
                }
                T.coo_n[s] = (uint16_t)(out.coo.size() - coo_before);
                out.coo_entries += T.coo_n[s];
                const int64_t nseg = (int64_t)segs.size() / 4;
                const int m = (int)((nseg + 63) / 64);
                T.m[s] = (uint16_t)m;
                size_t base = out.bwd.size();
                uint32_t empty[4] = {0, 0, 0, 0};                          // unused segment: entry 0 (never flushed), padding rows
                for (int j = 1; j < 12; j++) pack10(empty, j, pad_row);
                out.bwd.resize(base + (size_t)m * 64 * 4, 0u);
                for (int64_t g = 0; g < (int64_t)m * 64; g++) {
                    // logical segment g -> lane g / m, unit g % m ; physical int4 index (unit*64 + lane)
                    int64_t lane = g / m, unit = g % m;
                    size_t u0 = base + (size_t)((unit * 64 + lane) * 4);
                    const uint32_t *src = g < nseg ? &segs[(size_t)g * 4] : empty;
                    for (int w = 0; w < 4; w++) out.bwd[u0 + (size_t)w] = src[w];
                }
            }
            (void)nd;
            out.tiles.push_back(T);
            }
            for (int32_t t : distinct) stamp[(size_t)t] = -1;
            tile_id++;
            i0 = i1;
        }
        return 0;
    };
    {
        int64_t frag_rows = kFragRows;
        if (const char *e = getenv("EMSAR_HIP_FRAG_ROWS")) { long long v = atoll(e); if (v >= kTileRows) frag_rows = v; }   // tests: many fragments on small inputs
        const int64_t n_frag = std::max<int64_t>(1, (n_act + frag_rows - 1) / frag_rows);
        std::vector<TiledLayout> frag((size_t)n_frag);
        std::vector<int> frc((size_t)n_frag, 0);
        unsigned hw = std::thread::hardware_concurrency();
        int nthr = (int)std::min<int64_t>(n_frag, hw ? std::min(hw, 16u) : 1u);
        if (const char *e = getenv("EMSAR_HOST_THREADS")) { int v = atoi(e); if (v >= 1) nthr = (int)std::min<int64_t>(n_frag, v); }
        std::atomic<int64_t> next{0};
        auto worker = [&]() {
            for (;;) {
                const int64_t g = next.fetch_add(1);
                if (g >= n_frag) break;
                frc[(size_t)g] = form_tiles(g * frag_rows, std::min(n_act, (g + 1) * frag_rows), frag[(size_t)g]);
            }
        };
        run_on_threads(nthr, [&](int) { worker(); });
        for (int64_t g = 0; g < n_frag; g++) if (frc[(size_t)g] != 0) return frc[(size_t)g];
        const auto tp3 = t_now();
        if (dbg_t) fprintf(stderr, "build_tiled: classify %.0f ms, sort %.0f ms, tiles %.0f ms on %d thread(s)\n", t_ms(tp0, tp1), t_ms(tp1, tp2), t_ms(tp2, tp3), nthr);
        // concatenate: descriptors, COO pairs and far lists here (small), the three big arrays by the pool, each fragment
        // into its own range of the final arrays
        std::vector<size_t> slot_b((size_t)n_frag + 1, 0), fwd_b((size_t)n_frag + 1, 0), bwd_b((size_t)n_frag + 1, 0);
        {
            size_t nt_ = 0, nc_ = 0, nfar_ = 0;
            for (int64_t g = 0; g < n_frag; g++) {
                const TiledLayout &F = frag[(size_t)g];
                nt_ += F.tiles.size(); nc_ += F.coo.size(); nfar_ += F.far_tid.size();
                slot_b[(size_t)g + 1] = slot_b[(size_t)g] + F.slot_row.size();
                fwd_b[(size_t)g + 1] = fwd_b[(size_t)g] + F.fwd.size();
                bwd_b[(size_t)g + 1] = bwd_b[(size_t)g] + F.bwd.size();
            }
            if (slot_b[(size_t)n_frag] >= ((size_t)1 << 32)) return -1;
            out.tiles.reserve(nt_); out.coo.reserve(nc_); out.far_tid.reserve(nfar_);
            out.slot_row.resize(slot_b[(size_t)n_frag]); out.fwd.resize(fwd_b[(size_t)n_frag]); out.bwd.resize(bwd_b[(size_t)n_frag]);
        }
        for (int64_t g = 0; g < n_frag; g++) {
            TiledLayout &F = frag[(size_t)g];
            const uint32_t far_b = (uint32_t)out.far_tid.size(), coo_b = (uint32_t)out.coo.size();
            for (Tile t : F.tiles) {
                t.fwd_off += (uint64_t)fwd_b[(size_t)g] * 4; t.bwd_off += (uint64_t)bwd_b[(size_t)g] * 4;
                t.row_base += (uint32_t)slot_b[(size_t)g]; t.far_off += far_b; t.coo_off += coo_b;
                out.tiles.push_back(t);
            }
            out.coo.insert(out.coo.end(), F.coo.begin(), F.coo.end());
            out.far_tid.insert(out.far_tid.end(), F.far_tid.begin(), F.far_tid.end());
            out.tiled_entries += F.tiled_entries; out.tiled_ids += F.tiled_ids; out.far_entries += F.far_entries; out.coo_entries += F.coo_entries;
            out.n_fslices += F.n_fslices; out.padded_slots += F.padded_slots;
        }
        next.store(0);
        auto copier = [&]() {
            for (;;) {
                const int64_t g = next.fetch_add(1);
                if (g >= n_frag) break;
                TiledLayout &F = frag[(size_t)g];
                if (!F.slot_row.empty()) memcpy(out.slot_row.data() + slot_b[(size_t)g], F.slot_row.data(), F.slot_row.size() * sizeof(int64_t));
                if (!F.fwd.empty()) memcpy(out.fwd.data() + fwd_b[(size_t)g], F.fwd.data(), F.fwd.size() * 4);
                if (!F.bwd.empty()) memcpy(out.bwd.data() + bwd_b[(size_t)g], F.bwd.data(), F.bwd.size() * 4);
                F = TiledLayout();
            }
        };
        run_on_threads(nthr, [&](int) { copier(); });
    }
    // Largest tiles first: they start while the grid is full, the small ones fill the tail.
    auto work = [](const Tile &t) {
        int64_t w = 0;
        for (int s = 0; s < t.n_slices; s++) w += (int64_t)t.k[s] * kTileSliceRows + (int64_t)t.m[s] * 64 * 12 + t.coo_n[s] * 2;
        return w;
    };
    {   // the sort moves whole units (a tile and the one that follows it)
        struct Unit { uint32_t first, n; int64_t w; };
        std::vector<Unit> units;
        for (size_t i = 0; i < out.tiles.size(); i++) {
            if (out.tiles[i].follows && !units.empty()) { units.back().n++; units.back().w += work(out.tiles[i]); }
            else { out.tiles[i].follows = 0; units.push_back(Unit{(uint32_t)i, 1u, work(out.tiles[i])}); }
        }
        std::stable_sort(units.begin(), units.end(), [](const Unit &a, const Unit &b) { return a.w > b.w; });
        // The tail: with ~3.4 k units for 1024 workgroup slots the last workgroups run on a half-empty chip (timeline of config 3: 14 of
        // 114 us below 75 % occupancy).  The lightest units are therefore cut into their tiles -- every tile carries the dictionary
        // descriptor, so a tile that stops following is a unit of its own -- and the small pieces fill the tail.  Measured (EMSAR_HIP_TAIL_SPLIT =
        // share of the units cut, config 3): 0 / 10 / 20 / 35 / 50 % -> family law 0.1071 / 0.1055 / 0.1075 / 0.1077 / 0.1108 ms, window law
        // 0.0967 / 0.0976 / 0.0986 / 0.1000: what the tail gains the extra per-unit overhead takes back.  Off.
        int tail_pct = 0;
        if (const char *e = getenv("EMSAR_HIP_TAIL_SPLIT")) { int v = atoi(e); if (v >= 0 && v <= 100) tail_pct = v; }
        if (tail_pct > 0 && units.size() > 2048) {
            const size_t keep = units.size() - units.size() * (size_t)tail_pct / 100;
            std::vector<Unit> cut(units.begin(), units.begin() + (std::ptrdiff_t)keep);
            for (size_t q = keep; q < units.size(); q++)
                for (uint32_t j = 0; j < units[q].n; j++) {
                    out.tiles[units[q].first + j].follows = 0;
                    cut.push_back(Unit{units[q].first + j, 1u, work(out.tiles[units[q].first + j])});
                }
            std::stable_sort(cut.begin() + (std::ptrdiff_t)keep, cut.end(), [](const Unit &a, const Unit &b) { return a.w > b.w; });
            units.swap(cut);
        }
        if (dbg_t && !units.empty()) {     // how well the units fill 1024 workgroup slots in this (longest-first) order: greedy makespan over the mean load
            for (int64_t c : {(int64_t)0, (int64_t)20000}) {     // c: a fixed cost per unit (descriptor, dictionary, flush) in the units of `work`
                std::vector<int64_t> slot(1024, 0);
                int64_t tot = 0;
                for (const Unit &u : units) { auto it = std::min_element(slot.begin(), slot.end()); *it += u.w + c; tot += u.w; }
                const int64_t mk = *std::max_element(slot.begin(), slot.end());
                fprintf(stderr, "build_tiled: %zu units, work max %lld, median %lld, min %lld; with %lld per unit on top: greedy makespan on 1024 slots %.0f = %.3f of the mean load\n",
                        units.size(), (long long)units.front().w, (long long)units[units.size() / 2].w, (long long)units.back().w, (long long)c, (double)mk, (double)mk * 1024.0 / (double)tot);
            }
        }
        std::vector<Tile> sorted_tiles;
        sorted_tiles.reserve(out.tiles.size());
        out.unit_first.clear();
        for (const Unit &u : units) {
            out.unit_first.push_back((uint32_t)sorted_tiles.size());
            for (uint32_t j = 0; j < u.n; j++) sorted_tiles.push_back(out.tiles[u.first + j]);
        }
        out.unit_first.push_back((uint32_t)sorted_tiles.size());
        out.tiles.swap(sorted_tiles);
    }
    if (dbg_t) fprintf(stderr, "build_tiled: total %.0f ms; %zu tiles in %zu units, %lld slices, %lld entries in %lld padded operands, %lld far\n", t_ms(tp0, t_now()),
                       out.tiles.size(), out.unit_first.empty() ? (size_t)0 : out.unit_first.size() - 1, (long long)out.n_fslices,
                       (long long)out.tiled_entries, (long long)out.padded_slots, (long long)out.far_entries);
    if (dbg_t) {        // batches of 8 columns / segments per slice: what the E- and M-steps request after their first batch
        double n = 0, sk = 0, sm = 0, k8 = 0, k16 = 0, m8 = 0, m16 = 0, m24 = 0;
        for (const Tile &T : out.tiles) for (int s2 = 0; s2 < (int)T.n_slices; s2++) {
            n += 1; sk += T.k[s2]; sm += T.m[s2]; k8 += T.k[s2] > 8; k16 += T.k[s2] > 16; m8 += T.m[s2] > 8; m16 += T.m[s2] > 16; m24 += T.m[s2] > 24;
        }
        if (n > 0) fprintf(stderr, "build_tiled: per slice %.2f forward columns (%.0f %% > 8, %.0f %% > 16), %.2f backward segments per lane (%.0f %% > 8, %.0f %% > 16, %.0f %% > 24)\n",
                           sk / n, 100 * k8 / n, 100 * k16 / n, sm / n, 100 * m8 / n, 100 * m16 / n, 100 * m24 / n);
    }
    if (dbg_t && !out.unit_first.empty()) {         // how evenly a unit's slices load the four waves of its workgroup: sum of the loads / (4 x the largest)
        double eff = 0, wsum = 0, hist[13] = {0};
        const size_t nu = out.unit_first.size() - 1;
        for (size_t u = 0; u < nu; u++) {
            int64_t load[kTileSlices] = {0, 0, 0, 0}, tot = 0; int ns = 0;
            for (uint32_t t = out.unit_first[u]; t < out.unit_first[u + 1]; t++) {
                const Tile &T = out.tiles[t];
                for (int s2 = 0; s2 < (int)T.n_slices; s2++) {
                    const int64_t w = (int64_t)T.k[s2] * kTileSliceRows + (int64_t)T.m[s2] * 64 * 12;
                    load[(T.wave_of >> (2 * s2)) & 3] += w; tot += w; ns++;
                }
            }
            const int64_t mx = std::max(std::max(load[0], load[1]), std::max(load[2], load[3]));
            eff += (double)tot / 4.0; wsum += (double)mx; hist[std::min(ns, 12)] += 1;
        }
        fprintf(stderr, "build_tiled: the four waves of a unit are busy %.3f of the unit's longest wave on average; units by number of slices 1..12:", eff / wsum);
        for (int i = 1; i <= 12; i++) fprintf(stderr, " %.0f", hist[i]);
        fprintf(stderr, "\n");
    }
    const int ext = check_tiled_extents(out);       // nothing reaches the device unless every descriptor stays inside its arrays
    return ext == 0 ? 0 : ext;
}

// What k_pass_tiled_unit reads first, at an address that depends on the workgroup index alone (no unit -> tile index to chase):
//   utiles[u * stride + j]   the j-th tile of unit u, j < stride = the most tiles any unit has; absent tiles have n_slices = 0
struct UnitTables { std::vector<Tile> utiles; int stride = 1; };
inline void build_unit_tables(const TiledLayout &L, UnitTables &U) {
    const size_t nu = L.unit_first.empty() ? 0 : L.unit_first.size() - 1;
    int stride = 1;
    for (size_t u = 0; u < nu; u++) stride = std::max(stride, (int)(L.unit_first[u + 1] - L.unit_first[u]));
    U.stride = stride;
    Tile none;
    std::memset(&none, 0, sizeof none);
    U.utiles.assign(nu * (size_t)stride, none);
    for (size_t u = 0; u < nu; u++) {
        const uint32_t a = L.unit_first[u], b = L.unit_first[u + 1];
        for (uint32_t t = a; t < b; t++) U.utiles[u * (size_t)stride + (t - a)] = L.tiles[t];
    }
    if (getenv("EMSAR_HIP_DEBUG")) {
        std::vector<uint32_t> cnt((size_t)L.n_tx, 0);
        size_t slots = 0, with = 0; uint32_t mx = 0;
        for (size_t u = 0; u < nu; u++) {
            const Tile &T = L.tiles[L.unit_first[u]];
            for (uint32_t f = 0; f < T.far_n; f++) { cnt[(size_t)L.far_tid[(size_t)T.far_off + f]]++; slots++; }
        }
        for (uint32_t c : cnt) { with += c > 0; mx = std::max(mx, c); }
        fprintf(stderr, "unit tables: %zu units, stride %d, %zu far slots of %zu transcripts (most per transcript %u)\n", nu, stride, slots, with, mx);
    }
}

// the unit tables against the layout they were made from (host self-check): 0 = consistent
inline int check_unit_tables(const TiledLayout &L, const UnitTables &U) {
    const size_t nu = L.unit_first.empty() ? 0 : L.unit_first.size() - 1;
    if (U.stride < 1 || U.stride > kUnitMaxTiles) return -40;
    if (U.utiles.size() != nu * (size_t)U.stride) return -41;
    for (size_t u = 0; u < nu; u++) {
        const uint32_t a = L.unit_first[u], b = L.unit_first[u + 1];
        if (b - a > (uint32_t)U.stride) return -42;
        for (int j = 0; j < U.stride; j++) {
            const Tile &X = U.utiles[u * (size_t)U.stride + (size_t)j];
            if (a + (uint32_t)j < b) { if (std::memcmp(&X, &L.tiles[a + (uint32_t)j], sizeof(Tile)) != 0) return -43; }
            else if (X.n_slices != 0) return -44;               // the kernel stops at the first tile without slices
        }
    }
    return 0;
}

// Decode and compare with the input (host self-check, used by the CPU tests). 0 = identical.
inline int check_tiled(const TiledLayout &L, const uint64_t *row_ptr, const int32_t *col_idx_in) {
    if (const int ext = check_tiled_extents(L)) return ext;          // the decode below indexes the arrays by the descriptors
    // the layout is in the library's numbering: compare with the caller's rows mapped the same way (the map must be a permutation)
    std::vector<int32_t> mapped;
    const int32_t *col_idx = col_idx_in;
    if (!L.new_of_old.empty()) {
        if ((int64_t)L.new_of_old.size() != (int64_t)L.n_tx) return -11;
        std::vector<uint8_t> hit((size_t)L.n_tx, 0);
        for (int32_t v : L.new_of_old) { if (v < 0 || v >= L.n_tx || hit[(size_t)v]) return -11; hit[(size_t)v] = 1; }
        const uint64_t nnz = row_ptr[L.n_rows];
        mapped.resize((size_t)nnz);
        for (uint64_t k = 0; k < nnz; k++) mapped[(size_t)k] = L.new_of_old[(size_t)col_idx_in[k]];
        col_idx = mapped.data();
    }
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
    std::vector<int32_t> a, b;
    std::vector<uint32_t> pf, pb;
    for (const Tile &T : L.tiles) {
        const int nd = T.near_n + T.far_n;
        if (!dict_fits(T.near_n, T.far_n) || T.n_slices > kTileSlices || T.row_base % kTileSliceRows) return -3;
        const uint32_t far_base = (uint32_t)kBlkEntries * (uint32_t)near_blocks(T.near_n);
        auto tid_of = [&](int d) { return d < T.near_n ? T.lo + d : L.far_tid[(size_t)T.far_off + (size_t)(d - T.near_n)]; };
        size_t foff = (size_t)(T.fwd_off / 4), boff = (size_t)(T.bwd_off / 4), coff = T.coo_off;
        for (int s = 0; s < T.n_slices; s++) {
            pf.clear(); pb.clear();
            for (int i = 0; i < kTileSliceRows; i++) {
                int64_t r = L.slot_row[(size_t)T.row_base + (size_t)s * kTileSliceRows + (size_t)i];
                a.clear();
                for (int j = 0; j < T.k[s]; j++) {
                    const int fl = i & 63, fi = i >> 6;
                    const uint32_t e = (L.fwd[foff + (size_t)j * kSliceDwords + (size_t)(fl * 4 + fi / 3)] >> (10 * (fi % 3))) & 0x3FFu;
                    if (e >= far_base) a.push_back(tid_of((int)T.near_n + (int)(e - far_base)));
                    else {
                        const uint32_t msk = entry_mask(e);
                        if (msk == 0) { if (e != 0) return -4; continue; }   // entry 0 = padding
                        for (int bit = 0; bit < kBlk; bit++)
                            if (msk >> bit & 1u) a.push_back(tid_of((int)(e >> kBlk) * kBlk + bit));
                    }
                    pf.push_back((e << 16) | (uint32_t)i);
                }
                if (r < 0) { if (!a.empty()) return -5; continue; }
                std::sort(a.begin(), a.end());                    // entries are in block order: the decoded row is compared as a multiset
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
            foff += (size_t)T.k[s] * kSliceDwords;
            const int m = T.m[s];
            for (int64_t g = 0; g < (int64_t)m * 64; g++) {
                int64_t lane = g / m, unit = g % m;
                const uint32_t *q = &L.bwd[boff + (size_t)((unit * 64 + lane) * 4)];
                uint32_t d = unpack10(q, 0);
                for (int w = 1; w < 12; w++) {
                    uint32_t rl = unpack10(q, w);
                    if (rl == (uint32_t)kTileSliceRows) continue;
                    if (rl > (uint32_t)kTileSliceRows || d >= (uint32_t)kDictEntries || (d < far_base && entry_mask(d) == 0)) return -8;
                    pb.push_back((d << 16) | rl);
                }
            }
            boff += (size_t)m * 64 * 4;
            for (uint32_t q = 0; q < T.coo_n[s]; q++) {
                uint32_t p = L.coo[coff + q];
                pb.push_back(p);
            }
            coff += T.coo_n[s];
            std::sort(pf.begin(), pf.end()); std::sort(pb.begin(), pb.end());
            if (pf != pb) return -9;                                // the backward index is the transpose of the forward one
        }
    }
    for (int64_t r = 0; r < L.n_rows; r++)
        if (!seen[(size_t)r] && row_ptr[r + 1] != row_ptr[r]) return -10;
    return 0;
}

}  // namespace emsar
