// layout.hpp -- host-side construction of the WINDOWED HBM layout (pure C++, no HIP calls).
//
// Input is the incidence CT as CSR (what scan_rshbucket builds as ragged arrays,
// /root/reference/src/emsar_functions.c:2135-2192).  The EM only needs, per row, the multiset of its
// transcript ids, so rows may be stored in any order.  We choose the order for the hardware:
//
//   rows are sorted by (block(min tid), length class, min tid)      [two stable counting-sort passes]
//   consecutive groups of 256 sorted rows form a SLICE, stored column-major:
//        entry j of row i of the slice  ->  ent[slice_off + j*256 + i]           (-1 = padding)
//     lane l of the wave that owns the slice handles rows 4l..4l+3: its j-th load is ONE 16-byte int4, the
//     wave reads 1 KiB contiguous per load instruction, and no row_ptr is needed at run time
//   consecutive slices form a CHUNK (one workgroup); every tid of a chunk is >= chunk.lo, and tids in
//     [lo, lo+width) are served from LDS copies of theta / the M-step accumulator ("near"), the few
//     others ("far", cross-family alignments) go to L2/HBM directly.
#pragma once
#include <algorithm>
#include <cstdint>
#include <cstring>
#include <vector>

namespace emsar {

constexpr int kSliceRows = 256;    // one wavefront, 4 rows per lane (16-byte loads)
constexpr int kMinBlockTids = 256;  // smallest sort granularity in tid space
constexpr int kLenClasses = 64;

struct Chunk {             // 16 bytes, read once per workgroup
    uint32_t slice_begin;  // first slice (global slice index); sorted-row index = slice*256 + 4*lane + i
    uint32_t n_slices;
    int32_t lo;            // first tid of the LDS window
    int32_t width;         // window length in tids (<= window)
};

struct WindowedLayout {
    int64_t n_rows = 0, nnz = 0;
    int32_t n_tx = 0, window = 0;
    int32_t block_tids = kMinBlockTids;  // rows are bucketed by min_tid / block_tids
    int64_t n_sorted_rows = 0;           // rows with at least one tid
    std::vector<uint32_t> perm;          // sorted position -> original row
    std::vector<uint64_t> slice_off;     // n_slices+1 entry offsets (multiples of 256)
    std::vector<int32_t> ent;            // padded column-major entries
    std::vector<Chunk> chunks;
    int64_t far_entries = 0;
    int64_t n_slices() const { return (int64_t)slice_off.size() - 1; }
};

inline int len_class(int64_t len) {
    if (len <= 32) return (int)len;
    int64_t c = 32 + (len - 32 + 7) / 8;
    return (int)std::min<int64_t>(c, kLenClasses - 1);
}

// returns 0 ok, -1 malformed CSR
inline int validate_csr(int64_t n_rows, int32_t n_tx, const uint64_t *row_ptr, const int32_t *col_idx) {
    if (n_rows < 0 || n_tx <= 0 || !row_ptr) return -1;
    if (row_ptr[0] != 0) return -1;
    for (int64_t r = 0; r < n_rows; r++)
        if (row_ptr[r + 1] < row_ptr[r]) return -1;
    uint64_t nnz = row_ptr[n_rows];
    if (nnz && !col_idx) return -1;
    for (uint64_t k = 0; k < nnz; k++)
        if (col_idx[k] < 0 || col_idx[k] >= n_tx) return -1;
    return 0;
}

inline int build_windowed(int64_t n_rows, int32_t n_tx, const uint64_t *row_ptr, const int32_t *col_idx,
                          int32_t window, int64_t chunk_entries, WindowedLayout &out) {
    if (n_rows >= (int64_t)1 << 32) return -1;   // perm is 32-bit; 4.2e9 rows is beyond every BASELINE config
    out = WindowedLayout();
    out.n_rows = n_rows; out.n_tx = n_tx; out.window = window; out.nnz = (int64_t)row_ptr[n_rows];

    // ---- per-row keys ----
    std::vector<int32_t> mintid((size_t)n_rows);
    int64_t n_act = 0;
    for (int64_t r = 0; r < n_rows; r++) {
        uint64_t b = row_ptr[r], e = row_ptr[r + 1];
        if (e == b) { mintid[(size_t)r] = -1; continue; }
        int32_t m = col_idx[b];
        for (uint64_t k = b + 1; k < e; k++) m = std::min(m, col_idx[k]);
        mintid[(size_t)r] = m;
        n_act++;
    }
    out.n_sorted_rows = n_act;
    // Sort granularity: a (block, length-class) bucket should hold many 64-row slices, otherwise slices mix
    // lengths and pad.  Aim at >= 16384 rows per block; never wider than the LDS window.
    {
        int64_t want = n_act > 0 ? (16384 * (int64_t)n_tx + n_act - 1) / n_act : kMinBlockTids;
        int32_t b = kMinBlockTids;
        while (b < want && b * 2 <= window) b *= 2;
        out.block_tids = std::min(b, window);
    }
    const int32_t kBlockTids = out.block_tids;

    // ---- pass A: stable counting sort by min tid ----
    std::vector<uint32_t> pa((size_t)n_act);
    {
        std::vector<uint64_t> cnt((size_t)n_tx + 1, 0);
        for (int64_t r = 0; r < n_rows; r++) if (mintid[(size_t)r] >= 0) cnt[(size_t)mintid[(size_t)r] + 1]++;
        for (int32_t t = 0; t < n_tx; t++) cnt[(size_t)t + 1] += cnt[(size_t)t];
        for (int64_t r = 0; r < n_rows; r++) if (mintid[(size_t)r] >= 0) pa[(size_t)cnt[(size_t)mintid[(size_t)r]]++] = (uint32_t)r;
    }
    // ---- pass B: stable counting sort by (block, length class) ----
    out.perm.resize((size_t)n_act);
    {
        int64_t n_blocks = ((int64_t)n_tx + kBlockTids - 1) / kBlockTids;
        std::vector<uint64_t> cnt((size_t)(n_blocks * kLenClasses) + 1, 0);
        auto key = [&](uint32_t r) {
            return (size_t)(mintid[r] / kBlockTids) * kLenClasses + (size_t)len_class((int64_t)(row_ptr[r + 1] - row_ptr[r]));
        };
        for (int64_t i = 0; i < n_act; i++) cnt[key(pa[(size_t)i]) + 1]++;
        for (size_t i = 0; i + 1 < cnt.size(); i++) cnt[i + 1] += cnt[i];
        for (int64_t i = 0; i < n_act; i++) { uint32_t r = pa[(size_t)i]; out.perm[(size_t)cnt[key(r)]++] = r; }
    }
    std::vector<uint32_t>().swap(pa);

    // ---- slices ----
    int64_t n_slices = (n_act + kSliceRows - 1) / kSliceRows;
    out.slice_off.assign((size_t)n_slices + 1, 0);
    for (int64_t s = 0; s < n_slices; s++) {
        int64_t k = 0;
        for (int64_t i = s * kSliceRows; i < std::min(n_act, (s + 1) * kSliceRows); i++) {
            uint32_t r = out.perm[(size_t)i];
            k = std::max<int64_t>(k, (int64_t)(row_ptr[r + 1] - row_ptr[r]));
        }
        out.slice_off[(size_t)s + 1] = out.slice_off[(size_t)s] + (uint64_t)k * kSliceRows;
    }
    out.ent.assign((size_t)out.slice_off[(size_t)n_slices], -1);
    for (int64_t s = 0; s < n_slices; s++) {
        int32_t *base = out.ent.data() + out.slice_off[(size_t)s];
        for (int64_t i = s * kSliceRows; i < std::min(n_act, (s + 1) * kSliceRows); i++) {
            uint32_t r = out.perm[(size_t)i];
            int lane = (int)(i - s * kSliceRows);
            uint64_t b = row_ptr[r], e = row_ptr[r + 1];
            for (uint64_t k = b; k < e; k++) base[(k - b) * kSliceRows + lane] = col_idx[k];
        }
    }

    // ---- chunks ----
    int64_t s = 0;
    while (s < n_slices) {
        Chunk c;
        c.slice_begin = (uint32_t)s;
        int32_t first_min = mintid[out.perm[(size_t)(s * kSliceRows)]];
        c.lo = (first_min / kBlockTids) * kBlockTids;
        int64_t ents = 0, s1 = s;
        while (s1 < n_slices) {
            int64_t last = std::min(n_act, (s1 + 1) * kSliceRows) - 1;
            // the sort is by block first, so the last row of the slice carries the slice's largest block
            int64_t block_end = ((int64_t)mintid[out.perm[(size_t)last]] / kBlockTids + 1) * kBlockTids;
            if (s1 > s && block_end > (int64_t)c.lo + window) break;
            int64_t se = (int64_t)(out.slice_off[(size_t)s1 + 1] - out.slice_off[(size_t)s1]);
            if (s1 > s && ents + se > chunk_entries) break;
            ents += se;
            s1++;
        }
        c.n_slices = (uint32_t)(s1 - s);
        // tighten the window to the tids actually present below lo+window
        int32_t hi = c.lo;
        for (uint64_t k = out.slice_off[(size_t)s]; k < out.slice_off[(size_t)s1]; k++) {
            int32_t t = out.ent[(size_t)k];
            if (t < 0) continue;
            if ((int64_t)t < (int64_t)c.lo + window) hi = std::max(hi, t);
            else out.far_entries++;
            if (t < c.lo) return -2;   // cannot happen: rows are bucketed by their smallest tid
        }
        c.width = hi - c.lo + 1;
        out.chunks.push_back(c);
        s = s1;
    }
    return 0;
}

// Decode the layout back into per-row sorted tid lists and compare with the input (host self-check used by
// the CPU tests): returns 0 when the stored multiset of rows equals the input's.
inline int check_windowed(const WindowedLayout &L, const uint64_t *row_ptr, const int32_t *col_idx) {
    std::vector<uint8_t> seen((size_t)L.n_rows, 0);
    std::vector<int32_t> a, b;
    uint64_t covered = 0;
    for (const Chunk &c : L.chunks) {
        if ((uint64_t)c.slice_begin != covered) return -1;
        covered += c.n_slices;
        if (c.width < 1 || c.width > L.window || c.lo < 0 || (int64_t)c.lo + c.width > L.n_tx) return -2;
    }
    if ((int64_t)covered != L.n_slices()) return -3;
    for (int64_t s = 0; s < L.n_slices(); s++) {
        int64_t k = (int64_t)(L.slice_off[(size_t)s + 1] - L.slice_off[(size_t)s]) / kSliceRows;
        for (int lane = 0; lane < kSliceRows; lane++) {
            int64_t i = s * kSliceRows + lane;
            a.clear();
            for (int64_t j = 0; j < k; j++) {
                int32_t t = L.ent[(size_t)(L.slice_off[(size_t)s] + (uint64_t)(j * kSliceRows + lane))];
                if (t >= 0) a.push_back(t);
            }
            if (i >= L.n_sorted_rows) { if (!a.empty()) return -4; continue; }
            uint32_t r = L.perm[(size_t)i];
            if (seen[r]) return -5;
            seen[r] = 1;
            b.assign(col_idx + row_ptr[r], col_idx + row_ptr[r + 1]);
            if (a != b) return -6;
        }
    }
    for (int64_t r = 0; r < L.n_rows; r++)
        if (!seen[(size_t)r] && row_ptr[r + 1] != row_ptr[r]) return -7;
    return 0;
}

}  // namespace emsar
