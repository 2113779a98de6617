// sets.hpp -- host-side builder of the SET-RESIDENT solver's data (pure C++, no HIP: also compiled by the CPU
// sanitizer harness tools/layout_fuzz.cpp).
//
// The reference maximises each connected set of segments on its own (run_MLE_threads hands out sids,
// /root/reference/src/emsar_main.c:446-474; sets come from build_TC_from_CT_2/propagate_2,
// emsar_functions.c:2201-2259).  The EM decouples the same way: theta_t is updated from the rows that contain t
// only, and rows with R = 0 enter through den_t alone.  So the transcripts split into the connected components
// of the rows with R > 0 (and E > 0) -- finer than the reference's sets, which also link through R = 0 rows --
// and every component can be iterated to convergence independently.
//
//   * a component of one transcript has the closed form theta_t = (sum of its rows' R) / den_t;
//   * a component whose working set fits the 160 KiB LDS of a CU is packed here into a self-contained record
//     (local 16-bit ids, CSR for the E-step, CSC for the M-step, identical rows merged, single-transcript rows
//     folded into a per-transcript count) and solved by ONE workgroup with no global synchronisation;
//   * a component too large for that but with up to a few thousand transcripts is packed for a CLUSTER of 2, 4 or 8
//     workgroups (kind 3): the merged rows are dealt to the workgroups in contiguous ranges of equal nnz, each workgroup gets
//     the CSR of its rows and the CSC of its rows (transposed per workgroup), the transcripts are dealt in equal ranges
//     (kernels_cluster.hpp);
//   * anything larger is left to the streaming kernels (kind 2).
#ifndef EMSAR_SETS_HPP
#define EMSAR_SETS_HPP
#include <algorithm>
#include <cstddef>
#include <cstdint>
#include <cstdlib>
#include <numeric>
#include <vector>

namespace emsar {

struct SetDesc {                 // 32 bytes, read by the kernel
    uint32_t tid_off;            // into g_tid / g_u
    uint32_t row_off;            // into row_w
    uint32_t ent_off;            // into ent / crow
    uint32_t rp_off, cp_off;     // into rp (n_r + 1) / cp (n_t + 1)
    uint32_t n_t, n_r, nnz;
};

constexpr int kSetClasses = 3;
constexpr int kSetThreads[kSetClasses] = {64, 256, 512};
constexpr size_t kSetLdsCap[kSetClasses] = {6 * 1024, 48 * 1024, 156 * 1024};
constexpr int kSetRedDoubles = 64;   // LDS scratch of the workgroup reductions (4 values x 16 waves)

// bytes of LDS one resident set needs: 8 transcript vectors (A, B, C, den, u and the Newton step's z, Hp, 1/diag), 3 row
// vectors (w, R, R/S^2), the reduction scratch and the four 16-bit index arrays
inline size_t set_lds_bytes(size_t n_t, size_t n_r, size_t nnz) {
    size_t b = 8 * (8 * n_t + 3 * n_r + (size_t)kSetRedDoubles) + 2 * ((n_r + 1) + (n_t + 1) + 2 * nnz);
    return (b + 15) & ~(size_t)15;
}

enum TidKind : uint8_t { KIND_CLOSED = 0, KIND_RESIDENT = 1, KIND_STREAMED = 2, KIND_CLUSTER = 3 };

// ---- workgroup-cluster sets ----
constexpr int kClusterThreads = 512;
constexpr int kClusterMaxWg = 8;
constexpr size_t kClusterLdsCap = 156 * 1024;
constexpr size_t kClusterMaxT = 16384;
struct ClusterDesc {             // 64 bytes, read by the kernel
    uint32_t tid_off;            // into g_tid / g_u
    uint32_t n_t, n_r, nnz;
    uint32_t g;                  // workgroups: 2, 4 or 8
    uint32_t rp_off;             // into rp: n_r + 1 offsets into the set's ent (relative to ent_off)
    uint32_t ent_off;            // into ent (local transcript ids) and crow
    uint32_t row_off;            // into row_w
    uint32_t part_off;           // into part: g + 1 row boundaries
    uint32_t cp_off;             // into cp: g x (n_t + 1) offsets into the set's crow (relative to ent_off): workgroup h's rows that hold transcript i
    uint32_t blk0;               // first workgroup of the set in the launch
    uint32_t bar;                // index of the set's barrier word
    uint64_t scratch_off;        // doubles: 2 n_t (published points) + g n_t (partial column sums) + 16 g (partial scalars)
    uint32_t pad[2];
};
static_assert(sizeof(ClusterDesc) == 64, "ClusterDesc must stay 64 bytes");
// LDS of one workgroup: the whole current point, the weights of its own rows, six vectors over its own transcripts (three points,
// den, u, scratch), the reduction scratch
inline size_t cluster_lds_bytes(size_t n_t, size_t max_rows_wg, size_t g) {
    const size_t own = (n_t + g - 1) / g;
    return 8 * (n_t + max_rows_wg + 6 * own + (size_t)kSetRedDoubles + 16);
}
struct ClusterSets {
    std::vector<ClusterDesc> desc;
    std::vector<int32_t> g_tid;
    std::vector<double> g_u, row_w;
    std::vector<uint32_t> rp, cp, part, blk_set;     // blk_set: workgroup of the launch -> set
    std::vector<uint16_t> ent, crow;
    uint64_t scratch_doubles = 0;
    size_t max_lds = 0;
    int64_t n_tids = 0, rows_stored = 0;
};

struct ResidentSets {
    std::vector<SetDesc> desc[kSetClasses];
    size_t max_lds[kSetClasses] = {0, 0, 0};
    std::vector<int32_t> g_tid;
    std::vector<double> g_u, row_w;
    std::vector<uint16_t> rp, ent, cp, crow;
    std::vector<uint8_t> kind;        // [n_tx]
    std::vector<double> usum;         // [n_tx] sum of R over the rows whose only transcript is t
    int64_t n_components = 0;         // with at least two transcripts
    int64_t n_streamed_sets = 0, n_streamed_tids = 0, n_resident_tids = 0, n_closed_tids = 0;
    int64_t rows_in = 0, rows_stored = 0;   // multi-transcript rows with weight before / after merging
    bool giant = false;               // one component holds most of the transcripts: nothing was packed, stream everything
    ClusterSets CL;                   // components solved by a cluster of workgroups
    int64_t n_cluster_sets() const { return (int64_t)CL.desc.size(); }
    int64_t n_resident() const { return (int64_t)(desc[0].size() + desc[1].size() + desc[2].size()); }
};

namespace detail {
inline int32_t uf_find(std::vector<int32_t> &p, int32_t x) {
    while (p[(size_t)x] != x) { p[(size_t)x] = p[(size_t)p[(size_t)x]]; x = p[(size_t)x]; }
    return x;
}
// One component as a cluster record.  tids: its transcripts ascending; rep / repw: its distinct rows (indices into lptr) and their
// summed weights; lst / lptr: the rows' sorted local transcript ids.  False (nothing written) if no cluster size fits the LDS.
inline bool pack_cluster(ResidentSets &out, const int32_t *tids, size_t nt, const std::vector<uint32_t> &rep, const std::vector<double> &repw,
                         const std::vector<uint16_t> &lst, const std::vector<uint32_t> &lptr, size_t nnz) {
    ClusterSets &C = out.CL;
    const size_t nr = rep.size();
    if (nr == 0 || nt < 2 || nnz >= 0x7FFFFFFFull) return false;
    // the smallest cluster whose workgroups hold their share: rows dealt in contiguous ranges of about nnz / g entries
    size_t g = 0;
    std::vector<uint32_t> part;
    for (size_t cand : {(size_t)2, (size_t)4, (size_t)8}) {
        part.assign(1, 0);
        size_t acc = 0, max_rows = 0;
        for (size_t j = 0; j < nr; j++) {
            acc += lptr[rep[j] + 1] - lptr[rep[j]];
            if (part.size() < cand && acc * cand >= nnz * part.size()) part.push_back((uint32_t)(j + 1));
        }
        while (part.size() < cand) part.push_back((uint32_t)nr);
        part.push_back((uint32_t)nr);
        for (size_t h = 0; h < cand; h++) max_rows = std::max<size_t>(max_rows, part[h + 1] - part[h]);
        if (max_rows <= 65535 && cluster_lds_bytes(nt, max_rows, cand) <= kClusterLdsCap) { g = cand; break; }
    }
    if (g == 0) return false;
    if (C.ent.size() + nnz >= 0xFFFFFFFFull || C.cp.size() + g * (nt + 1) >= 0xFFFFFFFFull || C.rp.size() + nr + 1 >= 0xFFFFFFFFull) return false;
    ClusterDesc d{};
    d.tid_off = (uint32_t)C.g_tid.size(); d.n_t = (uint32_t)nt; d.n_r = (uint32_t)nr; d.nnz = (uint32_t)nnz; d.g = (uint32_t)g;
    d.rp_off = (uint32_t)C.rp.size(); d.ent_off = (uint32_t)C.ent.size(); d.row_off = (uint32_t)C.row_w.size();
    d.part_off = (uint32_t)C.part.size(); d.cp_off = (uint32_t)C.cp.size();
    d.blk0 = (uint32_t)C.blk_set.size(); d.bar = (uint32_t)C.desc.size();
    d.scratch_off = C.scratch_doubles;
    C.scratch_doubles += 2 * nt + g * nt + 16 * g;
    for (size_t h = 0; h <= g; h++) C.part.push_back(part[h]);
    uint32_t pos = 0;
    for (size_t j = 0; j < nr; j++) {
        C.rp.push_back(pos);
        C.row_w.push_back(repw[j]);
        for (uint32_t k = lptr[rep[j]]; k < lptr[rep[j] + 1]; k++) { C.ent.push_back(lst[k]); pos++; }
    }
    C.rp.push_back(pos);
    // per workgroup: the transpose of its rows (row ids local to the workgroup), laid out back to back in crow
    C.crow.resize(C.ent.size());
    uint32_t base = 0;
    std::vector<uint32_t> cnt(nt + 1);
    for (size_t h = 0; h < g; h++) {
        std::fill(cnt.begin(), cnt.end(), 0u);
        for (uint32_t j = part[h]; j < part[h + 1]; j++)
            for (uint32_t k = lptr[rep[j]]; k < lptr[rep[j] + 1]; k++) cnt[(size_t)lst[k] + 1]++;
        for (size_t i = 0; i < nt; i++) cnt[i + 1] += cnt[i];
        for (size_t i = 0; i <= nt; i++) C.cp.push_back(base + cnt[i]);
        std::vector<uint32_t> fill(cnt.begin(), cnt.end() - 1);
        for (uint32_t j = part[h]; j < part[h + 1]; j++)
            for (uint32_t k = lptr[rep[j]]; k < lptr[rep[j] + 1]; k++) C.crow[d.ent_off + base + fill[(size_t)lst[k]]++] = (uint16_t)(j - part[h]);
        base += cnt[nt];
    }
    for (size_t i = 0; i < nt; i++) {
        const int32_t t = tids[i];
        C.g_tid.push_back(t);
        C.g_u.push_back(out.usum[(size_t)t]);
        out.kind[(size_t)t] = KIND_CLUSTER;
    }
    for (size_t h = 0; h < g; h++) C.blk_set.push_back((uint32_t)C.desc.size());
    size_t max_rows = 0;
    for (size_t h = 0; h < g; h++) max_rows = std::max<size_t>(max_rows, part[h + 1] - part[h]);
    C.max_lds = std::max(C.max_lds, cluster_lds_bytes(nt, max_rows, g));
    C.desc.push_back(d);
    C.n_tids += (int64_t)nt; C.rows_stored += (int64_t)nr;
    return true;
}
}  // namespace detail

// wgt[r] >= 0: weight of row r inside the likelihood (0 = the row does not couple anything).
// Returns 0; the CSR is assumed validated (validate_csr).
inline int build_sets(int64_t n_rows, int32_t n_tx, const uint64_t *row_ptr, const int32_t *col_idx, const int32_t *wgt,
                      ResidentSets &out) {
    out = ResidentSets();
    // measured (tests/test_set_solver.py, 2500..9000-transcript families): 32 us per pass in a cluster against 19 us through the
    // streaming passes -- a cluster barrier is three dependent trips to memory that bypass the (non-coherent) L2s, ~2 us each.
    // Correct and bit-reproducible, but not faster: opt-in.
    bool use_cluster = false;
    if (const char *e = getenv("EMSAR_HIP_CLUSTER")) use_cluster = atoi(e) != 0;
    size_t par_one_wave = 128;           // up to this many rows / transcripts a set is run by ONE wave (no workgroup barriers): 0.351 -> 0.325 s on bench.py time_to_mle
    if (const char *e = getenv("EMSAR_HIP_SET_PAR")) { int v = atoi(e); if (v >= 64 && v <= 4096) par_one_wave = (size_t)v; }
    const size_t T = (size_t)n_tx;
    out.kind.assign(T, KIND_CLOSED);
    out.usum.assign(T, 0.0);
    std::vector<int32_t> parent(T), csize(T, 1);
    std::iota(parent.begin(), parent.end(), 0);
    std::vector<int64_t> multi_rows;   // rows with weight and >= 2 distinct transcripts
    int32_t largest = 1;
    for (int64_t r = 0; r < n_rows; r++) {
        // a read-level matrix with a few cross-family reads is ONE component: stop looking as soon as more than
        // half of the transcripts hang together (checked every 2^20 rows) -- the streaming passes take it all
        if ((r & 0xFFFFF) == 0xFFFFF && (int64_t)largest * 2 > (int64_t)n_tx && n_tx > 4096) {
            out.giant = true;
            std::fill(out.kind.begin(), out.kind.end(), (uint8_t)KIND_STREAMED);
            std::fill(out.usum.begin(), out.usum.end(), 0.0);
            out.n_components = 1; out.n_streamed_sets = 1; out.n_streamed_tids = (int64_t)T;
            return 0;
        }
        const int32_t x = wgt ? wgt[r] : 1;
        const uint64_t b = row_ptr[r], e = row_ptr[r + 1];
        if (x <= 0 || b == e) continue;
        const int32_t first = col_idx[b];
        bool single = true;
        for (uint64_t k = b + 1; k < e; k++) if (col_idx[k] != first) { single = false; break; }
        if (single) { out.usum[(size_t)first] += (double)x; continue; }
        multi_rows.push_back(r);
        int32_t ra = detail::uf_find(parent, first);
        for (uint64_t k = b + 1; k < e; k++) {
            int32_t rb = detail::uf_find(parent, col_idx[k]);
            if (rb != ra) {
                if (rb < ra) std::swap(ra, rb);
                parent[(size_t)rb] = ra;
                csize[(size_t)ra] += csize[(size_t)rb];
                largest = std::max(largest, csize[(size_t)ra]);
            }
        }
    }
    out.rows_in = (int64_t)multi_rows.size();
    // components with >= 2 transcripts, numbered by their smallest tid (the root: unions keep the smaller id)
    std::vector<int32_t> comp_of(T, -1);
    std::vector<int32_t> comp_nt;
    for (size_t t = 0; t < T; t++) {
        int32_t root = detail::uf_find(parent, (int32_t)t);
        if ((size_t)root == t) continue;
        if (comp_of[(size_t)root] < 0) { comp_of[(size_t)root] = (int32_t)comp_nt.size(); comp_nt.push_back(1); }
        comp_of[t] = comp_of[(size_t)root];
        comp_nt[(size_t)comp_of[t]]++;
    }
    const size_t NC = comp_nt.size();
    out.n_components = (int64_t)NC;
    if (NC == 0) { out.n_closed_tids = (int64_t)T; return 0; }
    // bucket transcripts and rows by component (counting sort keeps tids ascending)
    std::vector<uint64_t> tptr(NC + 1, 0), rptr(NC + 1, 0), nnz_of(NC, 0);
    for (size_t c = 0; c < NC; c++) tptr[c + 1] = tptr[c] + (uint64_t)comp_nt[c];
    std::vector<int32_t> tids(tptr[NC]);
    {
        std::vector<uint64_t> fill(tptr.begin(), tptr.end() - 1);
        for (size_t t = 0; t < T; t++) if (comp_of[t] >= 0) tids[fill[(size_t)comp_of[t]]++] = (int32_t)t;
    }
    for (int64_t r : multi_rows) { size_t c = (size_t)comp_of[(size_t)col_idx[row_ptr[r]]]; rptr[c + 1]++; nnz_of[c] += row_ptr[r + 1] - row_ptr[r]; }
    for (size_t c = 0; c < NC; c++) rptr[c + 1] += rptr[c];
    std::vector<int64_t> rows(rptr[NC]);
    {
        std::vector<uint64_t> fill(rptr.begin(), rptr.end() - 1);
        for (int64_t r : multi_rows) rows[fill[(size_t)comp_of[(size_t)col_idx[row_ptr[r]]]]++] = r;
    }
    std::vector<int64_t>().swap(multi_rows);
    std::vector<int32_t> local(T, -1);
    const size_t cap = kSetLdsCap[kSetClasses - 1];
    std::vector<uint16_t> lst;            // local sorted tid lists of the component's rows, back to back
    std::vector<uint32_t> lptr, order;
    for (size_t c = 0; c < NC; c++) {
        const size_t nt = (size_t)comp_nt[c];
        const size_t nr0 = (size_t)(rptr[c + 1] - rptr[c]);
        auto stream = [&]() {
            for (uint64_t q = tptr[c]; q < tptr[c + 1]; q++) out.kind[(size_t)tids[q]] = KIND_STREAMED;
            out.n_streamed_sets++; out.n_streamed_tids += (int64_t)nt;
        };
        // cheap bound first: even with every row merged away the transcript vectors must fit (one workgroup's LDS, or a cluster's),
        // and the local lists of a set worth packing are small
        const bool one_wg_possible = set_lds_bytes(nt, 1, 2) <= cap;
        const bool cluster_possible = use_cluster && nt <= kClusterMaxT && cluster_lds_bytes(nt, 1, kClusterMaxWg) <= kClusterLdsCap;
        if (nt > 65535 || (!one_wg_possible && !cluster_possible) || nnz_of[c] > (uint64_t)16 * 1024 * 1024) { stream(); continue; }
        for (size_t i = 0; i < nt; i++) local[(size_t)tids[tptr[c] + i]] = (int32_t)i;
        lst.clear(); lptr.assign(1, 0);
        for (size_t j = 0; j < nr0; j++) {
            const int64_t r = rows[rptr[c] + j];
            const size_t at = lst.size();
            for (uint64_t k = row_ptr[r]; k < row_ptr[r + 1]; k++) lst.push_back((uint16_t)local[(size_t)col_idx[k]]);
            std::sort(lst.begin() + (ptrdiff_t)at, lst.end());
            lptr.push_back((uint32_t)lst.size());
        }
        // merge identical rows: order rows by (length, list), add the weights of equal neighbours
        order.resize(nr0);
        std::iota(order.begin(), order.end(), 0u);
        auto cmp3 = [&](uint32_t a, uint32_t b) -> int {
            const uint32_t la = lptr[a + 1] - lptr[a], lb = lptr[b + 1] - lptr[b];
            if (la != lb) return la < lb ? -1 : 1;
            for (uint32_t k = 0; k < la; k++) {
                const uint16_t x = lst[lptr[a] + k], y = lst[lptr[b] + k];
                if (x != y) return x < y ? -1 : 1;
            }
            return 0;
        };
        std::sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { int q = cmp3(a, b); return q ? q < 0 : a < b; });
        std::vector<uint32_t> rep;        // representative (first) row of each distinct list
        std::vector<double> repw;
        size_t nnz = 0;
        for (size_t j = 0; j < nr0; j++) {
            const uint32_t a = order[j];
            const double x = (double)(wgt ? wgt[rows[rptr[c] + a]] : 1);
            if (!rep.empty() && cmp3(rep.back(), a) == 0) { repw.back() += x; continue; }
            rep.push_back(a); repw.push_back(x);
            nnz += lptr[a + 1] - lptr[a];
        }
        const size_t nr = rep.size();
        const size_t bytes = set_lds_bytes(nt, nr, nnz);
        const bool offsets_fit = out.ent.size() + nnz < 0xFFFFFFFFull && out.row_w.size() + nr < 0xFFFFFFFFull &&
                                 out.g_tid.size() + nt < 0xFFFFFFFFull && out.rp.size() + nr + 1 < 0xFFFFFFFFull;
        if (bytes > cap || nr > 65535 || nnz > 65535 || !offsets_fit) {
            bool packed = false;
            if (cluster_possible) packed = detail::pack_cluster(out, tids.data() + tptr[c], nt, rep, repw, lst, lptr, nnz);
            for (size_t i = 0; i < nt; i++) local[(size_t)tids[tptr[c] + i]] = -1;
            if (!packed) stream();
            continue;
        }
        // class by LDS footprint AND by parallelism: a pass is a chain of dependent LDS accesses per row / per
        // transcript, so a set wants about one row and one transcript per thread
        int cls = 0;
        while (bytes > kSetLdsCap[cls]) cls++;
        const size_t par = std::max(nt, nr);
        if (par > par_one_wave && cls < 1) cls = 1;
        if (par > 512 && cls < 2) cls = 2;
        SetDesc d;
        d.tid_off = (uint32_t)out.g_tid.size(); d.row_off = (uint32_t)out.row_w.size(); d.ent_off = (uint32_t)out.ent.size();
        d.rp_off = (uint32_t)out.rp.size(); d.cp_off = (uint32_t)out.cp.size();
        d.n_t = (uint32_t)nt; d.n_r = (uint32_t)nr; d.nnz = (uint32_t)nnz;
        std::vector<uint32_t> cnt(nt + 1, 0);
        uint32_t pos = 0;
        for (size_t j = 0; j < nr; j++) {
            out.rp.push_back((uint16_t)pos);
            out.row_w.push_back(repw[j]);
            for (uint32_t k = lptr[rep[j]]; k < lptr[rep[j] + 1]; k++) { out.ent.push_back(lst[k]); cnt[(size_t)lst[k] + 1]++; pos++; }
        }
        out.rp.push_back((uint16_t)pos);
        for (size_t i = 0; i < nt; i++) cnt[i + 1] += cnt[i];
        for (size_t i = 0; i <= nt; i++) out.cp.push_back((uint16_t)cnt[i]);
        out.crow.resize(out.ent.size());
        {
            std::vector<uint32_t> fill(cnt.begin(), cnt.end() - 1);
            for (size_t j = 0; j < nr; j++)
                for (uint32_t k = lptr[rep[j]]; k < lptr[rep[j] + 1]; k++) out.crow[d.ent_off + fill[(size_t)lst[k]]++] = (uint16_t)j;
        }
        for (size_t i = 0; i < nt; i++) {
            const int32_t t = tids[tptr[c] + i];
            out.g_tid.push_back(t);
            out.g_u.push_back(out.usum[(size_t)t]);
            out.kind[(size_t)t] = KIND_RESIDENT;
            local[(size_t)t] = -1;
        }
        out.desc[cls].push_back(d);
        out.max_lds[cls] = std::max(out.max_lds[cls], bytes);
        out.n_resident_tids += (int64_t)nt;
        out.rows_stored += (int64_t)nr;
    }
    // largest first inside a class: the long-running workgroups start first
    for (auto &v : out.desc)
        std::stable_sort(v.begin(), v.end(), [](const SetDesc &a, const SetDesc &b) { return a.nnz + a.n_t > b.nnz + b.n_t; });
    out.n_closed_tids = (int64_t)T - out.n_resident_tids - out.n_streamed_tids - out.CL.n_tids;
    return 0;
}

// reference check of a packed record against the CSR it came from: every weighted multi-transcript row of a
// resident component must be found (with its multiplicities) and the weights must add up.  Returns 0 if consistent.
inline int check_sets(int64_t n_rows, int32_t n_tx, const uint64_t *row_ptr, const int32_t *col_idx, const int32_t *wgt,
                      const ResidentSets &S) {
    const size_t T = (size_t)n_tx;
    if (S.kind.size() != T || S.usum.size() != T) return 1;
    if (S.giant) {                     // nothing packed, everything streamed
        for (size_t t = 0; t < T; t++) if (S.kind[t] != KIND_STREAMED) return 15;
        return S.n_resident() == 0 ? 0 : 15;
    }
    std::vector<int32_t> set_of(T, -1), loc(T, -1);
    std::vector<const SetDesc *> all;
    for (const auto &v : S.desc) for (const auto &d : v) all.push_back(&d);
    for (size_t s = 0; s < all.size(); s++) {
        const SetDesc &d = *all[s];
        if (d.n_t < 2 || d.n_r < 1) return 2;
        if (S.rp[d.rp_off] != 0 || S.rp[d.rp_off + d.n_r] != d.nnz || S.cp[d.cp_off] != 0 || S.cp[d.cp_off + d.n_t] != d.nnz) return 3;
        for (uint32_t i = 0; i < d.n_t; i++) {
            int32_t t = S.g_tid[d.tid_off + i];
            if (t < 0 || t >= n_tx || set_of[(size_t)t] >= 0 || S.kind[(size_t)t] != KIND_RESIDENT) return 4;
            if (i && t <= S.g_tid[d.tid_off + i - 1]) return 4;
            set_of[(size_t)t] = (int32_t)s; loc[(size_t)t] = (int32_t)i;
            if (S.g_u[d.tid_off + i] != S.usum[(size_t)t]) return 5;
        }
        // CSC is the transpose of CSR
        std::vector<uint32_t> seen(d.n_t, 0);
        for (uint32_t j = 0; j < d.n_r; j++) {
            if (S.rp[d.rp_off + j] > S.rp[d.rp_off + j + 1]) return 6;
            for (uint32_t k = S.rp[d.rp_off + j]; k < S.rp[d.rp_off + j + 1]; k++) {
                uint32_t i = S.ent[d.ent_off + k];
                if (i >= d.n_t) return 6;
                uint32_t q = S.cp[d.cp_off + i] + seen[i]++;
                if (q >= S.cp[d.cp_off + i + 1] || S.crow[d.ent_off + q] != j) return 7;
            }
        }
        for (uint32_t i = 0; i < d.n_t; i++) if (seen[i] != (uint32_t)(S.cp[d.cp_off + i + 1] - S.cp[d.cp_off + i])) return 7;
    }
    for (size_t t = 0; t < T; t++) if ((S.kind[t] == KIND_RESIDENT) != (set_of[t] >= 0)) return 8;
    // every weighted row: single -> usum; multi -> all tids of one kind; resident rows are present in their set
    std::vector<double> usum(T, 0.0), wsum(all.size(), 0.0), wfound(all.size(), 0.0);
    std::vector<uint16_t> key;
    for (int64_t r = 0; r < n_rows; r++) {
        const int32_t x = wgt ? wgt[r] : 1;
        const uint64_t b = row_ptr[r], e = row_ptr[r + 1];
        if (x <= 0 || b == e) continue;
        bool single = true;
        for (uint64_t k = b + 1; k < e; k++) if (col_idx[k] != col_idx[b]) single = false;
        if (single) { usum[(size_t)col_idx[b]] += x; continue; }
        const uint8_t kd = S.kind[(size_t)col_idx[b]];
        if (kd == KIND_CLOSED) return 9;
        for (uint64_t k = b; k < e; k++) if (S.kind[(size_t)col_idx[k]] != kd) return 10;
        if (kd != KIND_RESIDENT) continue;
        const int32_t s = set_of[(size_t)col_idx[b]];
        key.clear();
        for (uint64_t k = b; k < e; k++) { if (set_of[(size_t)col_idx[k]] != s) return 11; key.push_back((uint16_t)loc[(size_t)col_idx[k]]); }
        std::sort(key.begin(), key.end());
        const SetDesc &d = *all[(size_t)s];
        bool found = false;
        for (uint32_t j = 0; j < d.n_r && !found; j++) {
            const uint32_t a = S.rp[d.rp_off + j], len = S.rp[d.rp_off + j + 1] - a;
            if (len == key.size() && std::equal(key.begin(), key.end(), S.ent.begin() + d.ent_off + a)) found = true;
        }
        if (!found) return 12;
        wsum[(size_t)s] += x;
    }
    for (size_t t = 0; t < T; t++) if (usum[t] != S.usum[t]) return 13;
    for (size_t s = 0; s < all.size(); s++) {
        const SetDesc &d = *all[s];
        for (uint32_t j = 0; j < d.n_r; j++) wfound[s] += S.row_w[d.row_off + j];
        if (wfound[s] != wsum[s]) return 14;
    }
    // ---- cluster records: structure, per-workgroup transposes, and every weighted row of such a component present with its weight ----
    const ClusterSets &C = S.CL;
    std::vector<int32_t> cset(T, -1), cloc(T, -1);
    size_t blk = 0;
    for (size_t s = 0; s < C.desc.size(); s++) {
        const ClusterDesc &d = C.desc[s];
        if ((d.g != 2 && d.g != 4 && d.g != 8) || d.n_t < 2 || d.n_r < 1 || d.blk0 != blk || d.bar != s) return 16;
        for (uint32_t h = 0; h < d.g; h++) if (C.blk_set[blk + h] != s) return 16;
        blk += d.g;
        for (uint32_t i = 0; i < d.n_t; i++) {
            const int32_t t = C.g_tid[d.tid_off + i];
            if (t < 0 || t >= n_tx || cset[(size_t)t] >= 0 || S.kind[(size_t)t] != KIND_CLUSTER) return 16;
            if (i && t <= C.g_tid[d.tid_off + i - 1]) return 16;
            cset[(size_t)t] = (int32_t)s; cloc[(size_t)t] = (int32_t)i;
            if (C.g_u[d.tid_off + i] != S.usum[(size_t)t]) return 16;
        }
        const uint32_t *part = &C.part[d.part_off], *rp = &C.rp[d.rp_off];
        if (part[0] != 0 || part[d.g] != d.n_r || rp[0] != 0 || rp[d.n_r] != d.nnz) return 17;
        for (uint32_t h = 0; h < d.g; h++) if (part[h] > part[h + 1] || part[h + 1] - part[h] > 65535) return 17;
        for (uint32_t j = 0; j < d.n_r; j++) if (rp[j] > rp[j + 1]) return 17;
        size_t max_rows = 0;
        for (uint32_t h = 0; h < d.g; h++) max_rows = std::max<size_t>(max_rows, part[h + 1] - part[h]);
        if (cluster_lds_bytes(d.n_t, max_rows, d.g) > kClusterLdsCap) return 17;
        uint32_t base = 0;
        std::vector<uint32_t> seen(d.n_t);
        for (uint32_t h = 0; h < d.g; h++) {                  // workgroup h: the transpose of its rows
            const uint32_t *cp = &C.cp[d.cp_off + (size_t)h * (d.n_t + 1)];
            if (cp[0] != base) return 18;
            std::fill(seen.begin(), seen.end(), 0u);
            for (uint32_t j = part[h]; j < part[h + 1]; j++)
                for (uint32_t k = rp[j]; k < rp[j + 1]; k++) {
                    const uint32_t i = C.ent[d.ent_off + k];
                    if (i >= d.n_t) return 18;
                    const uint32_t q = cp[i] + seen[i]++;
                    if (q >= cp[i + 1] || C.crow[d.ent_off + q] != j - part[h]) return 18;
                }
            for (uint32_t i = 0; i < d.n_t; i++) if (seen[i] != cp[i + 1] - cp[i]) return 18;
            base = cp[d.n_t];
        }
        if (base != d.nnz) return 18;
    }
    if (blk != C.blk_set.size()) return 16;
    for (size_t t = 0; t < T; t++) if ((S.kind[t] == KIND_CLUSTER) != (cset[t] >= 0)) return 19;
    {
        std::vector<double> cw(C.desc.size(), 0.0), cf(C.desc.size(), 0.0);
        std::vector<std::vector<std::vector<uint16_t>>> keys(C.desc.size());     // the stored rows of each set, sorted, for the look-up
        for (size_t s = 0; s < C.desc.size(); s++) {
            const ClusterDesc &d = C.desc[s];
            for (uint32_t j = 0; j < d.n_r; j++) {
                keys[s].emplace_back(C.ent.begin() + d.ent_off + C.rp[d.rp_off + j], C.ent.begin() + d.ent_off + C.rp[d.rp_off + j + 1]);
                cf[s] += C.row_w[d.row_off + j];
            }
            std::sort(keys[s].begin(), keys[s].end());
        }
        for (int64_t r = 0; r < n_rows; r++) {
            const int32_t x = wgt ? wgt[r] : 1;
            const uint64_t b = row_ptr[r], e = row_ptr[r + 1];
            if (x <= 0 || b == e || S.kind[(size_t)col_idx[b]] != KIND_CLUSTER) continue;
            bool single = true;
            for (uint64_t k = b + 1; k < e; k++) if (col_idx[k] != col_idx[b]) single = false;
            if (single) continue;
            const int32_t s = cset[(size_t)col_idx[b]];
            key.clear();
            for (uint64_t k = b; k < e; k++) { if (cset[(size_t)col_idx[k]] != s) return 20; key.push_back((uint16_t)cloc[(size_t)col_idx[k]]); }
            std::sort(key.begin(), key.end());
            if (!std::binary_search(keys[(size_t)s].begin(), keys[(size_t)s].end(), key)) return 20;
            cw[(size_t)s] += x;
        }
        for (size_t s = 0; s < C.desc.size(); s++) if (cw[s] != cf[s]) return 21;
    }
    return 0;
}

}  // namespace emsar
#endif
