// renumber.hpp -- transcript numbering by co-occurrence (pure C++, no HIP calls).
//
// The TILED layout (layout_tiled.hpp) packs the transcripts a row hits into "block entries": an operand names a block of three
// CONSECUTIVE dictionary slots and a subset of it, and a tile's dictionary is a contiguous tid range.  Both rely on transcripts
// that occur together being neighbours in tid space.  The caller's numbering (the order of the '@' lines of the rsh, i.e. of the
// cDNA FASTA) usually has that property -- isoforms of a gene follow each other -- but nothing guarantees it, and what a row really
// is (update_ReadCounts, /root/reference/src/emsar_functions.c:838-943: the sorted tid multiset of one read) is an isoform SUBSET,
// not a run.  So the library numbers the transcripts itself at upload time and maps theta / den back at the ABI:
//
//   1. pair counts: over a strided sample of the multi-transcript rows, the pairs of neighbouring ids of every (sorted) row plus
//      the pair that closes the cycle, counted in an open-addressing table;
//   2. clusters: Kruskal over the pairs seen at least twice, heaviest first, with a cap on the cluster size -- a cross-family
//      hit of a single read (weight 1) joins nothing, a family of isoforms (or a family and its paralogs) becomes one cluster;
//   3. order inside a cluster: a greedy chain -- start at the member with the largest total weight, go to the unvisited
//      neighbour joined by the heaviest pair, restart at the heaviest unvisited member when there is none;
//   4. clusters (and the transcripts that belong to none) keep the order of their smallest original id: whatever global
//      locality the caller's numbering had survives;
//   5. the new numbering is used only if it needs fewer block entries for the sampled rows than the caller's (a numbering that is
//      already good -- consecutive windows -- is left alone).
#pragma once
#include <algorithm>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <new>
#include <numeric>
#include <system_error>
#include <thread>
#include <vector>

namespace emsar {

struct RenumberStats {
    int64_t rows_sampled = 0, pairs_sampled = 0, pairs_distinct = 0, pairs_kept = 0, clusters = 0, largest = 0;
    int64_t entries_before = 0, entries_after = 0, ids_sampled = 0;   // block entries of the sampled rows under the two numberings
    bool applied = false;
};

constexpr int kRenumberCap = 128;          // transcripts per cluster at most (the sort block of build_tiled; a dictionary holds 360)
constexpr int64_t kRenumberMaxPairs = (int64_t)12 << 20;   // sampled pairs at most: config 3 at 48 M pairs spent 3 s here on one thread, at 12 M 0.8 s with the
                                                           // same clusters on the shuffled family law (a pair is kept from two sightings on)

// block entries of one sorted row under a numbering (the rule of layout_tiled.hpp: slots_to_entries on tids), blk slots per block
inline int row_entries(const int32_t *ids, int n, int blk) {
    int n_ent = 0, cur_b = -1;
    uint32_t cur_m = 0;
    for (int i = 0; i < n; i++) {
        const int b = ids[i] / blk;
        const uint32_t bit = 1u << (ids[i] % blk);
        if (b != cur_b || (cur_m & bit)) { n_ent++; cur_b = b; cur_m = bit; } else cur_m |= bit;
    }
    return n_ent;
}

// new_of_old[t] = the library's id of the caller's transcript t; left empty when the caller's numbering is kept
inline void cooccurrence_order(int64_t n_rows, int32_t n_tx, const uint64_t *row_ptr, const int32_t *col_idx, int max_row_len, int blk,
                               std::vector<int32_t> &new_of_old, RenumberStats &st) {
    new_of_old.clear();
    st = RenumberStats();
    if (n_tx < 2 * blk || n_rows == 0) return;
    // ---- 1. sample + pair counts ----
    int64_t multi_ids = 0;
    for (int64_t r = 0; r < n_rows; r++) {
        const uint64_t len = row_ptr[r + 1] - row_ptr[r];
        if (len >= 2 && len <= (uint64_t)max_row_len) multi_ids += (int64_t)len;
    }
    if (multi_ids == 0) return;
    int64_t max_pairs = kRenumberMaxPairs;
    if (const char *ev = getenv("EMSAR_HIP_RENUMBER_PAIRS")) { const long long v = atoll(ev); if (v >= 1024) max_pairs = (int64_t)v; }
    const int64_t stride = std::max<int64_t>(1, (multi_ids + max_pairs - 1) / max_pairs);
    // the rows of the sample (one cheap sequential scan), then the pair counts on up to 16 host threads: every thread turns its share of
    // the sampled rows into pair keys (phase 1), then owns the keys of one hash class and counts them in a table of its own (phase 2: the
    // random-access inserts are what costs -- 2 s on one thread for 12 M pairs).  The set of (pair, count) is the same for any thread
    // count, and the edges are sorted by a total order below, so the numbering does not depend on the number of threads.
    std::vector<int64_t> sampled;
    {
        int64_t seen_multi = 0;
        for (int64_t r = 0; r < n_rows; r++) {
            const uint64_t len = row_ptr[r + 1] - row_ptr[r];
            if (len < 2 || len > (uint64_t)max_row_len) continue;
            if (seen_multi++ % stride) continue;
            sampled.push_back(r);
        }
    }
    st.rows_sampled = (int64_t)sampled.size();
    int nt = 1;
    {
        unsigned hw = std::thread::hardware_concurrency();
        nt = (int)(hw ? std::min(hw, 16u) : 1u);
        if (sampled.size() < ((size_t)1 << 15)) nt = 1;
        if (const char *ev = getenv("EMSAR_HOST_THREADS")) { const int v = atoi(ev); if (v >= 1) nt = std::min(v, 64); }
    }
    auto hash = [](uint64_t k) { k ^= k >> 31; k *= 0x9E3779B97F4A7C15ull; k ^= k >> 29; k *= 0xBF58476D1CE4E5B9ull; k ^= k >> 32; return k; };
    auto on_threads = [&](auto fn) {
        std::vector<std::thread> pool;
        bool failed = false;
        for (int t = 1; t < nt; t++) {
            try { pool.emplace_back([&, t] { try { fn(t); } catch (const std::bad_alloc &) { failed = true; } }); }
            catch (const std::system_error &) { try { fn(t); } catch (const std::bad_alloc &) { failed = true; } }
        }
        try { fn(0); } catch (const std::bad_alloc &) { failed = true; }
        for (auto &th : pool) th.join();
        if (failed) throw std::bad_alloc();
    };
    std::vector<std::vector<uint64_t>> keys((size_t)nt);
    on_threads([&](int t) {
        std::vector<int32_t> tmp;
        std::vector<uint64_t> &K = keys[(size_t)t];
        auto put = [&](int32_t a, int32_t b) { if (a != b) K.push_back(a < b ? (uint64_t)(uint32_t)a << 32 | (uint32_t)b : (uint64_t)(uint32_t)b << 32 | (uint32_t)a); };
        const size_t lo = sampled.size() * (size_t)t / (size_t)nt, hi = sampled.size() * (size_t)(t + 1) / (size_t)nt;
        for (size_t q = lo; q < hi; q++) {
            const int64_t r = sampled[q];
            tmp.assign(col_idx + row_ptr[r], col_idx + row_ptr[r + 1]);
            std::sort(tmp.begin(), tmp.end());
            tmp.erase(std::unique(tmp.begin(), tmp.end()), tmp.end());
            const size_t n = tmp.size();
            for (size_t i = 0; i + 1 < n; i++) put(tmp[i], tmp[i + 1]);
            if (n > 2) put(tmp[0], tmp[n - 1]);
        }
    });
    for (const auto &K : keys) st.pairs_sampled += (int64_t)K.size();
    const uint32_t min_cnt = st.pairs_sampled >= (1 << 20) ? 2u : 1u;
    struct Edge { int32_t a, b; uint32_t w; };
    std::vector<std::vector<Edge>> part((size_t)nt);
    std::vector<int64_t> part_used((size_t)nt, 0);
    on_threads([&](int t) {
        struct Slot { uint64_t key; uint32_t cnt; };
        uint64_t cap = 1 << 14, used = 0;
        std::vector<Slot> table((size_t)cap, Slot{~0ull, 0});
        for (const auto &K : keys)
            for (const uint64_t key : K) {
                const uint64_t hh = hash(key);
                if ((int)((hh >> 40) % (uint64_t)nt) != t) continue;
                uint64_t h = hh & (cap - 1);
                for (;;) {
                    Slot &sl = table[(size_t)h];
                    if (sl.key == key) { sl.cnt++; break; }
                    if (sl.key == ~0ull) {
                        sl.key = key; sl.cnt = 1;
                        if (++used * 2 > cap) {                    // grow
                            std::vector<Slot> old;
                            old.swap(table);
                            cap <<= 1;
                            table.assign((size_t)cap, Slot{~0ull, 0});
                            for (const Slot &o : old) {
                                if (o.key == ~0ull) continue;
                                uint64_t g = hash(o.key) & (cap - 1);
                                while (table[(size_t)g].key != ~0ull) g = (g + 1) & (cap - 1);
                                table[(size_t)g] = o;
                            }
                        }
                        break;
                    }
                    h = (h + 1) & (cap - 1);
                }
            }
        part_used[(size_t)t] = (int64_t)used;
        for (const Slot &sl : table)
            if (sl.key != ~0ull && sl.cnt >= min_cnt) part[(size_t)t].push_back(Edge{(int32_t)(sl.key >> 32), (int32_t)(sl.key & 0xFFFFFFFFu), sl.cnt});
    });
    std::vector<std::vector<uint64_t>>().swap(keys);
    std::vector<Edge> edges;
    for (int t = 0; t < nt; t++) { st.pairs_distinct += part_used[(size_t)t]; edges.insert(edges.end(), part[(size_t)t].begin(), part[(size_t)t].end()); }
    std::vector<std::vector<Edge>>().swap(part);
    st.pairs_kept = (int64_t)edges.size();
    if (edges.empty()) return;
    std::sort(edges.begin(), edges.end(), [](const Edge &x, const Edge &y) {      // heaviest first; ties by ids: the order is a function of the data
        if (x.w != y.w) return x.w > y.w;
        if (x.a != y.a) return x.a < y.a;
        return x.b < y.b;
    });
    std::vector<int32_t> parent((size_t)n_tx), csize((size_t)n_tx, 1);
    std::iota(parent.begin(), parent.end(), 0);
    auto find = [&](int32_t x) {
        while (parent[(size_t)x] != x) { parent[(size_t)x] = parent[(size_t)parent[(size_t)x]]; x = parent[(size_t)x]; }
        return x;
    };
    for (const Edge &e : edges) {
        int32_t ra = find(e.a), rb = find(e.b);
        if (ra == rb || csize[(size_t)ra] + csize[(size_t)rb] > kRenumberCap) continue;
        if (csize[(size_t)ra] < csize[(size_t)rb]) std::swap(ra, rb);
        parent[(size_t)rb] = ra;
        csize[(size_t)ra] += csize[(size_t)rb];
    }
    // ---- 3. adjacency inside the clusters, chain order ----
    std::vector<uint32_t> adj_ptr((size_t)n_tx + 1, 0);
    for (const Edge &e : edges) if (find(e.a) == find(e.b)) { adj_ptr[(size_t)e.a + 1]++; adj_ptr[(size_t)e.b + 1]++; }
    for (int32_t t = 0; t < n_tx; t++) adj_ptr[(size_t)t + 1] += adj_ptr[(size_t)t];
    struct Nb { int32_t t; uint32_t w; };
    std::vector<Nb> adj((size_t)adj_ptr[(size_t)n_tx]);
    {
        std::vector<uint32_t> fill(adj_ptr.begin(), adj_ptr.end() - 1);
        for (const Edge &e : edges)
            if (find(e.a) == find(e.b)) { adj[(size_t)fill[(size_t)e.a]++] = Nb{e.b, e.w}; adj[(size_t)fill[(size_t)e.b]++] = Nb{e.a, e.w}; }
    }
    std::vector<Edge>().swap(edges);
    // members of every cluster, clusters in the order of their smallest original id
    std::vector<int32_t> root_first((size_t)n_tx, -1), order;    // order: the roots by first member
    std::vector<uint32_t> mem_ptr;
    std::vector<int32_t> members((size_t)n_tx);
    {
        std::vector<int32_t> cid((size_t)n_tx, -1);
        int32_t nc = 0;
        for (int32_t t = 0; t < n_tx; t++) { const int32_t r = find(t); if (cid[(size_t)r] < 0) { cid[(size_t)r] = nc++; order.push_back(r); } }
        mem_ptr.assign((size_t)nc + 1, 0);
        for (int32_t t = 0; t < n_tx; t++) mem_ptr[(size_t)cid[(size_t)find(t)] + 1]++;
        for (int32_t c = 0; c < nc; c++) mem_ptr[(size_t)c + 1] += mem_ptr[(size_t)c];
        std::vector<uint32_t> fill(mem_ptr.begin(), mem_ptr.end() - 1);
        for (int32_t t = 0; t < n_tx; t++) members[(size_t)fill[(size_t)cid[(size_t)find(t)]]++] = t;
        st.clusters = nc;
    }
    std::vector<int32_t> cand((size_t)n_tx);
    std::vector<uint8_t> done((size_t)n_tx, 0);
    std::vector<uint64_t> tw((size_t)n_tx, 0);                  // total weight of a transcript inside its cluster
    for (int32_t t = 0; t < n_tx; t++) for (uint32_t q = adj_ptr[(size_t)t]; q < adj_ptr[(size_t)t + 1]; q++) tw[(size_t)t] += adj[(size_t)q].w;
    int32_t next_id = 0;
    std::vector<int32_t> byw;
    for (size_t c = 0; c + 1 < mem_ptr.size(); c++) {
        const uint32_t mb = mem_ptr[c], me = mem_ptr[c + 1];
        st.largest = std::max<int64_t>(st.largest, (int64_t)(me - mb));
        if (me - mb == 1) { cand[(size_t)members[(size_t)mb]] = next_id++; continue; }
        byw.assign(members.begin() + mb, members.begin() + me);          // restart candidates: heaviest first, ties by id
        std::sort(byw.begin(), byw.end(), [&](int32_t x, int32_t y) { return tw[(size_t)x] != tw[(size_t)y] ? tw[(size_t)x] > tw[(size_t)y] : x < y; });
        size_t restart = 0;
        uint32_t left = me - mb;
        int32_t cur = -1;
        while (left) {
            if (cur < 0) { while (done[(size_t)byw[restart]]) restart++; cur = byw[restart]; }
            done[(size_t)cur] = 1; cand[(size_t)cur] = next_id++; left--;
            int32_t best = -1; uint32_t bw = 0;
            for (uint32_t q = adj_ptr[(size_t)cur]; q < adj_ptr[(size_t)cur + 1]; q++) {
                const Nb &nb = adj[(size_t)q];
                if (done[(size_t)nb.t]) continue;
                if (nb.w > bw || (nb.w == bw && best >= 0 && nb.t < best)) { best = nb.t; bw = nb.w; }
            }
            cur = best;
        }
    }
    // ---- 5. keep it only if the sampled rows need fewer block entries ----
    int64_t e0 = 0, e1 = 0, ids = 0;
    {
        std::vector<int64_t> pe0((size_t)nt, 0), pe1((size_t)nt, 0), pid((size_t)nt, 0);
        on_threads([&](int t) {
            std::vector<int32_t> tmp, tmp2;
            const size_t lo = sampled.size() * (size_t)t / (size_t)nt, hi = sampled.size() * (size_t)(t + 1) / (size_t)nt;
            for (size_t q = lo; q < hi; q++) {
                const int64_t r = sampled[q];
                tmp.assign(col_idx + row_ptr[r], col_idx + row_ptr[r + 1]);
                std::sort(tmp.begin(), tmp.end());
                pe0[(size_t)t] += row_entries(tmp.data(), (int)tmp.size(), blk);
                tmp2.resize(tmp.size());
                for (size_t i = 0; i < tmp.size(); i++) tmp2[i] = cand[(size_t)tmp[i]];
                std::sort(tmp2.begin(), tmp2.end());
                pe1[(size_t)t] += row_entries(tmp2.data(), (int)tmp2.size(), blk);
                pid[(size_t)t] += (int64_t)tmp.size();
            }
        });
        for (int t = 0; t < nt; t++) { e0 += pe0[(size_t)t]; e1 += pe1[(size_t)t]; ids += pid[(size_t)t]; }
    }
    st.entries_before = e0; st.entries_after = e1; st.ids_sampled = ids;
    bool force = false, never = false;
    if (const char *ev = getenv("EMSAR_HIP_RENUMBER")) { const int v = atoi(ev); force = v == 2; never = v == 0; }     // 1 = decide by the sample (default)
    if (never || (!force && (double)e1 > 0.97 * (double)e0)) return;
    st.applied = true;
    new_of_old.swap(cand);
}

}  // namespace emsar
