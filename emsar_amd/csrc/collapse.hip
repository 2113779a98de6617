// collapse.hip -- read -> segment collapse on the device (the integer core of it).
//
// The reference folds reads into segments while it parses them: update_ReadCounts sorts a read's transcript ids,
// looks the tuple up in the rsh bucket and bumps that node's ReadCount (/root/reference/src/emsar_functions.c:838-943,
// update_rshbucket('r') 1597-1624).  Given a read-level incidence (one CSR row per read) this file does the same
// as a data-parallel pass: rows with the same multiset of transcript ids become ONE row whose weight is the sum of
// its members' weights.  Output rows are numbered by first occurrence (the order in which the reference would have
// met the segments), their ids sorted ascending (the reference's insertion order, emsar_functions.c:889).
//
// Round 2 hashed every row into ONE global table and counted with device-scope atomics: 8.7 ms on config 3, bound by ~45 M
// scattered atomics that execute at the memory side at ~20 G/s whatever they carry.  Round 3: PARTITION, THEN COUNT IN LDS.
//
//   k_row_hash    one lane per row.  Single-transcript rows (59 % of config 3) need no hash: their segment is named by the
//                 transcript; they are counted into per-transcript words, pre-aggregated over 4096 rows in an LDS table so that a
//                 workgroup sends one atomic per distinct transcript.  Rows of 2..8 ids are sorted in registers (written back only
//                 when they were out of order) and hashed -- a sum of per-id mixes, so the multiset decides, not the order; the
//                 workgroup appends its (hash, row) RECORDS to the record list with one atomic.  Longer rows are only LISTED.
//   k_long_hash   the listed long rows, one per lane, eight ids per step with all loads in flight; sorted in place when needed.
//   k_part_ids -> hipCUB radix sort of (partition id, record number), partition id = the TOP p BITS of the hash (two 8-bit passes up to
//                 65 k partitions) -> k_part_gather: the records in partition order.  A partition = the records of one hash prefix, ~1000 of
//                 them; all rows of one segment are in one partition.  (Sorting the 64-bit hashes themselves on bits [64 - p, 64) gave
//                 unsorted output with broken pairs in rocPRIM's small-input path on this image; 32-bit keys on bits [0, p) are the
//                 form the rest of this file has always used.)
//   k_part_bounds the first record of every partition (binary search in the sorted partition ids).
//   k_part_count  ONE WORKGROUP PER PARTITION with an LDS table of 2048 slots keyed by the 64-bit hash: insert (LDS compare-and-swap),
//                 first occurrence (LDS atomicMin over the row numbers), barrier; the used slots take consecutive "dense" numbers (one
//                 global atomic per workgroup); every record then compares its row with the slot's first row id by id (exactness: a
//                 64-bit hash collision is told apart here) and adds its weight to the slot's LDS counter; rows that differ, or found
//                 no place within 64 probes, go to an overflow list and are hashed again with the next seed (in practice never; the test
//                 hook EMSAR_HIP_COLLAPSE_WEAK_HASH makes every row of one length collide in round 0, EMSAR_HIP_COLLAPSE_PART_ROWS
//                 makes partitions several times larger than the table so that it overflows); barrier; first row and count of
//                 every used slot are STORED (not added) under its dense number: a segment lives in exactly one partition, so the
//                 counting needs no global atomics at all.
//   k_single_claim  the single-transcript segments that were met join the list of claimed segments
//   hipCUB radix sort of (first occurrence, segment) -> k_seg_len -> hipCUB exclusive sum -> k_row_emit -> k_row_map (when asked for)
// All integer / byte work.  Exact: a row is only ever counted into a segment after its ids were compared with the segment's.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include <algorithm>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>

#include "../../include/emsar_hip.h"
#include "internal.hpp"
#include <cstdlib>
#include <new>

#include "layout.hpp"

namespace {

__device__ __forceinline__ uint64_t mix64(uint64_t x) {   // splitmix64 finaliser
    x ^= x >> 30; x *= 0xbf58476d1ce4e5b9ull;
    x ^= x >> 27; x *= 0x94d049bb133111ebull;
    x ^= x >> 31;
    return x;
}

constexpr int kInline = 8;                    // ids handled in registers by k_row_hash
constexpr uint32_t kNoSlot = 0xFFFFFFFFu;
constexpr uint32_t kNoRow = 0x7FFFFFFFu;
constexpr uint32_t kSingle = 0x80000000u;     // seg_of / claimed: kSingle | tid = the single-transcript segment of tid
struct Counters { unsigned n_claimed, n_over[2], n_rec, n_long; };
constexpr int kHashThreads = 1024;            // rows per step of k_row_hash
constexpr int kHashSteps = 4;                 // steps per workgroup: the singles of 4096 rows are pre-aggregated in one LDS table
constexpr int kAggSlots = 2048;               // that table (keys 8 KB, first 8 KB, counts 16 KB)
constexpr int kAggProbes = 16;                // a single that finds no place this fast goes to memory directly
constexpr int kPartThreads = 512;             // workgroup of k_part_count
constexpr int kPartSlots = 2048;              // its LDS table: keys 16 KB + counts 16 KB + first 8 KB + dense 8 KB = 48 KB
constexpr int kPartProbes = 64;               // a key lives within this many slots of its home, or goes to the overflow list
constexpr int64_t kPartRows = 1024;           // records per partition aimed at (distinct keys <= records: the table is at most half full)

__device__ __forceinline__ void cswap(int32_t &a, int32_t &b) { const int32_t lo = a < b ? a : b, hi = a < b ? b : a; a = lo; b = hi; }
// sorting network for 8 keys (19 compare-exchanges, static indices: the keys stay in registers)
__device__ __forceinline__ void sort8(int32_t (&v)[8]) {
    cswap(v[0], v[1]); cswap(v[2], v[3]); cswap(v[4], v[5]); cswap(v[6], v[7]);
    cswap(v[0], v[2]); cswap(v[1], v[3]); cswap(v[4], v[6]); cswap(v[5], v[7]);
    cswap(v[1], v[2]); cswap(v[5], v[6]); cswap(v[0], v[4]); cswap(v[3], v[7]);
    cswap(v[1], v[5]); cswap(v[2], v[6]);
    cswap(v[1], v[4]); cswap(v[3], v[6]);
    cswap(v[2], v[4]); cswap(v[3], v[5]);
    cswap(v[3], v[4]);
}

// hash of a multiset of ids under a seed: a sum of per-id mixes (any order), never 0 (0 = empty key word)
__device__ __forceinline__ uint64_t id_mix(int32_t id, uint64_t seed) { return mix64((uint64_t)(uint32_t)id + 0x632be59bd9b4e019ull + seed * 0x9e3779b97f4a7c15ull); }
__device__ __forceinline__ uint64_t finish_hash(uint64_t a, uint64_t len, uint64_t seed, int weak) {
    if (weak && seed == 0) a = 0x9e3779b97f4a7c15ull * (len + 1);   // test hook: every row of one length collides in round 0
    a = mix64(a);                                                     // the TOP bits choose the partition: mix once more
    return a ? a : 1;
}

// every lane of the wave calls it: the items of the lanes that `have` one take consecutive places behind the counter -- one atomic per
// wave.  For counters in LDS only: 780 k wave-level atomics on ONE global word serialise at the memory side (k_row_hash: 1.75 -> 18 ms).
__device__ __forceinline__ unsigned wave_reserve(bool have, unsigned *lds_counter) {
    const unsigned long long m = __ballot(have);
    if (m == 0ull) return 0u;
    const unsigned lane = __lane_id();
    const int leader = __ffsll((long long)m) - 1;
    unsigned base = 0;
    if ((int)lane == leader) base = atomicAdd(lds_counter, (unsigned)__popcll(m));
    base = __shfl(base, leader);
    return base + (unsigned)__popcll(m & ((1ull << lane) - 1ull));
}
struct __attribute__((aligned(16))) Rec { unsigned long long key; uint32_t row, pad; };    // one 16-byte record: the gather into partition order touches one line per record

// ---- singles: LDS pre-aggregation of tid -> (first row, weight) ----
struct AggLds { uint32_t key[kAggSlots], first[kAggSlots]; unsigned long long cnt[kAggSlots]; };
__device__ __forceinline__ void single_add(AggLds &A, uint32_t tid, uint32_t r, unsigned long long w, uint32_t *first_1, unsigned long long *cnt_1) {
    uint32_t h = (tid * 0x9E3779B1u) >> 21;
    for (int p = 0; p < kAggProbes; p++) {
        uint32_t k = A.key[h];
        if (k == kNoSlot) k = atomicCAS(&A.key[h], kNoSlot, tid);
        if (k == kNoSlot || k == tid) { atomicMin(&A.first[h], r); atomicAdd(&A.cnt[h], w); return; }
        h = (h + 1) & (kAggSlots - 1);
    }
    if (r < first_1[tid]) atomicMin(&first_1[tid], r);           // the table is crowded around here: straight to memory
    atomicAdd(&cnt_1[tid], w);
}
__device__ __forceinline__ void single_flush(AggLds &A, int threads, uint32_t *first_1, unsigned long long *cnt_1) {
    for (int e = threadIdx.x; e < kAggSlots; e += threads) {
        const uint32_t t = A.key[e];
        if (t == kNoSlot) continue;
        // workgroups start roughly in row order, so the first occurrence has usually been seen: the word only ever decreases, a
        // stale read costs one atomic at most, and most entries skip theirs
        if (A.first[e] < first_1[t]) atomicMin(&first_1[t], A.first[e]);
        atomicAdd(&cnt_1[t], A.cnt[e]);
    }
}

// Round `seed` of the hashing: rows [0, n) or, with `list`, the rows it names.  Singles counted, short rows hashed, long rows listed.
// A workgroup takes kHashSteps x 1024 rows; what it appends to the record list and to the list of long rows is placed inside the
// workgroup first (wave ballots + one LDS atomic per wave) and reserved with ONE global atomic per list and workgroup.
__global__ __launch_bounds__(kHashThreads) void k_row_hash(int64_t n, const uint32_t *__restrict__ list, const uint64_t *__restrict__ rp, int32_t *__restrict__ ci,
                                                           const int32_t *__restrict__ wgt, uint32_t *__restrict__ seg_of /* or null */,
                                                           Rec *__restrict__ rec,
                                                           uint32_t *__restrict__ long_list, uint32_t *__restrict__ first_1, unsigned long long *__restrict__ cnt_1,
                                                           Counters *cnt, uint64_t seed, int weak) {
    __shared__ AggLds A;
    __shared__ unsigned n_rec_wg, n_long_wg, base_rec, base_long;
    for (int e = threadIdx.x; e < kAggSlots; e += kHashThreads) { A.key[e] = kNoSlot; A.first[e] = kNoRow; A.cnt[e] = 0ull; }
    if (threadIdx.x == 0) { n_rec_wg = 0; n_long_wg = 0; }
    __syncthreads();
    uint64_t a[kHashSteps];
    uint32_t row[kHashSteps], pos[kHashSteps];
    int kind[kHashSteps];                                          // 0 nothing to append, 1 a record, 2 a long row
#pragma unroll
    for (int step = 0; step < kHashSteps; step++) {
        const int64_t i = ((int64_t)blockIdx.x * kHashSteps + step) * kHashThreads + threadIdx.x;
        kind[step] = 0; a[step] = 0; row[step] = 0;
        if (i < n) {
            const int64_t r = list ? (int64_t)list[i] : i;
            row[step] = (uint32_t)r;
            const uint64_t b = rp[r], len = rp[r + 1] - b;
            const int64_t w = wgt ? wgt[r] : 1;
            uint32_t seg = kNoSlot;                                    // empty rows and rows without weight vanish
            if (len > (uint64_t)kInline && w > 0) kind[step] = 2;
            else if (len == 1 && w > 0) {
                // 59 % of config 3's reads hit one transcript only: their segment is named by the transcript itself
                const uint32_t t = (uint32_t)ci[b];
                seg = kSingle | t;
                single_add(A, t, row[step], (unsigned long long)w, first_1, cnt_1);
            } else if (len != 0 && w > 0) {
                int32_t v[kInline];
                uint64_t h = 0x9e3779b97f4a7c15ull * (len + 1 + seed);
#pragma unroll
                for (int j = 0; j < kInline; j++) v[j] = (uint64_t)j < len ? ci[b + j] : INT32_MAX;
                bool sorted = true;
#pragma unroll
                for (int j = 1; j < kInline; j++) sorted &= v[j - 1] <= v[j];
                if (!sorted) {
                    sort8(v);
#pragma unroll
                    for (int j = 0; j < kInline; j++) if ((uint64_t)j < len) ci[b + j] = v[j];
                }
#pragma unroll
                for (int j = 0; j < kInline; j++) if ((uint64_t)j < len) h += id_mix(v[j], seed);
                a[step] = finish_hash(h, len, seed, weak);
                kind[step] = 1;
            }
            if (seg_of && kind[step] == 0) seg_of[r] = seg;            // multi-transcript rows get theirs from k_part_count
        }
        const unsigned pr = wave_reserve(kind[step] == 1, &n_rec_wg), pl = wave_reserve(kind[step] == 2, &n_long_wg);
        pos[step] = kind[step] == 1 ? pr : pl;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        base_rec = n_rec_wg ? atomicAdd(&cnt->n_rec, n_rec_wg) : 0u;
        base_long = n_long_wg ? atomicAdd(&cnt->n_long, n_long_wg) : 0u;
    }
    __syncthreads();
#pragma unroll
    for (int step = 0; step < kHashSteps; step++) {
        if (kind[step] == 1) rec[base_rec + pos[step]] = Rec{a[step], row[step], 0u};
        else if (kind[step] == 2) long_list[base_long + pos[step]] = row[step];
    }
    single_flush(A, kHashThreads, first_1, cnt_1);
}

// the same for the listed long rows, one per lane
__global__ __launch_bounds__(256) void k_long_hash(const uint32_t *__restrict__ long_list, const uint64_t *__restrict__ rp, int32_t *__restrict__ ci,
                                                   Rec *__restrict__ rec, const Counters *cnt, uint64_t seed, int weak) {
    const unsigned n = cnt->n_long, rec0 = cnt->n_rec;         // the long rows' records follow the short rows' (k_row_hash is done): no append, no atomic
    const unsigned i = blockIdx.x * 256 + threadIdx.x;
    if (blockIdx.x * 256 >= n) return;                           // the grid covers the worst case (workgroup-uniform exit)
    uint64_t a = 0;
    uint32_t row = 0;
    if (i < n) {
        const int64_t r = (int64_t)long_list[i];
        row = (uint32_t)r;
        const uint64_t b = rp[r], len = rp[r + 1] - b;
        // eight ids per step, all eight loads in flight together (one id per step is one trip to memory per id: 50 us for 100 ids)
        int32_t *x = ci + b;
        a = 0x9e3779b97f4a7c15ull * (len + 1 + seed);
        bool sorted = true;
        int32_t prev = INT32_MIN;
        for (uint64_t j0 = 0; j0 < len; j0 += 8) {
            int32_t c[8];
#pragma unroll
            for (int j = 0; j < 8; j++) c[j] = j0 + (uint64_t)j < len ? x[j0 + (uint64_t)j] : INT32_MAX;
            sorted &= prev <= c[0];
#pragma unroll
            for (int j = 1; j < 8; j++) sorted &= c[j - 1] <= c[j];
#pragma unroll
            for (int j = 0; j < 8; j++) if (j0 + (uint64_t)j < len) { a += id_mix(c[j], seed); prev = c[j]; }
        }
        if (!sorted) {                                            // rare (aligners and our own parser emit sorted ids): insertion sort in place
            for (uint64_t j = 1; j < len; j++) {                  // (the hash is a sum: the order did not matter)
                const int32_t t = x[j];
                uint64_t k = j;
                while (k > 0 && x[k - 1] > t) { x[k] = x[k - 1]; k--; }
                if (k != j) x[k] = t;
            }
        }
        a = finish_hash(a, len, seed, weak);
    }
    if (i < n) rec[rec0 + i] = Rec{a, row, 0u};
}

// partition id of every record (the top `bits` bits of its hash) and its number
__global__ __launch_bounds__(256) void k_part_ids(unsigned n, const Rec *__restrict__ rec, int bits, uint32_t *__restrict__ pid, uint32_t *__restrict__ idx) {
    const unsigned i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) { pid[i] = (uint32_t)(rec[i].key >> (64 - bits)); idx[i] = i; }
}
// the records in partition order
__global__ __launch_bounds__(256) void k_part_gather(unsigned n, const uint32_t *__restrict__ idx, const Rec *__restrict__ rec, Rec *__restrict__ srec) {
    const unsigned i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) srec[i] = rec[idx[i]];
}
// bounds[b] = the first sorted record whose partition id is >= b, b = 0 .. P
__global__ __launch_bounds__(256) void k_part_bounds(const uint32_t *__restrict__ pid, unsigned n, int bits, unsigned P, unsigned *__restrict__ bounds) {
    const unsigned b = blockIdx.x * 256 + threadIdx.x;
    if (b > P) return;
    if (b == P || bits == 0) { bounds[b] = b == 0 ? 0u : n; return; }
    unsigned lo = 0, hi = n;
    while (lo < hi) {
        const unsigned mid = lo + (hi - lo) / 2;
        if (pid[mid] < b) lo = mid + 1; else hi = mid;
    }
    bounds[b] = lo;
}

// two rows of the (sorted) CSR, id by id
__device__ __forceinline__ bool rows_equal(const uint64_t *__restrict__ rp, const int32_t *__restrict__ ci, uint32_t r, uint32_t q) {
    const uint64_t b = rp[r], len = rp[(size_t)r + 1] - b, pb = rp[q], plen = rp[(size_t)q + 1] - pb;
    if (len != plen) return false;
    bool same = true;
    for (uint64_t j0 = 0; j0 < len && same; j0 += 8) {            // eight ids of either row per step, sixteen loads in flight
        int32_t x[8], y[8];
#pragma unroll
        for (int j = 0; j < 8; j++) { const bool in = j0 + (uint64_t)j < len; x[j] = in ? ci[b + j0 + j] : 0; y[j] = in ? ci[pb + j0 + j] : 0; }
#pragma unroll
        for (int j = 0; j < 8; j++) same &= x[j] == y[j];
    }
    return same;
}

struct PartLds {
    unsigned long long key[kPartSlots], cnt[kPartSlots];
    uint32_t first[kPartSlots], dense[kPartSlots];
    unsigned n_used, base;
};
// slot of key h (inserted when absent and `insert`), -1 when it is not within kPartProbes slots of its home
__device__ __forceinline__ int part_probe(PartLds &T, unsigned long long h, bool insert) {
    uint32_t s = (uint32_t)(h >> 7) & (kPartSlots - 1);            // not the top bits: those are equal inside a partition
    for (int p = 0; p < kPartProbes; p++) {
        unsigned long long k = T.key[s];
        if (k == 0ull && insert) k = atomicCAS(&T.key[s], 0ull, h);
        if (k == h || (k == 0ull && insert)) return (int)s;
        if (k == 0ull) return -1;
        s = (s + 1) & (kPartSlots - 1);
    }
    return -1;
}
__global__ __launch_bounds__(kPartThreads) void k_part_count(const unsigned *__restrict__ bounds, const Rec *__restrict__ rec,
                                                             const uint64_t *__restrict__ rp, const int32_t *__restrict__ ci, const int32_t *__restrict__ wgt,
                                                             uint32_t *__restrict__ seg_of /* or null */, uint32_t *__restrict__ first_d, unsigned long long *__restrict__ cnt_d,
                                                             uint32_t *__restrict__ over, unsigned *__restrict__ n_over, unsigned over_cap, Counters *cnt) {
    __shared__ PartLds T;
    const unsigned lo = bounds[blockIdx.x], hi = bounds[blockIdx.x + 1];
    if (lo >= hi) return;                                         // workgroup-uniform
    for (int e = threadIdx.x; e < kPartSlots; e += kPartThreads) { T.key[e] = 0ull; T.cnt[e] = 0ull; T.first[e] = kNoRow; T.dense[e] = kNoSlot; }
    if (threadIdx.x == 0) T.n_used = 0;
    __syncthreads();
    // 1. keys in, first occurrence of every key
    for (unsigned i = lo + threadIdx.x; i < hi; i += kPartThreads) {
        const Rec R = rec[i];
        const int s = part_probe(T, R.key, true);
        if (s >= 0) atomicMin(&T.first[s], R.row);
    }
    __syncthreads();
    // 2. the used slots take consecutive dense numbers
    for (int e = threadIdx.x; e < kPartSlots; e += kPartThreads) if (T.key[e] != 0ull) T.dense[e] = atomicAdd(&T.n_used, 1u);
    __syncthreads();
    if (threadIdx.x == 0) T.base = atomicAdd(&cnt->n_claimed, T.n_used);
    __syncthreads();
    // 3. every record against the first row of its slot; counted when equal
    for (unsigned i = lo + threadIdx.x; i < hi; i += kPartThreads) {
        const Rec R = rec[i];
        const uint32_t r = R.row;
        const int s = part_probe(T, R.key, false);
        bool same = false;
        if (s >= 0) {
            const uint32_t q = T.first[s];
            same = q == r || rows_equal(rp, ci, r, q);
        }
        if (same) {
            atomicAdd(&T.cnt[s], (unsigned long long)(wgt ? wgt[r] : 1));
            if (seg_of) seg_of[r] = T.base + T.dense[s];
        } else {
#ifdef EMSAR_COLLAPSE_TRACE
            const unsigned pos_ = atomicAdd(n_over, 1u);
            if (pos_ >= over_cap) continue;
            if (pos_ < 12) {
                const uint32_t q_ = s >= 0 ? T.first[s] : 0xFFFFFFFFu;
                printf("over: part %u rec %u key %llx row %u slot %d first %u len %llu plen %llu ids %d %d | %d %d tkey %llx\n", blockIdx.x, i, R.key, r, s, q_,
                       (unsigned long long)(rp[r + 1] - rp[r]), q_ < 0x7FFFFFFFu ? (unsigned long long)(rp[q_ + 1] - rp[q_]) : 0ull, ci[rp[r]], ci[rp[r] + 1],
                       q_ < 0x7FFFFFFFu ? ci[rp[q_]] : -1, q_ < 0x7FFFFFFFu ? ci[rp[q_] + 1] : -1, s >= 0 ? T.key[s] : 0ull);
            }
            over[pos_] = r;
#else
            // a 64-bit hash collision or a crowded table: next round, next seed.  (The list holds one entry per record; the bound only bites
            // if the partition bounds were ever inconsistent -- the host then sees n_over > n_rec and fails the call instead of a wild store.)
            const unsigned pos = atomicAdd(n_over, 1u);
            if (pos < over_cap) over[pos] = r;
#endif
        }
    }
    __syncthreads();
    // 4. one segment, one partition: plain stores.  A slot whose rows ALL went to the overflow list cannot exist: its first row equals itself.
    for (int e = threadIdx.x; e < kPartSlots; e += kPartThreads)
        if (T.key[e] != 0ull) { const uint32_t d = T.base + T.dense[e]; first_d[d] = T.first[e]; cnt_d[d] = T.cnt[e]; }
}

// the multi-transcript segments (dense numbers [0, n_multi)) and the single-transcript segments that were met: the claimed list
__global__ __launch_bounds__(256) void k_multi_claim(unsigned n_multi, uint32_t *__restrict__ claimed) {
    const unsigned d = blockIdx.x * 256 + threadIdx.x;
    if (d < n_multi) claimed[d] = d;
}
__global__ __launch_bounds__(256) void k_single_claim(int32_t n_tx, const uint32_t *__restrict__ first_1, const unsigned long long *__restrict__ cnt_1,
                                                      uint32_t *__restrict__ claimed, uint32_t *__restrict__ first_d, Counters *cnt) {
    __shared__ unsigned n_won, base;
    if (threadIdx.x == 0) n_won = 0;
    __syncthreads();
    const int32_t t = (int32_t)(blockIdx.x * 256 + threadIdx.x);
    const bool met = t < n_tx && cnt_1[t] != 0ull;
    unsigned pos = 0;
    if (met) pos = atomicAdd(&n_won, 1u);
    __syncthreads();
    if (threadIdx.x == 0 && n_won) base = atomicAdd(&cnt->n_claimed, n_won);
    __syncthreads();
    if (met) { claimed[base + pos] = kSingle | (uint32_t)t; first_d[base + pos] = first_1[t]; }
}
// after the sort by first occurrence: first_sorted[u] is the first row of unique row u -- for a multi-transcript segment the row its ids are copied from
__global__ __launch_bounds__(256) void k_seg_len(int64_t nu, const uint32_t *__restrict__ seg, const uint32_t *__restrict__ first_sorted, const uint64_t *__restrict__ rp,
                                                 uint64_t *__restrict__ len) {
    const int64_t u = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (u < nu) { const uint32_t r = first_sorted[u]; len[u] = (seg[u] & kSingle) ? 1 : rp[(size_t)r + 1] - rp[r]; }
}
// The unique rows are the first members' rows concatenated in order of first occurrence: one workgroup copies the ids of 256 consecutive
// unique rows as ONE contiguous range of the output (coalesced stores; the source of an element is found by a binary search over the
// 257 offsets of the workgroup's rows in LDS).  One lane per unique row with a loop over its ids took 1.05 ms for config 3's 5.4 M segments.
__global__ __launch_bounds__(256) void k_row_emit(int64_t nu, uint64_t nnz_u, const uint32_t *__restrict__ seg, const uint32_t *__restrict__ first_sorted,
                                                  const uint64_t *__restrict__ rp, const int32_t *__restrict__ ci, const unsigned long long *__restrict__ cnt_d,
                                                  const unsigned long long *__restrict__ cnt_1, uint32_t *__restrict__ uid_1, uint32_t *__restrict__ uid_d,
                                                  const uint64_t *__restrict__ uoff, uint64_t *__restrict__ out_rp, int32_t *__restrict__ out_ci,
                                                  long long *__restrict__ out_w) {
    __shared__ uint64_t off[257], src[256];       // output offset of every row of the workgroup; where its ids come from (kSingle | tid in the low word for singles)
    const int64_t u0 = (int64_t)blockIdx.x * 256, u = u0 + threadIdx.x;
    const int n_loc = (int)(nu - u0 < 256 ? nu - u0 : 256);
    if (u < nu) {
        const uint32_t sg = seg[u];
        const uint64_t o = uoff[u];
        out_rp[u] = o;
        off[threadIdx.x] = o;
        if (sg & kSingle) {
            const uint32_t t = sg & ~kSingle;
            out_w[u] = (long long)cnt_1[t]; uid_1[t] = (uint32_t)u;
            src[threadIdx.x] = ~0ull << 32 | t;                  // no source row: the id itself
        } else {
            out_w[u] = (long long)cnt_d[sg]; uid_d[sg] = (uint32_t)u;
            src[threadIdx.x] = rp[first_sorted[u]];
        }
    }
    if ((int)threadIdx.x == n_loc) off[n_loc] = u0 + n_loc < nu ? uoff[u0 + n_loc] : nnz_u;
    if (n_loc == 256 && threadIdx.x == 0) off[256] = u0 + 256 < nu ? uoff[u0 + 256] : nnz_u;
    __syncthreads();
    const uint64_t base = off[0], total = off[n_loc] - base;
    for (uint64_t p = threadIdx.x; p < total; p += 256) {
        int lo = 0, hi = n_loc - 1;                               // the last row whose offset is <= base + p (rows are never empty)
        while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (off[mid] - base <= p) lo = mid; else hi = mid - 1; }
        const uint64_t sb = src[lo];
        out_ci[base + p] = (sb >> 32) == 0xFFFFFFFFull ? (int32_t)(uint32_t)sb : ci[sb + (p - (off[lo] - base))];
    }
}
__global__ __launch_bounds__(256) void k_row_map(int64_t n_rows, const uint32_t *__restrict__ seg_of, const uint32_t *__restrict__ uid_1, const uint32_t *__restrict__ uid_d,
                                                 int32_t *__restrict__ row_map) {
    const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (r >= n_rows) return;
    const uint32_t s = seg_of[r];
    row_map[r] = s == kNoSlot ? -1 : (s & kSingle) ? (int32_t)uid_1[s & ~kSingle] : (int32_t)uid_d[s];
}

struct Events {
    hipEvent_t a = nullptr, b = nullptr;
    ~Events() { if (a) (void)hipEventDestroy(a); if (b) (void)hipEventDestroy(b); }
};
struct DevBuf {
    void *p = nullptr;
    ~DevBuf() { if (p) (void)hipFree(p); }
    hipError_t alloc(size_t bytes) { return hipMalloc(&p, bytes ? bytes : 16); }
    template <class T> T *as() { return reinterpret_cast<T *>(p); }
};

}  // namespace

extern "C" int emsar_hip_collapse_rows(emsar_hip_ctx *ctx, int64_t n_rows, int32_t n_tx, const uint64_t *row_ptr, const int32_t *col_idx,
                                       const int32_t *row_weight, int64_t *n_unique_out, uint64_t *row_ptr_out, int32_t *col_idx_out,
                                       int32_t *weight_out, int32_t *row_map_out, emsar_hip_collapse_stats *stats) {
    if (!ctx || !n_unique_out || !row_ptr_out || !weight_out || (!col_idx_out && n_rows > 0 && row_ptr && row_ptr[n_rows] > 0)) return EMSAR_HIP_ERR_ARG;
    if (emsar::validate_csr(n_rows, n_tx, row_ptr, col_idx) != 0) return EMSAR_HIP_ERR_ARG;
    if (n_rows >= (int64_t)0x7F7F7F7F) return EMSAR_HIP_ERR_ARG;          // row ids travel as 31-bit values below the 'no row yet' mark
    if (row_weight) for (int64_t r = 0; r < n_rows; r++) if (row_weight[r] < 0) return EMSAR_HIP_ERR_ARG;
    hipStream_t st = emsar_internal_stream(ctx);
#define CCHK(call)                                                                                          \
    do {                                                                                                    \
        hipError_t e_ = (call);                                                                             \
        if (e_ != hipSuccess) {                                                                             \
            emsar_internal_set_error(ctx, #call, hipGetErrorString(e_));                                    \
            return e_ == hipErrorOutOfMemory ? EMSAR_HIP_ERR_OOM : EMSAR_HIP_ERR_HIP;                        \
        }                                                                                                   \
    } while (0)
#define CFAIL(what) do { emsar_internal_set_error(ctx, "collapse", what); return EMSAR_HIP_ERR_HIP; } while (0)
    const bool dbg = getenv("EMSAR_HIP_DEBUG") != nullptr;      // stage by stage: wait for the device and say where the call is
#define STAGE(name) do { if (dbg) { CCHK(hipStreamSynchronize(st)); fprintf(stderr, "collapse: %s done\n", name); fflush(stderr); } } while (0)
    CCHK(hipSetDevice(emsar_internal_device(ctx)));
    auto t0 = std::chrono::steady_clock::now();
    const uint64_t nnz = row_ptr[n_rows];
    *n_unique_out = 0;
    row_ptr_out[0] = 0;
    if (n_rows == 0) return EMSAR_HIP_OK;
    // records per partition aimed at.  Test hook: a larger value makes partitions that overflow their LDS table, so that the overflow
    // rounds carry part of the call.  Clamped to [16, 2^20]; the partition count below is always between 1 and 2^24, never 0.
    int64_t part_rows = kPartRows;
    if (const char *e = getenv("EMSAR_HIP_COLLAPSE_PART_ROWS")) { const long long v = atoll(e); if (v >= 16 && v <= (1 << 20)) part_rows = (int64_t)v; }
    DevBuf d_rp, d_ci, d_w, d_seg, d_claimed, d_over0, d_over1, d_long, d_cnt, d_first, d_cntd, d_first1, d_cnt1, d_uid1, d_uidd, d_sseg, d_sfirst, d_ulen, d_uoff,
        d_orp, d_oci, d_ow, d_map, d_tmp, d_rec0, d_rec1, d_bounds, d_pid0, d_pid1, d_idx0, d_idx1;
    CCHK(d_rp.alloc((size_t)(n_rows + 1) * 8)); CCHK(d_ci.alloc((size_t)nnz * 4));
    if (row_weight) CCHK(d_w.alloc((size_t)n_rows * 4));
    if (row_map_out) { CCHK(d_seg.alloc((size_t)n_rows * 4)); CCHK(d_map.alloc((size_t)n_rows * 4)); }
    CCHK(d_claimed.alloc((size_t)n_rows * 4));
    CCHK(d_over0.alloc((size_t)n_rows * 4)); CCHK(d_over1.alloc((size_t)n_rows * 4)); CCHK(d_long.alloc((size_t)n_rows * 4)); CCHK(d_cnt.alloc(sizeof(Counters)));
    // everything the numbering and the emit need, at worst-case size (every row unique), so that nothing is allocated between the kernels
    CCHK(d_first.alloc((size_t)n_rows * 4)); CCHK(d_cntd.alloc((size_t)n_rows * 8)); CCHK(d_uidd.alloc((size_t)n_rows * 4));
    CCHK(d_first1.alloc((size_t)std::max(n_tx, 1) * 4)); CCHK(d_cnt1.alloc((size_t)std::max(n_tx, 1) * 8)); CCHK(d_uid1.alloc((size_t)std::max(n_tx, 1) * 4));
    CCHK(d_sseg.alloc((size_t)n_rows * 4)); CCHK(d_sfirst.alloc((size_t)n_rows * 4)); CCHK(d_ulen.alloc((size_t)n_rows * 8)); CCHK(d_uoff.alloc((size_t)n_rows * 8));
    CCHK(d_orp.alloc((size_t)(n_rows + 1) * 8)); CCHK(d_ow.alloc((size_t)n_rows * 8)); CCHK(d_oci.alloc((size_t)nnz * 4));
    CCHK(d_rec0.alloc((size_t)n_rows * sizeof(Rec))); CCHK(d_rec1.alloc((size_t)n_rows * sizeof(Rec)));
    CCHK(d_pid0.alloc((size_t)n_rows * 4)); CCHK(d_pid1.alloc((size_t)n_rows * 4)); CCHK(d_idx0.alloc((size_t)n_rows * 4)); CCHK(d_idx1.alloc((size_t)n_rows * 4));
    {
        int bits_max = 0;                                          // partition bounds at the largest partition count any round can have
        while (bits_max < 24 && ((int64_t)1 << bits_max) * part_rows < n_rows) bits_max++;
        CCHK(d_bounds.alloc((((size_t)1 << bits_max) + 1) * 4));
    }
    int end_bit = 1;
    while (end_bit < 32 && ((uint64_t)1 << end_bit) < (uint64_t)n_rows) end_bit++;
    size_t tb1 = 0, tb2 = 0;
    CCHK(hipcub::DeviceRadixSort::SortPairs(nullptr, tb1, (uint32_t *)nullptr, (uint32_t *)nullptr, (uint32_t *)nullptr, (uint32_t *)nullptr, (int)n_rows, 0, end_bit, st));
    CCHK(hipcub::DeviceScan::ExclusiveSum(nullptr, tb2, (uint64_t *)nullptr, (uint64_t *)nullptr, (int)n_rows, st));
    const size_t tmp_bytes = std::max(tb1, tb2) + ((size_t)1 << 20);
    CCHK(d_tmp.alloc(tmp_bytes));
    CCHK(hipMemcpyAsync(d_rp.p, row_ptr, (size_t)(n_rows + 1) * 8, hipMemcpyHostToDevice, st));
    if (nnz) CCHK(hipMemcpyAsync(d_ci.p, col_idx, (size_t)nnz * 4, hipMemcpyHostToDevice, st));
    if (row_weight) CCHK(hipMemcpyAsync(d_w.p, row_weight, (size_t)n_rows * 4, hipMemcpyHostToDevice, st));
    Events ev;
    CCHK(hipEventCreate(&ev.a)); CCHK(hipEventCreate(&ev.b));
    CCHK(hipEventRecord(ev.a, st));
    const char *weak_env = getenv("EMSAR_HIP_COLLAPSE_WEAK_HASH");      // tests: every row of one length collides in round 0 and is told apart by comparison
    const int weak_hash = weak_env && atoi(weak_env) != 0;
    const int32_t *dw = row_weight ? d_w.as<int32_t>() : nullptr;
    uint32_t *seg_of = row_map_out ? d_seg.as<uint32_t>() : nullptr;
    uint32_t *over[2] = {d_over0.as<uint32_t>(), d_over1.as<uint32_t>()};
    Counters hc{0, {0, 0}, 0, 0};
    CCHK(hipMemsetAsync(d_cnt.p, 0, sizeof(Counters), st));
    CCHK(hipMemsetAsync(d_first1.p, 0x7F, (size_t)n_tx * 4, st));       // 0x7F7F7F7F: larger than any row number
    CCHK(hipMemsetAsync(d_cnt1.p, 0, (size_t)n_tx * 8, st));
    int64_t n_cur = n_rows, max_parts = 0;
    const uint32_t *list = nullptr;
    int rounds = 0;
    for (uint64_t seed = 0; n_cur > 0; seed++) {
        const int o = (int)(seed & 1);
        CCHK(hipMemsetAsync(&d_cnt.as<Counters>()->n_long, 0, sizeof(unsigned), st));
        CCHK(hipMemsetAsync(&d_cnt.as<Counters>()->n_rec, 0, sizeof(unsigned), st));
        CCHK(hipMemsetAsync(&d_cnt.as<Counters>()->n_over[o], 0, sizeof(unsigned), st));
        const int64_t rows_per_wg = (int64_t)kHashThreads * kHashSteps;
        hipLaunchKernelGGL(k_row_hash, dim3((unsigned)((n_cur + rows_per_wg - 1) / rows_per_wg)), dim3(kHashThreads), 0, st, n_cur, list, d_rp.as<uint64_t>(),
                           d_ci.as<int32_t>(), dw, seg_of, d_rec0.as<Rec>(), d_long.as<uint32_t>(), d_first1.as<uint32_t>(),
                           d_cnt1.as<unsigned long long>(), d_cnt.as<Counters>(), seed, weak_hash);
        STAGE("k_row_hash");
        hipLaunchKernelGGL(k_long_hash, dim3((unsigned)((n_cur + 255) / 256)), dim3(256), 0, st, d_long.as<uint32_t>(), d_rp.as<uint64_t>(), d_ci.as<int32_t>(),
                           d_rec0.as<Rec>(), d_cnt.as<Counters>(), seed, weak_hash);
        STAGE("k_long_hash");
        CCHK(hipGetLastError());
        CCHK(hipMemcpyAsync(&hc, d_cnt.p, sizeof(Counters), hipMemcpyDeviceToHost, st));
        CCHK(hipStreamSynchronize(st));
        const int64_t n_rec = (int64_t)hc.n_rec + (int64_t)hc.n_long;          // short rows' records, then one per long row
        if (n_rec > n_cur) CFAIL("more records than rows");
        if (n_rec > 0) {
            // partitions: the records of one hash prefix; 2^bits of them so that one holds about part_rows records
            int bits = 0;
            while (bits < 24 && ((int64_t)1 << bits) * part_rows < n_rec) bits++;
            const unsigned P = 1u << bits;
            max_parts = std::max<int64_t>(max_parts, (int64_t)P);
            const Rec *srec = d_rec0.as<Rec>();
            if (bits > 0) {
                const dim3 gr((unsigned)((n_rec + 255) / 256)), br(256);
                hipLaunchKernelGGL(k_part_ids, gr, br, 0, st, (unsigned)n_rec, d_rec0.as<Rec>(), bits, d_pid0.as<uint32_t>(), d_idx0.as<uint32_t>());
                size_t t1 = 0;                                    // this call's own temporary storage size (rocPRIM picks its algorithm by the item count)
                CCHK(hipcub::DeviceRadixSort::SortPairs(nullptr, t1, d_pid0.as<uint32_t>(), d_pid1.as<uint32_t>(), d_idx0.as<uint32_t>(), d_idx1.as<uint32_t>(), (int)n_rec, 0, bits, st));
                if (t1 > tmp_bytes) CFAIL("temporary storage of the partition sort exceeds the worst case");
                CCHK(hipcub::DeviceRadixSort::SortPairs(d_tmp.p, t1, d_pid0.as<uint32_t>(), d_pid1.as<uint32_t>(), d_idx0.as<uint32_t>(), d_idx1.as<uint32_t>(), (int)n_rec, 0, bits, st));
                hipLaunchKernelGGL(k_part_gather, gr, br, 0, st, (unsigned)n_rec, d_idx1.as<uint32_t>(), d_rec0.as<Rec>(), d_rec1.as<Rec>());
                srec = d_rec1.as<Rec>();
                STAGE("partition sort");
            }
            if (dbg) fprintf(stderr, "collapse: round %d, %lld rows, %lld records, %u partitions\n", rounds, (long long)n_cur, (long long)n_rec, P);
            hipLaunchKernelGGL(k_part_bounds, dim3((P + 1 + 255) / 256), dim3(256), 0, st, d_pid1.as<uint32_t>(), (unsigned)n_rec, bits, P, d_bounds.as<unsigned>());
            STAGE("k_part_bounds");
            hipLaunchKernelGGL(k_part_count, dim3(P), dim3(kPartThreads), 0, st, d_bounds.as<unsigned>(), srec, d_rp.as<uint64_t>(), d_ci.as<int32_t>(), dw, seg_of,
                               d_first.as<uint32_t>(), d_cntd.as<unsigned long long>(), over[o], &d_cnt.as<Counters>()->n_over[o], (unsigned)n_rec, d_cnt.as<Counters>());
            STAGE("k_part_count");
            CCHK(hipGetLastError());
            CCHK(hipMemcpyAsync(&hc, d_cnt.p, sizeof(Counters), hipMemcpyDeviceToHost, st));
            CCHK(hipStreamSynchronize(st));
        }
        if ((int64_t)hc.n_over[o] > n_rec || (int64_t)hc.n_claimed > n_rows) CFAIL("overflow list or segment count out of range");
        n_cur = n_rec > 0 ? (int64_t)hc.n_over[o] : 0;           // rows whose ids differ from their slot's, or that found the table crowded: again, with the next seed
        list = over[o];
        if (++rounds > 64) CFAIL("hash rounds do not terminate");
    }
    const unsigned n_multi = hc.n_claimed;
    if (n_multi) hipLaunchKernelGGL(k_multi_claim, dim3((n_multi + 255) / 256), dim3(256), 0, st, n_multi, d_claimed.as<uint32_t>());
    if (n_tx > 0)
        hipLaunchKernelGGL(k_single_claim, dim3((unsigned)((n_tx + 255) / 256)), dim3(256), 0, st, n_tx, d_first1.as<uint32_t>(), d_cnt1.as<unsigned long long>(),
                           d_claimed.as<uint32_t>(), d_first.as<uint32_t>(), d_cnt.as<Counters>());
    CCHK(hipGetLastError());
    CCHK(hipMemcpyAsync(&hc, d_cnt.p, sizeof(Counters), hipMemcpyDeviceToHost, st));
    CCHK(hipStreamSynchronize(st));
    const int64_t nu = (int64_t)hc.n_claimed;
    // every unique row has a member row, and every id of a unique row is an id of the input: anything else means the bookkeeping on the
    // device went wrong -- say so instead of sizing copies by it
    if (nu > n_rows) CFAIL("more unique rows than rows");
    uint64_t nnz_u = 0;
    if (nu > 0) {
        // the segments in order of their first occurrence
        const dim3 gu((unsigned)((nu + 255) / 256)), bu(256);
        size_t t1 = 0, t2 = 0;
        CCHK(hipcub::DeviceRadixSort::SortPairs(nullptr, t1, d_first.as<uint32_t>(), d_sfirst.as<uint32_t>(), d_claimed.as<uint32_t>(), d_sseg.as<uint32_t>(), (int)nu, 0, end_bit, st));
        CCHK(hipcub::DeviceScan::ExclusiveSum(nullptr, t2, d_ulen.as<uint64_t>(), d_uoff.as<uint64_t>(), (int)nu, st));
        if (t1 > tmp_bytes || t2 > tmp_bytes) CFAIL("temporary storage of the numbering exceeds the worst case");
        CCHK(hipcub::DeviceRadixSort::SortPairs(d_tmp.p, t1, d_first.as<uint32_t>(), d_sfirst.as<uint32_t>(), d_claimed.as<uint32_t>(), d_sseg.as<uint32_t>(), (int)nu, 0, end_bit, st));
        STAGE("sort by first occurrence");
        hipLaunchKernelGGL(k_seg_len, gu, bu, 0, st, nu, d_sseg.as<uint32_t>(), d_sfirst.as<uint32_t>(), d_rp.as<uint64_t>(), d_ulen.as<uint64_t>());
        CCHK(hipcub::DeviceScan::ExclusiveSum(d_tmp.p, t2, d_ulen.as<uint64_t>(), d_uoff.as<uint64_t>(), (int)nu, st));
        // the size of the output is known, and checked, before anything is written into it
        uint64_t last_off = 0, last_len = 0;
        CCHK(hipMemcpyAsync(&last_off, d_uoff.as<uint64_t>() + (nu - 1), 8, hipMemcpyDeviceToHost, st));
        CCHK(hipMemcpyAsync(&last_len, d_ulen.as<uint64_t>() + (nu - 1), 8, hipMemcpyDeviceToHost, st));
        CCHK(hipStreamSynchronize(st));
        nnz_u = last_off + last_len;
        if (nnz_u > nnz) CFAIL("unique rows hold more ids than the input");
        hipLaunchKernelGGL(k_row_emit, gu, bu, 0, st, nu, nnz_u, d_sseg.as<uint32_t>(), d_sfirst.as<uint32_t>(), d_rp.as<uint64_t>(), d_ci.as<int32_t>(), d_cntd.as<unsigned long long>(),
                           d_cnt1.as<unsigned long long>(), d_uid1.as<uint32_t>(), d_uidd.as<uint32_t>(), d_uoff.as<uint64_t>(), d_orp.as<uint64_t>(), d_oci.as<int32_t>(),
                           d_ow.as<long long>());
    }
    if (row_map_out)
        hipLaunchKernelGGL(k_row_map, dim3((unsigned)((n_rows + 255) / 256)), dim3(256), 0, st, n_rows, d_seg.as<uint32_t>(), d_uid1.as<uint32_t>(), d_uidd.as<uint32_t>(), d_map.as<int32_t>());
    STAGE("emit + map");
    CCHK(hipGetLastError());
    CCHK(hipEventRecord(ev.b, st));
    std::vector<long long> w64;
    try { w64.resize((size_t)nu); } catch (const std::bad_alloc &) { return EMSAR_HIP_ERR_OOM; }
    if (nu) {
        CCHK(hipMemcpyAsync(row_ptr_out, d_orp.p, (size_t)nu * 8, hipMemcpyDeviceToHost, st));
        CCHK(hipMemcpyAsync(w64.data(), d_ow.p, (size_t)nu * 8, hipMemcpyDeviceToHost, st));
        if (nnz_u) CCHK(hipMemcpyAsync(col_idx_out, d_oci.p, (size_t)nnz_u * 4, hipMemcpyDeviceToHost, st));
    }
    if (row_map_out) CCHK(hipMemcpyAsync(row_map_out, d_map.p, (size_t)n_rows * 4, hipMemcpyDeviceToHost, st));
    CCHK(hipStreamSynchronize(st));
    row_ptr_out[nu] = nnz_u;
    for (int64_t i = 0; i < nu; i++) {
        if (w64[(size_t)i] > INT32_MAX) return EMSAR_HIP_ERR_ARG;        // a segment's count must fit ReadCount (int)
        weight_out[i] = (int32_t)w64[(size_t)i];
    }
    *n_unique_out = nu;
    if (stats) {
        float ms = 0;
        CCHK(hipEventElapsedTime(&ms, ev.a, ev.b));
        memset(stats, 0, sizeof(*stats));
        stats->kernel_ms = ms;
        stats->total_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        stats->n_rows = n_rows; stats->nnz = (int64_t)nnz; stats->n_unique = nu; stats->nnz_unique = (int64_t)nnz_u;
        stats->table_slots = max_parts * kPartSlots;       // LDS slots over all partitions of the largest round
        stats->rounds = rounds;
        // algorithmic bytes: the CSR once for the hash, once for the compare against the representative, the
        // weights, and the unique rows written
        stats->algorithmic_bytes = 2 * (int64_t)(4 * nnz + 8 * (uint64_t)(n_rows + 1)) + (row_weight ? 4 * n_rows : 0) + 4 * (int64_t)nnz_u + 16 * nu;
    }
#undef STAGE
#undef CFAIL
#undef CCHK
    return EMSAR_HIP_OK;
}
