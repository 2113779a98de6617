// collapse.hip -- read -> segment collapse on the device (the integer core of it).
//
// The reference folds reads into segments while it parses them: update_ReadCounts sorts a read's transcript ids,
// looks the tuple up in the rsh bucket and bumps that node's ReadCount (/root/reference/src/emsar_functions.c:838-943,
// update_rshbucket('r') 1597-1624).  Given a read-level incidence (one CSR row per read) this file does the same
// as a data-parallel pass: rows with the same multiset of transcript ids become ONE row whose weight is the sum of
// its members' weights.  Output rows are numbered by first occurrence (the order in which the reference would have
// met the segments), their ids sorted ascending (the reference's insertion order, emsar_functions.c:889).
//
// What bounds it is not the stream (the CSR is 1.4 GB on config 3) but (a) the number of RANDOM 64-byte lines a row touches and (b)
// the rate of scattered device-scope atomics, which execute at the memory side at about 20 G/s chip-wide whatever the footprint
// (MI355X_MICROARCH.md, Global float atomics; the same rate was measured here for 64-bit integer adds).  Round 1's version touched
// about eight random lines and made two to three atomics per row: 27.9 ms for config 3.  This one: 8.7 ms.
//
//   k_row_insert  one lane per row.  Single-transcript rows (59 % of config 3) need no table: their segment is named by the
//                 transcript.  Rows of 2..8 ids are sorted in registers (written back only when they were out of order), hashed, and
//                 probe an open-addressing table of 64-bit KEY words (the hash itself; one CAS claims a slot, nobody waits).  The
//                 winner of a slot writes the slot's 64-byte PAYLOAD line -- length, offset of its row, its first eight ids -- and the
//                 workgroup's winners take consecutive numbers ("dense", the order of claiming) under which the slot's count and
//                 first-occurrence words live.  Longer rows are only LISTED.                     [random line 1: the key word]
//   k_long_insert the listed long rows, one per lane (19 % of the rows, 69 % of the ids, most of the distinct segments): eight ids
//                 per step with all loads in flight, then the same probe.  A kernel of their own because next to short rows a wave
//                 lasts as long as its longest row while most lanes idle.
//   k_row_count / k_long_count   (after the kernel boundary the payloads are visible.)  Every row reads the payload line of its slot
//                 and compares ids: rows of up to eight ids against the line itself, longer ones against the winner's row in the CSR.
//                 Equal (always, but for a 64-bit hash collision): its weight and row number are combined with the other rows of
//                 its workgroup in an LDS table, and one atomicAdd (count) per distinct segment and workgroup goes out; the
//                 atomicMin (first occurrence) only when the row number is below the word's current value -- workgroups start
//                 roughly in row order, so it rarely is.  Not equal: the row goes to an overflow list.   [random line 2: the payload]
//   (overflow rounds: the four kernels again on the listed rows with the next hash seed, until the list is empty -- in practice never;
//    the test hook EMSAR_HIP_COLLAPSE_WEAK_HASH makes every row of one length collide in round 0.)
//   k_single_claim  the single-transcript segments that were met join the list of claimed slots
//   hipCUB radix sort of (first occurrence, slot) over the claimed slots -> k_slot_len -> hipCUB exclusive sum:
//                 the unique rows in order of first occurrence and the offsets of their ids
//   k_row_emit    one lane per unique row: ids from the payload line (or the winner's sorted row), count, offset
//   k_row_map     (only when asked for) original row -> unique row, through the payload line
// The table is sized for one segment per four rows first (probe chains that overrun start the call over with the worst-case size).
// All integer / byte work.  Exact: a row is only ever counted into a slot after its ids were compared with the slot's.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include <algorithm>
#include <chrono>
#include <cstdint>
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

constexpr int kInline = 8;                    // ids kept in the payload line
constexpr uint32_t kNoSlot = 0xFFFFFFFFu;
constexpr uint32_t kNoRow = 0x7FFFFFFFu;
constexpr uint32_t kSingle = 0x80000000u;     // slot_of / claimed: kSingle | tid = the single-transcript segment of tid (no table slot)
struct __attribute__((aligned(64))) Payload {
    uint32_t dense;                           // number of the slot in order of claiming: index of its first-occurrence and count words
    uint32_t len;
    uint64_t off;                             // the winner's row in the (sorted) CSR
    uint32_t uid;                             // number of the unique row (k_row_emit)
    uint32_t pad[3];
    int32_t ids[kInline];
};
static_assert(sizeof(Payload) == 64, "one line per slot");
struct Counters { unsigned n_claimed, n_over[2], full, n_long; };
constexpr int kInsertThreads = 1024;         // one append to the claimed list per workgroup: few of them, they all hit one counter
constexpr unsigned kMaxProbes = 4096;         // a probe chain this long means the (optimistically sized) table is too full

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

// Rows longer than kInline ids are 19 % of config 3's reads but carry 69 % of the ids, most of the distinct segments are among
// them (so they are the rows that claim slots and miss the caches), and a wave of 64 consecutive rows holds a dozen of them with
// lengths anywhere up to 100.  Lane by lane next to the short rows, the wave lasts as long as its longest row while most lanes idle;
// regrouped inside the workgroup, a fifth of the threads work while the rest wait at a barrier.  So the row kernels only LIST the
// long rows (one append per workgroup) and a kernel of its own takes one listed row per lane, every lane busy.
// every thread of the workgroup calls it (barriers inside)
__device__ __forceinline__ void list_append(bool have, uint32_t item, uint32_t *list, unsigned *n_list, unsigned &lds_n, unsigned &lds_base) {
    if (threadIdx.x == 0) lds_n = 0;
    __syncthreads();
    unsigned pos = 0;
    if (have) pos = atomicAdd(&lds_n, 1u);
    __syncthreads();
    if (threadIdx.x == 0 && lds_n) lds_base = atomicAdd(n_list, lds_n);
    __syncthreads();
    if (have) list[lds_base + pos] = item;
}

// probe for key a: the slot where it lives, claimed by this row (won) or by an earlier one; kNoSlot when the table is too full
__device__ __forceinline__ uint32_t probe_insert(unsigned long long *keys, uint64_t mask, uint64_t a, Counters *cnt, bool &won) {
    uint64_t idx = mix64(a) & mask;
    unsigned probes = 0;
    won = false;
    for (;;) {
        unsigned long long k = keys[idx];
        if (k == 0ull) {
            k = atomicCAS(&keys[idx], 0ull, (unsigned long long)a);
            if (k == 0ull) { won = true; break; }                 // claimed: this row's ids describe the slot
        }
        if (k == (unsigned long long)a) break;
        if (++probes > kMaxProbes) { cnt->full = 1u; return kNoSlot; }     // the host starts over with a table for the worst case
        idx = (idx + 1) & mask;
    }
    return (uint32_t)idx;
}
__device__ __forceinline__ void write_payload(Payload *payload, uint32_t slot, uint64_t len, uint64_t b, const int32_t (&v)[kInline]) {
    Payload P;
    P.dense = 0; P.len = (uint32_t)len; P.off = b; P.uid = 0; P.pad[0] = P.pad[1] = P.pad[2] = 0;
#pragma unroll
    for (int j = 0; j < kInline; j++) P.ids[j] = v[j];
    payload[slot] = P;
}

// the winners of a workgroup take consecutive places in the list of claimed slots; the slot's words in claim order are dense, so
// that the atomics of the count kernels stay in the caches
__device__ __forceinline__ void claim_append(bool won, uint32_t slot, uint32_t *claimed, uint32_t *first_d,
                                             unsigned long long *cnt_d, Payload *payload, Counters *cnt, unsigned &lds_n, unsigned &lds_base) {
    if (threadIdx.x == 0) lds_n = 0;
    __syncthreads();
    unsigned pos = 0;
    if (won) pos = atomicAdd(&lds_n, 1u);
    __syncthreads();
    if (threadIdx.x == 0 && lds_n) lds_base = atomicAdd(&cnt->n_claimed, lds_n);
    __syncthreads();
    if (won) { const uint32_t d = lds_base + pos; claimed[d] = slot; first_d[d] = kNoRow; cnt_d[d] = 0ull; payload[slot].dense = d; }
}

// Round `seed` of the insert: rows [0, n) or, with `list`, the rows it names.  Short rows here, long rows to long_list.
__global__ __launch_bounds__(kInsertThreads) void k_row_insert(int64_t n, const uint32_t *__restrict__ list, const uint64_t *__restrict__ rp, int32_t *__restrict__ ci,
                                                    const int32_t *__restrict__ wgt, unsigned long long *__restrict__ keys, Payload *__restrict__ payload,
                                                    uint64_t mask, uint32_t *__restrict__ slot_of, uint32_t *__restrict__ claimed, uint32_t *__restrict__ first_d,
                                                    unsigned long long *__restrict__ cnt_d, uint32_t *__restrict__ long_list, Counters *cnt, uint64_t seed, int weak) {
    __shared__ unsigned lds_n, lds_base;
    const int64_t i = (int64_t)blockIdx.x * kInsertThreads + threadIdx.x;
    bool won = false, is_long = false;
    uint32_t slot = kNoSlot, long_row = 0;
    if (i < n) {
        const int64_t r = list ? (int64_t)list[i] : i;
        const uint64_t b = rp[r], len = rp[r + 1] - b;
        const int64_t w = wgt ? wgt[r] : 1;
        if (len > (uint64_t)kInline && w > 0) { is_long = true; long_row = (uint32_t)r; }
        else {
            if (len == 1 && w > 0) {
                // 59 % of config 3's reads hit one transcript only: their segment is named by the transcript itself -- no hash, no
                // probe, nothing to compare; counted into per-transcript words by k_row_count
                slot = kSingle | (uint32_t)ci[b];
            } else if (len != 0 && w > 0) {                       // empty rows and rows without weight vanish
                int32_t v[kInline];
                uint64_t a = 0x9e3779b97f4a7c15ull * (len + 1 + seed);
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
                for (int j = 0; j < kInline; j++) if ((uint64_t)j < len) a += id_mix(v[j], seed);
                if (weak && seed == 0) a = 0x9e3779b97f4a7c15ull * (len + 1);   // test hook: every row of one length collides in round 0
                if (a == 0) a = 1;
                slot = probe_insert(keys, mask, a, cnt, won);
                if (won) write_payload(payload, slot, len, b, v);
            }
            slot_of[r] = slot;
        }
    }
    list_append(is_long, long_row, long_list, &cnt->n_long, lds_n, lds_base);
    claim_append(won, slot, claimed, first_d, cnt_d, payload, cnt, lds_n, lds_base);
}

// the same for the listed long rows, one per lane
__global__ __launch_bounds__(256) void k_long_insert(const uint32_t *__restrict__ long_list, const uint64_t *__restrict__ rp, int32_t *__restrict__ ci,
                                                     unsigned long long *__restrict__ keys, Payload *__restrict__ payload, uint64_t mask,
                                                     uint32_t *__restrict__ slot_of, uint32_t *__restrict__ claimed, uint32_t *__restrict__ first_d,
                                                     unsigned long long *__restrict__ cnt_d, Counters *cnt, uint64_t seed, int weak) {
    __shared__ unsigned lds_n, lds_base;
    const unsigned n = cnt->n_long;
    const unsigned i = blockIdx.x * 256 + threadIdx.x;
    if (blockIdx.x * 256 >= n) return;                           // the grid covers the worst case (workgroup-uniform exit)
    bool won = false;
    uint32_t slot = kNoSlot;
    if (i < n) {
        const int64_t r = (int64_t)long_list[i];
        const uint64_t b = rp[r], len = rp[r + 1] - b;
        // eight ids per step, all eight loads in flight together (one id per step is one trip to memory per id: 50 us for 100 ids)
        int32_t *x = ci + b;
        int32_t v[kInline];
        uint64_t a = 0x9e3779b97f4a7c15ull * (len + 1 + seed);
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
            if (j0 == 0) {
#pragma unroll
                for (int j = 0; j < kInline; j++) v[j] = c[j];
            }
        }
        if (!sorted) {                                            // rare (aligners and our own parser emit sorted ids): insertion sort in place,
            for (uint64_t j = 1; j < len; j++) {                  // then the hash again (a sum: the order did not matter) and the first ids
                const int32_t t = x[j];
                uint64_t k = j;
                while (k > 0 && x[k - 1] > t) { x[k] = x[k - 1]; k--; }
                if (k != j) x[k] = t;
            }
#pragma unroll
            for (int j = 0; j < kInline; j++) v[j] = x[j];
        }
        if (weak && seed == 0) a = 0x9e3779b97f4a7c15ull * (len + 1);
        if (a == 0) a = 1;
        slot = probe_insert(keys, mask, a, cnt, won);
        if (won) write_payload(payload, slot, len, b, v);
        slot_of[r] = slot;
    }
    claim_append(won, slot, claimed, first_d, cnt_d, payload, cnt, lds_n, lds_base);
}

constexpr int kAggSlots = 2048;               // LDS table of k_row_count: 1024 rows per workgroup
struct AggLds { uint32_t key[kAggSlots], first[kAggSlots]; unsigned long long cnt[kAggSlots]; };
__device__ __forceinline__ void agg_add(AggLds &A, uint32_t key, uint32_t r, unsigned long long w) {
    uint32_t h = (key * 0x9E3779B1u) >> 21;
    for (;;) {
        uint32_t k = A.key[h];
        if (k == kNoSlot) k = atomicCAS(&A.key[h], kNoSlot, key);
        if (k == kNoSlot || k == key) break;
        h = (h + 1) & (kAggSlots - 1);
    }
    atomicMin(&A.first[h], r);
    atomicAdd(&A.cnt[h], w);
}
__device__ __forceinline__ void agg_flush(AggLds &A, int threads, uint32_t *first_d, unsigned long long *cnt_d,
                                          uint32_t *first_1, unsigned long long *cnt_1) {
    for (int e = threadIdx.x; e < kAggSlots; e += threads) {
        const uint32_t d = A.key[e];
        if (d == kNoSlot) continue;
        uint32_t *f = (d & kSingle) ? first_1 + (d & ~kSingle) : first_d + d;
        unsigned long long *c = (d & kSingle) ? cnt_1 + (d & ~kSingle) : cnt_d + d;
        // workgroups start roughly in row order, so the first occurrence has usually been seen: the word only ever decreases, a
        // stale read costs one atomic at most, and most rows skip theirs (the atomics were two thirds of this kernel)
        if (A.first[e] < *f) atomicMin(f, A.first[e]);
        atomicAdd(c, A.cnt[e]);
    }
}
// short rows of [0, n) (or of `list`): compared with their slot's payload line and counted; long rows are k_long_count's
__global__ __launch_bounds__(1024) void k_row_count(int64_t n, const uint32_t *__restrict__ list, const uint64_t *__restrict__ rp, const int32_t *__restrict__ ci,
                                                    const int32_t *__restrict__ wgt, const Payload *__restrict__ payload, const uint32_t *__restrict__ slot_of,
                                                    uint32_t *__restrict__ first_d, unsigned long long *__restrict__ cnt_d,
                                                    uint32_t *__restrict__ first_1, unsigned long long *__restrict__ cnt_1,
                                                    uint32_t *__restrict__ over, unsigned *__restrict__ n_over) {
    __shared__ AggLds A;
    for (int e = threadIdx.x; e < kAggSlots; e += 1024) { A.key[e] = kNoSlot; A.first[e] = kNoRow; A.cnt[e] = 0ull; }
    const int64_t i = (int64_t)blockIdx.x * 1024 + threadIdx.x;
    __syncthreads();
    if (i < n) {
        const int64_t r = list ? (int64_t)list[i] : i;
        const uint32_t s = slot_of[r];
        // what a row is counted under: kSingle | tid, or the dense number of its slot once its ids were found equal to the slot's
        if (s != kNoSlot && (s & kSingle)) agg_add(A, s, (uint32_t)r, (unsigned long long)(wgt ? wgt[r] : 1));
        else if (s != kNoSlot) {
            const uint64_t b = rp[r], len = rp[r + 1] - b;
            if (len <= (uint64_t)kInline) {
                const int4 *P = reinterpret_cast<const int4 *>(payload + s);
                const int4 q0 = P[0];
                const uint64_t poff = (uint64_t)(uint32_t)q0.z | ((uint64_t)(uint32_t)q0.w << 32);
                bool same = (uint64_t)(uint32_t)q0.y == len;
                if (same && poff != b) {
                    const int4 q2 = P[2], q3 = P[3];
                    const int32_t pid[kInline] = {q2.x, q2.y, q2.z, q2.w, q3.x, q3.y, q3.z, q3.w};
#pragma unroll
                    for (int j = 0; j < kInline; j++) if ((uint64_t)j < len) same &= ci[b + j] == pid[j];
                }
                if (same) agg_add(A, (uint32_t)q0.x, (uint32_t)r, (unsigned long long)(wgt ? wgt[r] : 1));
                else over[atomicAdd(n_over, 1u)] = (uint32_t)r;  // a 64-bit hash collision: next round, next seed
            }
        }
    }
    __syncthreads();
    agg_flush(A, 1024, first_d, cnt_d, first_1, cnt_1);
}
// the listed long rows: compared with the winner's row in the CSR
__global__ __launch_bounds__(256) void k_long_count(const uint32_t *__restrict__ long_list, const Counters *__restrict__ cnt, const uint64_t *__restrict__ rp,
                                                    const int32_t *__restrict__ ci, const int32_t *__restrict__ wgt, const Payload *__restrict__ payload,
                                                    const uint32_t *__restrict__ slot_of, uint32_t *__restrict__ first_d, unsigned long long *__restrict__ cnt_d,
                                                    uint32_t *__restrict__ over, unsigned *__restrict__ n_over) {
    __shared__ AggLds A;
    const unsigned n = cnt->n_long;
    if (blockIdx.x * 256 >= n) return;                           // the grid covers the worst case (workgroup-uniform exit)
    for (int e = threadIdx.x; e < kAggSlots; e += 256) { A.key[e] = kNoSlot; A.first[e] = kNoRow; A.cnt[e] = 0ull; }
    const unsigned i = blockIdx.x * 256 + threadIdx.x;
    __syncthreads();
    if (i < n) {
        const int64_t r = (int64_t)long_list[i];
        const uint32_t s = slot_of[r];
        if (s != kNoSlot) {
            const uint64_t b = rp[r], len = rp[r + 1] - b;
            const int4 q0 = reinterpret_cast<const int4 *>(payload + s)[0];
            const uint64_t poff = (uint64_t)(uint32_t)q0.z | ((uint64_t)(uint32_t)q0.w << 32);
            bool same = (uint64_t)(uint32_t)q0.y == len;
            if (same && poff != b) {
                for (uint64_t j0 = 0; j0 < len && same; j0 += 8) {  // eight ids of either row per step, sixteen loads in flight
                    int32_t x[8], y[8];
#pragma unroll
                    for (int j = 0; j < 8; j++) { const bool in = j0 + (uint64_t)j < len; x[j] = in ? ci[b + j0 + j] : 0; y[j] = in ? ci[poff + j0 + j] : 0; }
#pragma unroll
                    for (int j = 0; j < 8; j++) same &= x[j] == y[j];
                }
            }
            if (same) agg_add(A, (uint32_t)q0.x, (uint32_t)r, (unsigned long long)(wgt ? wgt[r] : 1));
            else over[atomicAdd(n_over, 1u)] = (uint32_t)r;
        }
    }
    __syncthreads();
    agg_flush(A, 256, first_d, cnt_d, first_d, cnt_d);           // no kSingle keys here (null pointers for the last two crash hipcc 7.2 in the inliner)
}

// the single-transcript segments that were met join the claimed slots (after the last round), under the name kSingle | tid
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
__global__ __launch_bounds__(256) void k_slot_len(int64_t nu, const uint32_t *__restrict__ slot, const Payload *__restrict__ payload, uint64_t *__restrict__ len) {
    const int64_t u = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (u < nu) { const uint32_t s = slot[u]; len[u] = (s & kSingle) ? 1 : payload[s].len; }
}
__global__ __launch_bounds__(256) void k_row_emit(int64_t nu, const uint32_t *__restrict__ slot, Payload *__restrict__ payload, const int32_t *__restrict__ ci,
                                                  const unsigned long long *__restrict__ cnt_d, const unsigned long long *__restrict__ cnt_1,
                                                  uint32_t *__restrict__ uid_1, const uint64_t *__restrict__ uoff, uint64_t *__restrict__ out_rp,
                                                  int32_t *__restrict__ out_ci, long long *__restrict__ out_w) {
    const int64_t u = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (u >= nu) return;
    const uint32_t sl = slot[u];
    const uint64_t o = uoff[u];
    out_rp[u] = o;
    if (sl & kSingle) {
        const uint32_t t = sl & ~kSingle;
        out_ci[o] = (int32_t)t; out_w[u] = (long long)cnt_1[t]; uid_1[t] = (uint32_t)u;
        return;
    }
    Payload *P = payload + sl;
    const int4 q0 = reinterpret_cast<const int4 *>(P)[0];
    const uint64_t len = (uint32_t)q0.y;
    out_w[u] = (long long)cnt_d[(uint32_t)q0.x];
    P->uid = (uint32_t)u;
    if (len <= (uint64_t)kInline) {
        const int4 q2 = reinterpret_cast<const int4 *>(P)[2], q3 = reinterpret_cast<const int4 *>(P)[3];
        const int32_t pid[kInline] = {q2.x, q2.y, q2.z, q2.w, q3.x, q3.y, q3.z, q3.w};
#pragma unroll
        for (int j = 0; j < kInline; j++) if ((uint64_t)j < len) out_ci[o + j] = pid[j];
    } else {                                                      // from the winner's sorted row, eight ids per step
        const int32_t *x = ci + ((uint64_t)(uint32_t)q0.z | ((uint64_t)(uint32_t)q0.w << 32));
        for (uint64_t j0 = 0; j0 < len; j0 += 8) {
            int32_t c[8];
#pragma unroll
            for (int j = 0; j < 8; j++) c[j] = j0 + (uint64_t)j < len ? x[j0 + (uint64_t)j] : 0;
#pragma unroll
            for (int j = 0; j < 8; j++) if (j0 + (uint64_t)j < len) out_ci[o + j0 + j] = c[j];
        }
    }
}
__global__ __launch_bounds__(256) void k_row_map(int64_t n_rows, const uint32_t *__restrict__ slot_of, const Payload *__restrict__ payload,
                                                 const uint32_t *__restrict__ uid_1, int32_t *__restrict__ row_map) {
    const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (r >= n_rows) return;
    const uint32_t s = slot_of[r];
    row_map[r] = s == kNoSlot ? -1 : (s & kSingle) ? (int32_t)uid_1[s & ~kSingle] : (int32_t)payload[s].uid;
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
    CCHK(hipSetDevice(emsar_internal_device(ctx)));
    auto t0 = std::chrono::steady_clock::now();
    const uint64_t nnz = row_ptr[n_rows];
    *n_unique_out = 0;
    row_ptr_out[0] = 0;
    if (n_rows == 0) return EMSAR_HIP_OK;
    if (n_rows > (int64_t)1600000000) return EMSAR_HIP_ERR_ARG;          // slots are named by 32-bit numbers
    // The table is sized for the usual case first -- reads outnumber segments several times over, and a table that stays in the
    // Infinity Cache is what makes the probes cheap: one slot per FOUR rows (EMSAR_HIP_COLLAPSE_SHIFT: per 2^shift rows).  A probe
    // chain longer than kMaxProbes raises `full`; the call then starts over with the table no input can fill (load <= 0.8 with every
    // row unique).  Sorting the rows in place the first time round does no harm the second.
    uint64_t M_worst = 1024;
    while (M_worst < (uint64_t)n_rows + (uint64_t)n_rows / 4) M_worst <<= 1;
    int shift = 2;
    if (const char *e = getenv("EMSAR_HIP_COLLAPSE_SHIFT")) { const int v = atoi(e); if (v >= 0 && v <= 8) shift = v; }
    uint64_t M = std::max<uint64_t>(1024, M_worst >> (shift + 1));
    DevBuf d_rp, d_ci, d_w, d_keys, d_pay, d_slot, d_claimed, d_over0, d_over1, d_long, d_cnt, d_first, d_cntd, d_first1, d_cnt1, d_uid1, d_sslot, d_ulen, d_uoff, d_orp, d_oci, d_ow, d_map, d_tmp;
    CCHK(d_rp.alloc((size_t)(n_rows + 1) * 8)); CCHK(d_ci.alloc((size_t)nnz * 4));
    if (row_weight) CCHK(d_w.alloc((size_t)n_rows * 4));
    CCHK(d_slot.alloc((size_t)n_rows * 4)); CCHK(d_claimed.alloc((size_t)n_rows * 4));
    CCHK(d_over0.alloc((size_t)n_rows * 4)); CCHK(d_over1.alloc((size_t)n_rows * 4)); CCHK(d_long.alloc((size_t)n_rows * 4)); CCHK(d_cnt.alloc(sizeof(Counters)));
    // everything the numbering and the emit need, at worst-case size (every row unique), so that nothing is allocated between the kernels
    CCHK(d_first.alloc((size_t)n_rows * 4)); CCHK(d_cntd.alloc((size_t)n_rows * 8));
    CCHK(d_first1.alloc((size_t)std::max(n_tx, 1) * 4)); CCHK(d_cnt1.alloc((size_t)std::max(n_tx, 1) * 8)); CCHK(d_uid1.alloc((size_t)std::max(n_tx, 1) * 4));
    CCHK(d_sslot.alloc((size_t)n_rows * 4)); CCHK(d_ulen.alloc((size_t)n_rows * 8)); CCHK(d_uoff.alloc((size_t)n_rows * 8));
    CCHK(d_orp.alloc((size_t)(n_rows + 1) * 8)); CCHK(d_ow.alloc((size_t)n_rows * 8)); CCHK(d_oci.alloc((size_t)nnz * 4));
    if (row_map_out) CCHK(d_map.alloc((size_t)n_rows * 4));
    int end_bit = 1;
    while (end_bit < 32 && ((uint64_t)1 << end_bit) < (uint64_t)n_rows) end_bit++;
    size_t tb1 = 0, tb2 = 0;
    CCHK(hipcub::DeviceRadixSort::SortPairs(nullptr, tb1, (uint32_t *)nullptr, (uint32_t *)nullptr, (uint32_t *)nullptr, (uint32_t *)nullptr, (int)n_rows, 0, end_bit, st));
    CCHK(hipcub::DeviceScan::ExclusiveSum(nullptr, tb2, (uint64_t *)nullptr, (uint64_t *)nullptr, (int)n_rows, st));
    CCHK(d_tmp.alloc(std::max(tb1, tb2)));
    CCHK(hipMemcpyAsync(d_rp.p, row_ptr, (size_t)(n_rows + 1) * 8, hipMemcpyHostToDevice, st));
    if (nnz) CCHK(hipMemcpyAsync(d_ci.p, col_idx, (size_t)nnz * 4, hipMemcpyHostToDevice, st));
    if (row_weight) CCHK(hipMemcpyAsync(d_w.p, row_weight, (size_t)n_rows * 4, hipMemcpyHostToDevice, st));
    CCHK(d_keys.alloc((size_t)M * 8)); CCHK(d_pay.alloc((size_t)M * sizeof(Payload)));   // payload lines are written by the slot's winner: no memset
    Events ev;
    CCHK(hipEventCreate(&ev.a)); CCHK(hipEventCreate(&ev.b));
    CCHK(hipEventRecord(ev.a, st));
    const char *weak_env = getenv("EMSAR_HIP_COLLAPSE_WEAK_HASH");      // tests: every row of one length collides in round 0 and is told apart by comparison
    const int weak_hash = weak_env && atoi(weak_env) != 0;
    const int32_t *dw = row_weight ? d_w.as<int32_t>() : nullptr;
    uint32_t *over[2] = {d_over0.as<uint32_t>(), d_over1.as<uint32_t>()};
    Counters hc{0, {0, 0}, 0, 0};
    for (;;) {                                                    // once; twice when the optimistic table was too small
        CCHK(hipMemsetAsync(d_keys.p, 0, (size_t)M * 8, st));
        CCHK(hipMemsetAsync(d_cnt.p, 0, sizeof(Counters), st));
        CCHK(hipMemsetAsync(d_first1.p, 0x7F, (size_t)n_tx * 4, st));       // 0x7F7F7F7F: larger than any row number
        CCHK(hipMemsetAsync(d_cnt1.p, 0, (size_t)n_tx * 8, st));
        int64_t n_cur = n_rows;
        const uint32_t *list = nullptr;
        int rounds = 0;
        for (uint64_t seed = 0; n_cur > 0 && !hc.full; seed++) {
            const int o = (int)(seed & 1);
            const dim3 gl((unsigned)((n_cur + 255) / 256)), bl(256);         // long rows: at most all of them; the kernels read the count
            CCHK(hipMemsetAsync(&d_cnt.as<Counters>()->n_long, 0, sizeof(unsigned), st));
            hipLaunchKernelGGL(k_row_insert, dim3((unsigned)((n_cur + kInsertThreads - 1) / kInsertThreads)), dim3(kInsertThreads), 0, st, n_cur, list,
                               d_rp.as<uint64_t>(), d_ci.as<int32_t>(), dw, d_keys.as<unsigned long long>(), d_pay.as<Payload>(), M - 1, d_slot.as<uint32_t>(),
                               d_claimed.as<uint32_t>(), d_first.as<uint32_t>(), d_cntd.as<unsigned long long>(), d_long.as<uint32_t>(), d_cnt.as<Counters>(),
                               seed, weak_hash);
            hipLaunchKernelGGL(k_long_insert, gl, bl, 0, st, d_long.as<uint32_t>(), d_rp.as<uint64_t>(), d_ci.as<int32_t>(), d_keys.as<unsigned long long>(),
                               d_pay.as<Payload>(), M - 1, d_slot.as<uint32_t>(), d_claimed.as<uint32_t>(), d_first.as<uint32_t>(),
                               d_cntd.as<unsigned long long>(), d_cnt.as<Counters>(), seed, weak_hash);
            hipLaunchKernelGGL(k_row_count, dim3((unsigned)((n_cur + 1023) / 1024)), dim3(1024), 0, st, n_cur, list, d_rp.as<uint64_t>(), d_ci.as<int32_t>(), dw,
                               d_pay.as<Payload>(), d_slot.as<uint32_t>(), d_first.as<uint32_t>(), d_cntd.as<unsigned long long>(), d_first1.as<uint32_t>(),
                               d_cnt1.as<unsigned long long>(), over[o], &d_cnt.as<Counters>()->n_over[o]);
            hipLaunchKernelGGL(k_long_count, gl, bl, 0, st, d_long.as<uint32_t>(), d_cnt.as<Counters>(), d_rp.as<uint64_t>(), d_ci.as<int32_t>(), dw,
                               d_pay.as<Payload>(), d_slot.as<uint32_t>(), d_first.as<uint32_t>(), d_cntd.as<unsigned long long>(), over[o],
                               &d_cnt.as<Counters>()->n_over[o]);
            CCHK(hipGetLastError());
            CCHK(hipMemcpyAsync(&hc, d_cnt.p, sizeof(Counters), hipMemcpyDeviceToHost, st));
            CCHK(hipStreamSynchronize(st));
            n_cur = hc.n_over[o];                             // rows whose ids differ from their slot's: again, with the next seed
            list = over[o];
            if (n_cur) CCHK(hipMemsetAsync(&d_cnt.as<Counters>()->n_over[o ^ 1], 0, sizeof(unsigned), st));
            if (++rounds > 64) { emsar_internal_set_error(ctx, "collapse", "hash rounds do not terminate"); return EMSAR_HIP_ERR_HIP; }
        }
        if (!hc.full) break;
        if (M >= M_worst) { emsar_internal_set_error(ctx, "collapse", "hash table full"); return EMSAR_HIP_ERR_HIP; }
        M = M_worst;
        (void)hipFree(d_keys.p); d_keys.p = nullptr; (void)hipFree(d_pay.p); d_pay.p = nullptr;
        CCHK(d_keys.alloc((size_t)M * 8)); CCHK(d_pay.alloc((size_t)M * sizeof(Payload)));
        hc = Counters{0, {0, 0}, 0, 0};
    }
    if (n_tx > 0)
        hipLaunchKernelGGL(k_single_claim, dim3((unsigned)((n_tx + 255) / 256)), dim3(256), 0, st, n_tx, d_first1.as<uint32_t>(), d_cnt1.as<unsigned long long>(),
                       d_claimed.as<uint32_t>(), d_first.as<uint32_t>(), d_cnt.as<Counters>());
    CCHK(hipMemcpyAsync(&hc, d_cnt.p, sizeof(Counters), hipMemcpyDeviceToHost, st));
    CCHK(hipStreamSynchronize(st));
    const int64_t nu = (int64_t)hc.n_claimed;
    // every unique row has a member row, and every id of a unique row is an id of the input: anything else means the bookkeeping on the
    // device went wrong -- say so instead of sizing copies by it
    if (nu > n_rows) { emsar_internal_set_error(ctx, "collapse", "more unique rows than rows"); return EMSAR_HIP_ERR_HIP; }
    uint64_t nnz_u = 0;
    if (nu > 0) {
        // the slots in order of their first occurrence (the overflow lists are free now: sorted keys go there)
        const dim3 gu((unsigned)((nu + 255) / 256)), bu(256);
        size_t t1 = tb1, t2 = tb2;
        CCHK(hipcub::DeviceRadixSort::SortPairs(d_tmp.p, t1, d_first.as<uint32_t>(), d_over0.as<uint32_t>(), d_claimed.as<uint32_t>(), d_sslot.as<uint32_t>(), (int)nu, 0, end_bit, st));
        hipLaunchKernelGGL(k_slot_len, gu, bu, 0, st, nu, d_sslot.as<uint32_t>(), d_pay.as<Payload>(), d_ulen.as<uint64_t>());
        CCHK(hipcub::DeviceScan::ExclusiveSum(d_tmp.p, t2, d_ulen.as<uint64_t>(), d_uoff.as<uint64_t>(), (int)nu, st));
        hipLaunchKernelGGL(k_row_emit, gu, bu, 0, st, nu, d_sslot.as<uint32_t>(), d_pay.as<Payload>(), d_ci.as<int32_t>(), d_cntd.as<unsigned long long>(),
                           d_cnt1.as<unsigned long long>(), d_uid1.as<uint32_t>(), d_uoff.as<uint64_t>(), d_orp.as<uint64_t>(), d_oci.as<int32_t>(), d_ow.as<long long>());
    }
    if (row_map_out)
        hipLaunchKernelGGL(k_row_map, dim3((unsigned)((n_rows + 255) / 256)), dim3(256), 0, st, n_rows, d_slot.as<uint32_t>(), d_pay.as<Payload>(), d_uid1.as<uint32_t>(), d_map.as<int32_t>());
    CCHK(hipGetLastError());
    CCHK(hipEventRecord(ev.b, st));
    if (nu > 0) {
        uint64_t last_off = 0, last_len = 0;
        CCHK(hipMemcpyAsync(&last_off, d_uoff.as<uint64_t>() + (nu - 1), 8, hipMemcpyDeviceToHost, st));
        CCHK(hipMemcpyAsync(&last_len, d_ulen.as<uint64_t>() + (nu - 1), 8, hipMemcpyDeviceToHost, st));
        CCHK(hipStreamSynchronize(st));
        nnz_u = last_off + last_len;
        if (nnz_u > nnz) { emsar_internal_set_error(ctx, "collapse", "unique rows hold more ids than the input"); return EMSAR_HIP_ERR_HIP; }
    }
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
        stats->table_slots = (int64_t)M;        // of the attempt that went through
        // algorithmic bytes: the CSR once for the hash, once for the compare against the representative, the
        // weights, and the unique rows written
        stats->algorithmic_bytes = 2 * (int64_t)(4 * nnz + 8 * (uint64_t)(n_rows + 1)) + (row_weight ? 4 * n_rows : 0) + 4 * (int64_t)nnz_u + 16 * nu;
    }
#undef CCHK
    return EMSAR_HIP_OK;
}
