// collapse.hip -- read -> segment collapse on the device (the integer core of it).
//
// The reference folds reads into segments while it parses them: update_ReadCounts sorts a read's transcript ids,
// looks the tuple up in the rsh bucket and bumps that node's ReadCount (/root/reference/src/emsar_functions.c:838-943,
// update_rshbucket('r') 1597-1624).  Given a read-level incidence (one CSR row per read) this file does the same
// as a data-parallel pass: rows with the same multiset of transcript ids become ONE row whose weight is the sum of
// its members' weights.  Output rows are numbered by first occurrence (the order in which the reference would have
// met the segments), their ids sorted ascending (the reference's insertion order, emsar_functions.c:889).
//
//   k_row_hash    one lane per row: 2 x 64-bit order-independent hash of the multiset (no sort needed)
//   k_row_insert  open-addressing table of one 64-bit word per slot, {hash tag : representative row + 1}; a single CAS
//                 claims a slot AND names its representative, so nobody ever waits for anybody; a tag match is
//                 confirmed by comparing both hashes, the length and finally the multisets themselves -> exact
//   k_row_flag    the member with the smallest row id of every slot is its first occurrence
//   (hipCUB exclusive sums: unique id of every first occurrence, offsets of the output rows)
//   k_row_emit    copy the first occurrences out, sort each one's ids in place (insertion sort, rows are short)
//   k_row_map     original row -> unique row
// All integer / byte work, bound by HBM: the CSR is read twice (hash, compare) and the table is hit at random.
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

__global__ __launch_bounds__(256) void k_row_hash(int64_t n_rows, const uint64_t *__restrict__ rp, const int32_t *__restrict__ ci,
                                                  uint64_t *__restrict__ h1, uint64_t *__restrict__ h2, int weak) {
    const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (r >= n_rows) return;
    const uint64_t b = rp[r], e = rp[r + 1];
    uint64_t a = 0x9e3779b97f4a7c15ull * (e - b + 1), c = 0;
    if (weak) { h1[r] = a; h2[r] = 1ull; return; }   // test hook: every row of one length collides in both hashes and in the tag
    for (uint64_t k = b; k < e; k++) {
        const uint64_t m = mix64((uint64_t)(uint32_t)ci[k] + 0x632be59bd9b4e019ull);
        a += m;                      // sums of per-element mixes: invariant under permutation, sensitive to multiplicity
        c += mix64(m ^ 0xd6e8feb86659fd93ull) | 1ull;
    }
    h1[r] = a; h2[r] = c ? c : 1ull;     // 0 means 'not published yet' in a table slot
}

// multiset equality of two unsorted id lists of the same length n (rows are short; O(n^2) only on a full hash match)
__device__ bool same_multiset(const int32_t *x, const int32_t *y, uint64_t n) {
    for (uint64_t i = 0; i < n; i++) {
        const int32_t v = x[i];
        int cx = 0, cy = 0;
        for (uint64_t j = 0; j < n; j++) { cx += x[j] == v; cy += y[j] == v; }
        if (cx != cy) return false;
    }
    return true;
}

// The table is three parallel arrays on purpose.  Read counts are heavily skewed (the single-transcript row of a highly
// expressed transcript collects percent of all reads), so the lines of a hot segment are hit by hundreds of thousands of
// rows; what limits the kernel is the rate at which ONE L2 line can be served.  Measured on 10M rows (config 3 x 0.2):
// claim word, first occurrence and count in three arrays 4.1 ms; all of a segment's state packed into one 64-byte
// slot 19.8 ms (every load and atomic of a hot segment queues on the same line); counts and first occurrences
// pre-combined per workgroup in LDS with 8 rows per thread 7 ms (the probe chains of a thread run one after another);
// one leader per distinct row of a 1024-row workgroup elected in an LDS table, only leaders probing the global table
// 4.8 ms (total 6.2 against 5.6): duplicates inside a workgroup are too few, the cost is the ~8 random lines EVERY row
// touches (table word, the representative's two hashes, its row_ptr pair, its ids, first, count).
__global__ __launch_bounds__(256) void k_row_insert(int64_t n_rows, const uint64_t *__restrict__ rp, const int32_t *__restrict__ ci,
                                                    const int32_t *__restrict__ wgt, const uint64_t *__restrict__ h1, const uint64_t *__restrict__ h2,
                                                    unsigned long long *__restrict__ table, uint64_t mask, int32_t *__restrict__ slot_of,
                                                    int32_t *__restrict__ first, unsigned long long *__restrict__ cnt) {
    const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (r >= n_rows) return;
    const int64_t w = wgt ? wgt[r] : 1;
    const uint64_t b = rp[r], n = rp[r + 1] - b;
    if (n == 0 || w <= 0) { slot_of[r] = -1; return; }          // empty rows and rows without weight vanish
    const uint64_t a = h1[r], c = h2[r];
    const unsigned long long mine = ((a >> 32) << 32) | (unsigned long long)(uint32_t)(r + 1);
    uint64_t idx = mix64(a ^ c) & mask;
    for (;;) {
        unsigned long long v = table[idx];
        if (v == 0ull) {
            v = atomicCAS(&table[idx], 0ull, mine);
            if (v == 0ull) break;                               // claimed: this row is the slot's representative
        }
        if ((v >> 32) == (a >> 32)) {
            const int64_t rep = (int64_t)(uint32_t)v - 1;
            const uint64_t rb = rp[rep];
            if (h1[rep] == a && h2[rep] == c && rp[rep + 1] - rb == n && same_multiset(ci + b, ci + rb, n)) break;
        }
        idx = (idx + 1) & mask;
    }
    slot_of[r] = (int32_t)idx;
    if ((int32_t)r < first[idx]) atomicMin(&first[idx], (int32_t)r);     // first[] only decreases: a stale read costs one atomic at most
    atomicAdd(&cnt[idx], (unsigned long long)w);
}

__global__ __launch_bounds__(256) void k_row_flag(int64_t n_rows, const uint64_t *__restrict__ rp, const int32_t *__restrict__ slot_of,
                                                  const int32_t *__restrict__ first, int32_t *__restrict__ flag, uint64_t *__restrict__ flen) {
    const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (r >= n_rows) return;
    const int32_t s = slot_of[r];
    const bool f = s >= 0 && first[s] == (int32_t)r;
    flag[r] = f ? 1 : 0;
    flen[r] = f ? rp[r + 1] - rp[r] : 0;
}

__global__ __launch_bounds__(256) void k_row_emit(int64_t n_rows, const uint64_t *__restrict__ rp, const int32_t *__restrict__ ci,
                                                  const int32_t *__restrict__ slot_of, const int32_t *__restrict__ flag,
                                                  const int32_t *__restrict__ uid, const uint64_t *__restrict__ uoff,
                                                  const unsigned long long *__restrict__ cnt, uint64_t *__restrict__ out_rp,
                                                  int32_t *__restrict__ out_ci, long long *__restrict__ out_w) {
    const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (r >= n_rows || !flag[r]) return;
    const uint64_t b = rp[r], n = rp[r + 1] - b, o = uoff[r];
    int32_t *dst = out_ci + o;
    for (uint64_t i = 0; i < n; i++) {                           // insertion sort while copying
        const int32_t v = ci[b + i];
        uint64_t j = i;
        while (j > 0 && dst[j - 1] > v) { dst[j] = dst[j - 1]; j--; }
        dst[j] = v;
    }
    out_rp[uid[r]] = o;
    out_w[uid[r]] = (long long)cnt[slot_of[r]];
}

__global__ __launch_bounds__(256) void k_row_map(int64_t n_rows, const int32_t *__restrict__ slot_of, const int32_t *__restrict__ first,
                                                 const int32_t *__restrict__ uid, int32_t *__restrict__ row_map) {
    const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (r >= n_rows) return;
    const int32_t s = slot_of[r];
    row_map[r] = s < 0 ? -1 : uid[first[s]];
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
    uint64_t M = 1024;                                   // load factor <= 0.8 even if no two rows are equal
    while (M < (uint64_t)n_rows + (uint64_t)n_rows / 4) M <<= 1;
    DevBuf d_rp, d_ci, d_w, d_h1, d_h2, d_tab, d_slot, d_first, d_cnt, d_flag, d_flen, d_uid, d_uoff, d_orp, d_oci, d_ow, d_map, d_tmp;
    CCHK(d_rp.alloc((size_t)(n_rows + 1) * 8)); CCHK(d_ci.alloc((size_t)nnz * 4));
    if (row_weight) CCHK(d_w.alloc((size_t)n_rows * 4));
    CCHK(d_h1.alloc((size_t)n_rows * 8)); CCHK(d_h2.alloc((size_t)n_rows * 8));
    CCHK(d_tab.alloc((size_t)M * 8)); CCHK(d_slot.alloc((size_t)n_rows * 4)); CCHK(d_first.alloc((size_t)M * 4)); CCHK(d_cnt.alloc((size_t)M * 8));
    CCHK(d_flag.alloc((size_t)n_rows * 4)); CCHK(d_flen.alloc((size_t)n_rows * 8)); CCHK(d_uid.alloc((size_t)n_rows * 4)); CCHK(d_uoff.alloc((size_t)n_rows * 8));
    CCHK(d_orp.alloc((size_t)(n_rows + 1) * 8)); CCHK(d_oci.alloc((size_t)nnz * 4)); CCHK(d_ow.alloc((size_t)n_rows * 8)); CCHK(d_map.alloc((size_t)n_rows * 4));
    CCHK(hipMemcpyAsync(d_rp.p, row_ptr, (size_t)(n_rows + 1) * 8, hipMemcpyHostToDevice, st));
    if (nnz) CCHK(hipMemcpyAsync(d_ci.p, col_idx, (size_t)nnz * 4, hipMemcpyHostToDevice, st));
    if (row_weight) CCHK(hipMemcpyAsync(d_w.p, row_weight, (size_t)n_rows * 4, hipMemcpyHostToDevice, st));
    CCHK(hipMemsetAsync(d_tab.p, 0, (size_t)M * 8, st));
    CCHK(hipMemsetAsync(d_first.p, 0x7F, (size_t)M * 4, st));           // 0x7F7F7F7F: larger than any row id
    CCHK(hipMemsetAsync(d_cnt.p, 0, (size_t)M * 8, st));
    Events ev;
    CCHK(hipEventCreate(&ev.a)); CCHK(hipEventCreate(&ev.b));
    CCHK(hipEventRecord(ev.a, st));
    const dim3 grid((unsigned)((n_rows + 255) / 256)), block(256);
    const char *weak_env = getenv("EMSAR_HIP_COLLAPSE_WEAK_HASH");      // tests: force full hash collisions (small inputs only: probing becomes O(distinct rows))
    const int weak_hash = weak_env && atoi(weak_env) != 0;
    hipLaunchKernelGGL(k_row_hash, grid, block, 0, st, n_rows, d_rp.as<uint64_t>(), d_ci.as<int32_t>(), d_h1.as<uint64_t>(), d_h2.as<uint64_t>(), weak_hash);
    hipLaunchKernelGGL(k_row_insert, grid, block, 0, st, n_rows, d_rp.as<uint64_t>(), d_ci.as<int32_t>(), row_weight ? d_w.as<int32_t>() : nullptr,
                       d_h1.as<uint64_t>(), d_h2.as<uint64_t>(), d_tab.as<unsigned long long>(), M - 1, d_slot.as<int32_t>(), d_first.as<int32_t>(),
                       d_cnt.as<unsigned long long>());
    hipLaunchKernelGGL(k_row_flag, grid, block, 0, st, n_rows, d_rp.as<uint64_t>(), d_slot.as<int32_t>(), d_first.as<int32_t>(), d_flag.as<int32_t>(),
                       d_flen.as<uint64_t>());
    CCHK(hipGetLastError());
    size_t tb1 = 0, tb2 = 0;
    CCHK(hipcub::DeviceScan::ExclusiveSum(nullptr, tb1, d_flag.as<int32_t>(), d_uid.as<int32_t>(), (int)n_rows, st));
    CCHK(hipcub::DeviceScan::ExclusiveSum(nullptr, tb2, d_flen.as<uint64_t>(), d_uoff.as<uint64_t>(), (int)n_rows, st));
    CCHK(d_tmp.alloc(std::max(tb1, tb2)));
    CCHK(hipcub::DeviceScan::ExclusiveSum(d_tmp.p, tb1, d_flag.as<int32_t>(), d_uid.as<int32_t>(), (int)n_rows, st));
    CCHK(hipcub::DeviceScan::ExclusiveSum(d_tmp.p, tb2, d_flen.as<uint64_t>(), d_uoff.as<uint64_t>(), (int)n_rows, st));
    hipLaunchKernelGGL(k_row_emit, grid, block, 0, st, n_rows, d_rp.as<uint64_t>(), d_ci.as<int32_t>(), d_slot.as<int32_t>(), d_flag.as<int32_t>(),
                       d_uid.as<int32_t>(), d_uoff.as<uint64_t>(), d_cnt.as<unsigned long long>(), d_orp.as<uint64_t>(), d_oci.as<int32_t>(),
                       d_ow.as<long long>());
    hipLaunchKernelGGL(k_row_map, grid, block, 0, st, n_rows, d_slot.as<int32_t>(), d_first.as<int32_t>(), d_uid.as<int32_t>(), d_map.as<int32_t>());
    CCHK(hipGetLastError());
    CCHK(hipEventRecord(ev.b, st));
    // sizes: the last row's exclusive sums + its own flag / length
    int32_t last_uid = 0, last_flag = 0; uint64_t last_off = 0, last_len = 0;
    CCHK(hipMemcpyAsync(&last_uid, d_uid.as<int32_t>() + (n_rows - 1), 4, hipMemcpyDeviceToHost, st));
    CCHK(hipMemcpyAsync(&last_flag, d_flag.as<int32_t>() + (n_rows - 1), 4, hipMemcpyDeviceToHost, st));
    CCHK(hipMemcpyAsync(&last_off, d_uoff.as<uint64_t>() + (n_rows - 1), 8, hipMemcpyDeviceToHost, st));
    CCHK(hipMemcpyAsync(&last_len, d_flen.as<uint64_t>() + (n_rows - 1), 8, hipMemcpyDeviceToHost, st));
    CCHK(hipStreamSynchronize(st));
    const int64_t nu = (int64_t)last_uid + last_flag;
    const uint64_t nnz_u = last_off + last_len;
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
        stats->table_slots = (int64_t)M;
        // algorithmic bytes: the CSR once for the hash, once for the compare against the representative, the
        // weights, and the unique rows written
        stats->algorithmic_bytes = 2 * (int64_t)(4 * nnz + 8 * (uint64_t)(n_rows + 1)) + (row_weight ? 4 * n_rows : 0) + 4 * (int64_t)nnz_u + 16 * nu;
    }
#undef CCHK
    return EMSAR_HIP_OK;
}
