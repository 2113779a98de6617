// kernels_windowed.hpp -- k_pass_windowed (WINDOWED layout) and k_pass_csr (the caller's CSR): the earlier pass kernels, kept as layouts 2 and 1
#pragma once
// included by emsar_hip.hip only (one translation unit: the kernels live in its anonymous namespace)

namespace {

// ------------------------------------------------------------------------------------------------
// k_pass_windowed: one EM pass (or a plain row-value scatter) over the WINDOWED layout.
//   chunks[blockIdx.x]  -> slices [slice_begin, +n_slices), LDS window [lo, lo+width)
//   a wave owns one 256-row slice at a time; lane l handles rows 4l..4l+3 of it; the j-th tids of those four
//   rows are ONE int4 at ent[slice_off + j*256 + 4l]  (1 KiB contiguous per wave load)
// Up to 8 loads (8 KiB per wave) are issued before the first use, the tids then stay in registers for both
// the E-step sums and the M-step adds; rows longer than 8 are streamed in segments of 8 and re-read (L2) for
// the adds.
// HBM traffic per pass: ent once (4 B per stored slot), slice_off (8 B per 256 rows), optional row weights;
// theta window loads and acc window flushes are O(n_tx + chunks*family) and stay in L2.
// ------------------------------------------------------------------------------------------------
constexpr int kSeg = 8;  // int4 loads in flight per lane

struct Window {
    const double *th_w; double *acc_w; const double *theta; double *acc; int lo; unsigned width;
    __device__ __forceinline__ double get(int t) const {
        unsigned d = (unsigned)(t - lo);
        return d < width ? th_w[d] : theta[t];
    }
    __device__ __forceinline__ void add(int t, double v) const {
        unsigned d = (unsigned)(t - lo);
        if (d < width) lds_add_f64(&acc_w[d], v);
        else atomic_add_f64(&acc[t], v);
    }
};

// The segment length n (1..8) is wave-uniform; each length gets its own straight-line code so that all n
// loads are issued back to back (n KiB in flight per wave) with no control flow between them.
template <int N>
__device__ __forceinline__ void load_n(int4 (&q)[N], const int4 *e) {
#pragma unroll
    for (int j = 0; j < N; j++) q[j] = e[(size_t)j * 64];
}

template <int N>
__device__ __forceinline__ void sum_n(const int4 (&q)[N], const Window &W, double (&S)[4]) {
#pragma unroll
    for (int j = 0; j < N; j++) {
        int4 t = q[j];
        if (t.x >= 0) S[0] += W.get(t.x);
        if (t.y >= 0) S[1] += W.get(t.y);
        if (t.z >= 0) S[2] += W.get(t.z);
        if (t.w >= 0) S[3] += W.get(t.w);
    }
}

// M-step adds of one segment.  The four rows of a lane are neighbours in the sorted order and often carry
// the same tid in column j: equal neighbours are merged in registers first, so that one LDS atomic carries
// up to four contributions (an LDS f64 atomic costs ~6 cycles per extra lane on the same address).
template <int N>
__device__ __forceinline__ void add_n(const int4 (&q)[N], const Window &W, const double (&w)[4]) {
#pragma unroll
    for (int j = 0; j < N; j++) {
        int4 t = q[j];
        double v0 = t.x >= 0 ? w[0] : 0.0, v1 = t.y >= 0 ? w[1] : 0.0;
        double v2 = t.z >= 0 ? w[2] : 0.0, v3 = t.w >= 0 ? w[3] : 0.0;
        if (t.y == t.x) { v1 += v0; v0 = 0.0; }
        if (t.z == t.y) { v2 += v1; v1 = 0.0; }
        if (t.w == t.z) { v3 += v2; v2 = 0.0; }
        if (v0 != 0.0) W.add(t.x, v0);
        if (v1 != 0.0) W.add(t.y, v1);
        if (v2 != 0.0) W.add(t.z, v2);
        if (v3 != 0.0) W.add(t.w, v3);
    }
}

template <bool WEIGHTED, int MODE>
__device__ __forceinline__ void row_weights(const double (&S)[4], const int32_t *wgt, uint64_t row0, double (&w)[4], double &ll) {
    double r[4] = {1.0, 1.0, 1.0, 1.0};
    if (WEIGHTED) {
        int4 rw = *reinterpret_cast<const int4 *>(wgt + row0);
        r[0] = rw.x; r[1] = rw.y; r[2] = rw.z; r[3] = rw.w;
    }
#pragma unroll
    for (int i = 0; i < 4; i++) {
        bool live = (S[i] > 0.0) && (r[i] > 0.0);
        w[i] = live ? r[i] / S[i] : 0.0;
        if (MODE == MODE_EM_LL && live) ll += r[i] * log(S[i]);
    }
}

__device__ __forceinline__ void scatter_weights(const double *rowval, uint64_t row0, double (&w)[4]) {
    const double2 *rv = reinterpret_cast<const double2 *>(rowval + row0);
    double2 a = rv[0], b = rv[1];
    w[0] = a.x; w[1] = a.y; w[2] = b.x; w[3] = b.y;
}

// A slice whose rows have at most 8 tids: all N loads (N KiB per wave) are issued back to back, the tids stay
// in registers for the E-step sums and the M-step adds.  One straight-line instance per N.
template <int N, bool WEIGHTED, int MODE>
__device__ __forceinline__ void slice_short(const int4 *e, uint64_t row0, const Window &W, const int32_t *wgt,
                                            const double *rowval, double &ll) {
    int4 q[N];
    double w[4];
    load_n<N>(q, e);
    if (MODE == MODE_SCATTER) {
        scatter_weights(rowval, row0, w);
    } else {
        double S[4] = {0.0, 0.0, 0.0, 0.0};
        sum_n<N>(q, W, S);
        row_weights<WEIGHTED, MODE>(S, wgt, row0, w, ll);
    }
    add_n<N>(q, W, w);
}

// Rows with more than 8 tids: streamed in segments of 8 loads for the sums, re-read (L2) for the adds.
template <bool WEIGHTED, int MODE>
__device__ __forceinline__ void slice_long(const int4 *e, int k, uint64_t row0, const Window &W, const int32_t *wgt,
                                           const double *rowval, double &ll) {
    double w[4];
    const int nfull = k / kSeg, rem = k % kSeg;
    if (MODE == MODE_SCATTER) {
        scatter_weights(rowval, row0, w);
    } else {
        double S[4] = {0.0, 0.0, 0.0, 0.0};
        for (int g = 0; g < nfull; g++) {
            int4 q[kSeg];
            load_n<kSeg>(q, e + (size_t)g * kSeg * 64);
            sum_n<kSeg>(q, W, S);
        }
        for (int j = nfull * kSeg; j < k; j++) {
            int4 q[1];
            load_n<1>(q, e + (size_t)j * 64);
            sum_n<1>(q, W, S);
        }
        row_weights<WEIGHTED, MODE>(S, wgt, row0, w, ll);
    }
    for (int g = 0; g < nfull; g++) {
        int4 q[kSeg];
        load_n<kSeg>(q, e + (size_t)g * kSeg * 64);
        add_n<kSeg>(q, W, w);
    }
    for (int j = nfull * kSeg; j < k; j++) {
        int4 q[1];
        load_n<1>(q, e + (size_t)j * 64);
        add_n<1>(q, W, w);
    }
    (void)rem;
}

template <int THREADS, bool WEIGHTED, int MODE>
__global__ __launch_bounds__(THREADS) void k_pass_windowed(const Chunk *__restrict__ chunks,
                                                           const uint64_t *__restrict__ slice_off,
                                                           const int32_t *__restrict__ ent,
                                                           const int32_t *__restrict__ wgt,   // sorted rows, padded
                                                           const double *__restrict__ rowval, // MODE_SCATTER
                                                           const double *__restrict__ theta,
                                                           double *__restrict__ acc, double *__restrict__ ll_out,
                                                           int window) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    double *th_w = lds;            // [window]
    double *acc_w = lds + window;  // [window]
    __shared__ double red[THREADS / 64];

    const Chunk c = chunks[blockIdx.x];
    const int lo = c.lo, width = c.width;
    for (int i = threadIdx.x; i < width; i += THREADS) {
        if (MODE != MODE_SCATTER) th_w[i] = theta[lo + i];
        acc_w[i] = 0.0;
    }
    __syncthreads();
    const Window W{th_w, acc_w, theta, acc, lo, (unsigned)width};

    const int lane = threadIdx.x & 63;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // keep slice bookkeeping in SGPRs
    double ll = 0.0;
    for (uint32_t s = wave; s < c.n_slices; s += THREADS / 64) {
        const uint32_t gs = c.slice_begin + s;
        const uint64_t off = slice_off[gs];
        const int k = (int)((slice_off[gs + 1] - off) >> 8);
        const int4 *e = reinterpret_cast<const int4 *>(ent + off) + lane;
        const uint64_t row0 = (uint64_t)gs * 256 + 4 * lane;
#define EMSAR_SHORT(NN) case NN: slice_short<NN, WEIGHTED, MODE>(e, row0, W, wgt, rowval, ll); break;
        switch (k) {
            EMSAR_SHORT(1) EMSAR_SHORT(2) EMSAR_SHORT(3) EMSAR_SHORT(4)
            EMSAR_SHORT(5) EMSAR_SHORT(6) EMSAR_SHORT(7) EMSAR_SHORT(8)
            default: slice_long<WEIGHTED, MODE>(e, k, row0, W, wgt, rowval, ll); break;
        }
#undef EMSAR_SHORT
    }
    __syncthreads();
    for (int i = threadIdx.x; i < width; i += THREADS) {
        double v = acc_w[i];
        if (v != 0.0) atomic_add_f64(&acc[lo + i], v);
    }
    if (MODE == MODE_EM_LL) {
        double t = block_sum<THREADS>(ll, red);
        if (threadIdx.x == 0 && t != 0.0) atomic_add_f64(ll_out, t);
    }
}


// ------------------------------------------------------------------------------------------------
// k_pass_csr: the same pass on the caller's CSR (any row order), one lane per row.
// ------------------------------------------------------------------------------------------------
template <typename PTR, bool WEIGHTED, int MODE>
__global__ __launch_bounds__(256) void k_pass_csr(int64_t n_rows, const PTR *__restrict__ row_ptr,
                                                  const int32_t *__restrict__ col, const int32_t *__restrict__ wgt,
                                                  const double *__restrict__ rowval, const double *__restrict__ theta,
                                                  double *__restrict__ acc, double *__restrict__ ll_out) {
    __shared__ double red[4];
    double ll = 0.0;
    for (int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x; r < n_rows; r += (int64_t)gridDim.x * 256) {
        const uint64_t b = row_ptr[r], e = row_ptr[r + 1];
        double w;
        if (MODE == MODE_SCATTER) {
            w = rowval[r];
        } else {
            double S = 0.0;
            for (uint64_t k = b; k < e; k++) S += theta[col[k]];
            double rw = WEIGHTED ? (double)wgt[r] : 1.0;
            bool live = (S > 0.0) && (rw > 0.0);
            w = live ? rw / S : 0.0;
            if (MODE == MODE_EM_LL && live) ll += rw * log(S);
        }
        if (w != 0.0)
            for (uint64_t k = b; k < e; k++) atomic_add_f64(&acc[col[k]], w);
    }
    if (MODE == MODE_EM_LL) {
        double t = block_sum<256>(ll, red);
        if (threadIdx.x == 0 && t != 0.0) atomic_add_f64(ll_out, t);
    }
}

}  // namespace
