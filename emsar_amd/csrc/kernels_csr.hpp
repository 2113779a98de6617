// kernels_csr.hpp -- k_pass_csr: the EM pass on the caller's CSR as it is (layout 1; also the leftover rows of the TILED layout)
#pragma once
// included by emsar_hip.hip only (one translation unit: the kernels live in its anonymous namespace)

namespace {

// ------------------------------------------------------------------------------------------------
// k_pass_csr: the same pass on the caller's CSR (any row order), one lane per row.
// ------------------------------------------------------------------------------------------------
template <typename PTR, bool WEIGHTED, int MODE>
__global__ __launch_bounds__(256) void k_pass_csr(int64_t n_rows, const PTR *__restrict__ row_ptr,
                                                  const int32_t *__restrict__ col, const int32_t *__restrict__ wgt,
                                                  const double *__restrict__ rowval, const double *__restrict__ theta,
                                                  double *__restrict__ acc, double *__restrict__ ll_out, Fx fx) {
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
        if (w != 0.0) {
            if (MODE != MODE_SCATTER && fx.mass != 0.0) {        // deterministic mode: mass in fixed point (kernels_common.hpp)
                for (uint64_t k = b; k < e; k++) { const int32_t t = col[k]; atomic_add_i64(&acc[t], __double2ll_rn(w * theta[t] * fx.mass)); }
            } else {
                for (uint64_t k = b; k < e; k++) atomic_add_f64(&acc[col[k]], w);
            }
        }
    }
    if (MODE == MODE_EM_LL) {
        double t = block_sum<256>(ll, red);
        if (threadIdx.x == 0 && t != 0.0) ll_add(ll_out, t, fx.ll);
    }
}

}  // namespace
