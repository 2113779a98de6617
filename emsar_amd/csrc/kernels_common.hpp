// kernels_common.hpp -- device scalars of a solve, FP64 atomics, wave / workgroup reductions, pass modes
#pragma once
// included by emsar_hip.hip only (one translation unit: the kernels live in its anonymous namespace)

namespace {
using emsar::Tile;

// ------------------------------------------------------------------------------------------------
// device scalars of one solve (lives in HBM, polled by the host every check_every cycles)
// ------------------------------------------------------------------------------------------------
// A sum that hundreds of workgroups add to with same-address atomics (~10 ns each, served one after another per line)
// gets a 128-byte line of its own: the four sums of k_update_p2 then proceed side by side in four L2 channels.
struct alignas(128) LineF64 {
    double v;
    __host__ __device__ operator double() const { return v; }
    __host__ __device__ LineF64 &operator=(double x) { v = x; return *this; }
};
// A likelihood sum receives one add per workgroup of a pass kernel -- 4.4 k adds on config 3, and same-address atomics are served
// one after another (~5 ns each: 23 us of a 134 us likelihood pass).  So the sum is kept in kLlSlots words on lines of their own,
// workgroup b adds to word b % kLlSlots, and whoever needs the value adds the words up (ll_value / host_ll).
constexpr int kLlSlots = 64;
struct LlSum { LineF64 s[kLlSlots]; };
struct Scal {
    LlSum ll[4];                  // sum_c R_c log S_c at the input of pass 0/1/2 of the cycle; [3] scratch
    LineF64 sr2, sv2, pen1, penx; // SQUAREM norms, sum theta*den of th1 and of the extrapolated point
    double stepmax, s_used;
    unsigned long long delta_bits;  // max_t |dtheta|/(theta+floor) as IEEE bits (non-negative -> integer max)
    unsigned long long delta1_bits; // the same, frozen after the first (plain) pass of a SQUAREM cycle
    int32_t accepted, rejected;
    double sum_a, sum_b;          // generic reductions (normalise)
    long long passes;             // EM passes enqueued before the current cycle (k_cycle_begin keeps it: the cycle may replay from a hipGraph)
    double abs_step_cur;          // emsar_em_params.abs_step scaled to the pass count of the current cycle (0 = rule off)
    unsigned int bad;             // set (never cleared during a solve) when an update met NaN / Inf: the state may look finite again later
};

__device__ __forceinline__ void atomic_add_f64(double *p, double v) {
    // gfx950: global_atomic_add_f64 / ds_add_f64 (no CAS loop; compiled with -munsafe-fp-atomics)
    __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void lds_add_f64(double *p, double v) {
    __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// ---- deterministic mode (emsar_hip_set_deterministic) ------------------------------------------------------------------
// Floating atomics make a sum depend on the order in which its terms arrive.  In deterministic mode every sum that workgroups
// share is kept as a 64-bit INTEGER in fixed point (integer adds commute: any arrival order gives the same bits) at a scale
// fixed per sample:
//   * the M-step accumulators hold MASS, theta_t * sum_c m_ct R_c / S_c -- the reads assigned to t -- which is bounded by the
//     sample's total weight N whatever theta is: scale fx.mass = 2^(61 - ceil(log2(N + 1))), a resolution of N * 2^-61 reads;
//   * the log-likelihood sums are bounded by N * 745: scale fx.ll.
// fx.mass == 0 switches the mode off (plain FP64 atomics).  The sums inside one lane and the DPP / LDS reductions of one
// workgroup are in program order either way.
struct Fx { double mass, ll; };
__device__ __forceinline__ void atomic_add_i64(double *p, long long v) {
    __hip_atomic_fetch_add(reinterpret_cast<long long *>(p), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void lds_add_i64(double *p, long long v) {
    __hip_atomic_fetch_add(reinterpret_cast<long long *>(p), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
// a likelihood partial sum of one workgroup -> the shared word (fixed point when fx_ll != 0)
__device__ __forceinline__ void ll_add(double *ll_out /* word 0 of an LlSum */, double t, double fx_ll) {
    double *p = ll_out + (blockIdx.x % kLlSlots) * (sizeof(LineF64) / sizeof(double));
    if (fx_ll != 0.0) atomic_add_i64(p, __double2ll_rn(t * fx_ll));
    else atomic_add_f64(p, t);
}
// the value of such a sum: its words added in a fixed order
__device__ __forceinline__ double ll_value(const double *ll_word0, double fx_ll) {
    if (fx_ll != 0.0) {
        long long s = 0;
        for (int i = 0; i < kLlSlots; i++) s += __double_as_longlong(ll_word0[i * (sizeof(LineF64) / sizeof(double))]);
        return (double)s / fx_ll;
    }
    double s = 0.0;
    for (int i = 0; i < kLlSlots; i++) s += ll_word0[i * (sizeof(LineF64) / sizeof(double))];
    return s;
}

__device__ __forceinline__ double wave_sum(double v) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

template <int THREADS>
__device__ __forceinline__ double block_sum(double v, double *red /* THREADS/64 doubles of LDS */) {
    v = wave_sum(v);
    int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) red[wave] = v;
    __syncthreads();
    double t = 0;
    if (threadIdx.x == 0)
        for (int i = 0; i < THREADS / 64; i++) t += red[i];
    return t;  // valid in thread 0
}

// 1/x for normal positive x: v_rcp_f64 (about half the mantissa) refined by two Newton steps -- five dependent
// instructions instead of the dozen of the IEEE division sequence (scaling, fix-up).  The per-set solver is a chain of
// dependent steps; theta, den and the row sums it divides by are far from the exponent range where the fix-ups matter.
__device__ __forceinline__ double fast_rcp(double x) {
    double r = __builtin_amdgcn_rcp(x);
    double e = __builtin_fma(-x, r, 1.0);
    r = __builtin_fma(r, e, r);
    e = __builtin_fma(-x, r, 1.0);
    return __builtin_fma(r, e, r);
}

enum PassMode { MODE_EM = 0, MODE_EM_LL = 1, MODE_SCATTER = 2 };

}  // namespace
