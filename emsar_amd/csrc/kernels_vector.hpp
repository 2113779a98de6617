// kernels_vector.hpp -- T-sized vector kernels: the M-step update, the SQUAREM cycle, normalisation; compute_adjEUMA
#pragma once
// included by emsar_hip.hip only (one translation unit: the kernels live in its anonymous namespace)

namespace {

// ------------------------------------------------------------------------------------------------
// T-sized vector kernels
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ double em_new_theta(double x, double a, double dn, const double *u, int t, double fx);
__global__ void k_fill_start(int n, const double *__restrict__ den, double *__restrict__ theta) {
    int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n) theta[t] = den[t] > 0.0 ? 1.0 : 0.0;  // uniform interior start; tids outside F are defined 0
}

// theta_out = theta_in * acc / den ; acc <- 0 ; scal.delta = max |dtheta| / (theta_out + floor)
// grid-stride, one atomicMax per workgroup (hundreds of same-address atomics cost ~12 ns each)
__global__ __launch_bounds__(256) void k_update(int n, const double *__restrict__ th_in, double *__restrict__ acc,
                                                const double *__restrict__ den, const double *__restrict__ u /* folded single-tid rows, may be null */,
                                                double *__restrict__ th_out, double abs_floor, double count_floor, double zero_cut, Scal *scal,
                                                const uint8_t *__restrict__ kind /* non-null: only KIND_STREAMED transcripts enter the stopping rule */,
                                                int to_delta1 /* the first (plain) step of a SQUAREM cycle: the cycle's stopping rule */, double fx) {
    __shared__ double red[4];
    double d = 0.0;
    const double abs_step = scal->abs_step_cur;      // written by k_cycle_begin, nobody writes it during a pass
    for (int t = blockIdx.x * 256 + threadIdx.x; t < n; t += gridDim.x * 256) {
        double a = acc[t], dn = den[t], x = th_in[t];
        // a row {t} contributes R/theta_t to acc_t, i.e. R to theta_t*acc_t: added analytically (TILED layout)
        double y = em_new_theta(x, a, dn, u, t, fx);
        th_out[t] = y;
        acc[t] = 0.0;
        double fl = abs_floor;
        if (count_floor > 0.0 && dn > 0.0) fl = fmax(fl, count_floor / dn);    // floor expressed in inferred reads
        double dd = fabs(y - x) / (fabs(y) + fl);
        if (!(dd == dd)) dd = __builtin_huge_val();  // NaN -> +inf so that the host sees it
        // ... also when the next check is many passes away: an overflowed theta is NaN one pass later and 0 the pass after
        // (NaN > 0 is false in the update), and an all-zero answer looks converged
        if (dd == __builtin_huge_val() || !(y - y == 0.0)) scal->bad = 1u;
        if (y < zero_cut && y <= x) dd = 0.0;        // below the print quantum and still falling: prints as 0.000000 either way
        if (fabs(y - x) < abs_step) dd = 0.0;         // moves by less than abs_step per pass (emsar_em_params.abs_step)
        if (kind && kind[t] != emsar::KIND_STREAMED) dd = 0.0;
        d = fmax(d, dd);
    }
    for (int o = 32; o > 0; o >>= 1) d = fmax(d, __shfl_xor(d, o, 64));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = d;
    __syncthreads();
    if (threadIdx.x == 0) {
        d = fmax(fmax(red[0], red[1]), fmax(red[2], red[3]));
        // the word only grows during a kernel: workgroups whose maximum is already covered skip the same-address atomic
        unsigned long long *dst = to_delta1 ? &scal->delta1_bits : &scal->delta_bits;
        const unsigned long long bits = (unsigned long long)__double_as_longlong(d);
        if (d > 0.0 && bits > __hip_atomic_load(dst, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(dst, bits);
    }
}

// abs_step_base > 0: the projected-drift bound of emsar_em_params.abs_step, |dtheta| < base * 2e5 / K at pass K >= 1000.  K is
// counted here, on the device, so that a cycle recorded once in a hipGraph carries the right bound at every replay.
__global__ __launch_bounds__(kLlSlots) void k_cycle_begin(Scal *s, double abs_step_base, int passes_in_cycle) {   // one workgroup of kLlSlots threads
    for (int i = 0; i < 4; i++) s->ll[i].s[threadIdx.x] = 0.0;
    if (threadIdx.x != 0) return;
    s->sr2 = 0.0; s->sv2 = 0.0; s->pen1 = 0.0; s->penx = 0.0;
    s->delta_bits = 0ull; s->delta1_bits = 0ull;
    const long long done = s->passes;
    s->abs_step_cur = abs_step_base > 0.0 ? abs_step_base * 2e5 / (double)(done + 1 > 1000 ? done + 1 : 1000) : 0.0;
    s->passes = done + passes_in_cycle;
}
__global__ void k_scal_init(Scal *s) {
    s->stepmax = 1.0; s->s_used = 1.0; s->accepted = 0; s->rejected = 0; s->sum_a = s->sum_b = 0.0;
    s->passes = 0; s->abs_step_cur = 0.0; s->bad = 0u;
}

// ---- the SQUAREM cycle of the streaming solve with the O(T) work folded into the three update kernels ----
// (8 launches per cycle instead of 13: on a problem of a few thousand rows the cycle is pure launch latency)
//   k_update_p1   th1 = EM(th0); stopping rule of the cycle -> delta1_bits
//   k_update_p2   th2 = EM(th1); F(th1) terms: sum u log th1 -> ll[1], sum th1*den -> pen1; |r|^2, |v|^2
//   k_sq_extrap_ll thx = th0 + 2 s r + s^2 v  (Varadhan & Roland 2008, S3: s = |r|/|v| clamped to [1, stepmax]); components that
//                 would leave the interior keep the plain EM value th2; s <= 1.01 -> thx = th2; + sum u log thx -> ll[2]
//   k_update_p3   th0 = accepted ? EM(thx) : th2, accepted iff F(thx) >= F(th1), F = ll - sum theta*den; step bounds x4 / :4
// a = the accumulator word of t: sum_c m_ct R_c / S_c, or in deterministic mode (fx != 0) the MASS theta_t * that sum in fixed point
__device__ __forceinline__ double em_new_theta(double x, double a, double dn, const double *u, int t, double fx) {
    const double mass = fx != 0.0 ? (double)__double_as_longlong(a) / fx : x * a;
    return dn > 0.0 ? (u ? (x > 0.0 ? (mass + u[t]) / dn : 0.0) : mass / dn) : 0.0;
}
// Reductions of the SQUAREM cycle that every workgroup of the NEXT kernel needs (|r|^2, |v|^2, sum theta*den): each workgroup of the
// producer writes its partial sum to its own word, each workgroup of the consumer adds the words up in the same fixed order -- no
// same-address atomics (hundreds of them cost ~12 ns each), and the same bits in every run.
constexpr int kSqPart = 1024;        // most workgroups of a SQUAREM vector kernel (ctx->sq_grid is clamped to it)
__device__ __forceinline__ double sq_sum(const double *__restrict__ part, int n, double *red) {
    double s = 0.0;
    for (int i = threadIdx.x; i < n; i += 256) s += part[i];
    const double tot = block_sum<256>(s, red);      // valid in thread 0: handed to everybody through LDS
    if (threadIdx.x == 0) red[0] = tot;
    __syncthreads();
    const double all = red[0];
    __syncthreads();
    return all;
}
// the value of a likelihood sum (kLlSlots words, kernels_common.hpp) for every thread of a 256-thread workgroup: the first wave
// loads one word per lane and adds them up in a fixed butterfly, the result goes round through LDS
__device__ __forceinline__ double ll_value_wg(const double *ll_word0, double fx_ll, double *red) {
    static_assert(kLlSlots == 64, "one word per lane of a wave");
    if (threadIdx.x < 64) {
        const double w = ll_word0[threadIdx.x * (sizeof(LineF64) / sizeof(double))];
        double v;
        if (fx_ll != 0.0) {
            long long x = __double_as_longlong(w);
            for (int o = 32; o > 0; o >>= 1) x += __shfl_xor(x, o, 64);
            v = (double)x / fx_ll;
        } else {
            v = w;
            for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
        }
        if (threadIdx.x == 0) red[0] = v;
    }
    __syncthreads();
    const double all = red[0];
    __syncthreads();
    return all;
}
__global__ __launch_bounds__(256) void k_update_p2(int n, const double *__restrict__ th0, const double *__restrict__ th1, double *__restrict__ acc,
                                                   const double *__restrict__ den, const double *__restrict__ u, double *__restrict__ th2, Scal *scal,
                                                   double *__restrict__ part, Fx fx) {
    __shared__ double red[4];
    double r2 = 0, v2 = 0, p1 = 0, l1 = 0;
    for (int t = blockIdx.x * 256 + threadIdx.x; t < n; t += gridDim.x * 256) {
        const double x = th1[t], dn = den[t];
        const double y = em_new_theta(x, acc[t], dn, u, t, fx.mass);
        th2[t] = y;
        acc[t] = 0.0;
        const double r = x - th0[t], v = (y - x) - r;
        r2 += r * r; v2 += v * v; p1 += x * dn;
        if (u) { const double c = u[t]; if (c > 0.0 && x > 0.0) l1 += c * log(x); }
    }
    double a = block_sum<256>(r2, red); __syncthreads();
    double b = block_sum<256>(v2, red); __syncthreads();
    double c = block_sum<256>(p1, red); __syncthreads();
    double d = block_sum<256>(l1, red);
    if (threadIdx.x == 0) {       // |r|^2, |v|^2, sum theta*den: one word per workgroup, summed in a fixed order by the kernels that use them (sq_sum)
        part[blockIdx.x] = a; part[kSqPart + blockIdx.x] = b; part[2 * kSqPart + blockIdx.x] = c;
        if (d != 0.0) ll_add(&scal->ll[1].s[0].v, d, fx.ll);
    }
}
__global__ __launch_bounds__(256) void k_sq_extrap_ll(int n, const double *__restrict__ th0, const double *__restrict__ th1,
                                                      const double *__restrict__ th2, const double *__restrict__ den, const double *__restrict__ u,
                                                      double *__restrict__ thx, Scal *scal, double *__restrict__ part, int n_part, Fx fx) {
    __shared__ double red[4];
    const double sr2 = sq_sum(part, n_part, red), sv2 = sq_sum(part + kSqPart, n_part, red);
    double s = sv2 > 0.0 ? sqrt(sr2 / sv2) : 1.0;
    s = fmin(fmax(s, 1.0), scal->stepmax);
    const bool extrap = s > 1.01;
    double px = 0, lx = 0;
    for (int t = blockIdx.x * 256 + threadIdx.x; t < n; t += gridDim.x * 256) {
        double x2 = th2[t], x = x2;
        if (extrap) {
            double r = th1[t] - th0[t], v = (x2 - th1[t]) - r;
            double y = th0[t] + 2.0 * s * r + s * s * v;
            x = (y > 0.0 && x2 > 0.0) ? y : x2;
        }
        thx[t] = x;
        px += x * den[t];
        if (u) { const double c = u[t]; if (c > 0.0 && x > 0.0) lx += c * log(x); }
    }
    double p = block_sum<256>(px, red); __syncthreads();
    double l = block_sum<256>(lx, red);
    if (threadIdx.x == 0) {
        part[3 * kSqPart + blockIdx.x] = p;
        if (l != 0.0) ll_add(&scal->ll[2].s[0].v, l, fx.ll);
        if (blockIdx.x == 0) { scal->s_used = extrap ? s : 1.0; scal->sr2 = sr2; scal->sv2 = sv2; }
    }
}
__global__ __launch_bounds__(256) void k_update_p3(int n, const double *__restrict__ thx, const double *__restrict__ th2, double *__restrict__ acc,
                                                   const double *__restrict__ den, const double *__restrict__ u, double *__restrict__ th0, Scal *scal,
                                                   const double *__restrict__ part, int n_part, Fx fx) {
    __shared__ double red[4];
    const double s = scal->s_used;
    const bool extrap = s > 1.0;
    const double pen1 = sq_sum(part + 2 * kSqPart, n_part, red), penx = sq_sum(part + 3 * kSqPart, n_part, red);
    const double ll2 = ll_value_wg(&scal->ll[2].s[0].v, fx.ll, red), ll1 = ll_value_wg(&scal->ll[1].s[0].v, fx.ll, red);
    const bool ok = !extrap || (ll2 - penx >= ll1 - pen1);
    for (int t = blockIdx.x * 256 + threadIdx.x; t < n; t += gridDim.x * 256) {
        const double y = em_new_theta(thx[t], acc[t], den[t], u, t, fx.mass);
        acc[t] = 0.0;
        th0[t] = ok ? y : th2[t];
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {      // nobody reads these three during this kernel
        double sm = scal->stepmax;
        if (!ok) { scal->rejected++; if (s >= sm) sm = fmax(1.0, sm / 4.0); }
        else { scal->accepted++; }
        if ((ok ? s : 1.0) >= sm) sm *= 4.0;
        scal->stepmax = sm;
        scal->pen1 = pen1; scal->penx = penx;
    }
}


// EUMA [rows][nfl] -> [nfl][rows] through a 64 x 64 LDS tile (once per rsh)
__global__ __launch_bounds__(256) void k_transpose_i32(int64_t n_rows, int nfl, const int32_t *__restrict__ in, int32_t *__restrict__ out) {
    __shared__ int32_t tile[64][65];
    const int64_t r0 = (int64_t)blockIdx.x * 64;
    const int c0 = (int)blockIdx.y * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int j = ty; j < 64; j += 4) {
        const int64_t r = r0 + j; const int c = c0 + tx;
        tile[j][tx] = (r < n_rows && c < nfl) ? in[(size_t)r * (size_t)nfl + (size_t)c] : 0;
    }
    __syncthreads();
    for (int j = ty; j < 64; j += 4) {
        const int c = c0 + j; const int64_t r = r0 + tx;
        if (r < n_rows && c < nfl) out[(size_t)c * (size_t)n_rows + (size_t)r] = tile[tx][j];
    }
}
// compute_adjEUMA (emsar_functions.c:2517-2523): one lane per row, fragment lengths in ascending order, product and sum
// rounded separately (no FMA) -- bit-identical to the reference's scalar loop; every load is a coalesced 256 B per wave
__global__ __launch_bounds__(256) void k_adj_euma(int64_t n_rows, int nfl, const int32_t *__restrict__ euma_t, const double *__restrict__ wf,
                                                  double *__restrict__ out) {
#pragma clang fp contract(off)   // hipcc fuses a + x*y into an FMA by default (one rounding instead of the reference's two);
                                 // plain operators: the __dmul_rn / __dadd_rn wrappers carry their own contraction flag
    const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (r >= n_rows) return;
    double a = 0.0;
    int i = 0;
    for (; i + 8 <= nfl; i += 8) {
        int32_t e[8];
#pragma unroll
        for (int j = 0; j < 8; j++) e[j] = __builtin_nontemporal_load(&euma_t[(size_t)(i + j) * (size_t)n_rows + (size_t)r]);   // read once per sample
#pragma unroll
        for (int j = 0; j < 8; j++) { const double p = wf[i + j] * (double)e[j]; a = a + p; }
    }
    for (; i < nfl; i++) { const double p = wf[i] * (double)euma_t[(size_t)i * (size_t)n_rows + (size_t)r]; a = a + p; }
    out[r] = a;
}

// transcripts outside every multi-transcript set: theta = (reads of its single-transcript rows) / den
__global__ void k_closed_form(int n, const uint8_t *__restrict__ kind, const double *__restrict__ usum,
                              const double *__restrict__ den, double *__restrict__ theta) {
    int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n && kind[t] == emsar::KIND_CLOSED) theta[t] = den[t] > 0.0 ? usum[t] / den[t] : 0.0;
}

// sum of the mean FPKM (the TPM denominator, emsar_functions.c:3176-3181): ONE workgroup, fixed order -- the printed TPM
// column must not depend on the arrival order of atomics (the per-set solver is bit-reproducible, its output should be too)
__global__ __launch_bounds__(1024) void k_sum(int n, const double *__restrict__ x, double *out) {
    __shared__ double red[16];
    double s = 0.0;
    for (int t = threadIdx.x; t < n; t += 1024) s += x[t];
    double tot = block_sum<1024>(s, red);
    if (threadIdx.x == 0) *out = *out + tot;
}
// sum x*y: ONE workgroup, fixed order (the penalty term of the printed log-likelihood)
__global__ __launch_bounds__(1024) void k_dot(int n, const double *__restrict__ x, const double *__restrict__ y, double *out) {
    __shared__ double red[16];
    double s = 0.0;
    for (int t = threadIdx.x; t < n; t += 1024) s += x[t] * y[t];
    const double tot = block_sum<1024>(s, red);
    if (threadIdx.x == 0) *out = tot;
}
// print_FPKMfinal arithmetic (emsar_functions.c:3203-3207): TPM, iReadcount, Round_off
__global__ void k_normalise(int n, const double *__restrict__ mean, const double *__restrict__ ieuma, double nreads_m,
                            const double *total, double *__restrict__ tpm, double *__restrict__ ir, int32_t *__restrict__ iri) {
    int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    double m = mean[t];
    tpm[t] = m * 1E6 / *total;
    double x = (ieuma[t] / 1E3) * m * nreads_m;
    ir[t] = x;
    int xi = (int)x;
    iri[t] = (x - xi >= 0.5) ? xi + 1 : xi;
}

}  // namespace
