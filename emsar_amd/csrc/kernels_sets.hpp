// kernels_sets.hpp -- k_solve_sets: the set-resident solver (one workgroup per connected set, everything in LDS)
#pragma once
// included by emsar_hip.hip only (one translation unit: the kernels live in its anonymous namespace)

namespace {

// ------------------------------------------------------------------------------------------------
// k_solve_sets: the SET-RESIDENT solver.  One workgroup owns one connected set (sets.hpp) and runs the whole
// SQUAREM-accelerated EM on it out of LDS: theta vectors, den, the folded single-row counts, the row weights and
// the 16-bit CSR/CSC indices are loaded once, then every pass is two LDS sweeps (E-step over rows, M-step over
// transcripts through the CSC -- no atomics, so the result is reproducible bit for bit) and a workgroup
// reduction.  No global synchronisation, no kernel launch per pass: a pass costs ~1 us instead of the ~30 us
// launch-latency floor of the streaming kernels, and every set stops at its own convergence.
// Same update, same start (theta = 1 where den > 0), same stopping rule and the same likelihood-safeguarded S3
// step as the streaming solve below, so both reach the same fixed point.
// ------------------------------------------------------------------------------------------------
struct SetStat { int32_t passes, converged; double delta; };
struct SetSolveParams { double tol, abs_floor, count_floor, zero_cut, abs_step; int32_t max_iter, accel, newton_after /* < 0: off */; };

// Wave-wide reductions on the DPP path (row shifts inside rows of 16 lanes, then row broadcasts; the total lands in lane
// 63 and is read back as a scalar): ~6 cross-lane moves per value instead of the twelve ds_bpermute round trips of a
// shuffle butterfly.  The per-set solver is a chain of dependent steps, so the latency of its reductions is pass time.
template <int CTRL>
__device__ __forceinline__ double dpp_move(double v, double identity) {
    const unsigned long long b = (unsigned long long)__double_as_longlong(v), id = (unsigned long long)__double_as_longlong(identity);
    const int lo = __builtin_amdgcn_update_dpp((int)(unsigned)id, (int)(unsigned)b, CTRL, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp((int)(unsigned)(id >> 32), (int)(unsigned)(b >> 32), CTRL, 0xf, 0xf, false);
    return __longlong_as_double((long long)(((unsigned long long)(unsigned)hi << 32) | (unsigned long long)(unsigned)lo));
}
__device__ __forceinline__ double wave_bcast63(double v) {
    const unsigned long long b = (unsigned long long)__double_as_longlong(v);
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)b, 63), hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(b >> 32), 63);
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}
__device__ __forceinline__ double wave_sum_dpp(double v) {
    v += dpp_move<0x111>(v, 0.0);      // row_shr:1
    v += dpp_move<0x112>(v, 0.0);      // row_shr:2
    v += dpp_move<0x114>(v, 0.0);      // row_shr:4
    v += dpp_move<0x118>(v, 0.0);      // row_shr:8   -> lane 15 of every row holds its row's sum
    v += dpp_move<0x142>(v, 0.0);      // row_bcast:15 -> lane 31 / 63 hold the sums of rows 0-1 / 2-3 (plus their own rows)
    v += dpp_move<0x143>(v, 0.0);      // row_bcast:31 -> lane 63 holds the wave's sum
    return wave_bcast63(v);
}
__device__ __forceinline__ double wave_max_dpp(double v) {      // v >= 0
    v = fmax(v, dpp_move<0x111>(v, 0.0));
    v = fmax(v, dpp_move<0x112>(v, 0.0));
    v = fmax(v, dpp_move<0x114>(v, 0.0));
    v = fmax(v, dpp_move<0x118>(v, 0.0));
    v = fmax(v, dpp_move<0x142>(v, 0.0));
    v = fmax(v, dpp_move<0x143>(v, 0.0));
    return wave_bcast63(v);
}

// Workgroup barrier of the set solver.  A one-wave workgroup needs none: the DS operations of a wave execute in order, so
// only the compiler has to be kept from moving LDS accesses across the point (same hand-over as between the E- and the
// M-step of the tiled pass kernel); s_barrier would cost its issue + wait on every one of the ~10 phases of a cycle.
template <int THREADS>
__device__ __forceinline__ void set_sync() {
    if (THREADS > 64) __syncthreads();
    else {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
}

template <int THREADS, int N>
__device__ __forceinline__ void set_reduce_sum(double (&v)[N], double *red) {
#pragma unroll
    for (int i = 0; i < N; i++) v[i] = wave_sum_dpp(v[i]);
    if (THREADS > 64) {
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        __syncthreads();                       // red may still be read from the previous reduction
        if (lane == 0)
#pragma unroll
            for (int i = 0; i < N; i++) red[wave * N + i] = v[i];
        __syncthreads();
#pragma unroll
        for (int i = 0; i < N; i++) {
            double t = 0;
            for (int w = 0; w < THREADS / 64; w++) t += red[w * N + i];
            v[i] = t;
        }
    }
}
template <int THREADS>
__device__ __forceinline__ double set_reduce_max(double v, double *red) {
    v = wave_max_dpp(v);
    if (THREADS > 64) {
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        __syncthreads();
        if (lane == 0) red[wave] = v;
        __syncthreads();
        double t = 0;
        for (int w = 0; w < THREADS / 64; w++) t = fmax(t, red[w]);
        v = t;
    }
    return v;
}

struct SetLds {
    double *den, *u, *w, *rw, *red;
    double *z, *hp, *mi, *hrow;      // Newton step: direction, Hessian-vector product / trial point, preconditioner + masks, R/S^2 per row
    const uint16_t *rp, *ent, *cp, *crow;
    int nt, nr;
};

// E-step over the rows at x, then M-step: y = (x*acc + u)/den.  Returns this thread's share of sum R log S (+ the
// folded single rows' u log x) when LL.
template <int THREADS, bool LL>
__device__ __forceinline__ double set_em_estep(const SetLds &L, const double *x) {
    double ll = 0.0;
    for (int j = threadIdx.x; j < L.nr; j += THREADS) {
        double S = 0.0;
        const int b = L.rp[j], e = L.rp[j + 1];
        for (int k = b; k < e; k += 4) {  // four independent index -> value chains in flight, also for the last 1..3 entries
            const int l = e - 1;
            const int i0 = L.ent[k], i1 = L.ent[k + 1 < e ? k + 1 : l], i2 = L.ent[k + 2 < e ? k + 2 : l], i3 = L.ent[k + 3 < e ? k + 3 : l];
            const double v0 = x[i0], v1 = x[i1], v2 = x[i2], v3 = x[i3];
            S += (v0 + (k + 1 < e ? v1 : 0.0)) + ((k + 2 < e ? v2 : 0.0) + (k + 3 < e ? v3 : 0.0));
        }
        const double r = L.rw[j];
        const bool live = S > 0.0;
        L.w[j] = live ? r * fast_rcp(S) : 0.0;
        if (LL && live) ll += r * log(S);
    }
    set_sync<THREADS>();
    return ll;
}
__device__ __forceinline__ double set_em_acc(const SetLds &L, int i) {
    double a = 0.0;
    const int b = L.cp[i], e = L.cp[i + 1];
    for (int k = b; k < e; k += 4) {
        const int l = e - 1;
        const int j0 = L.crow[k], j1 = L.crow[k + 1 < e ? k + 1 : l], j2 = L.crow[k + 2 < e ? k + 2 : l], j3 = L.crow[k + 3 < e ? k + 3 : l];
        const double v0 = L.w[j0], v1 = L.w[j1], v2 = L.w[j2], v3 = L.w[j3];
        a += (v0 + (k + 1 < e ? v1 : 0.0)) + ((k + 2 < e ? v2 : 0.0) + (k + 3 < e ? v3 : 0.0));
    }
    return a;
}
__device__ __forceinline__ double set_em_update(double x, double a, double u, double dn) {
    return dn > 0.0 ? (x > 0.0 ? (x * a + u) * fast_rcp(dn) : 0.0) : 0.0;
}

// ------------------------------------------------------------------------------------------------
// The safeguarded second-order step.  The set objective F(theta) = sum_c R_c log S_c + sum_t u_t log theta_t - sum_t theta_t den_t
// (Fp, emsar_functions.c:2946-2975, with E folded into den) is concave; the EM -- with or without SQUAREM -- creeps along
// nearly flat directions (10^4 .. 10^5 passes for a badly conditioned family of a few dozen isoforms) and decays like 1/k
// towards a boundary optimum theta_t = 0 with zero gradient.  Once a set has used newton_after passes without converging,
// every SQUAREM cycle is followed by one projected Newton step (Bertsekas 1982):
//   * components that are tiny (theta_t den_t < 1e-6 reads), carry no single-transcript reads and have a negative gradient are
//     BOUND: they go to exactly 0 (an EM fixed point; the stopping rule re-opens one whose gradient at 0 turns positive);
//   * for the FREE components the direction solves (-H) d = g by preconditioned conjugate gradients, matrix-free:
//     (-H) v = A^T diag(R/S^2) A v + diag(u/theta^2) v is one row sweep and one column sweep over the set's CSR / CSC -- the
//     cost of an EM pass, and counted as one; Jacobi preconditioner; at most n_free iterations (exact in exact arithmetic);
//   * the step is projected onto theta >= 0 and ACCEPTED ONLY IF F DOES NOT FALL (step 1, 1/4, 1/16, 1/64); otherwise the point is
//     left alone and the next eight cycles are plain SQUAREM.
// Convergence is still declared by the plain EM step's relative change (same rule, same tol as without the Newton steps), so
// the fixed point reached is the EM's.  tools/newton_proto.py is the numpy prototype of exactly this procedure.
// ------------------------------------------------------------------------------------------------
template <int THREADS>
__device__ __forceinline__ double set_row_dot(const SetLds &L, const double *v, int j) {
    double S = 0.0;
    const int b = L.rp[j], e = L.rp[j + 1];
    for (int k = b; k < e; k += 4) {
        const int l = e - 1;
        const int i0 = L.ent[k], i1 = L.ent[k + 1 < e ? k + 1 : l], i2 = L.ent[k + 2 < e ? k + 2 : l], i3 = L.ent[k + 3 < e ? k + 3 : l];
        const double v0 = v[i0], v1 = v[i1], v2 = v[i2], v3 = v[i3];
        S += (v0 + (k + 1 < e ? v1 : 0.0)) + ((k + 2 < e ? v2 : 0.0) + (k + 3 < e ? v3 : 0.0));
    }
    return S;
}
__device__ __forceinline__ double set_col_sum(const SetLds &L, const double *rowv, int i) {
    double a = 0.0;
    const int b = L.cp[i], e = L.cp[i + 1];
    for (int k = b; k < e; k++) a += rowv[L.crow[k]];
    return a;
}
// x: current point (left untouched unless the step is accepted), r / p: two transcript vectors of scratch.
// Returns true if x was replaced; adds its pass-equivalents to `passes`.
template <int THREADS>
__device__ __forceinline__ bool set_newton_step(const SetLds &L, double *x, double *r, double *p, int &passes) {
    const int nt = L.nt, nr = L.nr;
    constexpr double kBoundReads = 1e-6;
    // ---- E-step quantities at x: w = R/S, h = R/S^2, F(x) ----
    double s3[3] = {0.0, 0.0, 0.0};        // sum R log S (+ u log x), sum x den, r.q
    for (int j = threadIdx.x; j < nr; j += THREADS) {
        const double S = set_row_dot<THREADS>(L, x, j), rw = L.rw[j];
        const bool live = S > 0.0;
        const double inv = live ? fast_rcp(S) : 0.0;
        L.w[j] = rw * inv; L.hrow[j] = rw * inv * inv;
        if (live) s3[0] += rw * log(S);
    }
    set_sync<THREADS>();
    for (int i = threadIdx.x; i < nt; i += THREADS) {
        const double xi = x[i], dn = L.den[i], ui = L.u[i];
        const double acc = set_col_sum(L, L.w, i);
        const double g = acc - dn + (xi > 0.0 ? ui / xi : 0.0);
        if (ui > 0.0 && xi > 0.0) s3[0] += ui * log(xi);
        s3[1] += xi * dn;
        double mi = 0.0, ri = 0.0;                                   // mi: 0 = outside F (den = 0), -1 = bound, > 0 = 1 / diag(-H)
        if (dn > 0.0) {
            if (g < 0.0 && xi * dn < kBoundReads && ui == 0.0) mi = -1.0;
            else {
                const double diag = set_col_sum(L, L.hrow, i) + (xi > 0.0 ? ui / (xi * xi) : 0.0);
                mi = diag > 0.0 ? 1.0 / diag : 1.0;
                ri = g;
            }
        }
        L.mi[i] = mi; r[i] = ri; L.z[i] = 0.0;
        const double q = mi > 0.0 ? mi * ri : 0.0;
        p[i] = q;
        s3[2] += ri * q;
    }
    set_reduce_sum<THREADS, 3>(s3, L.red);
    const double Fx = s3[0] - s3[1];
    double rq = s3[2];
    const double rq_stop = 1e-8 * rq;                               // |r| down by 1e-4 in the preconditioned norm
    passes++;
    set_sync<THREADS>();
    // ---- preconditioned CG on the free components ----
    for (int it = 0; it < nt && rq > rq_stop && rq > 0.0; it++) {
        for (int j = threadIdx.x; j < nr; j += THREADS) L.w[j] = L.hrow[j] * set_row_dot<THREADS>(L, p, j);
        set_sync<THREADS>();
        double s1[1] = {0.0};
        for (int i = threadIdx.x; i < nt; i += THREADS) {
            double hp = 0.0;
            if (L.mi[i] > 0.0) {
                const double xi = x[i];
                hp = set_col_sum(L, L.w, i) + (xi > 0.0 ? L.u[i] / (xi * xi) : 0.0) * p[i];
            }
            L.hp[i] = hp;
            s1[0] += p[i] * hp;
        }
        set_reduce_sum<THREADS, 1>(s1, L.red);
        passes++;
        if (!(s1[0] > 0.0)) break;
        const double al = rq / s1[0];
        double s2[1] = {0.0};
        for (int i = threadIdx.x; i < nt; i += THREADS) {
            L.z[i] += al * p[i];
            const double ri = r[i] - al * L.hp[i];
            r[i] = ri;
            const double mi = L.mi[i];
            if (mi > 0.0) s2[0] += ri * (mi * ri);
        }
        set_reduce_sum<THREADS, 1>(s2, L.red);
        const double beta = s2[0] / rq;
        rq = s2[0];
        for (int i = threadIdx.x; i < nt; i += THREADS) { const double mi = L.mi[i]; p[i] = (mi > 0.0 ? mi * r[i] : 0.0) + beta * p[i]; }
        set_sync<THREADS>();
    }
    // ---- projected step, accepted only if F does not fall ----
    double alpha = 1.0;
    for (int tr = 0; tr < 4; tr++, alpha *= 0.25) {
        for (int i = threadIdx.x; i < nt; i += THREADS) {
            const double mi = L.mi[i], xi = x[i];
            double xn = 0.0;
            if (mi > 0.0) xn = fmax(xi + alpha * L.z[i], 0.0);
            else if (mi < 0.0) xn = alpha == 1.0 ? 0.0 : xi * (1.0 - alpha);
            L.hp[i] = xn;
        }
        set_sync<THREADS>();
        double s4[3] = {0.0, 0.0, 0.0};    // sum R log S + u log x, sum x den, infeasible
        for (int j = threadIdx.x; j < nr; j += THREADS) {
            const double S = set_row_dot<THREADS>(L, L.hp, j), rw = L.rw[j];
            if (S > 0.0) s4[0] += rw * log(S); else if (rw > 0.0) s4[2] += 1.0;
        }
        for (int i = threadIdx.x; i < nt; i += THREADS) {
            const double xn = L.hp[i], ui = L.u[i];
            if (ui > 0.0) { if (xn > 0.0) s4[0] += ui * log(xn); else s4[2] += 1.0; }
            s4[1] += xn * L.den[i];
        }
        set_reduce_sum<THREADS, 3>(s4, L.red);
        passes++;
        const double Fn = s4[0] - s4[1];
        if (s4[2] == 0.0 && Fn >= Fx - 1e-13 * fabs(Fx)) {
            for (int i = threadIdx.x; i < nt; i += THREADS) x[i] = L.hp[i];
            set_sync<THREADS>();
            return true;
        }
        set_sync<THREADS>();
    }
    return false;
}

template <int THREADS>
__global__ __launch_bounds__(THREADS) void k_solve_sets(const emsar::SetDesc *__restrict__ desc, const int32_t *__restrict__ g_tid,
                                                        const double *__restrict__ g_u, const double *__restrict__ row_w,
                                                        const uint16_t *__restrict__ rp_g, const uint16_t *__restrict__ ent_g,
                                                        const uint16_t *__restrict__ cp_g, const uint16_t *__restrict__ crow_g,
                                                        const double *__restrict__ den_g, double *__restrict__ theta_g,
                                                        SetStat *__restrict__ stat, SetSolveParams P) {
    extern __shared__ double smem[];
    const emsar::SetDesc d = desc[blockIdx.x];
    const int nt = (int)d.n_t, nr = (int)d.n_r, nnz = (int)d.nnz;
    double *A = smem, *B = A + nt, *Cc = B + nt;
    SetLds L;
    L.den = Cc + nt; L.u = L.den + nt; L.z = L.u + nt; L.hp = L.z + nt; L.mi = L.hp + nt;
    L.w = L.mi + nt; L.rw = L.w + nr; L.hrow = L.rw + nr; L.red = L.hrow + nr;
    uint16_t *rp = (uint16_t *)(L.red + emsar::kSetRedDoubles), *ent = rp + (nr + 1), *cp = ent + nnz, *crow = cp + (nt + 1);
    L.rp = rp; L.ent = ent; L.cp = cp; L.crow = crow; L.nt = nt; L.nr = nr;
    for (int i = threadIdx.x; i < nt; i += THREADS) {
        const double dn = den_g[g_tid[d.tid_off + i]];
        L.den[i] = dn; L.u[i] = g_u[d.tid_off + i];
        A[i] = dn > 0.0 ? 1.0 : 0.0;
    }
    for (int j = threadIdx.x; j < nr; j += THREADS) L.rw[j] = row_w[d.row_off + j];
    for (int j = threadIdx.x; j <= nr; j += THREADS) rp[j] = rp_g[d.rp_off + j];
    for (int i = threadIdx.x; i <= nt; i += THREADS) cp[i] = cp_g[d.cp_off + i];
    for (int k = threadIdx.x; k < nnz; k += THREADS) { ent[k] = ent_g[d.ent_off + k]; crow[k] = crow_g[d.ent_off + k]; }
    set_sync<THREADS>();

    double stepmax = 1.0, delta = __builtin_huge_val();
    int passes = 0, converged = 0, cooldown = 0;
    double *res = A;
    for (;;) {
        // pass 1 (plain): B = EM(A); the stopping rule is measured on this step only
        (void)set_em_estep<THREADS, false>(L, A);
        double dloc = 0.0;
        for (int i = threadIdx.x; i < nt; i += THREADS) {
            const double x = A[i], dn = L.den[i];
            const double acc = set_em_acc(L, i);
            double y = set_em_update(x, acc, L.u[i], dn);
            // a component the Newton step has put on the boundary stays there under the EM (0 is a fixed point); if the gradient
            // at 0 has turned positive since, it is re-opened just above 0 and the set is not done
            const bool reopen = x == 0.0 && dn > 0.0 && acc > dn * (1.0 + 1e-9);
            if (reopen) y = 1e-9 * fast_rcp(dn);
            B[i] = y;
            double fl = P.abs_floor;
            if (P.count_floor > 0.0 && dn > 0.0) fl = fmax(fl, P.count_floor / dn);
            double dd = fabs(y - x) * fast_rcp(fabs(y) + fl);
            if (!(dd == dd)) dd = __builtin_huge_val();
            if (y < P.zero_cut && y <= x) dd = 0.0;
            if (fabs(y - x) * (double)(passes + 1 > 1000 ? passes + 1 : 1000) < P.abs_step * 2e5) dd = 0.0;   // projected drift, see emsar_em_params.abs_step
            if (reopen) dd = 1.0;
            dloc = fmax(dloc, dd);
        }
        delta = set_reduce_max<THREADS>(dloc, L.red);
        set_sync<THREADS>();
        passes++;
        res = B;
        if (delta < P.tol) { converged = 1; break; }
        if (passes >= P.max_iter || delta == __builtin_huge_val()) break;
        if (!P.accel) { double *t = A; A = B; B = t; continue; }
        // pass 2: C = EM(B) with F(B); r = B-A, v = (C-B)-r
        double s4[4];
        s4[0] = set_em_estep<THREADS, true>(L, B);
        s4[1] = s4[2] = s4[3] = 0.0;
        for (int i = threadIdx.x; i < nt; i += THREADS) {
            const double x = B[i], dn = L.den[i], u = L.u[i];
            const double y = set_em_update(x, set_em_acc(L, i), u, dn);
            Cc[i] = y;
            if (u > 0.0 && x > 0.0) s4[0] += u * log(x);
            s4[1] += x * dn;
            const double r = x - A[i], v = (y - x) - r;
            s4[2] += r * r; s4[3] += v * v;
        }
        set_reduce_sum<THREADS, 4>(s4, L.red);
        const double F1 = s4[0] - s4[1];
        double s = s4[3] > 0.0 ? sqrt(s4[2] / s4[3]) : 1.0;
        s = fmin(fmax(s, 1.0), stepmax);
        const bool extrap = s > 1.01;
        // extrapolated point, in place of B
        double s2[2] = {0.0, 0.0};
        set_sync<THREADS>();
        for (int i = threadIdx.x; i < nt; i += THREADS) {
            const double x2 = Cc[i];
            double x = x2;
            if (extrap) {
                const double r = B[i] - A[i], v = (x2 - B[i]) - r;
                const double y = A[i] + 2.0 * s * r + s * s * v;
                x = (y > 0.0 && x2 > 0.0) ? y : x2;
            }
            B[i] = x;
            s2[1] += x * L.den[i];
        }
        set_sync<THREADS>();
        // pass 3: A = EM(B) with F(B)
        s2[0] = set_em_estep<THREADS, true>(L, B);
        for (int i = threadIdx.x; i < nt; i += THREADS) {
            const double x = B[i], u = L.u[i];
            A[i] = set_em_update(x, set_em_acc(L, i), u, L.den[i]);
            if (u > 0.0 && x > 0.0) s2[0] += u * log(x);
        }
        set_reduce_sum<THREADS, 2>(s2, L.red);
        const bool ok = !extrap || (s2[0] - s2[1] >= F1);
        set_sync<THREADS>();
        if (!ok) {
            for (int i = threadIdx.x; i < nt; i += THREADS) A[i] = Cc[i];
            if (s >= stepmax) stepmax = fmax(1.0, stepmax / 4.0);
        }
        if ((ok ? s : 1.0) >= stepmax) stepmax *= 4.0;
        set_sync<THREADS>();
        passes += 2;
        res = A;
        if (passes >= P.max_iter) break;
        if (P.newton_after >= 0 && passes >= P.newton_after) {
            if (cooldown > 0) cooldown--;
            else if (!set_newton_step<THREADS>(L, A, B, Cc, passes)) cooldown = 8;
        }
    }
    for (int i = threadIdx.x; i < nt; i += THREADS) theta_g[g_tid[d.tid_off + i]] = res[i];
    if (threadIdx.x == 0) { stat[blockIdx.x].passes = passes; stat[blockIdx.x].converged = converged; stat[blockIdx.x].delta = delta; }
}

}  // namespace
