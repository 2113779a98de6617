// kernels_cluster.hpp -- k_solve_cluster: connected sets too large for ONE workgroup's LDS, solved by a CLUSTER of 2, 4 or 8
// workgroups that stay inside one launch for the whole solve (sets.hpp packs the records: ClusterDesc).
#pragma once
// included by emsar_hip.hip only (one translation unit: the kernels live in its anonymous namespace)
//
// Why: such a set (a gene family of a few thousand isoforms, 10^4 .. 10^5 distinct segments) used to go through the streaming
// passes -- a SQUAREM cycle there is 8 dependent launches, 20-25 us per pass whatever the size, and the slowest set needs 10^4
// passes.  Inside one launch a pass costs two cluster barriers and L2 round trips instead.
//
// Division of labour inside a cluster of g workgroups (512 threads each):
//   rows         dealt in g contiguous ranges of about nnz / g entries.  A workgroup streams the CSR of ITS rows from global memory
//                (read in order: coalesced, L2-resident) and gathers theta from its LDS copy of the WHOLE current point; the
//                weights w_r = R_r / S_r of its rows stay in its LDS.
//   column sums  every workgroup adds up, for EVERY transcript, the w_r of its own rows that hold it (the transpose of its rows,
//                also streamed in order) and writes that partial vector to global memory.            -- cluster barrier 1 --
//   transcripts  dealt in g equal ranges.  A workgroup adds the g partial sums of its transcripts IN FIXED ORDER (no atomics:
//                bit-reproducible like the one-workgroup solver), makes the EM update, the stopping rule and the SQUAREM terms for
//                them, and publishes its piece of the new point and its partial scalars.             -- cluster barrier 2 --
//   everybody reads the whole new point into LDS and adds the partial scalars in fixed order.
// The SQUAREM cycle (same S3 step, same likelihood safeguard and step bounds as k_solve_sets) takes 7 barriers for its 3 passes.
//
// The cluster barrier: one counter per set in global memory that only ever grows; a workgroup arrives with an atomic add by
// one lane and polls (s_sleep) until the counter reaches g x (barriers so far); see cl_barrier for the visibility rules.  Every workgroup of a
// launch is resident at the same time -- the host launches at most as many cluster workgroups as the device has CUs (one per
// CU: each asks for most of a CU's LDS) -- so nobody waits for a workgroup that cannot start.  A spin that lasts longer than
// half a minute raises the set's abort word instead of hanging the device; the solve then reports an error.

namespace {

struct ClusterStat { int32_t passes, converged; double delta; int32_t aborted, pad; };

constexpr int kClT = emsar::kClusterThreads;
constexpr unsigned kClSpinLimit = 1u << 24;   // about half a minute of polling

struct Cl {                       // one workgroup's view of its set
    int g, G, nt, t0, t1, r0, r1;             // own transcripts [t0, t1), own rows [r0, r1)
    double *X, *w, *Ao, *Bo, *Co, *den, *u, *tmp, *red;      // LDS: whole point; own rows' weights; own transcripts' vectors
    const uint32_t *rp; const uint16_t *ent; const double *rw;    // the set's CSR (global)
    const uint32_t *cp; const uint16_t *crow;                     // transpose of THIS workgroup's rows (global), row ids local to it
    double *P0, *P1, *partial, *pscal;                            // global scratch of the set
    unsigned *bar, *abort;
    unsigned gen;                                                 // barriers passed so far
};

// Everything the workgroups of a cluster hand to each other (partial column sums, published points, partial scalars) is written
// and read with device-scope accesses that bypass the non-coherent caches (`sc1` stores and loads: the per-XCD L2s are not
// coherent with each other and a CU's L1 is never refreshed by another CU's stores) -- so the barrier needs no cache
// write-back / invalidate (an agent-scope fence per wave on either side of it cost ~10 us per barrier): every wave waits for its
// own stores, the workgroup meets, one lane adds to the set's counter and polls it.
__device__ __forceinline__ void cl_st(double *p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ double cl_ld(const double *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// returns false if the cluster has been aborted (every workgroup sees the same answer after the same barrier)
__device__ __forceinline__ bool cl_barrier(Cl &C) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's stores have been performed
    __syncthreads();
    C.gen++;
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(C.bar, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned want = C.gen * (unsigned)C.G;
        unsigned spins = 0;
        while (__hip_atomic_load(C.bar, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) {
            if (__hip_atomic_load(C.abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) break;
            if (++spins > kClSpinLimit) { __hip_atomic_store(C.abort, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
            __builtin_amdgcn_s_sleep(1);
        }
    }
    __syncthreads();
    return __hip_atomic_load(C.abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u;
}

// workgroup-wide sums of N values, the result in every thread (fixed order)
template <int N>
__device__ __forceinline__ void cl_wg_sum(double (&v)[N], double *red) {
    set_reduce_sum<kClT, N>(v, red);
}

// E-step on the own rows at the point in C.X, then the partial column sums of those rows for every transcript -> global.
// Returns this thread's share of sum R log S when LL.
template <bool LL>
__device__ __forceinline__ double cl_estep_partial(const Cl &C) {
    double ll = 0.0;
    for (int j = C.r0 + (int)threadIdx.x; j < C.r1; j += kClT) {
        const uint32_t b = C.rp[j], e = C.rp[j + 1];
        double S = 0.0;
        for (uint32_t k = b; k < e; k += 4) {      // four independent index -> value chains in flight
            const uint32_t l = e - 1;
            const int i0 = C.ent[k], i1 = C.ent[k + 1 < e ? k + 1 : l], i2 = C.ent[k + 2 < e ? k + 2 : l], i3 = C.ent[k + 3 < e ? k + 3 : l];
            const double v0 = C.X[i0], v1 = C.X[i1], v2 = C.X[i2], v3 = C.X[i3];
            S += (v0 + (k + 1 < e ? v1 : 0.0)) + ((k + 2 < e ? v2 : 0.0) + (k + 3 < e ? v3 : 0.0));
        }
        const double r = C.rw[j];
        const bool live = S > 0.0;
        C.w[j - C.r0] = live ? r * fast_rcp(S) : 0.0;
        if (LL && live) ll += r * log(S);
    }
    __syncthreads();
    double *mine = C.partial + (size_t)C.g * (size_t)C.nt;
    for (int i = threadIdx.x; i < C.nt; i += kClT) {
        const uint32_t b = C.cp[i], e = C.cp[i + 1];
        double a = 0.0;
        for (uint32_t k = b; k < e; k += 4) {
            const uint32_t l = e - 1;
            const int j0 = C.crow[k], j1 = C.crow[k + 1 < e ? k + 1 : l], j2 = C.crow[k + 2 < e ? k + 2 : l], j3 = C.crow[k + 3 < e ? k + 3 : l];
            const double v0 = C.w[j0], v1 = C.w[j1], v2 = C.w[j2], v3 = C.w[j3];
            a += (v0 + (k + 1 < e ? v1 : 0.0)) + ((k + 2 < e ? v2 : 0.0) + (k + 3 < e ? v3 : 0.0));
        }
        cl_st(&mine[i], a);
    }
    return ll;
}
// column sum of own transcript i (local index io = i - t0): the g partial sums in fixed order
__device__ __forceinline__ double cl_acc(const Cl &C, int i) {
    double a = 0.0;
    double v[emsar::kClusterMaxWg];
    for (int h = 0; h < C.G; h++) v[h] = cl_ld(&C.partial[(size_t)h * (size_t)C.nt + (size_t)i]);     // all in flight, added in fixed order
    for (int h = 0; h < C.G; h++) a += v[h];
    return a;
}
// publish a vector over the own transcripts into a published point / read a whole published point into X
__device__ __forceinline__ void cl_publish(const Cl &C, const double *own, double *P) {
    for (int i = C.t0 + (int)threadIdx.x; i < C.t1; i += kClT) cl_st(&P[i], own[i - C.t0]);
}
__device__ __forceinline__ void cl_load_point(const Cl &C, const double *P) {
    for (int i = threadIdx.x; i < C.nt; i += kClT) C.X[i] = cl_ld(&P[i]);
    __syncthreads();
}
// N partial scalars of this workgroup -> global; after the next barrier cl_read_scal adds the g partials in fixed order
template <int N>
__device__ __forceinline__ void cl_write_scal(const Cl &C, const double (&v)[N], int slot0) {
    if (threadIdx.x == 0)
#pragma unroll
        for (int i = 0; i < N; i++) cl_st(&C.pscal[(size_t)C.g * 16 + (size_t)(slot0 + i)], v[i]);
}
__device__ __forceinline__ double cl_read_scal(const Cl &C, int slot) {
    double s = 0.0;
    for (int h = 0; h < C.G; h++) s += cl_ld(&C.pscal[(size_t)h * 16 + (size_t)slot]);
    return s;
}
__device__ __forceinline__ double cl_read_scal_max(const Cl &C, int slot) {
    double s = 0.0;
    for (int h = 0; h < C.G; h++) s = fmax(s, cl_ld(&C.pscal[(size_t)h * 16 + (size_t)slot]));
    return s;
}

__global__ __launch_bounds__(kClT) void k_solve_cluster(const emsar::ClusterDesc *__restrict__ desc, const uint32_t *__restrict__ blk_set, uint32_t blk_base,
                                                        const int32_t *__restrict__ g_tid, const double *__restrict__ g_u,
                                                        const double *__restrict__ row_w, const uint32_t *__restrict__ rp_g,
                                                        const uint16_t *__restrict__ ent_g, const uint32_t *__restrict__ cp_g,
                                                        const uint16_t *__restrict__ crow_g, const uint32_t *__restrict__ part_g,
                                                        double *__restrict__ scratch, unsigned *__restrict__ bars, unsigned *__restrict__ aborts,
                                                        const double *__restrict__ den_g, double *__restrict__ theta_g,
                                                        ClusterStat *__restrict__ stat, SetSolveParams P) {
    extern __shared__ double smem[];
    const uint32_t blk = blk_base + blockIdx.x;      // a launch holds whole sets: workgroups blk_base .. of the list
    const uint32_t set = blk_set[blk];
    const emsar::ClusterDesc d = desc[set];
    Cl C;
    C.G = (int)d.g; C.g = (int)(blk - d.blk0); C.nt = (int)d.n_t;
    const int own = (C.nt + C.G - 1) / C.G;
    C.t0 = C.g * own < C.nt ? C.g * own : C.nt;
    C.t1 = C.t0 + own < C.nt ? C.t0 + own : C.nt;
    C.r0 = (int)part_g[d.part_off + (uint32_t)C.g]; C.r1 = (int)part_g[d.part_off + (uint32_t)C.g + 1];
    int max_rows = 0;
    for (int h = 0; h < C.G; h++) { const int n = (int)(part_g[d.part_off + h + 1] - part_g[d.part_off + h]); max_rows = n > max_rows ? n : max_rows; }
    C.X = smem; C.w = C.X + C.nt; C.Ao = C.w + max_rows; C.Bo = C.Ao + own; C.Co = C.Bo + own; C.den = C.Co + own; C.u = C.den + own;
    C.tmp = C.u + own; C.red = C.tmp + own;
    C.rp = rp_g + d.rp_off; C.ent = ent_g + d.ent_off; C.rw = row_w + d.row_off;
    C.cp = cp_g + d.cp_off + (size_t)C.g * (size_t)(C.nt + 1); C.crow = crow_g + d.ent_off;
    C.P0 = scratch + d.scratch_off; C.P1 = C.P0 + C.nt; C.partial = C.P1 + C.nt; C.pscal = C.partial + (size_t)C.G * (size_t)C.nt;
    C.bar = bars + d.bar; C.abort = aborts + d.bar; C.gen = 0;
    const int nown = C.t1 - C.t0;
    // start: theta = 1 where den > 0 (every workgroup fills the whole point itself: no barrier needed)
    for (int i = threadIdx.x; i < C.nt; i += kClT) C.X[i] = den_g[g_tid[d.tid_off + i]] > 0.0 ? 1.0 : 0.0;
    for (int i = threadIdx.x; i < nown; i += kClT) {
        const double dn = den_g[g_tid[d.tid_off + C.t0 + i]];
        C.den[i] = dn; C.u[i] = g_u[d.tid_off + C.t0 + i]; C.Ao[i] = dn > 0.0 ? 1.0 : 0.0;
    }
    __syncthreads();

    double stepmax = 1.0, delta = __builtin_huge_val();
    int passes = 0, converged = 0;
    bool alive = true;
    const double *res = C.Ao;                       // the own piece of the point to return
    for (;;) {
        // ---- pass 1 (plain): B = EM(A); the stopping rule is measured on this step only ----
        (void)cl_estep_partial<false>(C);
        if (!(alive = cl_barrier(C))) break;
        double dloc = 0.0;
        for (int i = threadIdx.x; i < nown; i += kClT) {
            const double x = C.Ao[i], dn = C.den[i];
            const double y = set_em_update(x, cl_acc(C, C.t0 + i), C.u[i], dn);
            C.Bo[i] = y;
            double fl = P.abs_floor;
            if (P.count_floor > 0.0 && dn > 0.0) fl = fmax(fl, P.count_floor / dn);
            double dd = fabs(y - x) * fast_rcp(fabs(y) + fl);
            if (!(dd == dd)) dd = __builtin_huge_val();
            if (y < P.zero_cut && y <= x) dd = 0.0;
            if (fabs(y - x) * (double)(passes + 1 > 1000 ? passes + 1 : 1000) < P.abs_step * 2e5) dd = 0.0;
            dloc = fmax(dloc, dd);
        }
        { double m[1] = {set_reduce_max<kClT>(dloc, C.red)}; cl_write_scal<1>(C, m, 0); }
        cl_publish(C, C.Bo, C.P0);
        if (!(alive = cl_barrier(C))) break;
        delta = cl_read_scal_max(C, 0);
        passes++;
        res = C.Bo;
        if (delta < P.tol) { converged = 1; break; }
        if (passes >= P.max_iter || delta == __builtin_huge_val()) break;
        cl_load_point(C, C.P0);                      // X = B
        if (!P.accel) { for (int i = threadIdx.x; i < nown; i += kClT) C.Ao[i] = C.Bo[i]; __syncthreads(); res = C.Ao; continue; }
        // ---- pass 2: C = EM(B) with F(B); r = B - A, v = (C - B) - r ----
        double s4[4];
        s4[0] = cl_estep_partial<true>(C);
        s4[1] = s4[2] = s4[3] = 0.0;
        if (!(alive = cl_barrier(C))) break;
        for (int i = threadIdx.x; i < nown; i += kClT) {
            const double x = C.Bo[i], dn = C.den[i], u = C.u[i];
            const double y = set_em_update(x, cl_acc(C, C.t0 + i), u, dn);
            C.Co[i] = y;
            if (u > 0.0 && x > 0.0) s4[0] += u * log(x);
            s4[1] += x * dn;
            const double r = x - C.Ao[i], v = (y - x) - r;
            s4[2] += r * r; s4[3] += v * v;
        }
        cl_wg_sum<4>(s4, C.red);
        cl_write_scal<4>(C, s4, 1);                  // slots 1..4
        if (!(alive = cl_barrier(C))) break;
        const double F1 = cl_read_scal(C, 1) - cl_read_scal(C, 2);
        const double sr2 = cl_read_scal(C, 3), sv2 = cl_read_scal(C, 4);
        double s = sv2 > 0.0 ? sqrt(sr2 / sv2) : 1.0;
        s = fmin(fmax(s, 1.0), stepmax);
        const bool extrap = s > 1.01;
        // the extrapolated point over the own transcripts (kept in tmp), published as the next point to evaluate
        double s2[2] = {0.0, 0.0};
        for (int i = threadIdx.x; i < nown; i += kClT) {
            const double x2 = C.Co[i];
            double x = x2;
            if (extrap) {
                const double r = C.Bo[i] - C.Ao[i], v = (x2 - C.Bo[i]) - r;
                const double y = C.Ao[i] + 2.0 * s * r + s * s * v;
                x = (y > 0.0 && x2 > 0.0) ? y : x2;
            }
            C.tmp[i] = x;
            s2[1] += x * C.den[i];
        }
        __syncthreads();
        cl_publish(C, C.tmp, C.P1);
        if (!(alive = cl_barrier(C))) break;
        cl_load_point(C, C.P1);                      // X = extrapolated point
        // ---- pass 3: A = EM(X) with F(X); accepted iff F(X) >= F(B) ----
        s2[0] = cl_estep_partial<true>(C);
        if (!(alive = cl_barrier(C))) break;
        for (int i = threadIdx.x; i < nown; i += kClT) {
            const double x = C.tmp[i], u = C.u[i];
            C.Ao[i] = set_em_update(x, cl_acc(C, C.t0 + i), u, C.den[i]);
            if (u > 0.0 && x > 0.0) s2[0] += u * log(x);
        }
        cl_wg_sum<2>(s2, C.red);
        cl_write_scal<2>(C, s2, 5);                  // slots 5, 6
        cl_publish(C, C.Ao, C.P0);                   // both candidates for the next point: EM(X) ...
        cl_publish(C, C.Co, C.P1);                   // ... and the plain C, taken if the extrapolation is rejected
        if (!(alive = cl_barrier(C))) break;
        const bool ok = !extrap || (cl_read_scal(C, 5) - cl_read_scal(C, 6) >= F1);
        if (!ok) {
            for (int i = threadIdx.x; i < nown; i += kClT) C.Ao[i] = C.Co[i];
            if (s >= stepmax) stepmax = fmax(1.0, stepmax / 4.0);
        }
        if ((ok ? s : 1.0) >= stepmax) stepmax *= 4.0;
        cl_load_point(C, ok ? C.P0 : C.P1);          // X = A
        passes += 2;
        res = C.Ao;
        if (passes >= P.max_iter) break;
    }
    if (alive)
        for (int i = threadIdx.x; i < nown; i += kClT) theta_g[g_tid[d.tid_off + C.t0 + i]] = res[i];
    if (threadIdx.x == 0 && C.g == 0) {
        stat[set].passes = passes; stat[set].converged = alive ? converged : 0; stat[set].delta = delta; stat[set].aborted = alive ? 0 : 1;
    }
}

}  // namespace
