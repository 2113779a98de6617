"""Prototype (numpy, CPU) of the safeguarded second-order step for the set-resident solver: SQUAREM cycles as in
k_solve_sets, plus -- once a set has not converged after `start` passes -- projected Newton steps whose direction comes
from preconditioned CG on the free variables (Hessian-vector products = one row sweep + one column sweep, like an EM pass).
Counts pass-equivalents per set with and without the Newton steps on bench.py's time_to_mle problem."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from emsar_amd import synth


def build_problem():
    rng = np.random.default_rng(11)
    sizes = np.minimum(rng.zipf(1.6, size=40000), 60)
    sizes = sizes[np.cumsum(sizes) <= 100000]
    n_tx, rp, ci, _ = synth.family_matrix([int(x) for x in sizes], rows_per_tid=3, seed=11, dup=0.0)
    E = rng.uniform(0.5, 2.0, size=len(rp) - 1)
    theta_true = np.where(rng.random(n_tx) < 0.3, 0.0, rng.lognormal(0.0, 2.0, size=n_tx))
    R = rng.poisson(E * np.add.reduceat(theta_true[ci], rp[:-1].astype(np.int64))).astype(np.int32)
    return n_tx, rp.astype(np.int64), ci, R, E, sizes


class SetP:
    """dense per-set problem: A [rows x nt] multiplicities, R rows, den, u"""
    def __init__(self, A, R, den, u):
        self.A, self.R, self.den, self.u = A, R, den, u
        self.passes = 0

    def em(self, x, want_F=False):
        S = self.A @ x
        live = S > 0
        w = np.where(live, self.R / np.where(live, S, 1), 0.0)
        acc = self.A.T @ w
        y = np.where(self.den > 0, np.where(x > 0, (x * acc + self.u) / np.where(self.den > 0, self.den, 1), 0.0), 0.0)
        self.passes += 1
        if want_F:
            F = (self.R[live] * np.log(S[live])).sum() + (self.u[(self.u > 0) & (x > 0)] * np.log(x[(self.u > 0) & (x > 0)])).sum() - (x * self.den).sum()
            if np.any((self.R > 0) & ~live) or np.any((self.u > 0) & (x <= 0)): F = -np.inf
            return y, F, acc
        return y

    def F(self, x):
        S = self.A @ x
        self.passes += 1
        if np.any((self.R > 0) & (S <= 0)) or np.any((self.u > 0) & (x <= 0)): return -np.inf
        live = S > 0
        m = (self.u > 0)
        return (self.R[live] * np.log(S[live])).sum() + (self.u[m] * np.log(x[m])).sum() - (x * self.den).sum()


def delta_of(x, y, tol_floor=1e-6):
    return np.max(np.abs(y - x) / (np.abs(y) + tol_floor)) if len(x) else 0.0


def newton_step(P, x, Fx, eps_reads=1e-6, cg_max=None, cg_tol=1e-4):
    """one projected Newton step from x (F(x) = Fx).  returns (x_new, F_new, accepted)"""
    A, R, den, u = P.A, P.R, P.den, P.u
    S = A @ x
    live = S > 0
    w = np.where(live, R / np.where(live, S, 1), 0.0)
    h = np.where(live, w / np.where(live, S, 1), 0.0)           # R / S^2
    acc = A.T @ w
    P.passes += 1
    up = np.where(x > 0, u / np.where(x > 0, x, 1), 0.0)
    g = acc - den + up
    # bound set: tiny components that the gradient pushes down
    bound = (den > 0) & (g < 0) & (x * den < eps_reads) & (u == 0)
    dead = den <= 0
    free = ~bound & ~dead
    d = np.zeros_like(x)
    d[bound] = -x[bound]
    if free.any():
        diag = (A * A).T @ h + np.where(x > 0, u / np.where(x > 0, x * x, 1), 0.0)
        diag = np.where(diag > 0, diag, 1.0)
        Minv = 1.0 / diag
        def Hv(v):
            P.passes += 1
            return A.T @ (h * (A @ v)) + np.where(x > 0, u / np.where(x > 0, x * x, 1), 0.0) * v
        b = np.where(free, g, 0.0)
        z = np.zeros_like(x); r = b.copy(); q = np.where(free, Minv * r, 0.0); p = q.copy()
        rq = r @ q; rq0 = rq
        n_it = cg_max or int(free.sum())
        for _ in range(n_it):
            if rq <= cg_tol * cg_tol * rq0 or rq <= 0: break
            Hp = np.where(free, Hv(p), 0.0)
            pHp = p @ Hp
            if pHp <= 0: break
            al = rq / pHp
            z += al * p; r -= al * Hp
            q = np.where(free, Minv * r, 0.0)
            rq_new = r @ q
            p = q + (rq_new / rq) * p
            rq = rq_new
        d[free] = z[free]
    alpha = 1.0
    for _ in range(4):
        xn = np.maximum(x + alpha * d, 0.0)
        xn[bound] = 0.0 if alpha == 1.0 else xn[bound]
        Fn = P.F(xn)
        if Fn >= Fx - 1e-13 * abs(Fx):
            return xn, Fn, True
        alpha *= 0.25
    return x, Fx, False


def solve(P, newton, tol=1e-10, max_iter=200000, start=30, every=1):
    nt = len(P.den)
    A_ = np.where(P.den > 0, 1.0, 0.0)
    stepmax = 1.0
    cycles = 0
    cooldown = 0
    while P.passes < max_iter:
        B = P.em(A_)
        if delta_of(A_, B) < tol:
            # KKT check for components at exactly zero (only Newton creates them)
            z = (B == 0) & (P.den > 0)
            if newton and z.any():
                S = P.A @ B; live = S > 0
                acc = P.A.T @ np.where(live, P.R / np.where(live, S, 1), 0.0)
                bad = z & (acc - P.den > 1e-9 * P.den)
                if bad.any():
                    B = B.copy(); B[bad] = 1e-9 / P.den[bad]; A_ = B; continue
            return B, True
        Cc, F1, _ = P.em(B, want_F=True)
        r = B - A_; v = (Cc - B) - r
        sv = v @ v
        s = np.sqrt((r @ r) / sv) if sv > 0 else 1.0
        s = min(max(s, 1.0), stepmax)
        extrap = s > 1.01
        X = Cc
        if extrap:
            Y = A_ + 2 * s * r + s * s * v
            X = np.where((Y > 0) & (Cc > 0), Y, Cc)
        An, Fx, _ = P.em(X, want_F=True)
        ok = (not extrap) or Fx >= F1
        if not ok:
            An = Cc
            if s >= stepmax: stepmax = max(1.0, stepmax / 4)
        if (s if ok else 1.0) >= stepmax: stepmax *= 4
        A_ = An
        cycles += 1
        if newton and P.passes >= start and cooldown == 0:
            Fa = P.F(A_)
            xn, Fn, acc_ = newton_step(P, A_, Fa)
            if acc_: A_ = xn
            else: cooldown = 8
        elif cooldown: cooldown -= 1
    return A_, False


def main():
    n_tx, rp, ci, R, E, sizes = build_problem()
    # components = families (block diagonal by construction); rows with R>0 only
    base = np.concatenate([[0], np.cumsum(sizes)])
    fam_of_tid = np.repeat(np.arange(len(sizes)), sizes)
    row_fam = fam_of_tid[ci[rp[:-1]]]
    den = np.zeros(n_tx); np.add.at(den, ci, np.repeat(E, np.diff(rp)))
    order = np.argsort(row_fam, kind="stable")
    starts = np.searchsorted(row_fam[order], np.arange(len(sizes) + 1))
    todo = [f for f in range(len(sizes)) if sizes[f] >= 2]
    sel = sys.argv[1:] and int(sys.argv[1]) or 400
    rng = np.random.default_rng(0)
    big = [f for f in todo if sizes[f] >= (20 if sel < 5000 else 2)]
    pick = list(rng.choice(big, size=min(sel, len(big)), replace=False))
    res = []
    t0 = time.time()
    for f in pick:
        rows = order[starts[f]:starts[f + 1]]
        nt = sizes[f]; t0_ = base[f]
        A = np.zeros((len(rows), nt)); Rr = R[rows].astype(float)
        for i, r_ in enumerate(rows):
            for k in range(rp[r_], rp[r_ + 1]): A[i, ci[k] - t0_] += 1
        single = (A > 0).sum(1) == 1
        u = np.zeros(nt)
        for i in np.where(single)[0]:
            u[np.argmax(A[i])] += Rr[i]
        keep = ~single & (Rr > 0)
        out = []
        for newton in (False, True):
            P = SetP(A[keep], Rr[keep], den[t0_:t0_ + nt], u)
            th, conv = solve(P, newton, max_iter=200000)
            out.append((P.passes, conv, P.F(th), th))
        res.append((f, nt, out[0][0], out[1][0], out[0][1], out[1][1], out[1][2] - out[0][2], np.abs(out[1][3] - out[0][3]).max()))
    res.sort(key=lambda x: -x[2])
    print("family nt | passes EM-SQUAREM | passes with Newton | conv | dF (newton - em) | max|dtheta|")
    for r_ in res[:25]:
        print("%6d %3d | %7d | %6d | %s %s | %+.3e | %.2e" % r_)
    a = np.array([r_[2] for r_ in res]); b = np.array([r_[3] for r_ in res])
    print("sets %d: max passes %d -> %d, sum %d -> %d, unconverged with newton %d, time %.0fs" % (len(res), a.max(), b.max(), a.sum(), b.sum(), sum(1 for r_ in res if not r_[5]), time.time() - t0))
    print("worst dF %.3e, worst max|dtheta| %.3e" % (min(r_[6] for r_ in res), max(r_[7] for r_ in res)))


if __name__ == "__main__":
    main()
