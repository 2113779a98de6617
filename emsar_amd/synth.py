"""Seeded synthetic read->transcript compatibility matrices of the BASELINE.json shapes.

A read picks its transcript with probability proportional to theta*_t * len_t (theta* ~ LogNormal(0,2), 30 %
exact zeros) and is compatible with a window of k neighbouring transcript ids around it (isoforms of one gene
are neighbours in a cDNA FASTA, hence in EMSAR's tid space); a small fraction also hits one transcript
anywhere (cross-family paralog).  k follows the per-config alignment-count law.  Rows are emitted in random
read order, exactly like an unsorted alignment file: any locality the kernels exploit has to be created by the
library at upload time.

The generator is ours (the reference's readgenerator.c is not built and works at sequence level); it emits the
CSR directly.  den_t = len_t/1e3 * N/1e6 plays the role of sum_c m_ct E_c (effective length in kb x million reads).
"""
import numpy as np

CONFIGS = {
    # name: (n_tx, n_reads, k law, cross-family fraction, seed)
    "cfg2": dict(n_tx=80_000, n_reads=5_000_000, law="poisson2", xfam=0.0, seed=2),
    "cfg3": dict(n_tx=200_000, n_reads=50_000_000, law="human", xfam=0.02, seed=3),
    "cfg4": dict(n_tx=200_000, n_reads=20_000_000, law="human", xfam=0.02, seed=40),
    "cfg5": dict(n_tx=250_000, n_reads=200_000_000, law="repeats", xfam=0.02, seed=5),
}


def _draw_k(rng, n, law, geo_p=None):
    if law == "poisson2":      # 1 + Poisson(2) truncated to [1,100]: mean 3
        k = 1 + rng.poisson(2.0, n)
    elif law == "human":       # 60 % unique, rest 1 + geometric (mean 10): overall mean ~5, cap 100 (-k 100)
        k = np.ones(n, dtype=np.int64)
        multi = rng.random(n) >= 0.6
        k[multi] = 1 + rng.geometric(geo_p or 0.1, int(multi.sum()))
    elif law == "repeats":     # heavy repeats: 10 % of reads hit 50-100 members, rest 1 + geometric(mean 13): mean ~20
        k = 1 + rng.geometric(1.0 / 13.0, n)
        big = rng.random(n) < 0.1
        k[big] = rng.integers(50, 101, int(big.sum()))
    else:
        raise ValueError(law)
    return np.clip(k, 1, 100).astype(np.int32)


def make_abundance(n_tx, seed):
    rng = np.random.default_rng(seed)
    theta = rng.lognormal(0.0, 2.0, n_tx)
    theta[rng.random(n_tx) < 0.3] = 0.0
    length = np.clip(rng.lognormal(np.log(1800.0), 0.6, n_tx), 200, 20000)
    return theta, length


def make_families(n_tx, seed):
    """Gene families of SURVEY.md 8d: sizes ~ Zipf(1.6) capped at 60 isoforms, laid out as consecutive tid ranges
    (isoforms of one gene are neighbours in a cDNA FASTA).  Returns (fam_start[F+1] int32, fam_of[n_tx] int32)."""
    rng = np.random.default_rng(seed + 2000)
    sizes = []
    tot = 0
    while tot < n_tx:
        z = np.minimum(rng.zipf(1.6, size=max(1024, n_tx // 4)), 60)
        sizes.append(z)
        tot += int(z.sum())
    sizes = np.concatenate(sizes)
    cs = np.cumsum(sizes)
    nf = int(np.searchsorted(cs, n_tx, side="left")) + 1
    sizes = sizes[:nf].astype(np.int64)
    sizes[-1] -= int(cs[nf - 1]) - n_tx          # the last family is cut to fit
    start = np.zeros(nf + 1, dtype=np.int32)
    np.cumsum(sizes, out=start[1:])
    fam_of = np.repeat(np.arange(nf, dtype=np.int32), sizes)
    return start, fam_of


def _family_subsets(rng, t0, k, fam_start, fam_of):
    """Per read: a uniformly random k-subset of its family that contains t0, as a bitmask over the family's members
    (families hold <= 60 transcripts).  Floyd's sampling of k-1 of the n-1 other members, vectorised over the reads."""
    f = fam_of[t0]
    fs = fam_start[f]
    nf = (fam_start[f + 1] - fs).astype(np.int64)
    p0 = (t0 - fs).astype(np.uint64)
    m = np.minimum(k.astype(np.int64), nf) - 1           # others to draw
    N = nf - 1
    S = np.zeros(len(t0), dtype=np.uint64)
    idx = np.nonzero(m > 0)[0]
    i = 0
    one = np.uint64(1)
    while idx.size:
        j = (N[idx] - m[idx] + i)                          # Floyd: j runs over N-m .. N-1
        t = np.minimum((rng.random(idx.size) * (j + 1)).astype(np.int64), j)
        taken = (S[idx] >> t.astype(np.uint64)) & one
        pick = np.where(taken == 1, j, t).astype(np.uint64)
        S[idx] |= one << pick
        i += 1
        idx = idx[m[idx] > i]
    # open a gap at p0 for the read's own transcript
    low = (one << p0) - one
    S = (S & low) | ((S & ~low) << one) | (one << p0)
    return S, fs


def make_matrix(n_tx, n_reads, law="human", xfam=0.02, seed=1, block=None, structure="window", geo_p=None):
    """Returns dict(row_ptr u64[n_reads+1], col_idx i32[nnz], den f64[n_tx], theta_true, n_tx, n_reads).

    structure: what the compatible set of a multi-read looks like in tid space
      "window"           k consecutive tids around the read's transcript (rounds 1-2; flatters any layout that relies on runs)
      "family"           SURVEY.md 8d as written: gene families ~ Zipf(1.6) capped at 60 isoforms, a multi-read hits a random
                         SUBSET of its family (size by the config's law, capped at the family size) -- what update_ReadCounts
                         sees: sorted isoform subsets, not runs (emsar_functions.c:838-943)
      "family_shuffled"  the same matrix with the transcripts numbered in random order (tid order carries no information)
    """
    if structure != "window":
        return _make_matrix_family(n_tx, n_reads, law, xfam, seed, block or 2_000_000, structure == "family_shuffled", geo_p)
    block = block or 5_000_000
    theta, length = make_abundance(n_tx, seed)
    p = theta * length
    cdf = np.cumsum(p)
    cdf /= cdf[-1]
    rng = np.random.default_rng(seed + 1000)
    lens = np.empty(n_reads, dtype=np.int32)
    cols = []
    for b0 in range(0, n_reads, block):
        n = min(block, n_reads - b0)
        t0 = np.searchsorted(cdf, rng.random(n), side="right").astype(np.int32)
        np.minimum(t0, n_tx - 1, out=t0)
        k = np.minimum(_draw_k(rng, n, law), n_tx)
        # window of k consecutive tids that contains t0
        start = t0 - (rng.random(n) * k).astype(np.int32)
        np.clip(start, 0, n_tx - k, out=start)
        x = (rng.random(n) < xfam) if xfam > 0 else np.zeros(n, dtype=bool)
        kk = k + x.astype(np.int32)
        lens[b0:b0 + n] = kk
        rp = np.zeros(n + 1, dtype=np.int64)
        np.cumsum(kk, out=rp[1:])
        nnz = int(rp[-1])
        row = np.repeat(np.arange(n, dtype=np.int32), kk)
        j = (np.arange(nnz, dtype=np.int64) - rp[row]).astype(np.int32)
        c = start[row] + j
        # the extra cross-family hit sits in the last slot of its row
        last = rp[1:][x] - 1
        c[last] = rng.integers(0, n_tx, int(x.sum()), dtype=np.int32)
        cols.append(c.astype(np.int32))
        del row, j, c, rp
    col_idx = np.concatenate(cols) if len(cols) > 1 else cols[0]
    del cols
    row_ptr = np.zeros(n_reads + 1, dtype=np.uint64)
    np.cumsum(lens, out=row_ptr[1:], dtype=np.uint64)
    den = length / 1e3 * (n_reads / 1e6)
    return {"n_tx": n_tx, "n_reads": n_reads, "row_ptr": row_ptr, "col_idx": col_idx, "den": den,
            "theta_true": theta}


def _make_matrix_family(n_tx, n_reads, law, xfam, seed, block, shuffled, geo_p):
    if geo_p is None and law == "human":
        geo_p = 0.076      # the family size caps a row: 1 + geometric(mean 13) gives SURVEY 8d's target mean of 5 alignments per read
    theta, length = make_abundance(n_tx, seed)
    fam_start, fam_of = make_families(n_tx, seed)
    p = theta * length
    cdf = np.cumsum(p)
    cdf /= cdf[-1]
    rng = np.random.default_rng(seed + 1000)
    lens = np.empty(n_reads, dtype=np.int32)
    cols = []
    for b0 in range(0, n_reads, block):
        n = min(block, n_reads - b0)
        t0 = np.searchsorted(cdf, rng.random(n), side="right").astype(np.int32)
        np.minimum(t0, n_tx - 1, out=t0)
        k = _draw_k(rng, n, law, geo_p)
        S, fs = _family_subsets(rng, t0, k, fam_start, fam_of)
        bits = np.unpackbits(S.view(np.uint8).reshape(n, 8), axis=1, bitorder="little")
        row, bit = np.nonzero(bits)                       # row-major: CSR order, members ascending inside a row
        del bits
        kk = np.bincount(row, minlength=n).astype(np.int32)
        x = (rng.random(n) < xfam) if xfam > 0 else np.zeros(n, dtype=bool)
        tot = kk + x.astype(np.int32)
        lens[b0:b0 + n] = tot
        rp = np.zeros(n + 1, dtype=np.int64)
        np.cumsum(tot, out=rp[1:])
        rp_main = np.zeros(n + 1, dtype=np.int64)
        np.cumsum(kk, out=rp_main[1:])
        c = np.empty(int(rp[-1]), dtype=np.int32)
        pos = rp[row] + (np.arange(len(row), dtype=np.int64) - rp_main[row])
        c[pos] = fs[row] + bit.astype(np.int32)
        # the extra cross-family hit sits in the last slot of its row
        c[rp[1:][x] - 1] = rng.integers(0, n_tx, int(x.sum()), dtype=np.int32)
        cols.append(c)
        del row, bit, pos, rp, rp_main, S
    col_idx = np.concatenate(cols) if len(cols) > 1 else cols[0]
    del cols
    row_ptr = np.zeros(n_reads + 1, dtype=np.uint64)
    np.cumsum(lens, out=row_ptr[1:], dtype=np.uint64)
    den = length / 1e3 * (n_reads / 1e6)
    out = {"n_tx": n_tx, "n_reads": n_reads, "row_ptr": row_ptr, "col_idx": col_idx, "den": den, "theta_true": theta,
           "fam_start": fam_start}
    if shuffled:
        new_of_old = np.random.default_rng(seed + 3000).permutation(n_tx).astype(np.int32)
        out["col_idx"] = new_of_old[col_idx]
        d = np.empty_like(den); d[new_of_old] = den
        th = np.empty_like(theta); th[new_of_old] = theta
        out["den"], out["theta_true"], out["new_of_old"] = d, th, new_of_old
    return out


STRUCTURES = ("window", "family", "family_shuffled")


def make_config(name, scale=1.0, structure="window"):
    """BASELINE.json config by name; scale < 1 shrinks reads AND transcripts proportionally (parity tests)."""
    c = dict(CONFIGS[name])
    c["n_reads"] = max(1000, int(c["n_reads"] * scale))
    c["n_tx"] = max(500, int(c["n_tx"] * scale))
    return make_matrix(structure=structure, **c)


def collapse(row_ptr, col_idx):
    """Read-level -> segment-level: unique rows (as tid multisets in the given order) + counts.
    What the reference's update_ReadCounts does read by read (emsar_functions.c:838-943).  Small inputs only
    (python dict); used by tests to check that the collapsed and the read-level solve agree."""
    seen, order, counts = {}, [], []
    rp = row_ptr.astype(np.int64)
    for r in range(len(rp) - 1):
        key = tuple(sorted(col_idx[rp[r]:rp[r + 1]].tolist()))
        i = seen.get(key)
        if i is None:
            seen[key] = len(order)
            order.append(key)
            counts.append(1)
        else:
            counts[i] += 1
    new_rp = np.zeros(len(order) + 1, dtype=np.uint64)
    new_rp[1:] = np.cumsum([len(k) for k in order])
    new_col = np.fromiter((t for k in order for t in k), dtype=np.int32, count=int(new_rp[-1]))
    return new_rp, new_col, np.array(counts, dtype=np.int32)


def family_matrix(sizes, rows_per_tid=3, seed=0, dup=0.3, singles=0.3):
    """Block-diagonal incidence: one family of transcripts per entry of `sizes`; rows draw 1..4 tids (with repeats)
    from one family; a share `dup` of the rows repeats an earlier row (identical tid multiset)."""
    rng = np.random.default_rng(seed)
    rp, ci, w = [0], [], []
    base = 0
    for n in sizes:
        fam_rows = []
        for _ in range(max(1, rows_per_tid * n)):
            if fam_rows and rng.random() < dup:
                row = fam_rows[rng.integers(len(fam_rows))]
            elif rng.random() < singles:
                row = [base + int(rng.integers(n))] * int(rng.integers(1, 3))
            else:
                row = list(base + rng.integers(0, n, size=int(rng.integers(2, 5))))
            fam_rows.append(row)
            ci.extend(int(x) for x in row)
            rp.append(len(ci))
            w.append(int(rng.integers(0, 40)))
        base += n
    return base, np.array(rp, dtype=np.uint64), np.array(ci, dtype=np.int32), np.array(w, dtype=np.int32)
