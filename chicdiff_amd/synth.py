"""Synthetic inputs of the shapes BASELINE.json names.

NumPy ``Generator(PCG64)`` seeded from ``SeedSequence(20190123)``; rows are generated in
65 536-row chunks from spawned child sequences so that any shard (rank r of N) of the global
matrix can be regenerated independently and identically.

Region level (n x S): mean mu_i ~ LogNormal(ln 19, 1.4); dispersion
alpha_i = (0.05 + 2/mu_i) * LogNormal(0, 0.5); log2 fold change 0 for 90 % of rows, N(0,1)
otherwise; offsets nf_ij = s_j * r_ij (s_j ~ LogNormal(0, .2), r_ij ~ LogNormal(0, .25)),
rows rescaled to geometric mean 1 as ``sc`` is at chicdiff.R:1669; counts
k_ij ~ NB(mean = mu_i 2^(lfc_i g_j) nf_ij, size = 1/alpha_i).

Fragment level (for the window kernels): F fragments per region, counts split
multinomially with Dirichlet(0.3) weights, FullMean per fragment
LogNormal(ln(mu_i/F), 0.5), 1 % of rows with one NA FullMean.
"""
from __future__ import annotations

import numpy as np

SEED = 20190123
CHUNK = 65_536


def groups(S: int) -> np.ndarray:
    """First S/2 samples condition A (0), rest condition B (1)."""
    g = np.zeros(S, dtype=np.int32)
    g[S // 2:] = 1
    return g


def _sample_scales(S: int) -> np.ndarray:
    rng = np.random.Generator(np.random.PCG64(np.random.SeedSequence([SEED, 7, S])))
    return np.exp(rng.normal(0.0, 0.2, S))


def _chunk(ci: int, rows: int, S: int, fragments: int | None):
    rng = np.random.Generator(np.random.PCG64(np.random.SeedSequence([SEED, S, ci])))
    g = groups(S)
    mu = np.exp(rng.normal(np.log(19.0), 1.4, rows))
    alpha = (0.05 + 2.0 / mu) * np.exp(rng.normal(0.0, 0.5, rows))
    lfc = np.where(rng.random(rows) < 0.10, rng.normal(0.0, 1.0, rows), 0.0)
    nf = _sample_scales(S)[None, :] * np.exp(rng.normal(0.0, 0.25, (rows, S)))
    nf /= np.exp(np.log(nf).mean(axis=1, keepdims=True))
    mean = mu[:, None] * np.exp2(lfc[:, None] * g[None, :]) * nf
    size = (1.0 / alpha)[:, None]
    k = rng.negative_binomial(size, size / (size + mean)).astype(np.int32)
    out = {"counts": k, "nf": nf, "mu": mu, "alpha": alpha, "lfc": lfc}
    if fragments:
        F = fragments
        w = rng.dirichlet(np.full(F, 0.3), size=rows)  # (rows, F), shared by the samples of a region
        fragN = np.empty((rows, F, S), dtype=np.int32)
        for j in range(S):
            fragN[:, :, j] = rng.multinomial(k[:, j], w)
        fm = np.exp(rng.normal(np.log(mu / F)[:, None, None], 0.5, (rows, F, S)))
        na_rows = rng.random(rows) < 0.01
        idx = np.nonzero(na_rows)[0]
        fm[idx, rng.integers(0, F, len(idx)), rng.integers(0, S, len(idx))] = np.nan
        out["fragN"] = fragN
        out["fragFullMean"] = fm
    return out


def make(n: int, S: int, start: int = 0, fragments: int | None = None) -> dict:
    """Rows [start, start+n) of the global synthetic matrix for S samples.

    Returns C-ordered (n, S) ``counts`` (int32) and ``nf`` (float64), ``group`` (S,), the
    generating truths, and with ``fragments=F`` the long-form ``fragN`` / ``fragFullMean``
    of shape (n*F, S) plus ``region_ptr`` (n+1,).
    """
    parts = []
    pos = start
    end = start + n
    while pos < end:
        ci, off = divmod(pos, CHUNK)
        take = min(CHUNK - off, end - pos)
        c = _chunk(ci, CHUNK, S, fragments)  # always the whole chunk: a row's value never depends on n
        parts.append({k: v[off:off + take] for k, v in c.items()})
        pos += take
    out = {k: np.concatenate([p[k] for p in parts]) for k in parts[0]}
    out["group"] = groups(S)
    if fragments:
        F = fragments
        out["fragN"] = out["fragN"].reshape(n * F, S)
        out["fragFullMean"] = out["fragFullMean"].reshape(n * F, S)
        out["region_ptr"] = np.arange(0, (n + 1) * F, F, dtype=np.int64)
    return out
