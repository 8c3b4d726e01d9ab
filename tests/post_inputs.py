"""Shared inputs of the f3 / f4 tests (IHW application side, region universe)."""
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


def ihw_tables_from_golden(golden):
    """distLookup as far as the reference's result table determines it: per-group avWeights, and breaks anywhere
    between the largest log|avDist| of one group and the smallest of the next (chicdiff.R:2039 puts them at the
    midpoints of the *control* set's group ranges, which the table does not hold); first break 0 =
    (minLogDist[1] + 0)/2 with minLogDist[1] <- 0 (:2030), last Inf."""
    g = golden["group"].astype(np.int64)
    x = np.log(np.abs(golden["avDist"]))
    ng = int(g.max())
    w = np.array([np.unique(golden["avWeights"][g == k])[0] for k in range(1, ng + 1)])
    breaks = [0.0]
    for k in range(1, ng):
        breaks.append(0.5 * (x[g == k].max() + x[g == k + 1].min()))
    breaks.append(np.inf)
    return np.array(breaks), w


def region_universe_case(seed=3, n=4000):
    """Peaks on the real chr19 HindIII fragment IDs of the reference's design folder (first 3000 fragments, real
    bait IDs), cut into three pretend chromosomes with a few IDs missing from the map, so that the
    chromosome / map-end / bait-avoidance rules of getRegionUniverse all fire."""
    rng = np.random.default_rng(seed)
    rmap = np.loadtxt(os.path.join(HERE, "golden", "chr19_HindIII_first3000.rmap"), dtype=str)
    ids = rmap[:, 3].astype(np.int64)
    baits = np.loadtxt(os.path.join(HERE, "golden", "chr19_baitIDs_first3000.txt"), dtype=np.int64)
    id0 = int(ids.min()) - 1  # renumber 1..3000 (the kernels index chr_of by ID)
    ids, baits = ids - id0, baits - id0
    maxfrag = int(ids.max())
    chr_of = np.full(maxfrag + 1, -1, dtype=np.int32)
    chr_of[ids] = np.where(ids <= 1200, 0, np.where(ids <= 2100, 1, 2))
    holes = rng.choice(ids[10:-10], 6, replace=False)
    holes = holes[~np.isin(holes, baits)]
    chr_of[holes] = -1  # IDs the map does not hold
    b = rng.choice(baits, n)
    d = rng.integers(1, 40, n) * rng.choice([-1, 1], n)
    d[: n // 4] = rng.integers(1, 9, n // 4) * rng.choice([-1, 1], n // 4)  # many peaks right next to the bait
    oe = b + d
    oe[n // 2] = maxfrag + 3  # beyond the map
    oe[n // 2 + 1] = maxfrag - 1
    oe[n // 2 + 2] = 2
    keep = oe != b
    return b[keep].astype(np.int32), oe[keep].astype(np.int32), chr_of


def region_universe_literal(bait, oe, s, chr_of):
    """getRegionUniverse, statement by statement (chicdiff.R:353-401), on Python lists: the test's own twin."""
    maxfrag = len(chr_of) - 1
    rows = []
    for region, (b, o) in enumerate(zip(bait.tolist(), oe.tolist()), start=1):
        if abs(b - o) > s + 1:
            a, e = o - s, o + s
        elif o > b:
            a, e = b + 2, o + s
        elif o < b:
            a, e = o - s, b - 2
        else:
            raise ValueError("Invalid parameters")
        seq = range(a, e + 1) if a <= e else range(a, e - 1, -1)  # R's a:b
        rows += [(b, region, x) for x in seq]
    rows = [r for r in rows if r[2] <= maxfrag]                                    # :384
    on_map = lambda i: 1 <= i <= maxfrag and chr_of[i] >= 0
    rows = sorted(rows, key=lambda r: r[2])                                         # setkey(otherEndID), stable
    rows = [(r, chr_of[r[2]] if on_map(r[2]) else None) for r in rows]              # rmap[RU.DT]: chr NA when not on the map
    rows = sorted(rows, key=lambda t: t[0][0])                                      # setkey(baitID), stable
    out = [r for r, c in rows if c is not None and on_map(r[0]) and chr_of[r[0]] == c]   # chr == i.chr (NA drops)
    return np.array(out, dtype=np.int32).reshape(-1, 3)


def golden_regions_as_ru(golden):
    """The 24 863 regions of the reference's own chr19 run as CSR-ordered RU rows: a region is the contiguous
    otherEndID range [minOE, maxOE] of its bait (chicdiff.R:353-367, :1703-1705) on the full chr19 restriction map
    (tests/golden/chr19_design.npz).  Returns ru_bait, ru_oe, region_ptr, id_min, midsum (= start + end per ID)."""
    d = np.load(os.path.join(HERE, "golden", "chr19_design.npz"))
    ids = d["rmap_id"].astype(np.int64)
    assert np.array_equal(ids, np.arange(ids[0], ids[0] + len(ids)))  # dense, ascending: table index = ID - id_min
    lo, hi = golden["minOE"].astype(np.int64), golden["maxOE"].astype(np.int64)
    F = hi - lo + 1
    ptr = np.concatenate([[0], np.cumsum(F)]).astype(np.int64)
    ru_oe = (np.repeat(lo, F) + (np.arange(ptr[-1]) - np.repeat(ptr[:-1], F))).astype(np.int32)
    ru_bait = np.repeat(golden["baitID"], F).astype(np.int32)
    midsum = d["rmap_start"].astype(np.int64) + d["rmap_end"].astype(np.int64)
    return ru_bait, ru_oe, ptr, int(ids[0]), midsum
