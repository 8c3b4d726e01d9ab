"""f2, text part: the .chinput parser of the library (chicdiff_amd/csrc/chinput.hip, host threads) against a
line-by-line Python reading of the same file, in the format the reference reads with fread() at chicdiff.R:828
('#' comment line, header `baitID otherEndID N otherEndLen distSign`, tab-separated rows, distSign
may be NA).  CPU only: the parser needs no device."""
import ctypes as C
import os

import numpy as np
import pytest


@pytest.fixture(scope="module")
def lib():
    import __graft_entry__ as g
    g.build()
    from chicdiff_amd import hip
    return hip.load_library()


def parse(lib, path, threads=4, cap=10_000_000):
    b, o, n = (np.empty(cap, np.int32) for _ in range(3))
    nrows = C.c_int64(0)
    err = C.create_string_buffer(256)
    P = C.POINTER(C.c_int32)
    rc = lib.chicdiff_hip_selftest_chinput(os.fsencode(path), threads, cap, b.ctypes.data_as(P), o.ctypes.data_as(P), n.ctypes.data_as(P),
                                           C.byref(nrows), err, 256)
    if rc:
        raise ValueError(err.value.decode())
    k = nrows.value
    return b[:k].copy(), o[:k].copy(), n[:k].copy()


def write_chinput(path, bait, oe, N, sep="\t", comment=True, header=("baitID", "otherEndID", "N", "otherEndLen", "distSign"),
                  crlf=False, trailing_newline=True, blank_every=0, rng=None):
    rng = rng or np.random.default_rng(0)
    nl = "\r\n" if crlf else "\n"
    cols = {"baitID": bait, "otherEndID": oe, "N": N, "otherEndLen": rng.integers(100, 20000, len(bait)),
            "distSign": np.where(rng.random(len(bait)) < 0.1, -1, rng.integers(-10 ** 6, 10 ** 6, len(bait)))}
    lines = []
    if comment:
        lines.append("#\tsamplename=x\tbamname=x.bam\tbaitmapfile=b.baitmap\tdigestfile=d.rmap")
    lines.append(sep.join(header))
    for i in range(len(bait)):
        fields = []
        for h in header:
            v = cols[h][i]
            fields.append("NA" if (h == "distSign" and v == -1) else str(int(v)))
        lines.append(sep.join(fields))
        if blank_every and i % blank_every == blank_every - 1:
            lines.append("")
    text = nl.join(lines) + (nl if trailing_newline else "")
    with open(path, "w", newline="") as f:
        f.write(text)


def make_rows(n, seed=1):
    rng = np.random.default_rng(seed)
    bait = np.sort(rng.integers(1, 800_000, n)).astype(np.int32)
    oe = rng.integers(1, 840_000, n).astype(np.int32)
    N = rng.geometric(0.3, n).astype(np.int32)
    return bait, oe, N


@pytest.mark.parametrize("kw", [dict(), dict(sep=" "), dict(crlf=True), dict(trailing_newline=False), dict(comment=False),
                                dict(blank_every=97), dict(header=("otherEndID", "N", "distSign", "baitID", "otherEndLen"))])
def test_parser_reads_what_was_written(lib, tmp_path, kw):
    bait, oe, N = make_rows(50_000)
    path = tmp_path / "s.chinput"
    write_chinput(path, bait, oe, N, **kw)
    for threads in (1, 3, 16):
        b, o, n = parse(lib, path, threads)
        assert np.array_equal(b, bait) and np.array_equal(o, oe) and np.array_equal(n, N), (kw, threads)


def test_parser_small_and_bad_files(lib, tmp_path):
    path = tmp_path / "tiny.chinput"
    write_chinput(path, np.array([5]), np.array([7]), np.array([3]))
    assert [a.tolist() for a in parse(lib, path, 8)] == [[5], [7], [3]]
    path.write_text("#c\nbaitID\totherEndID\tN\n")  # header only: no rows
    assert len(parse(lib, path)[0]) == 0
    path.write_text("#c\nbait\totherEndID\tN\n1\t2\t3\n")
    with pytest.raises(ValueError, match="baitID"):
        parse(lib, path)
    path.write_text("baitID\totherEndID\tN\n1\t2\tx\n")
    with pytest.raises(ValueError, match="malformed"):
        parse(lib, path)
    path.write_text("baitID\totherEndID\tN\n1\t2\n")
    with pytest.raises(ValueError, match="malformed"):
        parse(lib, path)
    path.write_text("baitID\totherEndID\tN\n1\t2\t99999999999\n")
    with pytest.raises(ValueError, match="malformed"):
        parse(lib, path)
    with pytest.raises(ValueError, match="cannot open"):
        parse(lib, tmp_path / "missing.chinput")


def test_parsed_columns_feed_the_oracle_key_table(lib, tmp_path):
    """Parsed columns -> oracle.count_table -> oracle.count_join reproduce the reference's left join of RU with the
    chinput rows of the RU baits, N := 0 where absent (chicdiff.R:828-831, :843-858), restated with pandas."""
    import pandas as pd
    from oracle import oracle
    rng = np.random.default_rng(3)
    bait, oe, N = make_rows(20_000, seed=4)
    keep = np.unique(np.stack([bait, oe], 1), axis=0, return_index=True)[1]  # a chinput holds one row per pair
    bait, oe, N = bait[np.sort(keep)], oe[np.sort(keep)], N[np.sort(keep)]
    path = tmp_path / "s.chinput"
    write_chinput(path, bait, oe, N)
    b, o, n = parse(lib, path, 5)
    ru_baits = np.unique(rng.choice(bait, 300))
    pick = rng.choice(len(bait), 3000)
    ru = pd.DataFrame({"baitID": np.concatenate([bait[pick], rng.choice(ru_baits, 500)]),
                       "otherEndID": np.concatenate([oe[pick], rng.integers(1, 840_000, 500)])}).drop_duplicates()
    ru = ru[ru["baitID"].isin(ru_baits)].sort_values(["baitID", "otherEndID"]).reset_index(drop=True)
    flags = np.zeros(int(b.max()) + 1, np.uint8)
    flags[ru_baits] = 1
    keys, vals = oracle.count_table(b, o, n, flags)
    got = oracle.count_join(ru["baitID"].to_numpy(np.int32), ru["otherEndID"].to_numpy(np.int32), keys, vals)
    x = pd.DataFrame({"baitID": bait, "otherEndID": oe, "N": N})
    x = x[x["baitID"].isin(ru_baits)]
    want = ru.merge(x, how="left", on=["baitID", "otherEndID"])["N"].fillna(0).to_numpy(np.int32)
    assert np.array_equal(got, want) and (want > 0).sum() > 20 and (want == 0).sum() > 20
