"""Minimal reader for R's XDR serialisation format (version 2/3, gzip .Rds).

Only what is needed to lift *data* (numeric/integer/logical/character vectors,
lists, pairlist attributes, factors) out of the fixture files that ship with the
reference's data package (``ChicdiffData/inst/extdata/CD4_Mono_results/*.Rds``).
It is fixture tooling: it is used by ``tools/make_golden_from_rds.py`` in the
authoring container (where ``/root/reference`` exists) and by nothing at run time.

Format notes (R internals manual, "Serialization Formats"): header ``X\\n``, three
ints (format version, writer R version, min reader version), then a recursive item
stream.  Each item starts with a flags word: bits 0-7 SEXPTYPE, bit 8 "is object",
bit 9 "has attributes", bit 10 "has tag".
"""
from __future__ import annotations

import gzip
import struct
from typing import Any

import numpy as np

NA_INTEGER = -2147483648


class RObject:
    """A decoded R value plus its attributes (names, class, levels, ...)."""

    __slots__ = ("value", "attrs")

    def __init__(self, value: Any, attrs: dict | None = None):
        self.value = value
        self.attrs = attrs or {}

    def names(self):
        n = self.attrs.get("names")
        return None if n is None else list(n.value)

    def __repr__(self):  # pragma: no cover - debugging aid
        return f"RObject({type(self.value).__name__}, attrs={list(self.attrs)})"


class _Reader:
    def __init__(self, buf: bytes):
        self.b = buf
        self.p = 0
        self.refs: list[Any] = []

    def i32(self) -> int:
        v = struct.unpack_from(">i", self.b, self.p)[0]
        self.p += 4
        return v

    def raw(self, n: int) -> bytes:
        v = self.b[self.p:self.p + n]
        self.p += n
        return v

    def length(self) -> int:
        n = self.i32()
        if n == -1:  # long vector: two ints (hi, lo)
            hi, lo = self.i32(), self.i32()
            n = (hi << 32) | (lo & 0xFFFFFFFF)
        return n

    def item(self) -> Any:
        flags = self.i32()
        t = flags & 0xFF
        has_attr = bool(flags & (1 << 9))
        has_tag = bool(flags & (1 << 10))
        if t == 254:  # NILVALUE_SXP
            return None
        if t in (253, 252, 251, 250, 249, 248, 242, 241):  # env/namespace sentinels
            return None
        if t == 255:  # REFSXP
            idx = flags >> 8
            if idx == 0:
                idx = self.i32()
            return self.refs[idx - 1]
        if t == 1:  # SYMSXP
            name = self.item()
            self.refs.append(name)
            return name
        if t == 9:  # CHARSXP
            n = self.i32()
            if n == -1:
                return None  # NA_character_
            return self.raw(n).decode("utf-8", errors="replace")
        if t in (2, 6):  # LISTSXP / LANGSXP (pairlist): attr?, tag?, car, cdr
            out = []
            while True:
                attrs = self.item() if has_attr else None  # noqa: F841 (rare; ignored)
                tag = self.item() if has_tag else None
                car = self.item()
                out.append((tag, car))
                flags = self.i32()
                t2 = flags & 0xFF
                if t2 == 254:
                    break
                if t2 not in (2, 6):
                    raise ValueError(f"pairlist cdr of type {t2} unsupported")
                has_attr = bool(flags & (1 << 9))
                has_tag = bool(flags & (1 << 10))
            return out
        if t == 10 or t == 13:  # LGLSXP / INTSXP
            n = self.length()
            v = np.frombuffer(self.b, dtype=">i4", count=n, offset=self.p).astype(np.int32)
            self.p += 4 * n
            val: Any = v
        elif t == 14:  # REALSXP
            n = self.length()
            v = np.frombuffer(self.b, dtype=">f8", count=n, offset=self.p).astype(np.float64)
            self.p += 8 * n
            val = v
        elif t == 16:  # STRSXP
            n = self.length()
            val = [self.item() for _ in range(n)]
        elif t in (19, 20):  # VECSXP / EXPRSXP
            n = self.length()
            val = [self.item() for _ in range(n)]
        elif t == 24:  # RAWSXP
            n = self.length()
            val = self.raw(n)
        elif t == 22:  # EXTPTRSXP: protected value, tag (data.table's .internal.selfref)
            obj = RObject(None)
            self.refs.append(obj)
            self.item()
            self.item()
            val = None
            if has_attr:
                self.item()
            return obj
        else:
            raise ValueError(f"unsupported SEXPTYPE {t} at byte {self.p}")
        attrs = {}
        if has_attr:
            for tag, car in self.item():
                attrs[tag] = car
        return RObject(val, attrs)


def read_rds(path: str) -> RObject:
    """Decode a gzip-compressed XDR ``.Rds`` file into nested :class:`RObject` s."""
    with gzip.open(path, "rb") as fh:
        buf = fh.read()
    if buf[:2] != b"X\n":
        raise ValueError("not an XDR-format R serialisation")
    r = _Reader(buf)
    r.p = 2
    version = r.i32()
    r.i32()  # writer version
    r.i32()  # min reader version
    if version == 3:
        n = r.i32()
        r.raw(n)  # native encoding name
    return r.item()


def as_columns(df: RObject) -> dict[str, Any]:
    """data.frame / data.table -> {column name: numpy array or list of str}.

    Factors are expanded to their string labels; integer/logical NA stays INT_MIN.
    """
    out = {}
    for name, col in zip(df.names(), df.value):
        v = col.value
        if "levels" in col.attrs and isinstance(v, np.ndarray):
            lev = col.attrs["levels"].value
            v = [None if k == NA_INTEGER else lev[k - 1] for k in v]
        out[name] = v
    return out
