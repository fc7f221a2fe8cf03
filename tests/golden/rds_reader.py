"""Minimal reader for R's XDR-serialised ``.rds`` files (gzip or plain).

Used ONLY by ``make_fixtures.py`` to turn the reference's validation data
(`/root/reference/validation/data/*.rds`, produced by the reference's
``introspect::capture_sim0_data`` -- `crates/engine-core/src/introspect.rs:104-155`)
into small ``.npz``/``.json`` fixtures.  It executes nothing from the file: it
only walks the documented SEXP tags of R's serialisation format version 2/3.
"""
from __future__ import annotations

import gzip
import struct
from typing import Any

import numpy as np

NILVALUE_SXP = 254
REFSXP = 255
SYMSXP, LISTSXP, CHARSXP, LGLSXP, INTSXP, REALSXP, STRSXP, VECSXP = 1, 2, 9, 10, 13, 14, 16, 19
ALTREP_SXP = 238


class RObj:
    """A decoded R value plus its attributes."""

    def __init__(self, value: Any, attrs: dict | None = None):
        self.value = value
        self.attrs = attrs or {}

    def __repr__(self) -> str:  # pragma: no cover - debugging aid
        return f"RObj({self.value!r}, attrs={list(self.attrs)})"


class _Reader:
    def __init__(self, buf: bytes):
        self.b = buf
        self.p = 0
        self.refs: list[Any] = []

    def u32(self) -> int:
        v = struct.unpack_from(">i", self.b, self.p)[0]
        self.p += 4
        return v

    def raw(self, n: int) -> bytes:
        v = self.b[self.p : self.p + n]
        self.p += n
        return v

    def length(self) -> int:
        n = self.u32()
        if n == -1:  # long vector
            hi, lo = self.u32(), self.u32()
            n = (hi << 32) | (lo & 0xFFFFFFFF)
        return n

    def item(self) -> Any:
        flags = self.u32()
        ty = flags & 0xFF
        has_attr = bool(flags & (1 << 9))
        has_tag = bool(flags & (1 << 10))
        if ty == NILVALUE_SXP:
            return None
        if ty == REFSXP:
            idx = flags >> 8
            if idx == 0:
                idx = self.u32()
            return self.refs[idx - 1]
        if ty == SYMSXP:
            name = self.item()
            self.refs.append(name)
            return name
        if ty == LISTSXP:
            # pairlist: (attr?) (tag?) car cdr ; flatten into list of (tag, value)
            out = []
            while True:
                if has_attr:
                    self.item()
                tag = self.item() if has_tag else None
                car = self.item()
                out.append((tag, car))
                flags = self.u32()
                ty = flags & 0xFF
                has_attr = bool(flags & (1 << 9))
                has_tag = bool(flags & (1 << 10))
                if ty == NILVALUE_SXP:
                    break
                if ty != LISTSXP:
                    raise ValueError(f"unexpected pairlist cdr type {ty}")
            return out
        if ty == CHARSXP:
            n = self.u32()
            if n == -1:
                return None
            return self.raw(n).decode("utf-8", "replace")
        if ty == LGLSXP or ty == INTSXP:
            n = self.length()
            v = np.frombuffer(self.raw(4 * n), dtype=">i4").astype(np.int32)
            val: Any = v
        elif ty == REALSXP:
            n = self.length()
            val = np.frombuffer(self.raw(8 * n), dtype=">f8").astype(np.float64)
        elif ty == STRSXP:
            n = self.length()
            val = [self.item() for _ in range(n)]
        elif ty == VECSXP:
            n = self.length()
            val = [self.item() for _ in range(n)]
        elif ty == ALTREP_SXP:
            info = self.item()
            state = self.item()
            self.item()  # attr
            cls = info[0][1] if info else None
            # compact_intseq / compact_realseq: state = [n, start, step]
            st = state.value if isinstance(state, RObj) else state
            if cls in ("compact_intseq", "compact_realseq"):
                n, start, step = int(st[0]), st[1], st[2]
                arr = start + step * np.arange(n)
                return RObj(arr.astype(np.int32 if cls == "compact_intseq" else np.float64))
            raise ValueError(f"unsupported ALTREP class {cls}")
        else:
            raise ValueError(f"unsupported SEXP type {ty} at byte {self.p}")
        attrs = {}
        if has_attr:
            for tag, v in self.item():
                attrs[tag] = v
        return RObj(val, attrs)


def read_rds(path: str) -> Any:
    with open(path, "rb") as fh:
        head = fh.read(2)
    if head == b"\x1f\x8b":
        with gzip.open(path, "rb") as fh:
            buf = fh.read()
    else:
        with open(path, "rb") as fh:
            buf = fh.read()
    if buf[:2] != b"X\n":
        raise ValueError("not an XDR rds stream")
    r = _Reader(buf)
    r.p = 2
    version = r.u32()
    r.u32()  # writer version
    r.u32()  # min reader version
    if version == 3:
        n = r.u32()
        r.raw(n)  # native encoding
    return r.item()


def to_py(obj: Any) -> Any:
    """Collapse RObj trees into dict / list / ndarray / str."""
    if isinstance(obj, RObj):
        names = obj.attrs.get("names")
        v = obj.value
        if isinstance(v, list):
            items = [to_py(x) for x in v]
            if names is not None:
                nm = to_py(names)
                if len(nm) == len(items) and all(isinstance(s, str) and s for s in nm):
                    return dict(zip(nm, items))
            if all(isinstance(x, (str, type(None))) for x in items):
                return items
            return items
        if isinstance(v, np.ndarray):
            dim = obj.attrs.get("dim")
            out: Any = v
            if dim is not None:
                d = to_py(dim)
                out = v.reshape(tuple(int(x) for x in d), order="F")
            if names is not None:
                nm = to_py(names)
                if len(nm) == v.shape[0] and v.ndim == 1:
                    return {k: out[i] for i, k in enumerate(nm)}
            return out
        return v
    return obj
