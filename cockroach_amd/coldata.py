"""Host-side mirror of the reference's columnar containers (pkg/col/coldata).

Only the canonical type families the MI355X path handles are modelled
(SURVEY.md 8(a) "Type coverage decision"): Bool, Int16/32/64 (incl. DATE),
Float64 and Bytes (STRING/UUID/ENUM/CHAR).

Two physical layouts exist for a Bytes column:

* the *device / Arrow* layout used inside the library and by the bulk
  device API: ``offsets[n+1]`` (uint32) into a flat ``data`` buffer;
* the *coldata* layout that crosses the cgo boundary untouched
  (pkg/col/coldata/bytes.go:45-74,156-165): 32-byte ``element``s with values of
  <= 30 bytes stored inline and longer ones as {offset,len,cap} into
  ``Bytes.buffer``.  ``bytes_to_elements`` / ``elements_to_bytes`` convert.

Nulls follow coldata.Nulls (pkg/col/coldata/nulls.go:35-41,184-201): a
validity bitmap, LSB first, bit = 1 means the value is present.
"""
from __future__ import annotations

import numpy as np

BOOL, INT16, INT32, INT64, FLOAT64, BYTES = 0, 1, 2, 3, 4, 5
TYPE_NAMES = {BOOL: "bool", INT16: "int16", INT32: "int32", INT64: "int64", FLOAT64: "float64", BYTES: "bytes"}
TYPE_BY_NAME = {v: k for k, v in TYPE_NAMES.items()}
# aliases used by the reference's test tables (types.Int == INT8 == int64, types.Float == float64 ...)
TYPE_BY_NAME.update({"int": INT64, "int2": INT16, "int4": INT32, "int8": INT64, "float": FLOAT64, "string": BYTES, "date": INT64})
NP_DTYPE = {BOOL: np.uint8, INT16: np.int16, INT32: np.int32, INT64: np.int64, FLOAT64: np.float64}
WIDTH = {BOOL: 1, INT16: 2, INT32: 4, INT64: 8, FLOAT64: 8}

BATCH_SIZE = 1024       # coldata.BatchSize() default, pkg/col/coldata/batch.go:79
MAX_BATCH_SIZE = 4096   # coldata.MaxBatchSize, batch.go:102
BYTES_ELEMENT_SIZE = 32          # bytes.go:76-78
BYTES_MAX_INLINE_LENGTH = 30     # bytes.go:80-82


def pack_valid(mask_valid: np.ndarray) -> np.ndarray:
    """bool[n] (True = present) -> validity bitmap, trailing bits set to 1."""
    n = len(mask_valid)
    nb = (n + 7) // 8
    padded = np.ones(nb * 8, dtype=bool)
    padded[:n] = mask_valid
    return np.packbits(padded, bitorder="little")


def unpack_valid(bitmap, n: int) -> np.ndarray:
    if bitmap is None:
        return np.ones(n, dtype=bool)
    return np.unpackbits(np.asarray(bitmap, dtype=np.uint8), bitorder="little")[:n].astype(bool)


class Vec:
    """One column: mirrors coldata.Vec (pkg/col/coldata/vec.go:44-49)."""

    __slots__ = ("type", "data", "valid", "offsets", "n")

    def __init__(self, typ, data, valid=None, offsets=None, n=None):
        self.type = typ
        if typ == BYTES:
            self.data = np.ascontiguousarray(data, dtype=np.uint8)
            self.offsets = np.ascontiguousarray(offsets, dtype=np.uint32)
            self.n = len(self.offsets) - 1 if n is None else n
        else:
            self.data = np.ascontiguousarray(data, dtype=NP_DTYPE[typ])
            self.offsets = None
            self.n = len(self.data) if n is None else n
        self.valid = None if valid is None else np.ascontiguousarray(valid, dtype=np.uint8)

    def __len__(self):
        return self.n

    # -- construction -------------------------------------------------------
    @staticmethod
    def from_values(typ, values) -> "Vec":
        """Build from a python list where None means NULL."""
        n = len(values)
        present = np.array([v is not None for v in values], dtype=bool) if n else np.zeros(0, dtype=bool)
        valid = None if present.all() else pack_valid(present)
        if typ == BYTES:
            chunks, offs = [], [0]
            for v in values:
                b = b"" if v is None else (v.encode() if isinstance(v, str) else bytes(v))
                chunks.append(b)
                offs.append(offs[-1] + len(b))
            data = np.frombuffer(b"".join(chunks), dtype=np.uint8).copy() if offs[-1] else np.zeros(0, np.uint8)
            return Vec(BYTES, data, valid, np.array(offs, dtype=np.uint32))
        arr = np.zeros(n, dtype=NP_DTYPE[typ])
        for i, v in enumerate(values):
            if v is not None:
                arr[i] = v
        return Vec(typ, arr, valid)

    @staticmethod
    def from_numpy(typ, arr, present=None) -> "Vec":
        valid = None if present is None or bool(np.all(present)) else pack_valid(np.asarray(present, dtype=bool))
        return Vec(typ, arr, valid)

    @staticmethod
    def bytes_from_fixed(mat: np.ndarray, present=None) -> "Vec":
        """uint8[n, w] -> Bytes column of n values of width w."""
        n, w = mat.shape
        offs = (np.arange(n + 1, dtype=np.uint64) * w).astype(np.uint32)
        valid = None if present is None or bool(np.all(present)) else pack_valid(np.asarray(present, dtype=bool))
        return Vec(BYTES, mat.reshape(-1), valid, offs)

    # -- access -------------------------------------------------------------
    def present(self) -> np.ndarray:
        return unpack_valid(self.valid, self.n)

    def get_bytes(self, i) -> bytes:
        return self.data[self.offsets[i]:self.offsets[i + 1]].tobytes()

    def to_values(self):
        pres = self.present()
        out = []
        for i in range(self.n):
            if not pres[i]:
                out.append(None)
            elif self.type == BYTES:
                out.append(self.get_bytes(i))
            elif self.type == BOOL:
                out.append(bool(self.data[i]))
            elif self.type == FLOAT64:
                out.append(float(self.data[i]))
            else:
                out.append(int(self.data[i]))
        return out

    def take(self, idx) -> "Vec":
        """Vec.Copy with a selection vector (pkg/col/coldata/vec_tmpl.go:187)."""
        idx = np.asarray(idx, dtype=np.int64)
        pres = self.present()[idx] if self.valid is not None else None
        if self.type == BYTES:
            lens = (self.offsets[1:].astype(np.int64) - self.offsets[:-1].astype(np.int64))[idx]
            offs = np.zeros(len(idx) + 1, dtype=np.int64)
            np.cumsum(lens, out=offs[1:])
            data = np.zeros(int(offs[-1]), dtype=np.uint8)
            for k, i in enumerate(idx):
                data[offs[k]:offs[k + 1]] = self.data[self.offsets[i]:self.offsets[i + 1]]
            v = Vec(BYTES, data, None, offs.astype(np.uint32))
        else:
            v = Vec(self.type, self.data[idx])
        if pres is not None and not pres.all():
            v.valid = pack_valid(pres)
        return v

    def slice(self, start, end) -> "Vec":
        return self.take(np.arange(start, end, dtype=np.int64))


class Batch:
    """coldata.Batch (pkg/col/coldata/batch.go:24): columns + length + optional selection vector."""

    def __init__(self, cols, length=None, sel=None):
        self.cols = list(cols)
        self.sel = None if sel is None else np.ascontiguousarray(sel, dtype=np.int64)
        if length is None:
            length = len(self.sel) if self.sel is not None else (len(self.cols[0]) if self.cols else 0)
        self.length = int(length)

    def dense(self) -> "Batch":
        """Materialise the selection vector (colexecutils.Deselector)."""
        if self.sel is None:
            if all(len(c) == self.length for c in self.cols):
                return self
            return Batch([c.slice(0, self.length) for c in self.cols], self.length)
        return Batch([c.take(self.sel[: self.length]) for c in self.cols], self.length)


def bytes_to_elements(vec: Vec):
    """Arrow layout -> coldata.Bytes (elements uint8[n,32], buffer uint8[])."""
    n = len(vec)
    elems = np.zeros((n, BYTES_ELEMENT_SIZE), dtype=np.uint8)
    buf_chunks, buf_len = [], 0
    hdr = elems.view(np.int64).reshape(n, 4)
    for i in range(n):
        a, b = int(vec.offsets[i]), int(vec.offsets[i + 1])
        ln = b - a
        if ln <= BYTES_MAX_INLINE_LENGTH:
            elems[i, :ln] = vec.data[a:b]
            elems[i, 30] = ln
            elems[i, 31] = 1
        else:
            hdr[i, 0], hdr[i, 1], hdr[i, 2] = buf_len, ln, ln
            elems[i, 31] = 0
            buf_chunks.append(vec.data[a:b])
            buf_len += ln
    buf = np.concatenate(buf_chunks) if buf_chunks else np.zeros(0, dtype=np.uint8)
    return elems, np.ascontiguousarray(buf, dtype=np.uint8)


def elements_to_bytes(elems: np.ndarray, buf: np.ndarray, n: int, valid=None) -> Vec:
    """coldata.Bytes -> Arrow layout."""
    elems = np.asarray(elems, dtype=np.uint8).reshape(-1, BYTES_ELEMENT_SIZE)[:n]
    hdr = np.ascontiguousarray(elems).view(np.int64).reshape(n, 4) if n else np.zeros((0, 4), np.int64)
    chunks, offs = [], [0]
    for i in range(n):
        if elems[i, 31]:
            ln = int(elems[i, 30])
            chunks.append(elems[i, :ln].tobytes())
        else:
            o, ln = int(hdr[i, 0]), int(hdr[i, 1])
            chunks.append(np.asarray(buf[o:o + ln], dtype=np.uint8).tobytes())
        offs.append(offs[-1] + len(chunks[-1]))
    data = np.frombuffer(b"".join(chunks), dtype=np.uint8).copy() if offs[-1] else np.zeros(0, np.uint8)
    return Vec(BYTES, data, valid, np.array(offs, dtype=np.uint32))
