"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the reference's batch wire format, used by tests/ to produce and to
check serialized RecordBatch messages for cockroach_amd/csrc/arrow.hip.  Nothing in the product imports this module.

Follows
  coldata.Bytes.Serialize / Deserialize                       pkg/col/coldata/bytes.go:560-640 (element layout :20-53)
  colserde.ArrowBatchConverter.BatchToArrow / ArrowToBatch    pkg/col/colserde/arrowbatchconverter.go:130-470
  colserde.RecordBatchSerializer.Serialize / Deserialize      pkg/col/colserde/record_batch.go:96-314
  arrowserde Message / RecordBatch / FieldNode / Buffer        pkg/col/colserde/arrowserde/message_generated.go

Parity unpinned for the metadata BYTES: the reference's tests (record_batch_test.go, arrowbatchconverter_test.go) hold
no literal serialized message, only random round trips, and there is no Go toolchain here to produce one.  What IS
pinned by the reference's text: the field slots of the two flatbuffers tables (vtable offsets 4/6/8/10 and 4/6/8), the
16-byte FieldNode / Buffer structs, the 4-byte length prefix, the 8-byte padding rule and the [elements | buffer] form
of BYTES values with offsets [0, ..., 0, 32 n, total].  The writer below lays the flatbuffer out as the flatbuffers
builder does (objects prepended back to front: buffers vector, nodes vector, RecordBatch, Message, root offset), which
differs from the layout arrow.hip writes: each reader is therefore tested against the other side's writer.
"""
import struct

import numpy as np

from cockroach_amd import coldata as cd

ELEM = 32
MAX_INLINE = 30


def bytes_serialize(values):
    """Bytes.Serialize for n values (None = NULL, kept as the zero-length inlined element Set(i, nil) leaves).
    Returns (data bytes, int32 offsets list)."""
    n = len(values)
    if n == 0:
        return b"", [0]
    if n == 1:
        v = values[0] or b""
        return bytes(v), [0, len(v)]
    elems, buf = bytearray(), bytearray()
    for v in values:
        v = v or b""
        if len(v) <= MAX_INLINE:
            e = bytearray(ELEM)
            e[: len(v)] = v
            e[30] = len(v)
            e[31] = 1
        else:
            e = bytearray(struct.pack("<qqq", len(buf), len(v), len(v)) + bytes(8))
            buf += v
        elems += e
    data = bytes(elems) + bytes(buf)
    return data, [0] * (n - 1) + [len(elems), len(data)]


def bytes_deserialize(data, offsets):
    """Bytes.Deserialize: list of n byte strings."""
    n = len(offsets) - 1
    if n == 0:
        return []
    if n == 1:
        return [bytes(data[offsets[0]:offsets[1]])]
    assert offsets[n - 1] == n * ELEM, "offsets are not the [0, ..., 0, 32 n, total] form"
    buf = data[n * ELEM:offsets[n]]
    out = []
    for i in range(n):
        e = data[i * ELEM:(i + 1) * ELEM]
        if e[31]:
            out.append(bytes(e[: e[30]]))
        else:
            off, ln, _cap = struct.unpack("<qqq", e[:24])
            out.append(bytes(buf[off:off + ln]))
    return out


def _column_buffers(vec: cd.Vec, n):
    """BatchToArrow for one vector: the 2 or 3 body buffers."""
    bitmap = b""
    if vec.valid is not None:
        bm = np.array(vec.valid[: (n + 7) // 8], dtype=np.uint8, copy=True)
        if n % 8 and bm.size:
            bm[-1] &= (1 << (n % 8)) - 1  # nulls.Truncate(batch.Length())
        bitmap = bm.tobytes()
    if vec.type == cd.BYTES:
        vals = vec.to_values()[:n]
        data, offs = bytes_serialize(vals)
        return [bitmap, np.array(offs, dtype=np.int32).tobytes(), data]
    return [bitmap, np.ascontiguousarray(vec.data[:n]).tobytes()]


class _Builder:
    """The flatbuffers builder, as far as the two tables need it: bytes are PREPENDED, offsets are measured from the
    end of the buffer."""

    def __init__(self):
        self.buf = bytearray()
        self.minalign = 1

    def offset(self):
        return len(self.buf)

    def prep(self, size, additional):
        self.minalign = max(self.minalign, size)
        pad = (-(len(self.buf) + additional)) % size
        self.buf[:0] = bytes(pad)

    def prepend(self, fmt, v):
        size = struct.calcsize(fmt)
        self.prep(size, 0)
        self.buf[:0] = struct.pack("<" + fmt, v)

    def prepend_uoffset(self, target):
        self.prep(4, 0)
        self.buf[:0] = struct.pack("<I", len(self.buf) - target + 4)

    def struct_vector(self, items):
        """vector of 16-byte structs (two int64)."""
        self.prep(4, 16 * len(items))
        self.prep(8, 16 * len(items))
        for a, b in reversed(items):
            self.prep(8, 16)
            self.buf[:0] = struct.pack("<qq", a, b)
        self.buf[:0] = struct.pack("<I", len(items))
        return self.offset()

    def table(self, fields):
        """fields: list of (slot, fmt or 'uoffset', value) in the order they are added; returns the table's offset."""
        obj_end = self.offset()
        slots = {}
        for slot, fmt, v in fields:
            if fmt == "uoffset":
                self.prepend_uoffset(v)
            else:
                self.prepend(fmt, v)
            slots[slot] = self.offset()
        self.prepend("i", 0)  # soffset placeholder
        tab = self.offset()
        nslots = max(slots) + 1
        vt = struct.pack("<HH", 4 + 2 * nslots, tab - obj_end)
        for s in range(nslots):
            vt += struct.pack("<H", tab - slots[s] if s in slots else 0)
        self.buf[:0] = vt  # (2-byte aligned: every table here ends on a 4-byte boundary)
        vt_off = self.offset()
        pos = len(self.buf) - tab
        self.buf[pos:pos + 4] = struct.pack("<i", vt_off - tab)
        return tab

    def finish(self, root):
        self.prep(self.minalign, 4)
        self.prepend_uoffset(root)
        return bytes(self.buf)


def serialize(vecs, n=None):
    """ArrowBatchConverter.BatchToArrow + RecordBatchSerializer.Serialize of one batch -> message bytes."""
    n = (len(vecs[0]) if vecs else 0) if n is None else n
    cols = [_column_buffers(v, n) for v in vecs]
    lens = [len(b) for c in cols for b in c]
    total = sum(lens)
    b = _Builder()
    nodes = b.struct_vector([(n, 0)] * len(vecs))
    offs, at = [], 0
    for ln in lens:
        offs.append((at, ln))
        at += ln
    buffers = b.struct_vector(offs)
    rb = b.table([(0, "q", n), (1, "uoffset", nodes), (2, "uoffset", buffers)])
    pad_body = (8 - (total & 7)) & 7
    # MessageAddVersion(MetadataVersionV1 = 0) is the slot's default: the builder leaves it out of the table
    msg = b.table([(1, "B", 3), (2, "uoffset", rb), (3, "q", total)])  # totalBufferLen, without the body padding (record_batch.go:174)
    meta = b.finish(msg)
    pad_meta = (8 - ((4 + len(meta)) & 7)) & 7
    out = struct.pack("<I", len(meta) + pad_meta) + meta + bytes(pad_meta)
    for c in cols:
        for x in c:
            out += x
    return out + bytes(pad_body)


def _table_field(buf, tab, slot_off):
    so = struct.unpack_from("<i", buf, tab)[0]
    vt = tab - so
    vt_len = struct.unpack_from("<H", buf, vt)[0]
    if slot_off + 2 > vt_len:
        return 0
    o = struct.unpack_from("<H", buf, vt + slot_off)[0]
    return tab + o if o else 0


def deserialize(msg, types):
    """RecordBatchSerializer.Deserialize + ArrowBatchConverter.ArrowToBatch -> list of Vecs."""
    meta_len = struct.unpack_from("<I", msg, 0)[0]
    assert (4 + meta_len) % 8 == 0, "metadata is padded to 8 bytes"
    fb = msg[4:4 + meta_len]
    mtab = struct.unpack_from("<I", fb, 0)[0]
    f = _table_field(fb, mtab, 6)
    assert f and fb[f] == 3, "header_type must be RecordBatch"
    f = _table_field(fb, mtab, 10)
    body_len = struct.unpack_from("<q", fb, f)[0] if f else 0
    body = msg[4 + meta_len:4 + meta_len + body_len]
    assert len(msg) >= 4 + meta_len + body_len
    f = _table_field(fb, mtab, 8)
    rb = f + struct.unpack_from("<I", fb, f)[0]
    f = _table_field(fb, rb, 4)
    n = struct.unpack_from("<q", fb, f)[0] if f else 0
    f = _table_field(fb, rb, 6)
    nodes = f + struct.unpack_from("<I", fb, f)[0]
    f = _table_field(fb, rb, 8)
    bufs = f + struct.unpack_from("<I", fb, f)[0]
    assert struct.unpack_from("<I", fb, nodes)[0] == len(types)
    k, out = 0, []
    for c, t in enumerate(types):
        assert struct.unpack_from("<q", fb, nodes + 4 + 16 * c)[0] == n
        nb = 3 if t == cd.BYTES else 2
        parts = []
        for _ in range(nb):
            off, ln = struct.unpack_from("<qq", fb, bufs + 4 + 16 * k)
            parts.append(body[off:off + ln])
            k += 1
        valid = np.frombuffer(parts[0], dtype=np.uint8)[: (n + 7) // 8].copy() if len(parts[0]) else None
        if valid is not None and n % 8:
            valid[-1] |= (0xFF << (n % 8)) & 0xFF  # coldata keeps the bits past the length set
        if t == cd.BYTES:
            offs = np.frombuffer(parts[1], dtype=np.int32)[: n + 1].tolist()
            vals = bytes_deserialize(parts[2], offs)
            pres = None if valid is None else [(valid[i >> 3] >> (i & 7)) & 1 for i in range(n)]
            out.append(cd.Vec.from_values(cd.BYTES, [v if pres is None or pres[i] else None for i, v in enumerate(vals)]))
        else:
            data = np.frombuffer(parts[1], dtype=cd.NP_DTYPE[t])[:n].copy() if len(parts[1]) else np.zeros(n, dtype=cd.NP_DTYPE[t])
            out.append(cd.Vec(t, data, valid))
    return out
