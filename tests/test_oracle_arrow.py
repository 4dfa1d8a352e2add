"""The oracle's restatement of the batch wire format (oracle/arrow_serde.py) against the facts the reference's text
pins: coldata.Bytes.Serialize's [elements | buffer] form with offsets [0, ..., 0, 32 n, total] and its n = 0 / n = 1
special cases (bytes.go:577-615), the 30-byte inline limit and element layout (bytes.go:20-53,78-83), the 4-byte
length prefix and 8-byte padding of metadata and body (record_batch.go:84-88,190-232), body length = sum of the
buffers (record_batch.go:174), 2 buffers for Bool / Int / Float columns and 3 otherwise (record_batch.go:33-45), an
empty validity buffer for vectors without NULLs (arrowbatchconverter.go:338-341).  No serialized message is held by
the reference's tests (they are random round trips), so the metadata bytes are parity unpinned."""
import struct

import numpy as np

from cockroach_amd import coldata as cd
from oracle import arrow_serde as A


def test_bytes_serialize_forms():
    assert A.bytes_serialize([]) == (b"", [0])
    assert A.bytes_serialize([b"hello"]) == (b"hello", [0, 5])
    long = bytes(range(31))  # one byte over BytesMaxInlineLength
    data, offs = A.bytes_serialize([b"ab", long, None, b"x" * 30])
    assert offs == [0, 0, 0, 4 * 32, 4 * 32 + 31]
    e0, e1, e2, e3 = (data[i * 32:(i + 1) * 32] for i in range(4))
    assert e0[:2] == b"ab" and e0[30] == 2 and e0[31] == 1 and not any(e0[2:30])
    assert struct.unpack("<qqq", e1[:24]) == (0, 31, 31) and e1[31] == 0
    assert e2 == bytes(30) + b"\x00\x01"
    assert e3[:30] == b"x" * 30 and e3[30] == 30 and e3[31] == 1
    assert data[128:] == long
    assert A.bytes_deserialize(data, offs) == [b"ab", long, b"", b"x" * 30]


def test_message_framing_and_round_trip():
    rng = np.random.default_rng(2)
    for n in (0, 1, 2, 9, 1024):
        vecs = [cd.Vec.from_numpy(cd.INT64, rng.integers(-9, 9, n, dtype=np.int64), rng.random(n) > 0.3 if n else None),
                cd.Vec.from_numpy(cd.INT16, rng.integers(-9, 9, n).astype(np.int16)),
                cd.Vec.from_values(cd.BYTES, [None if i % 7 == 3 else bytes([65 + i % 26]) * (i % 50) for i in range(n)]),
                cd.Vec.from_numpy(cd.BOOL, rng.integers(0, 2, n).astype(np.uint8), rng.random(n) > 0.5 if n else None),
                cd.Vec.from_numpy(cd.FLOAT64, rng.standard_normal(n))]
        types = [v.type for v in vecs]
        msg = A.serialize(vecs, n)
        meta_len = struct.unpack_from("<I", msg, 0)[0]
        assert (4 + meta_len) % 8 == 0 and len(msg) % 8 == 0
        back = A.deserialize(msg, types)
        for a, b in zip(vecs, back):
            assert a.to_values() == b.to_values()
        # a vector without NULLs carries an empty validity buffer: INT16 and FLOAT64 here
        fb = msg[4:4 + meta_len]
        mtab = struct.unpack_from("<I", fb, 0)[0]
        f = A._table_field(fb, mtab, 8)
        rb = f + struct.unpack_from("<I", fb, f)[0]
        f = A._table_field(fb, rb, 8)
        bufs = f + struct.unpack_from("<I", fb, f)[0]
        assert struct.unpack_from("<I", fb, bufs)[0] == 2 + 2 + 3 + 2 + 2
        lens = [struct.unpack_from("<qq", fb, bufs + 4 + 16 * k)[1] for k in range(11)]
        assert lens[2] == 0 and lens[9] == 0
        assert lens[3] == 2 * n and lens[10] == 8 * n
        f = A._table_field(fb, mtab, 10)
        assert (struct.unpack_from("<q", fb, f)[0] if f else 0) == sum(lens)
