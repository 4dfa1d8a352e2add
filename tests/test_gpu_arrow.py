"""GPU parity of the batch wire format (csrc/arrow.hip): serialized RecordBatch messages as the
reference's cFetcherWrapper / Outbox produce them (colserde/record_batch.go, arrowbatchconverter.go, coldata/bytes.go)
to device columns and back, against the oracle's restatement (oracle/arrow_serde.py).  Each side's reader is checked
against the OTHER side's writer (their flatbuffer layouts differ on purpose), plus GPU -> GPU round trips at size."""
import numpy as np
import pytest

from cockroach_amd import _lib
from cockroach_amd import coldata as cd
from cockroach_amd import device as dev
from oracle import arrow_serde as A
from tests import helpers as H

pytestmark = pytest.mark.gpu


def make_batch(rng, n, with_nulls=True):
    def nulls():
        return (rng.random(n) > 0.15) if (with_nulls and n) else None
    words = [b"", b"a", b"\x00", b"x" * 30, b"y" * 31, b"hello world", bytes(range(256)), b"z" * 70]
    bytes_vals = [words[int(j)] + (b"%d" % i if i % 3 else b"") for i, j in enumerate(rng.integers(0, len(words), size=n))]
    if with_nulls:
        bytes_vals = [None if i % 11 == 5 else v for i, v in enumerate(bytes_vals)]
    return [cd.Vec.from_numpy(cd.INT64, rng.integers(-2**62, 2**62, n, dtype=np.int64), nulls()),
            cd.Vec.from_values(cd.BYTES, bytes_vals),
            cd.Vec.from_numpy(cd.BOOL, rng.integers(0, 2, n).astype(np.uint8), nulls()),
            cd.Vec.from_numpy(cd.INT16, rng.integers(-2**15, 2**15, n).astype(np.int16)),
            cd.Vec.from_numpy(cd.INT32, rng.integers(-2**31, 2**31, n).astype(np.int32), nulls()),
            cd.Vec.from_numpy(cd.FLOAT64, rng.standard_normal(n), nulls()),
            cd.Vec.from_values(cd.BYTES, [b"k%d" % (i % 5) for i in range(n)])]


def same(got, want, msg=""):
    assert len(got) == len(want)
    for c, (g, w) in enumerate(zip(got, want)):
        assert g.type == w.type and g.to_values() == w.to_values(), f"{msg}: column {c}"


@pytest.mark.parametrize("n", [0, 1, 2, 7, 8, 9, 1024, 70_001])
@pytest.mark.parametrize("with_nulls", [False, True])
def test_arrow_read_reference_form(n, with_nulls):
    """Deserialize + ArrowToBatch on the device: messages written by the oracle (flatbuffers-builder layout)."""
    rng = np.random.default_rng(n + 17)
    vecs = make_batch(rng, n, with_nulls)
    types = [v.type for v in vecs]
    h = dev.ArrowSerde(types)
    got = dev.result_to_vecs(h.read(A.serialize(vecs, n)))
    same(got, vecs, f"n={n}")
    # a second, smaller message on the same handle (buffers are reused)
    m = n // 3
    small = [v.slice(0, m) for v in vecs]
    same(dev.result_to_vecs(h.read(A.serialize(small, m))), small, f"n={m} after n={n}")
    h.close()


@pytest.mark.parametrize("n", [0, 1, 2, 5, 8, 1023, 70_001])
@pytest.mark.parametrize("with_nulls", [False, True])
def test_arrow_write_read_by_the_oracle(n, with_nulls):
    """BatchToArrow + Serialize on the device: the oracle's Deserialize + ArrowToBatch reads the message back."""
    rng = np.random.default_rng(n + 29)
    vecs = make_batch(rng, n, with_nulls)
    types = [v.type for v in vecs]
    h = dev.ArrowSerde(types)
    msg = h.write([dev.upload(v) for v in vecs], n)
    assert len(msg) % 8 == 0
    same(A.deserialize(msg, types), vecs, f"n={n}")
    # trailing validity bits are zeroed ("to conform to the Arrow spec", arrowbatchconverter.go:138-141) and the BYTES
    # values are in the [elements | buffer] form
    same(dev.result_to_vecs(h.read(msg)), vecs, f"n={n}, read back on the device")
    h.close()


def test_arrow_round_trip_at_size():
    """2^20 rows x (INT64, FLOAT64 with NULLs, 16-byte BYTES keys, BYTES of 0..79 bytes): device -> message -> device,
    compared array by array."""
    rng = np.random.default_rng(5)
    n = 1 << 20
    lens = rng.integers(0, 80, size=n)
    offl = np.zeros(n + 1, dtype=np.uint32)
    offl[1:] = np.cumsum(lens)
    vecs = [cd.Vec.from_numpy(cd.INT64, rng.integers(-2**62, 2**62, n, dtype=np.int64)),
            cd.Vec.from_numpy(cd.FLOAT64, rng.random(n), rng.random(n) > 0.1),
            cd.Vec.bytes_from_fixed(rng.integers(0, 256, (n, 16)).astype(np.uint8)),
            cd.Vec(cd.BYTES, rng.integers(0, 256, int(offl[-1])).astype(np.uint8), None, offl)]
    h = dev.ArrowSerde([v.type for v in vecs])
    msg = h.write([dev.upload(v) for v in vecs], n)
    assert len(msg) >= n * (8 + 8 + 32 + 32)
    back = dev.result_to_vecs(h.read(msg))
    H.assert_vecs_equal(back, vecs, "round trip")
    h.close()


def test_decoded_kv_batch_as_serialized_scan_response(oracle):
    """The direct-columnar-scan shape (cfetcher_wrapper.go:85-143): KV bytes -> crgpu_dev_decode_kv -> crgpu_dev_arrow_write;
    the message deserializes (oracle) to the columns the oracle's cFetcher restatement decodes."""
    rng = np.random.default_rng(41)
    n = 5000
    kt = [cd.INT64, cd.BYTES]
    vt = [cd.INT64, cd.FLOAT64, cd.BYTES, cd.BOOL, cd.INT16]
    ids = [3, 4, 7, 8, 20]
    keys, vals = H.kv_random_rows(rng, n, kt, vt)
    t = oracle.KvTable(120, 1, kt, [0, 1], ids, vt)
    data, offs = oracle.kv_encode(t, keys, vals, max_batch_bytes=32 << 10)
    spec = (t.prefix_len(), kt, [0, 1], [0, 1], ids, vt, [2, 3, 4, 5, 6], kt + vt)
    want = oracle.kv_decode(*spec, data, offs)
    d = dev.Decoder(*spec)
    pad = (-len(data)) % 16
    buf = dev.DeviceBuffer.from_numpy(np.concatenate([data, np.zeros(pad + 16, np.uint8)]))
    res = d.decode(buf, offs)
    h = dev.ArrowSerde(kt + vt)
    msg = h.write(dev.result_columns(res), int(res.n))
    same(A.deserialize(msg, kt + vt), want, "decode -> serialize")
    h.close()
    d.close()


def test_arrow_malformed_messages():
    rng = np.random.default_rng(3)
    vecs = make_batch(rng, 100)
    types = [v.type for v in vecs]
    msg = bytearray(A.serialize(vecs, 100))
    h = dev.ArrowSerde(types)

    def expect_invalid(m, tys=None):
        hh = h if tys is None else dev.ArrowSerde(tys)
        with pytest.raises(_lib.CrgpuError) as e:
            hh.read(bytes(m))
        assert e.value.code == _lib.ERR_INVALID

    expect_invalid(msg[:6])                      # shorter than the prefix + anything
    expect_invalid(msg[: len(msg) // 2])         # body cut off
    expect_invalid(msg, types[:-1])              # schema with fewer columns
    expect_invalid(msg, [cd.INT64] * len(types))  # buffer count differs
    bad = bytearray(msg)
    bad[0:4] = (2**31).to_bytes(4, "little")     # metadata length beyond the message
    expect_invalid(bad)
    # an element that points outside its buffer
    two = [cd.Vec.from_values(cd.BYTES, [b"q" * 40, b"r" * 50])]
    m2 = bytearray(A.serialize(two, 2))
    at = m2.find(b"q" * 40) - 64  # start of the elements
    m2[at:at + 8] = (10**6).to_bytes(8, "little")
    expect_invalid(m2, [cd.BYTES])
    with pytest.raises(_lib.CrgpuError) as e:
        dev.ArrowSerde([99])
    assert e.value.code == _lib.ERR_UNSUPPORTED
    h.close()
