"""The oracle's KV codec (oracle/crdb_kv.c) against the reference's literal vectors
(tests/golden/encoding_vectors.json) and through encode -> decode round trips."""
import json
import math
import os

import numpy as np
import pytest

from cockroach_amd import coldata as cd
from tests import helpers as H

with open(os.path.join(H.GOLDEN, "encoding_vectors.json")) as f:
    VEC = json.load(f)


@pytest.mark.parametrize("kind", ["varint_asc", "varint_desc", "uvarint_asc"])
def test_int_key_encoding_vectors(oracle, kind):
    for v, hx in VEC[kind]:
        enc = oracle.kv_enc(kind, int(v))
        assert enc.hex() == hx, (kind, v)
        dec, n = oracle.kv_dec(kind, enc + b"tail")
        assert (dec, n) == (int(v), len(enc))


def test_bytes_key_encoding_vectors(oracle):
    prev = None
    for raw, hx in VEC["bytes_asc"]:
        enc = oracle.kv_enc("bytes_asc", bytes.fromhex(raw))
        assert enc.hex() == hx
        assert prev is None or prev < enc  # order preserving (encoding_test.go:522-527)
        prev = enc
        assert oracle.kv_dec("bytes_asc", enc + b"remainder") == (bytes.fromhex(raw), len(enc))
        d = oracle.kv_enc("bytes_desc", bytes.fromhex(raw))
        assert d[0] == 0x13 and bytes(255 - b for b in d[1:]) == enc[1:]
        assert oracle.kv_dec("bytes_desc", d) == (bytes.fromhex(raw), len(d))


def test_float_key_encoding_vectors(oracle):
    prev = None
    for v, hx in VEC["float_asc"]:
        f = float(v)
        enc = oracle.kv_enc("float_asc", f)
        assert enc.hex() == hx, v
        if prev is not None and hx != "04":
            assert prev <= enc
        prev = enc
        dec, n = oracle.kv_dec("float_asc", enc)
        assert n == len(enc) and (math.isnan(dec) if math.isnan(f) else dec == f)
        dd, n = oracle.kv_dec("float_desc", oracle.kv_enc("float_desc", f))
        assert math.isnan(dd) if math.isnan(f) else dd == f


def test_value_side_primitives_round_trip(oracle):
    rng = np.random.default_rng(1)
    vals = [0, 1, -1, 63, 64, -64, -65, 127, 128, 2**31, -2**31, 2**63 - 1, -2**63] + \
           [int(x) for x in rng.integers(-2**62, 2**62, 200)]
    for v in vals:
        enc = oracle.kv_enc("stdlib_varint", v)
        assert oracle.kv_dec("stdlib_varint", enc + b"\x00") == (v, len(enc))
    assert oracle.kv_enc("stdlib_varint", 1).hex() == "02" and oracle.kv_enc("stdlib_varint", -1).hex() == "01"
    for v in [0, 1, 127, 128, 300, 16383, 16384, 2**35, 2**63, 2**64 - 1]:
        enc = oracle.kv_enc("nonsorting_uvarint", v)
        assert oracle.kv_dec("nonsorting_uvarint", enc) == (v, len(enc))
    assert oracle.kv_enc("nonsorting_uvarint", 300).hex() == "822c"  # most significant group first


def _record(key: bytes, value: bytes) -> bytes:
    mvcc = key + b"\x00" + (1489504989617188491).to_bytes(8, "big") + b"\x09"
    return len(value).to_bytes(4, "little") + len(mvcc).to_bytes(4, "little") + mvcc + value


def test_decode_reference_kv_dump(oracle):
    """docs/tech-notes/encoding.md:380-384: the primary index of accounts(id, owner, balance).
    owner (column 2) is fetched, balance (column 3, DECIMAL) is skipped by its encoded length."""
    data = b""
    for i, kv in enumerate(VEC["kv_values"]):
        key = oracle.kv_enc("uvarint_asc", 51) + oracle.kv_enc("uvarint_asc", 1) + oracle.kv_enc("varint_asc", i + 1) + b"\x88"
        data += _record(key, bytes.fromhex(kv["value"]))
    arr = np.frombuffer(data, dtype=np.uint8)
    out = oracle.kv_decode(2, [cd.INT64], [0], [0], [2], [cd.BYTES], [1], [cd.INT64, cd.BYTES], arr, [0, len(arr)])
    assert out[0].to_values() == [1, 2, 3, 4, 5]
    want = [None if kv["owner"] is None else bytes.fromhex(kv["owner"]) for kv in VEC["kv_values"]]
    assert out[1].to_values() == want


@pytest.mark.parametrize("dirs", [(0, 0, 0, 0), (1, 1, 1, 1), (0, 1, 0, 1)])
def test_encode_decode_round_trip(oracle, dirs):
    rng = np.random.default_rng(sum(dirs) + 5)
    n = 3000
    kt = [cd.INT64, cd.BYTES, cd.FLOAT64, cd.INT32]
    vt = [cd.INT64, cd.FLOAT64, cd.BYTES, cd.BOOL, cd.INT16, cd.BYTES]
    ids = [5, 6, 9, 10, 40, 300]
    keys, vals = H.kv_random_rows(rng, n, kt, vt)
    want_keys = [H.kv_key_canonical(k) for k in keys]
    t = oracle.KvTable(106, 1, kt, list(dirs), ids, vt)
    data, offs = oracle.kv_encode(t, keys, vals, max_batch_bytes=64 << 10, explicit_null_every=3)
    assert len(offs) > 3
    out = oracle.kv_decode(t.prefix_len(), kt, list(dirs), [0, 1, 2, 3], ids, vt, [4, 5, 6, 7, 8, 9], kt + vt, data, offs)
    H.assert_vecs_equal(out, want_keys + vals, "all columns")
    # a projection: two key columns skipped, value columns 6, 10 and 300 only, in another output order
    out = oracle.kv_decode(t.prefix_len(), kt, list(dirs), [-1, 2, -1, 0], [6, 10, 300], [vt[1], vt[3], vt[5]],
                           [1, 4, 3], [cd.INT32, cd.FLOAT64, cd.BYTES, cd.BYTES, cd.BOOL], data, offs)
    H.assert_vecs_equal(out, [want_keys[3], vals[1], want_keys[1], vals[5], vals[3]], "projection")


def test_decode_errors(oracle):
    t = oracle.KvTable(60, 1, [cd.INT64], [0], [2], [cd.INT64])
    k = cd.Vec.from_numpy(cd.INT64, np.arange(5, dtype=np.int64))
    data, offs = oracle.kv_encode(t, [k], [k])
    args = (t.prefix_len(), [cd.INT64], [0], [0], [2], [cd.INT64], [1], [cd.INT64, cd.INT64])
    assert oracle.kv_decode(*args, data, offs)[1].to_values() == [0, 1, 2, 3, 4]
    with pytest.raises(oracle.KvDecodeError) as e:
        oracle.kv_decode(*args, data[:-3], [0, len(data) - 3])  # truncated batch
    assert e.value.code == -1
    bad = data.copy()
    rec0_key_len = int.from_bytes(bytes(bad[4:8]), "little")
    bad[8 + rec0_key_len + 4] = 3  # value tag BYTES instead of TUPLE in family 0: the row sentinel of the pre-family
    out = oracle.kv_decode(*args, bad, offs)  # format, "a value is not expected" (cfetcher.go:1192-1196): the column is NULL
    assert out[1].to_values() == [None, 1, 2, 3, 4] and out[0].to_values() == [0, 1, 2, 3, 4]


def _family_key(oracle, table_id, index_id, pk, family):
    key = oracle.kv_enc("uvarint_asc", table_id) + oracle.kv_enc("uvarint_asc", index_id) + oracle.kv_enc("varint_asc", pk)
    if family == 0:
        return key + b"\x88"  # keys.MakeFamilyKey: family 0 has no length suffix
    fam = oracle.kv_enc("uvarint_asc", family)
    return key + fam + oracle.kv_enc("uvarint_asc", len(fam))


def test_decode_reference_kv_dump_with_families(oracle):
    """docs/tech-notes/encoding.md:169-235: accounts(id, owner, balance) with FAMILY f0 (id, balance), FAMILY f1 (owner).
    f1 holds one column that is its DefaultColumnID: its KVs carry a BYTES-tagged single value, and rows whose owner is
    NULL have no f1 KV at all.  balance (DECIMAL, column 3) sits in f0's TUPLE and is skipped by its encoded length."""
    dump = VEC["kv_families_dump"]
    data = b""
    for kv in dump["kvs"]:
        data += _record(_family_key(oracle, dump["table_id"], dump["index_id"], kv["id"], kv["family"]), bytes.fromhex(kv["value"]))
    arr = np.frombuffer(data, dtype=np.uint8)
    fam = oracle.KvFamilies([0, 1], [0, 2])
    out = oracle.kv_decode(2, [cd.INT64], [0], [0], [2], [cd.BYTES], [1], [cd.INT64, cd.BYTES], arr, [0, len(arr)], families=fam)
    assert out[0].to_values() == [1, 2, 3, 4, 5]
    assert out[1].to_values() == [None if o is None else o.encode() for o in dump["owner"]]


FAM_LAYOUTS = {
    # family ids, DefaultColumnID per family, family (index) of each of the six value columns 5, 6, 9, 10, 40, 300
    "tuple+singles": ([0, 1, 4, 7], [0, 6, 40, 300], [0, 1, 0, 0, 2, 3]),   # three single-column families: FLOAT, INT16, BYTES
    "two tuples": ([0, 3], [0, 0], [0, 1, 1, 0, 1, 0]),
    "sparse ids": ([0, 130, 70000], [0, 10, 0], [0, 2, 2, 1, 0, 2]),          # family ids beyond one byte; BOOL single
}


@pytest.mark.parametrize("layout", sorted(FAM_LAYOUTS))
@pytest.mark.parametrize("mvcc_header_every", [0, 3])
def test_encode_decode_round_trip_families(oracle, layout, mvcc_header_every):
    """Rows of several KVs: one per column family (rowenc.EncodePrimaryIndex), TUPLE values and single-column
    ("legacy") values, KVs suppressed when all their columns are NULL, every third value behind an MVCC extended header."""
    fam_ids, defaults, val_family = FAM_LAYOUTS[layout]
    rng = np.random.default_rng(len(layout) + mvcc_header_every)
    n = 4000
    kt, dirs = [cd.INT64, cd.BYTES], [0, 1]
    vt = [cd.INT64, cd.FLOAT64, cd.BYTES, cd.BOOL, cd.INT16, cd.BYTES]
    ids = [5, 6, 9, 10, 40, 300]
    keys, vals = H.kv_random_rows(rng, n, kt, vt)
    keys[0] = cd.Vec.from_numpy(cd.INT64, rng.permutation(n * 3)[:n].astype(np.int64) - n)  # a primary key: unique, no NULLs
    want_keys = [H.kv_key_canonical(k) for k in keys]
    t = oracle.KvTable(106, 1, kt, dirs, ids, vt)
    fam = oracle.KvFamilies(fam_ids, defaults, val_family, mvcc_header_every)
    data, offs = oracle.kv_encode(t, keys, vals, max_batch_bytes=64 << 10, explicit_null_every=3, families=fam)
    one, _ = oracle.kv_encode(t, keys, vals, max_batch_bytes=64 << 10, explicit_null_every=3)
    assert len(offs) > 3 and len(data) > len(one)  # more KVs than rows
    out = oracle.kv_decode(t.prefix_len(), kt, dirs, [0, 1], ids, vt, [2, 3, 4, 5, 6, 7], kt + vt, data, offs, families=fam)
    H.assert_vecs_equal(out, want_keys + vals, "all columns")
    out = oracle.kv_decode(t.prefix_len(), kt, dirs, [-1, 1], [6, 300], [vt[1], vt[5]], [0, 2], [cd.FLOAT64, cd.BYTES, cd.BYTES],
                           data, offs, families=fam)
    H.assert_vecs_equal(out, [vals[1], want_keys[1], vals[5]], "projection")
