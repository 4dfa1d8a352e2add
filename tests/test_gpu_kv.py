"""GPU parity of the KV fetch decode (crgpu_dev_decode_kv) against the oracle's
cFetcher restatement on the same encoded batches, bit-exact, plus the reference's
own KV dump (docs/tech-notes/encoding.md) and the error paths."""
import json
import os

import numpy as np
import pytest

from cockroach_amd import _lib
from cockroach_amd import coldata as cd
from cockroach_amd import device as dev
from tests import helpers as H

pytestmark = pytest.mark.gpu


def gpu_decode(prefix_len, kt, kd, ko, ids, vt, vo, out_types, data, offs, families=None):
    """families: an oracle.KvFamilies -> IndexFetchSpec.MaxKeysPerRow / MaxFamilyID / FamilyDefaultColumns."""
    kw = {}
    if families is not None:
        kw = dict(max_keys_per_row=len(families.family_ids), max_family_id=families.family_ids[-1],
                  family_defaults=[(f, c) for f, c in zip(families.family_ids, families.default_col_ids) if c != 0])
    d = dev.Decoder(prefix_len, kt, kd, ko, ids, vt, vo, out_types, **kw)
    pad = (-len(data)) % 16
    buf = dev.DeviceBuffer.from_numpy(np.concatenate([data, np.zeros(pad + 16, np.uint8)]))
    try:
        return dev.result_to_vecs(d.decode(buf, offs))
    finally:
        d.close()


@pytest.mark.parametrize("dirs", [(0, 0, 0, 0), (1, 1, 1, 1), (0, 1, 0, 1)])
@pytest.mark.parametrize("batch_bytes", [4 << 10, 1 << 20])
def test_decode_matches_oracle(oracle, dirs, batch_bytes):
    rng = np.random.default_rng(sum(dirs) * 7 + batch_bytes % 13)
    n = 20_000
    kt = [cd.INT64, cd.BYTES, cd.FLOAT64, cd.INT32]
    vt = [cd.INT64, cd.FLOAT64, cd.BYTES, cd.BOOL, cd.INT16, cd.BYTES]
    ids = [5, 6, 9, 10, 40, 300]
    keys, vals = H.kv_random_rows(rng, n, kt, vt)
    t = oracle.KvTable(106, 1, kt, list(dirs), ids, vt)
    data, offs = oracle.kv_encode(t, keys, vals, max_batch_bytes=batch_bytes, explicit_null_every=3)
    full = (t.prefix_len(), kt, list(dirs), [0, 1, 2, 3], ids, vt, [4, 5, 6, 7, 8, 9], kt + vt)
    H.assert_vecs_equal(gpu_decode(*full, data, offs), oracle.kv_decode(*full, data, offs), "all columns")
    proj = (t.prefix_len(), kt, list(dirs), [-1, 2, -1, 0], [6, 10, 300], [vt[1], vt[3], vt[5]], [1, 4, 3],
            [cd.INT32, cd.FLOAT64, cd.BYTES, cd.BYTES, cd.BOOL])
    got = gpu_decode(*proj, data, offs)
    H.assert_vecs_equal(got, oracle.kv_decode(*proj, data, offs), "projection")
    H.assert_vecs_equal(got, [H.kv_key_canonical(keys[3]), vals[1], H.kv_key_canonical(keys[1]), vals[5], vals[3]],
                        "projection vs source rows")


def test_decode_reference_kv_dump(oracle):
    with open(os.path.join(H.GOLDEN, "encoding_vectors.json")) as f:
        vec = json.load(f)
    data = b""
    for i, kv in enumerate(vec["kv_values"]):
        key = oracle.kv_enc("uvarint_asc", 51) + oracle.kv_enc("uvarint_asc", 1) + oracle.kv_enc("varint_asc", i + 1) + b"\x88"
        mvcc = key + b"\x00" + (1489504989617188491).to_bytes(8, "big") + b"\x09"
        value = bytes.fromhex(kv["value"])
        data += len(value).to_bytes(4, "little") + len(mvcc).to_bytes(4, "little") + mvcc + value
    arr = np.frombuffer(data, dtype=np.uint8)
    out = gpu_decode(2, [cd.INT64], [0], [0], [2], [cd.BYTES], [1], [cd.INT64, cd.BYTES], arr, [0, len(arr)])
    assert out[0].to_values() == [1, 2, 3, 4, 5]
    assert out[1].to_values() == [None if kv["owner"] is None else bytes.fromhex(kv["owner"]) for kv in vec["kv_values"]]


def test_decode_lineitem_shape_large(oracle):
    """TPC-H lineitem-like rows (2 int key columns, 14 value columns: ints/dates, 4 floats, 6 strings),
    2e5 rows in ~64 KiB batches; only the Q1 columns are fetched."""
    rng = np.random.default_rng(3)
    n = 200_000
    kt, kd = [cd.INT64, cd.INT64], [0, 0]
    vt = [cd.INT64, cd.INT64, cd.FLOAT64, cd.FLOAT64, cd.FLOAT64, cd.FLOAT64, cd.BYTES, cd.BYTES, cd.INT64, cd.INT64,
          cd.INT64, cd.BYTES, cd.BYTES, cd.BYTES]
    ids = list(range(3, 17))
    keys = [cd.Vec.from_numpy(cd.INT64, np.arange(n, dtype=np.int64) // 4 + 1),
            cd.Vec.from_numpy(cd.INT64, np.arange(n, dtype=np.int64) % 4 + 1)]
    vals = []
    for t in vt:
        if t == cd.INT64:
            vals.append(cd.Vec.from_numpy(cd.INT64, rng.integers(0, 200_000, n, dtype=np.int64)))
        elif t == cd.FLOAT64:
            vals.append(cd.Vec.from_numpy(cd.FLOAT64, np.round(rng.random(n) * 1e5, 2)))
        else:
            w = int(rng.integers(1, 26))
            vals.append(cd.Vec.bytes_from_fixed(rng.integers(65, 91, (n, w)).astype(np.uint8)))
    t = oracle.KvTable(110, 1, kt, kd, ids, vt)
    data, offs = oracle.kv_encode(t, keys, vals, max_batch_bytes=64 << 10)
    assert len(offs) > 300
    # l_quantity(5), l_extendedprice(6), l_discount(7), l_tax(8), l_returnflag(9), l_linestatus(10), l_shipdate(11)
    sel = [2, 3, 4, 5, 6, 7, 8]
    spec = (t.prefix_len(), kt, kd, [-1, -1], [ids[i] for i in sel], [vt[i] for i in sel], list(range(7)), [vt[i] for i in sel])
    got = gpu_decode(*spec, data, offs)
    H.assert_vecs_equal(got, [vals[i] for i in sel], "lineitem Q1 columns")


def test_decode_errors_and_empty(oracle):
    t = oracle.KvTable(60, 1, [cd.INT64], [0], [2], [cd.INT64])
    k = cd.Vec.from_numpy(cd.INT64, np.arange(50, dtype=np.int64))
    data, offs = oracle.kv_encode(t, [k], [k])
    args = (t.prefix_len(), [cd.INT64], [0], [0], [2], [cd.INT64], [1], [cd.INT64, cd.INT64])
    assert gpu_decode(*args, data, offs)[1].to_values() == list(range(50))
    out = gpu_decode(*args, np.zeros(0, np.uint8), [0, 0])
    assert len(out) == 2 and len(out[0]) == 0
    with pytest.raises(_lib.CrgpuError) as e:
        gpu_decode(*args, data[:-3], [0, len(data) - 3])
    assert e.value.code == _lib.ERR_INVALID and "truncated" in str(e.value)
    bad = data.copy()
    rec0_key_len = int.from_bytes(bytes(bad[4:8]), "little")
    bad[8 + rec0_key_len + 4] = 3  # a BYTES-tagged value in family 0: the pre-family row sentinel, no value expected
    out = gpu_decode(*args, bad, offs)  # (cfetcher.go:1192-1196): the column is NULL, as in the oracle
    H.assert_vecs_equal(out, oracle.kv_decode(*args, bad, offs), "sentinel value")
    assert out[1].to_values()[:3] == [None, 1, 2]
    bad[8 + rec0_key_len + 5] = 0xff  # a TUPLE whose first tag never ends
    bad[8 + rec0_key_len + 4] = 10
    with pytest.raises(_lib.CrgpuError) as e:
        gpu_decode(*args, bad, offs)
    assert "value column decode error" in str(e.value) and "row 0" in str(e.value)
    with pytest.raises(_lib.CrgpuError):  # value column ids must increase
        dev.Decoder(2, [cd.INT64], [0], [0], [5, 3], [cd.INT64, cd.INT64], [1, 2], [cd.INT64] * 3)


def test_decode_segmented_walk_misleading_values(oracle):
    """Batches are walked in 32 KiB segments whose first record is speculated and then verified.  Here many
    values are byte-for-byte copies of encoded records of the SAME table (same key prefix, plausible lengths
    and timestamps): a segment that begins inside such a value finds a convincing but wrong first record, and
    only the verification against the previous segment's walk puts it right."""
    rng = np.random.default_rng(11)
    kt, kd, vt, ids = [cd.INT64], [0], [cd.INT64, cd.BYTES], [2, 3]
    t = oracle.KvTable(77, 1, kt, kd, ids, vt)
    n0 = 600
    k0 = cd.Vec.from_numpy(cd.INT64, np.arange(n0, dtype=np.int64))
    inner, _ = oracle.kv_encode(t, [k0], [k0, cd.Vec.from_values(cd.BYTES, [b"x" * int(rng.integers(0, 40)) for _ in range(n0)])])
    blob = bytes(inner)
    assert len(blob) > 20_000
    n = 300
    payload = []
    for i in range(n):
        r = rng.random()
        if r < 0.4:
            a = int(rng.integers(0, len(blob) // 2))
            payload.append(blob[a:a + int(rng.integers(9_000, 70_000))])  # tens of fake records, often across a segment boundary
        elif r < 0.5:
            payload.append(None)
        else:
            payload.append(bytes(rng.integers(0, 256, size=int(rng.integers(0, 200)), dtype=np.uint8)))
    keys = [cd.Vec.from_numpy(cd.INT64, np.arange(n, dtype=np.int64) * 3 - 100)]
    vals = [cd.Vec.from_numpy(cd.INT64, rng.integers(-5, 5, size=n, dtype=np.int64)), cd.Vec.from_values(cd.BYTES, payload)]
    for batch_bytes in (1 << 20, 64 << 20):
        data, offs = oracle.kv_encode(t, keys, vals, max_batch_bytes=batch_bytes)
        assert len(data) > 40 * 32 * 1024 // 10
        spec = (t.prefix_len(), kt, kd, [0], ids, vt, [1, 2], kt + vt)
        dev.prof_enable(True)
        dev.prof_reset()
        got = gpu_decode(*spec, data, offs)
        walks = dev.prof_get("kv_walk")[1]
        dev.prof_enable(False)
        assert walks > 2, "expected at least one segment to be re-walked after a wrong speculation"
        H.assert_vecs_equal(got, oracle.kv_decode(*spec, data, offs), "misleading values")
        H.assert_vecs_equal(got, [keys[0], vals[0], vals[1]], "misleading values vs source rows")
    # truncation inside a later segment of a large batch is still reported
    with pytest.raises(_lib.CrgpuError) as e:
        gpu_decode(*spec, data[:-5], [0, len(data) - 5])
    assert "truncated" in str(e.value)


@pytest.mark.parametrize("walk_twice", [False, True])
def test_decode_one_large_batch_many_segments(oracle, walk_twice, monkeypatch):
    if walk_twice:  # the form a segment of more than 2048 records takes: the record offsets are written by a second walk
        monkeypatch.setenv("CRGPU_KV_WALK_TWICE", "1")
    rng = np.random.default_rng(12)
    n = 150_000
    kt, kd, vt, ids = [cd.INT64, cd.BYTES], [0, 1], [cd.FLOAT64, cd.BYTES, cd.INT32], [4, 8, 9]
    keys, vals = H.kv_random_rows(rng, n, kt, vt)
    t = oracle.KvTable(300, 2, kt, kd, ids, vt)
    data, offs = oracle.kv_encode(t, keys, vals, max_batch_bytes=1 << 30)
    assert len(offs) == 2 and len(data) > 100 * 32 * 1024
    spec = (t.prefix_len(), kt, kd, [0, 1], ids, vt, [2, 3, 4], kt + vt)
    H.assert_vecs_equal(gpu_decode(*spec, data, offs), oracle.kv_decode(*spec, data, offs), "one batch, many segments")


# ---------------------------------------------------------------- rows of several KVs: column families
def test_decode_reference_kv_dump_with_families(oracle):
    """docs/tech-notes/encoding.md:169-235 (tests/golden/encoding_vectors.json: kv_families_dump): two families, the second
    one a single-column family whose KVs carry a BYTES-tagged value and are missing where the column is NULL."""
    from tests.test_oracle_kv import _family_key, _record
    with open(os.path.join(H.GOLDEN, "encoding_vectors.json")) as f:
        dump = json.load(f)["kv_families_dump"]
    data = b""
    for kv in dump["kvs"]:
        data += _record(_family_key(oracle, dump["table_id"], dump["index_id"], kv["id"], kv["family"]), bytes.fromhex(kv["value"]))
    arr = np.frombuffer(data, dtype=np.uint8)
    fam = oracle.KvFamilies([0, 1], [0, 2])
    out = gpu_decode(2, [cd.INT64], [0], [0], [2], [cd.BYTES], [1], [cd.INT64, cd.BYTES], arr, [0, len(arr)], families=fam)
    assert out[0].to_values() == [1, 2, 3, 4, 5]
    assert out[1].to_values() == [None if o is None else o.encode() for o in dump["owner"]]


@pytest.mark.parametrize("layout", ["tuple+singles", "two tuples", "sparse ids"])
@pytest.mark.parametrize("mvcc_header_every,batch_bytes", [(0, 4 << 10), (3, 1 << 20)])
def test_decode_families_matches_oracle(oracle, layout, mvcc_header_every, batch_bytes):
    """Seeded tables with 2-4 column families (TUPLE families, INT / FLOAT / BYTES / BOOL single-column families, family
    ids of one and several bytes), NULLs that suppress whole KVs, values behind MVCC extended headers: bit-exact against
    the oracle's cFetcher state machine, all columns and a projection (a BYTES key column: the re-parsing BYTES pass)."""
    from tests.test_oracle_kv import FAM_LAYOUTS
    fam_ids, defaults, val_family = FAM_LAYOUTS[layout]
    rng = np.random.default_rng(len(layout) * 3 + mvcc_header_every)
    n = 30_000
    kt, dirs = [cd.INT64, cd.BYTES], [0, 1]
    vt = [cd.INT64, cd.FLOAT64, cd.BYTES, cd.BOOL, cd.INT16, cd.BYTES]
    ids = [5, 6, 9, 10, 40, 300]
    keys, vals = H.kv_random_rows(rng, n, kt, vt)
    keys[0] = cd.Vec.from_numpy(cd.INT64, rng.permutation(n * 3)[:n].astype(np.int64) - n)
    t = oracle.KvTable(106, 1, kt, dirs, ids, vt)
    fam = oracle.KvFamilies(fam_ids, defaults, val_family, mvcc_header_every)
    data, offs = oracle.kv_encode(t, keys, vals, max_batch_bytes=batch_bytes, explicit_null_every=3, families=fam)
    full = (t.prefix_len(), kt, dirs, [0, 1], ids, vt, [2, 3, 4, 5, 6, 7], kt + vt)
    got = gpu_decode(*full, data, offs, families=fam)
    H.assert_vecs_equal(got, oracle.kv_decode(*full, data, offs, families=fam), "all columns")
    H.assert_vecs_equal(got, [H.kv_key_canonical(k) for k in keys] + vals, "all columns vs source rows")
    proj = (t.prefix_len(), kt, dirs, [-1, -1], [6, 10, 300], [vt[1], vt[3], vt[5]], [0, 2, 1], [cd.FLOAT64, cd.BYTES, cd.BOOL])
    got = gpu_decode(*proj, data, offs, families=fam)  # no BYTES key column fetched: the copy pass
    H.assert_vecs_equal(got, oracle.kv_decode(*proj, data, offs, families=fam), "projection")
    H.assert_vecs_equal(got, [vals[1], vals[5], vals[3]], "projection vs source rows")


def test_decode_families_errors(oracle):
    """A single-column value of a family that has no default column is an error ("single entry value with no default
    column", cfetcher.go:1207-1212); a wrong tag for the column's type too (Value.GetInt, data.go:781-790)."""
    t = oracle.KvTable(60, 1, [cd.INT64], [0], [2, 3], [cd.INT64, cd.INT64])
    k = cd.Vec.from_numpy(cd.INT64, np.arange(40, dtype=np.int64))
    fam = oracle.KvFamilies([0, 1], [0, 3], [0, 1])
    data, offs = oracle.kv_encode(t, [k], [k, k], families=fam)
    args = (t.prefix_len(), [cd.INT64], [0], [0], [2, 3], [cd.INT64, cd.INT64], [1, 2], [cd.INT64] * 3)
    H.assert_vecs_equal(gpu_decode(*args, data, offs, families=fam), [k, k, k], "two families")
    nodefault = oracle.KvFamilies([0, 1], [0, 0], [0, 1])
    with pytest.raises(_lib.CrgpuError) as e:
        gpu_decode(*args, data, offs, families=nodefault)
    assert "no default column" in str(e.value) and "row 0" in str(e.value)
    with pytest.raises(oracle.KvDecodeError) as eo:
        oracle.kv_decode(*args, data, offs, families=nodefault)
    assert eo.value.code == -4
    bad = data.copy()
    kl0 = int.from_bytes(bytes(bad[4:8]), "little")
    vl0 = int.from_bytes(bytes(bad[0:4]), "little")
    rec1 = 8 + kl0 + vl0
    kl1 = int.from_bytes(bytes(bad[rec1 + 4:rec1 + 8]), "little")
    assert bad[rec1 + 8 + kl1 + 4] == 1  # INT-tagged single value of family 1
    bad[rec1 + 8 + kl1 + 4] = 2          # ... now FLOAT-tagged: not what an INT column takes
    with pytest.raises(_lib.CrgpuError) as e:
        gpu_decode(*args, bad, offs, families=fam)
    assert "value column decode error" in str(e.value)
    with pytest.raises(oracle.KvDecodeError) as eo:
        oracle.kv_decode(*args, bad, offs, families=fam)
    assert eo.value.code == -5
