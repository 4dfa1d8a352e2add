"""GPU parity: the device hash join (crgpu_dev_join_*) against
 (a) the reference's literal case tables (tests/golden), and
 (b) the oracle on seeded random inputs, compared as multisets, bit-exact."""
import numpy as np
import pytest

from cockroach_amd import coldata as cd
from tests import helpers as H
from tests import gpu_util as G

pytestmark = pytest.mark.gpu

CASES = H.join_cases()
ALL_JOIN_TYPES = ["INNER", "LEFT_OUTER", "RIGHT_OUTER", "FULL_OUTER", "LEFT_SEMI", "LEFT_ANTI", "INTERSECT_ALL",
                  "EXCEPT_ALL", "RIGHT_SEMI", "RIGHT_ANTI"]


@pytest.mark.parametrize("case", CASES, ids=[c["id"] for c in CASES])
def test_join_golden_case(case):
    expected = H.expected_join_rows(case)
    left = H.tuples_to_vecs(case["left"], case["left_types"])
    right = H.tuples_to_vecs(case["right"], case["right_types"])
    for chunks in (1, 3):
        rows = G.gpu_join(case["join_type"], left, right, case["left_eq"], case["right_eq"],
                          case["right_eq_cols_are_key"], probe_chunks=chunks if len(case["left"]) >= 3 else 1)
        H.assert_same_multiset(H.project_join_output(case, rows), expected, case["id"])


def _random_tables(rng, nl, nr, key_domain, null_frac, unique_right):
    if unique_right:
        rk = rng.permutation(max(key_domain, nr))[:nr].astype(np.int64)
    else:
        rk = rng.integers(0, key_domain, size=nr, dtype=np.int64)
    lk = rng.integers(0, key_domain * 2, size=nl, dtype=np.int64)
    lp = rng.random(nl) >= null_frac
    rp = rng.random(nr) >= null_frac
    left = [cd.Vec.from_numpy(cd.INT64, lk, lp), cd.Vec.from_numpy(cd.INT64, np.arange(nl, dtype=np.int64))]
    right = [cd.Vec.from_numpy(cd.INT64, rk, rp), cd.Vec.from_numpy(cd.FLOAT64, rng.standard_normal(nr))]
    return left, right


@pytest.mark.parametrize("jt", ALL_JOIN_TYPES)
@pytest.mark.parametrize("unique_right", [False, True])
def test_join_random_int_keys_vs_oracle(oracle, jt, unique_right):
    rng = np.random.default_rng(hash((jt, unique_right)) % 2**32)
    left, right = _random_tables(rng, 40_000, 6_000, 5_000 if not unique_right else 20_000, 0.05, unique_right)
    if jt in ("INTERSECT_ALL", "EXCEPT_ALL"):
        left, right = [left[0]], [right[0]]  # set-op joins: every column is an equality column
    want = oracle.hash_join(H.JOIN_TYPE[jt], left, right, [0], [0], unique_right)
    got_probe, got_right = G.gpu_join_vecs(jt, left, right, [0], [0], unique_right)
    got = np.concatenate([G.rows_as_sorted_array(got_probe), G.rows_as_sorted_array(got_right)]) \
        if len(got_probe) else G.rows_as_sorted_array(got_right)
    got = got[np.lexsort(got.T[::-1])]
    assert np.array_equal(got, G.rows_as_sorted_array(want)), jt


@pytest.mark.parametrize("jt", ["INNER", "LEFT_OUTER", "RIGHT_OUTER", "FULL_OUTER"])
@pytest.mark.parametrize("key_type", ["INT64", "INT32_nullable", "FLOAT64"])
@pytest.mark.parametrize("match_cache", ["1", "0"])
def test_join_duplicate_counts_around_the_match_cache(oracle, jt, key_type, match_cache, monkeypatch):
    """Build keys with 0..24 duplicates each: the probe's first pass hands the first 8 matches of a row to the pass that
    writes the pairs, rows with more walk their bucket again (join.hip, kChainCache); a probe wave mixes all of them.
    FLOAT64 keys take the generic walk (representative marks, chain_mark_kernel), one key has 156 rows."""
    monkeypatch.setenv("CRGPU_JOIN_MATCH_CACHE", match_cache)  # "0": the path probe chunks beyond 2^28 rows take
    rng = np.random.default_rng(5)
    rk = np.repeat(np.arange(1, 25 * 8 + 1), np.tile(np.arange(25), 8)).astype(np.int64)  # key k has (k-1) % 25 rows
    rk = np.concatenate([rk, np.full(150, 7, dtype=np.int64)])  # one bucket beyond the 64 entries that carry representative marks
    rng.shuffle(rk)
    lk = rng.integers(-5, 25 * 8 + 10, size=3000).astype(np.int64)
    nl, nr = len(lk), len(rk)
    if key_type == "INT64":
        lkey, rkey = cd.Vec.from_numpy(cd.INT64, lk), cd.Vec.from_numpy(cd.INT64, rk)
    elif key_type == "INT32_nullable":
        lkey = cd.Vec.from_numpy(cd.INT32, lk.astype(np.int32), rng.random(nl) >= 0.05)
        rkey = cd.Vec.from_numpy(cd.INT32, rk.astype(np.int32), rng.random(nr) >= 0.05)
    else:
        lkey, rkey = cd.Vec.from_numpy(cd.FLOAT64, lk * 0.5), cd.Vec.from_numpy(cd.FLOAT64, rk * 0.5)
    left = [lkey, cd.Vec.from_numpy(cd.INT64, np.arange(nl, dtype=np.int64))]
    right = [rkey, cd.Vec.from_numpy(cd.INT64, np.arange(nr, dtype=np.int64) + 10_000)]
    want = H.vecs_to_tuples(oracle.hash_join(H.JOIN_TYPE[jt], left, right, [0], [0], False))
    for chunks in (1, 2):
        got = G.gpu_join(jt, left, right, [0], [0], False, probe_chunks=chunks)
        H.assert_same_multiset(got, want, f"{jt} {key_type} chunks={chunks}")


@pytest.mark.parametrize("jt", ["INNER", "LEFT_OUTER", "FULL_OUTER", "LEFT_SEMI", "LEFT_ANTI", "RIGHT_SEMI"])
def test_join_multi_col_bytes_keys_vs_oracle(oracle, jt):
    rng = np.random.default_rng(11)
    nl, nr = 5_000, 1_500
    words = [b"", b"a", b"foo", b"foo1", b"x" * 31, b"y" * 40, b"z" * 100]

    def side(n):
        k1 = cd.Vec.from_values(cd.BYTES, [None if rng.random() < 0.03 else words[rng.integers(len(words))] for _ in range(n)])
        k2 = cd.Vec.from_numpy(cd.INT32, rng.integers(0, 6, size=n).astype(np.int32))
        k3 = cd.Vec.from_numpy(cd.FLOAT64, rng.choice([0.0, -0.0, np.nan, 1.5, 2.5], size=n))
        pay = cd.Vec.from_values(cd.BYTES, [b"p%d" % i for i in range(n)])
        return [k1, k2, k3, pay]

    left, right = side(nl), side(nr)
    right[1] = cd.Vec.from_numpy(cd.INT16, right[1].data.astype(np.int16))  # mixed-width integer equality
    want = H.vecs_to_tuples(oracle.hash_join(H.JOIN_TYPE[jt], left, right, [0, 1, 2], [0, 1, 2], False))
    got = G.gpu_join(jt, left, right, [0, 1, 2], [0, 1, 2], False, probe_chunks=2)
    H.assert_same_multiset(got, want, jt)


def test_join_packed_float_and_bool_keys(oracle):
    rng = np.random.default_rng(5)
    lf = cd.Vec.from_numpy(cd.FLOAT64, rng.choice([0.0, -0.0, np.nan, 1.0, 2.0, 3.5, -7.25], size=2000))
    rf = cd.Vec.from_numpy(cd.FLOAT64, np.array([-0.0, np.nan, 2.0, 3.5, 99.0]))
    rp = cd.Vec.from_numpy(cd.INT64, np.arange(5, dtype=np.int64) * 10)
    for jt in ("INNER", "LEFT_OUTER", "LEFT_SEMI", "LEFT_ANTI", "FULL_OUTER", "RIGHT_OUTER"):
        want = H.vecs_to_tuples(oracle.hash_join(H.JOIN_TYPE[jt], [lf], [rf, rp], [0], [0], True))
        got = G.gpu_join(jt, [lf], [rf, rp], [0], [0], True)
        H.assert_same_multiset(got, want, jt)
    lb = cd.Vec.from_values(cd.BOOL, [True, False, None, True])
    rb = cd.Vec.from_values(cd.BOOL, [True])
    want = H.vecs_to_tuples(oracle.hash_join(0, [lb], [rb], [0], [0], True))
    H.assert_same_multiset(G.gpu_join("INNER", [lb], [rb], [0], [0], True), want)


def test_join_packed_empty_key_sentinel(oracle):
    # the packed table's EMPTY sentinel (-1 as int64) must still be joinable
    lk = cd.Vec.from_numpy(cd.INT64, np.array([-1, 5, -1, 7, -2], dtype=np.int64))
    rk = cd.Vec.from_numpy(cd.INT64, np.array([7, -1, 100], dtype=np.int64))
    rv = cd.Vec.from_numpy(cd.INT64, np.array([70, -10, 1000], dtype=np.int64))
    for jt in ("INNER", "LEFT_OUTER", "LEFT_ANTI", "LEFT_SEMI"):
        want = H.vecs_to_tuples(oracle.hash_join(H.JOIN_TYPE[jt], [lk], [rk, rv], [0], [0], True))
        H.assert_same_multiset(G.gpu_join(jt, [lk], [rk, rv], [0], [0], True), want, jt)


def test_join_empty_sides(oracle):
    e = cd.Vec.from_numpy(cd.INT64, np.zeros(0, dtype=np.int64))
    a = cd.Vec.from_numpy(cd.INT64, np.array([1, 2, 3], dtype=np.int64))
    for jt in ALL_JOIN_TYPES:
        for left, right in (([e], [a]), ([a], [e]), ([e], [e])):
            want = H.vecs_to_tuples(oracle.hash_join(H.JOIN_TYPE[jt], left, right, [0], [0], False))
            H.assert_same_multiset(G.gpu_join(jt, left, right, [0], [0], False), want, f"{jt} {len(left[0])}x{len(right[0])}")


def test_join_packed_covering_large():
    """The bench shape (config J, scaled down): unique int64 build keys with one
    int64 payload, 100 % hit probes; checked against a direct numpy evaluation
    and through the size-independent checksum property sum(v_out) == sum(3*k+1)."""
    from cockroach_amd import device as dev
    rng = np.random.default_rng(42)
    nb, npr = 1 << 20, 1 << 22
    bk = rng.permutation(nb).astype(np.int64)
    bv = bk * 3 + 1
    pk = rng.integers(0, nb, size=npr, dtype=np.int64)
    pw = np.arange(npr, dtype=np.int64)
    j = dev.Join(0, [cd.INT64, cd.INT64], [cd.INT64, cd.INT64], [0], [0], True)
    j.build([dev.upload(cd.Vec.from_numpy(cd.INT64, bk)), dev.upload(cd.Vec.from_numpy(cd.INT64, bv))], nb)
    res = j.probe([dev.upload(cd.Vec.from_numpy(cd.INT64, pk)), dev.upload(cd.Vec.from_numpy(cd.INT64, pw))], npr)
    assert res.n == npr and res.ncols == 4
    out = dev.result_to_vecs(res)
    k_out, w_out, rk_out, v_out = (o.data for o in out)
    order = np.argsort(w_out, kind="stable")
    assert np.array_equal(w_out[order], pw)
    assert np.array_equal(k_out[order], pk)
    assert np.array_equal(rk_out[order], pk)
    assert np.array_equal(v_out[order], pk * 3 + 1)
    j.close()


@pytest.mark.parametrize("bitmap", ["1", "0", "prefilter"])
@pytest.mark.parametrize("two_cols", [False, True])
def test_join_packed_key_bitmap_selective_probe(oracle, bitmap, two_cols, monkeypatch):
    """A packed table beyond 2 MiB gets a key bitmap in front of it (join.hip, bloom_bit): a probe row whose bit is clear
    never reads the table.  Selective probes (5 % hits), then a chunk of hits only (which switches the bitmap off for
    INNER / LEFT_SEMI), with the key equal to the table's empty marker on both sides; two handles in a row."""
    monkeypatch.setenv("CRGPU_JOIN_BLOOM", "0" if bitmap == "0" else "1")
    if bitmap == "prefilter":  # the bitmap evaluated by a pass of its own, the probe kernel over the survivors (large chunks)
        monkeypatch.setenv("CRGPU_JOIN_PREFILTER_MIN", "1")
    rng = np.random.default_rng(17)
    nb, npr = 150_000, 400_000
    space = nb * 20
    bk = rng.permutation(space)[:nb].astype(np.int64)
    pk = rng.integers(0, space, size=npr, dtype=np.int64)
    pk[150_000:200_000] = rng.choice(bk, size=50_000)  # the 4th of 8 chunks: hits only
    if not two_cols:
        bk[0], pk[5], pk[-7] = -1, -1, -1
        lkeys, rkeys = [cd.Vec.from_numpy(cd.INT64, pk, rng.random(npr) >= 0.02)], [cd.Vec.from_numpy(cd.INT64, bk)]
    else:
        lkeys = [cd.Vec.from_numpy(cd.INT64, pk // 1000), cd.Vec.from_numpy(cd.INT32, (pk % 1000).astype(np.int32), rng.random(npr) >= 0.02)]
        rkeys = [cd.Vec.from_numpy(cd.INT64, bk // 1000), cd.Vec.from_numpy(cd.INT32, (bk % 1000).astype(np.int32))]
    nk = len(lkeys)
    left = lkeys + [cd.Vec.from_numpy(cd.INT64, np.arange(npr, dtype=np.int64))]
    right = rkeys + [cd.Vec.from_numpy(cd.FLOAT64, rng.standard_normal(nb))]
    eq = list(range(nk))
    for jt in ("INNER", "LEFT_SEMI", "LEFT_ANTI", "LEFT_OUTER"):
        want = G.rows_as_sorted_array(oracle.hash_join(H.JOIN_TYPE[jt], left, right, eq, eq, True))
        parts, got_right = G.gpu_join_vecs(jt, left, right, eq, eq, True, probe_chunks=8)
        assert len(got_right) == 0 or len(got_right[0]) == 0
        got = np.concatenate([G.rows_as_sorted_array(p) for p in parts])
        assert np.array_equal(got[np.lexsort(got.T[::-1])], want), (jt, bitmap, two_cols)


@pytest.mark.parametrize("keys", ["two_cols", "one_col_direct"])
@pytest.mark.parametrize("outputs", ["pairs", "fused_payload"])
def test_join_radix_first_key_bitmap_prefilter(oracle, keys, outputs, monkeypatch):
    """The radix path's prefilter (join.hip, key0_filter_kernel): a bitmap over the first key column's value range
    rejects probe rows before anything is scattered; the radix probe sees the survivors' compacted key columns and
    either their compacted payload column or their ORIGINAL row numbers (the pairs the later gathers use).  Selective
    chunks (5 % of the first-column values exist on the build side, and not every survivor matches on the second
    column), then a chunk of hits only, which switches the bitmap off; against the oracle, with the prefilter and
    without it."""
    monkeypatch.setenv("CRGPU_RADIX_MIN_BUILD", "1")
    monkeypatch.setenv("CRGPU_RADIX_MIN_PROBE", "1")
    monkeypatch.setenv("CRGPU_JOIN_PREFILTER_MIN", "1000")
    rng = np.random.default_rng(23)
    nb, npr = 20_000, 400_000
    if keys == "two_cols":
        parts = rng.permutation(400_000)[:5_000].astype(np.int64) + 7           # 5000 "green parts" of 400 000
        bk0 = np.repeat(parts, 4)
        bk1 = np.tile(np.arange(4, dtype=np.int64), 5_000) * 1000 + bk0 % 1000    # 4 suppliers per part
        pk0 = rng.integers(7, 400_007, npr, dtype=np.int64)
        pk1 = rng.integers(0, 5, npr, dtype=np.int64) * 1000 + pk0 % 1000         # every fifth survivor misses on column 2
        pk0[200_000:250_000] = rng.choice(parts, 50_000)                          # a chunk in which every row survives
        lkeys = [cd.Vec.from_numpy(cd.INT64, pk0), cd.Vec.from_numpy(cd.INT64, pk1)]
        rkeys = [cd.Vec.from_numpy(cd.INT64, bk0), cd.Vec.from_numpy(cd.INT64, bk1)]
    else:
        bk = (rng.permutation(60_000)[:nb].astype(np.int64)) + 1_000_000          # dense enough for direct-addressed tables
        pk = rng.integers(1_000_000 - 50_000, 1_000_000 + 600_000, npr, dtype=np.int64)
        pk[200_000:250_000] = rng.choice(bk, 50_000)
        lkeys, rkeys = [cd.Vec.from_numpy(cd.INT64, pk)], [cd.Vec.from_numpy(cd.INT64, bk)]
    nk = len(lkeys)
    left = lkeys + [cd.Vec.from_numpy(cd.INT64, np.arange(npr, dtype=np.int64))]
    if outputs == "pairs":  # two more probe columns (one nullable): gathered by pair afterwards
        left += [cd.Vec.from_numpy(cd.FLOAT64, rng.standard_normal(npr)), cd.Vec.from_numpy(cd.INT32, (np.arange(npr) % 97).astype(np.int32), rng.random(npr) > 0.1)]
    right = rkeys + [cd.Vec.from_numpy(cd.FLOAT64, rng.standard_normal(len(rkeys[0])))]
    eq = list(range(nk))
    for jt in ("INNER", "LEFT_SEMI"):
        want = G.rows_as_sorted_array(oracle.hash_join(H.JOIN_TYPE[jt], left, right, eq, eq, True))
        for on in ("1", "0"):
            monkeypatch.setenv("CRGPU_JOIN_KEY0_BITMAP", on)
            parts_out, _ = G.gpu_join_vecs(jt, left, right, eq, eq, True, probe_chunks=8)
            got = np.concatenate([G.rows_as_sorted_array(p) for p in parts_out])
            assert np.array_equal(got[np.lexsort(got.T[::-1])], want), (jt, keys, outputs, on)


@pytest.mark.parametrize("jt", ALL_JOIN_TYPES)
@pytest.mark.parametrize("table", ["packed", "packed_bitmap_prefilter", "radix", "chained_dups", "chained_bytes"])
def test_join_probe_selected_equals_deselect_then_probe(oracle, jt, table, monkeypatch):
    """crgpu_dev_join_probe_selected: a probe batch with a selection (a BOOL column, what the selection operators leave
    on a coldata.Batch) gives what the reference's joiner gives on that batch -- the join of the selected rows only,
    for every join type; checked on every table form (the packed table probes in place, the others copy the rows out)."""
    from cockroach_amd import device as dev
    rng = np.random.default_rng(hash((jt, table)) % 2**32)
    nl, nr = 60_000, 9_000
    unique = table in ("packed", "packed_bitmap_prefilter", "radix")
    if table == "radix":
        monkeypatch.setenv("CRGPU_RADIX_MIN_BUILD", "1")
        monkeypatch.setenv("CRGPU_RADIX_MIN_PROBE", "1")
    if table == "packed_bitmap_prefilter":
        nr = 150_000  # a table beyond 2 MiB: key bitmap
        monkeypatch.setenv("CRGPU_JOIN_PREFILTER_MIN", "1")
    if table == "chained_bytes":
        lk = cd.Vec.from_values(cd.BYTES, [None if rng.random() < 0.02 else b"k%d" % rng.integers(0, 3000) for _ in range(nl)])
        rk = cd.Vec.from_values(cd.BYTES, [b"k%d" % rng.integers(0, 6000) for _ in range(nr)])
        left, right = [lk, cd.Vec.from_numpy(cd.INT64, np.arange(nl, dtype=np.int64))], [rk, cd.Vec.from_numpy(cd.FLOAT64, rng.standard_normal(nr))]
    else:
        left, right = _random_tables(rng, nl, nr, nr * 2 if unique else nr // 3, 0.03, unique)
    if jt in ("INTERSECT_ALL", "EXCEPT_ALL"):
        left, right = [left[0]], [right[0]]
    keep = rng.random(nl) < 0.4
    sel_rows = np.nonzero(keep)[0].astype(np.int64)
    want = oracle.hash_join(H.JOIN_TYPE[jt], [v.take(sel_rows) for v in left], right, [0], [0], unique)
    j = dev.Join(H.JOIN_TYPE[jt], [v.type for v in left], [v.type for v in right], [0], [0], unique)
    rcols = [dev.upload(v) for v in right]
    j.build(rcols, nr)
    lcols = [dev.upload(v) for v in left]
    kcol = dev.upload(cd.Vec.from_numpy(cd.BOOL, keep))
    got = H.vecs_to_tuples(dev.result_to_vecs(j.probe_selected(lcols, kcol, nl)))
    got += H.vecs_to_tuples(dev.result_to_vecs(j.emit_right()))
    j.close()
    H.assert_same_multiset(got, H.vecs_to_tuples(want), f"{jt} {table}")


# ---------------------------------------------------------------------------
# radix-partitioned path (csrc/radix.hip): forced on for small inputs through the
# CRGPU_RADIX_MIN_BUILD / CRGPU_RADIX_MIN_PROBE environment overrides
# ---------------------------------------------------------------------------
@pytest.fixture(params=["partitioned", "direct", "partitioned-chd", "direct-chd", "partitioned-dense", "direct-dense"])
def radix_env(request, monkeypatch):
    """partitioned / direct: the probe side is radix-partitioned first / looked up row by row; -chd: the sub-partition
    tables are perfect-hash tables (CRGPU_RADIX_TABLE=chd, one slot read per lookup) instead of 3-way cuckoo tables;
    -dense: the library's default choice -- direct-addressed tables (key - min is the address, nothing hashed) when the
    build side's one integer key column is dense, hashed tables otherwise; the other forms pin the hashed tables
    (CRGPU_JOIN_DIRECT=0)."""
    monkeypatch.setenv("CRGPU_RADIX_MIN_BUILD", "1")
    monkeypatch.setenv("CRGPU_RADIX_MIN_PROBE", "1" if request.param.startswith("partitioned") else str(1 << 40))
    if request.param.endswith("-chd"):
        monkeypatch.setenv("CRGPU_RADIX_TABLE", "chd")
    if not request.param.endswith("-dense"):
        monkeypatch.setenv("CRGPU_JOIN_DIRECT", "0")
    return request.param


@pytest.mark.parametrize("case", CASES, ids=[c["id"] for c in CASES])
def test_join_golden_case_radix(case, radix_env):
    expected = H.expected_join_rows(case)
    left = H.tuples_to_vecs(case["left"], case["left_types"])
    right = H.tuples_to_vecs(case["right"], case["right_types"])
    rows = G.gpu_join(case["join_type"], left, right, case["left_eq"], case["right_eq"], case["right_eq_cols_are_key"])
    H.assert_same_multiset(H.project_join_output(case, rows), expected, case["id"])


@pytest.mark.parametrize("jt", ["INNER", "LEFT_OUTER", "LEFT_SEMI", "LEFT_ANTI"])
@pytest.mark.parametrize("shape", ["i64_cover", "i32_nullable", "f64_key", "bytes_payload"])
def test_join_radix_vs_oracle(oracle, radix_env, jt, shape):
    rng = np.random.default_rng(hash((jt, shape)) % 2**32)
    nl, nr = 60_000, 9_000
    if shape == "i64_cover":
        rk = cd.Vec.from_numpy(cd.INT64, rng.permutation(30_000)[:nr].astype(np.int64) - 5_000)
        right = [rk, cd.Vec.from_numpy(cd.INT64, rk.data * 3 + 1)]
        left = [cd.Vec.from_numpy(cd.INT64, rng.integers(-6_000, 26_000, nl, dtype=np.int64)),
                cd.Vec.from_numpy(cd.INT64, np.arange(nl, dtype=np.int64))]
    elif shape == "i32_nullable":
        rk = cd.Vec.from_numpy(cd.INT32, rng.permutation(30_000)[:nr].astype(np.int32), rng.random(nr) > 0.05)
        right = [rk, cd.Vec.from_numpy(cd.FLOAT64, rng.standard_normal(nr), rng.random(nr) > 0.1)]
        left = [cd.Vec.from_numpy(cd.INT16, rng.integers(0, 30_000, nl).astype(np.int16), rng.random(nl) > 0.05),
                cd.Vec.from_numpy(cd.INT32, np.arange(nl, dtype=np.int32))]
    elif shape == "f64_key":
        vals = np.concatenate([np.array([0.0, np.nan, 1.5, -2.25, 1e300]), rng.standard_normal(nr - 5)])
        right = [cd.Vec.from_numpy(cd.FLOAT64, vals), cd.Vec.from_numpy(cd.INT64, np.arange(nr, dtype=np.int64))]
        lk = rng.choice(np.concatenate([vals, np.array([-0.0, 7.0, 8.0])]), nl)
        left = [cd.Vec.from_numpy(cd.FLOAT64, lk), cd.Vec.from_numpy(cd.BOOL, rng.integers(0, 2, nl).astype(np.uint8))]
    else:
        rk = cd.Vec.from_numpy(cd.INT64, rng.permutation(1 << 40 if False else 30_000)[:nr].astype(np.int64) * 977)
        right = [rk, cd.Vec.from_values(cd.BYTES, [b"r%d" % i * (1 + i % 5) for i in range(nr)])]
        left = [cd.Vec.from_numpy(cd.INT64, rng.integers(0, 30_000, nl, dtype=np.int64) * 977),
                cd.Vec.from_values(cd.BYTES, [None if i % 13 == 0 else b"l%d" % i for i in range(nl)]),
                cd.Vec.from_numpy(cd.FLOAT64, rng.standard_normal(nl))]
    want = oracle.hash_join(H.JOIN_TYPE[jt], left, right, [0], [0], True)
    got = G.gpu_join(jt, left, right, [0], [0], True, probe_chunks=2)
    H.assert_same_multiset(got, H.vecs_to_tuples(want), f"{jt} {shape} {radix_env}")


@pytest.mark.parametrize("jt", ["INNER", "LEFT_OUTER", "LEFT_SEMI", "LEFT_ANTI"])
def test_join_direct_addressed_tables_at_real_thresholds(oracle, jt):
    """2^22 dense build keys (a shuffled range with holes, negative lower bound, one key held twice) x 2^23 probe rows
    with keys inside the range, in its holes, outside it on both sides and NULL: the library picks direct-addressed
    tables by itself (no environment overrides).  Against numpy row by row."""
    from cockroach_amd import device as dev
    rng = np.random.default_rng(77)
    nb, npr = 1 << 22, 1 << 23
    lo = -123_456
    keys = rng.permutation(nb + nb // 8)[:nb].astype(np.int64) + lo  # 1/9 of the range are holes
    bv = keys * 5 + 2
    pk = rng.integers(lo - 50_000, lo + nb + nb // 8 + 50_000, size=npr, dtype=np.int64)
    pvalid = rng.random(npr) > 0.01
    pw = np.arange(npr, dtype=np.int64)
    j = dev.Join(H.JOIN_TYPE[jt], [cd.INT64, cd.INT64], [cd.INT64, cd.INT64], [0], [0], True)
    dev.prof_enable(True)
    dev.prof_reset()
    j.build([dev.upload(cd.Vec.from_numpy(cd.INT64, keys)), dev.upload(cd.Vec.from_numpy(cd.INT64, bv))], nb)
    res = dev.result_to_vecs(j.probe([dev.upload(cd.Vec.from_numpy(cd.INT64, pk, pvalid)), dev.upload(cd.Vec.from_numpy(cd.INT64, pw))], npr))
    assert dev.prof_get("radix_build_table")[1] == 1 and dev.prof_get("radix_build_split")[1] == 1
    dev.prof_enable(False)
    j.close()
    present = np.zeros(nb + nb // 8, dtype=bool)
    present[keys - lo] = True
    inr = pvalid & (pk >= lo) & (pk < lo + nb + nb // 8)
    hit = inr.copy()
    hit[inr] = present[pk[inr] - lo]
    order = np.argsort(res[1].data, kind="stable")  # by w = the probe row id
    if jt in ("INNER", "LEFT_SEMI"):
        rows = np.nonzero(hit)[0]
    elif jt == "LEFT_ANTI":
        rows = np.nonzero(~hit)[0]
    else:
        rows = np.arange(npr)
    assert len(res[0]) == len(rows)
    assert np.array_equal(res[1].data[order], pw[rows])
    kv = res[0].present()[order] if res[0].valid is not None else np.ones(len(rows), dtype=bool)
    assert np.array_equal(kv, pvalid[rows])                                    # the probe key column: NULL where it was NULL ...
    assert np.array_equal(res[0].data[order][kv], pk[rows][pvalid[rows]])       # ... its own value everywhere else, out of range or not
    if jt == "INNER":
        assert np.array_equal(res[2].data[order], pk[rows]) and np.array_equal(res[3].data[order], pk[rows] * 5 + 2)
    if jt == "LEFT_OUTER":
        bp = res[3].present()[order]
        assert np.array_equal(bp, hit) and np.array_equal(res[3].data[order][bp], pk[hit] * 5 + 2)
        assert np.array_equal(res[2].present()[order], hit) and np.array_equal(res[2].data[order][bp], pk[hit])


@pytest.mark.parametrize("tables", ["hashed", "dense"])
@pytest.mark.parametrize("skew", ["zipf", "burst"])
def test_join_probe_cells_fall_back_to_histogram(monkeypatch, tables, skew):
    """The probe side is scattered into fixed-capacity (partition, chunk) cells without a histogram pass; skewed keys
    must take the histogram form instead.  zipf: the 65536-row sample sees the hot partitions and the cells are not
    tried; burst: a key repeated 3000 times inside one 2^20-row chunk is invisible to the sample, overflows its cell,
    the pass raises its flag and the probe is partitioned again with the histogram (radix.hip: radix_partition_cells ->
    cells_failed).  Real thresholds (2^22 build rows), results against numpy row by row."""
    from cockroach_amd import device as dev
    if tables == "hashed":
        monkeypatch.setenv("CRGPU_JOIN_DIRECT", "0")
    rng = np.random.default_rng(5)
    nb, npr = 1 << 22, (1 << 22) + (1 << 21)
    bk = rng.permutation(nb).astype(np.int64)
    bv = bk * 7 - 3
    if skew == "zipf":
        r = np.exp(rng.random(npr) * np.log(nb)).astype(np.int64).clip(1, nb - 1)
        pk = (r * 0x9E3779B1) % nb
    else:
        pk = rng.integers(0, nb, size=npr, dtype=np.int64)
        pk[(1 << 20) + 1000:(1 << 20) + 4000] = 123_457
    pw = np.arange(npr, dtype=np.int64)
    j = dev.Join(0, [cd.INT64, cd.INT64], [cd.INT64, cd.INT64], [0], [0], True, out_cols=[1, 3])
    j.build([dev.upload(cd.Vec.from_numpy(cd.INT64, bk)), dev.upload(cd.Vec.from_numpy(cd.INT64, bv))], nb)
    dev.prof_enable(True)
    dev.prof_reset()
    pcols = [dev.upload(cd.Vec.from_numpy(cd.INT64, pk)), dev.upload(cd.Vec.from_numpy(cd.INT64, pw))]
    for chunk in range(2):  # the second chunk goes straight to the histogram form (cells_failed is remembered)
        res = j.probe(pcols, npr)
        assert dev.prof_get("radix_probe_hist")[1] == chunk + 1, "expected the histogram form of the probe-side partition"
        w_out, v_out = (o.data for o in dev.result_to_vecs(res))
        assert res.n == npr
        order = np.argsort(w_out, kind="stable")
        assert np.array_equal(w_out[order], pw) and np.array_equal(v_out[order], pk * 7 - 3)
    scatters = dev.prof_get("radix_probe_scatter")[1]
    assert scatters == (3 if skew == "burst" else 2), "burst: the cells pass ran once and was thrown away"
    dev.prof_enable(False)
    j.close()
    # and with the cells switched off through the environment (read per probe) the histogram form runs without a first try
    monkeypatch.setenv("CRGPU_RADIX_CELLS", "0")
    j = dev.Join(0, [cd.INT64, cd.INT64], [cd.INT64, cd.INT64], [0], [0], True, out_cols=[1, 3])
    j.build([dev.upload(cd.Vec.from_numpy(cd.INT64, bk)), dev.upload(cd.Vec.from_numpy(cd.INT64, bv))], nb)
    dev.prof_enable(True)
    dev.prof_reset()
    res = j.probe(pcols, npr)
    assert dev.prof_get("radix_probe_hist")[1] == 1 and dev.prof_get("radix_probe_scatter")[1] == 1 and res.n == npr
    dev.prof_enable(False)
    j.close()


def test_join_radix_duplicate_build_keys_fall_back(oracle, radix_env):
    """A build side that is not unique (declared key or not) may overflow a sub-partition
    of the radix table; the joiner must then fall back to the classic table, not lose rows."""
    rng = np.random.default_rng(8)
    rk = cd.Vec.from_numpy(cd.INT64, np.concatenate([np.full(5000, 7, dtype=np.int64), np.arange(100, dtype=np.int64) + 100]))
    rv = cd.Vec.from_numpy(cd.INT64, np.arange(5100, dtype=np.int64))
    lk = cd.Vec.from_numpy(cd.INT64, rng.integers(0, 220, 3000, dtype=np.int64))
    for jt in ("LEFT_SEMI", "LEFT_ANTI"):
        want = H.vecs_to_tuples(oracle.hash_join(H.JOIN_TYPE[jt], [lk], [rk, rv], [0], [0], False))
        H.assert_same_multiset(G.gpu_join(jt, [lk], [rk, rv], [0], [0], False), want, jt)


def test_join_radix_large_checksums(radix_env):
    """2^20 x 2^22 rows through the radix path: many chunks, 64 rows per partition and
    chunk, XCD-mapped probe; verified row by row against numpy."""
    from cockroach_amd import device as dev
    rng = np.random.default_rng(42)
    nb, npr = 1 << 20, (1 << 22) + 12345
    bk = rng.permutation(nb).astype(np.int64) * 3 - 1000
    bv = bk * 3 + 1
    pk = (rng.integers(0, nb + nb // 4, size=npr, dtype=np.int64)) * 3 - 1000  # 80 % hit
    pw = np.arange(npr, dtype=np.int64)
    j = dev.Join(0, [cd.INT64, cd.INT64], [cd.INT64, cd.INT64], [0], [0], True, out_cols=[1, 3])
    j.build([dev.upload(cd.Vec.from_numpy(cd.INT64, bk)), dev.upload(cd.Vec.from_numpy(cd.INT64, bv))], nb)
    res = j.probe([dev.upload(cd.Vec.from_numpy(cd.INT64, pk)), dev.upload(cd.Vec.from_numpy(cd.INT64, pw))], npr)
    hit = (pk + 1000) // 3 < nb
    assert res.n == int(hit.sum())
    w_out, v_out = (o.data for o in dev.result_to_vecs(res))
    order = np.argsort(w_out, kind="stable")
    assert np.array_equal(w_out[order], pw[hit])
    assert np.array_equal(v_out[order], pk[hit] * 3 + 1)
    j.close()


# ---------------------------------------------------------------------------
# several integer key columns packed into one key word (KeyPack, csrc/device.h): packed table, radix table and the
# exact chained table; the reference compares the columns one by one (hashtable_tmpl.go:96-223)
# ---------------------------------------------------------------------------
def _multi_col_tables(rng, nl, nr, shape, unique_right):
    if shape == "two_i64":      # ranges of 13 + 9 bits, negative values, probe values outside the build range
        a = rng.integers(-3000, 3000, nr, dtype=np.int64)
        b = rng.integers(100, 400, nr, dtype=np.int64)
        la = rng.integers(-4000, 4000, nl, dtype=np.int64)
        lb = rng.integers(50, 450, nl, dtype=np.int64)
        rkeys = [cd.Vec.from_numpy(cd.INT64, a, rng.random(nr) > 0.03), cd.Vec.from_numpy(cd.INT64, b)]
        lkeys = [cd.Vec.from_numpy(cd.INT64, la, rng.random(nl) > 0.03), cd.Vec.from_numpy(cd.INT64, lb)]
    elif shape == "mixed_widths_bool":  # INT16 x INT32 / INT64, plus a BOOL column (hash_utils.eg.go:363-366, :69-73)
        a = rng.integers(-200, 200, nr).astype(np.int16)
        b = rng.integers(0, 50, nr).astype(np.int32)
        c = rng.integers(0, 2, nr).astype(np.uint8)
        rkeys = [cd.Vec.from_numpy(cd.INT16, a), cd.Vec.from_numpy(cd.INT32, b, rng.random(nr) > 0.05), cd.Vec.from_numpy(cd.BOOL, c)]
        lkeys = [cd.Vec.from_numpy(cd.INT64, rng.integers(-250, 250, nl, dtype=np.int64)),
                 cd.Vec.from_numpy(cd.INT16, rng.integers(0, 60, nl).astype(np.int16)),
                 cd.Vec.from_numpy(cd.BOOL, rng.integers(0, 2, nl).astype(np.uint8), rng.random(nl) > 0.05)]
    elif shape == "too_wide":   # 2 x 40 bits do not fit one word: the generic chained table must take over
        a = rng.integers(0, 300, nr, dtype=np.int64) << 32
        b = rng.integers(0, 20, nr, dtype=np.int64) << 36
        rkeys = [cd.Vec.from_numpy(cd.INT64, a), cd.Vec.from_numpy(cd.INT64, b)]
        lkeys = [cd.Vec.from_numpy(cd.INT64, rng.integers(0, 320, nl, dtype=np.int64) << 32),
                 cd.Vec.from_numpy(cd.INT64, rng.integers(0, 22, nl, dtype=np.int64) << 36)]
    else:                       # "full_word": 32 + 32 bits use the whole key word, including the all-ones pattern of the EMPTY marker
        a = np.concatenate([np.array([0xFFFFFFFF, 0, 0xFFFFFFFF], dtype=np.int64), rng.integers(0, 1 << 32, nr - 3, dtype=np.int64)])
        b = np.concatenate([np.array([0xFFFFFFFF, 0xFFFFFFFF, 0], dtype=np.int64), rng.integers(0, 1 << 32, nr - 3, dtype=np.int64)])
        rkeys = [cd.Vec.from_numpy(cd.INT64, a), cd.Vec.from_numpy(cd.INT64, b)]
        pick = rng.integers(0, nr, nl)
        la, lb = a[pick].copy(), b[pick].copy()
        miss = rng.random(nl) < 0.3
        la[miss] ^= 1
        lkeys = [cd.Vec.from_numpy(cd.INT64, la), cd.Vec.from_numpy(cd.INT64, lb)]
    if unique_right:  # keep the first row of every build key
        rows = list(zip(*[k.to_values() for k in rkeys]))
        seen, keep = set(), []
        for i, r in enumerate(rows):
            if r not in seen:
                seen.add(r)
                keep.append(i)
        rkeys = [k.take(np.array(keep)) for k in rkeys]
        nr = len(keep)
    right = rkeys + [cd.Vec.from_numpy(cd.INT64, np.arange(nr, dtype=np.int64) * 7)]
    left = lkeys + [cd.Vec.from_numpy(cd.FLOAT64, rng.standard_normal(nl))]
    return left, right, list(range(len(lkeys)))


@pytest.mark.parametrize("jt", ALL_JOIN_TYPES)
@pytest.mark.parametrize("shape", ["two_i64", "mixed_widths_bool", "too_wide", "full_word"])
@pytest.mark.parametrize("unique_right", [False, True])
def test_join_multi_col_int_keys_vs_oracle(oracle, jt, shape, unique_right):
    rng = np.random.default_rng(hash((jt, shape, unique_right)) % 2**32)
    left, right, eq = _multi_col_tables(rng, 30_000, 5_000, shape, unique_right)
    if jt in ("INTERSECT_ALL", "EXCEPT_ALL"):
        left, right = left[:len(eq)], right[:len(eq)]
    want = H.vecs_to_tuples(oracle.hash_join(H.JOIN_TYPE[jt], left, right, eq, eq, unique_right))
    got = G.gpu_join(jt, left, right, eq, eq, unique_right, probe_chunks=2)
    H.assert_same_multiset(got, want, f"{jt} {shape} unique={unique_right}")


@pytest.mark.parametrize("jt", ["INNER", "LEFT_OUTER", "LEFT_SEMI", "LEFT_ANTI"])
@pytest.mark.parametrize("shape", ["two_i64", "mixed_widths_bool", "full_word"])
def test_join_multi_col_int_keys_radix_vs_oracle(oracle, radix_env, jt, shape):
    rng = np.random.default_rng(hash((jt, shape, "radix")) % 2**32)
    left, right, eq = _multi_col_tables(rng, 50_000, 8_000, shape, True)
    want = H.vecs_to_tuples(oracle.hash_join(H.JOIN_TYPE[jt], left, right, eq, eq, True))
    got = G.gpu_join(jt, left, right, eq, eq, True, probe_chunks=2)
    H.assert_same_multiset(got, want, f"{jt} {shape} {radix_env}")


def test_join_two_col_keys_large_radix_checksums():
    """Q9's shape: (l_partkey, l_suppkey) = (ps_partkey, ps_suppkey), a two-column key that is unique on the build side,
    2^22 build x 2^24 probe rows through the radix path; verified row by row against numpy."""
    from cockroach_amd import device as dev
    rng = np.random.default_rng(9)
    nb, npr = 1 << 22, 1 << 24
    part = np.arange(nb, dtype=np.int64) // 4 + 1
    supp = (part * 7 + np.arange(nb, dtype=np.int64) % 4 * 2501) % 10_000 + 1
    cost = rng.integers(100, 100_000, nb).astype(np.float64) / 100.0
    pick = rng.integers(0, nb, npr)
    lp, ls = part[pick].copy(), supp[pick].copy()
    miss = rng.random(npr) < 0.25
    ls[miss] += 10_001  # outside the build side's range of the second column
    lw = np.arange(npr, dtype=np.int64)
    j = dev.Join(0, [cd.INT64] * 3, [cd.INT64, cd.INT64, cd.FLOAT64], [0, 1], [0, 1], True, out_cols=[2, 5])
    j.build([dev.upload(cd.Vec.from_numpy(cd.INT64, part)), dev.upload(cd.Vec.from_numpy(cd.INT64, supp)),
             dev.upload(cd.Vec.from_numpy(cd.FLOAT64, cost))], nb)
    res = j.probe([dev.upload(cd.Vec.from_numpy(cd.INT64, lp)), dev.upload(cd.Vec.from_numpy(cd.INT64, ls)),
                   dev.upload(cd.Vec.from_numpy(cd.INT64, lw))], npr)
    assert res.n == int((~miss).sum())
    w_out, c_out = (o.data for o in dev.result_to_vecs(res))
    order = np.argsort(w_out, kind="stable")
    assert np.array_equal(w_out[order], lw[~miss])
    assert np.array_equal(c_out[order], cost[pick][~miss])
    j.close()


# ---------------------------------------------------------------------------
# duplicate build keys on the radix path (radix_build_dup / radix_probe_dup_*, csrc/radix.hip): INNER and LEFT OUTER
# over a non-distinct build side; the reference walks its Same chains (hashjoiner_tmpl.go:16-337)
# ---------------------------------------------------------------------------
@pytest.mark.parametrize("jt", ["INNER", "LEFT_OUTER"])
@pytest.mark.parametrize("shape", ["i64_cover", "nullable_rowid", "two_cols", "bytes_payload", "empty_marker_key", "f64_key"])
def test_join_radix_duplicate_build_keys_vs_oracle(oracle, radix_env, jt, shape):
    rng = np.random.default_rng(hash((jt, shape, "dup")) % 2**32)
    nl, nr = 40_000, 12_000
    leq = req = [0]
    if shape == "i64_cover":
        rk = rng.integers(0, 3000, nr, dtype=np.int64) - 500  # ~4 rows per key, some keys with many more below
        rk[:300] = 77
        right = [cd.Vec.from_numpy(cd.INT64, rk), cd.Vec.from_numpy(cd.INT64, np.arange(nr, dtype=np.int64) * 3 + 1)]
        left = [cd.Vec.from_numpy(cd.INT64, rng.integers(-700, 3200, nl, dtype=np.int64)), cd.Vec.from_numpy(cd.INT64, np.arange(nl, dtype=np.int64))]
    elif shape == "nullable_rowid":  # NULL keys on both sides, two build payload columns (row-id table + gathers)
        right = [cd.Vec.from_numpy(cd.INT32, rng.integers(0, 2000, nr).astype(np.int32), rng.random(nr) > 0.05),
                 cd.Vec.from_numpy(cd.FLOAT64, rng.standard_normal(nr), rng.random(nr) > 0.1),
                 cd.Vec.from_numpy(cd.INT16, rng.integers(-9, 9, nr).astype(np.int16))]
        left = [cd.Vec.from_numpy(cd.INT16, rng.integers(0, 2500, nl).astype(np.int16), rng.random(nl) > 0.05),
                cd.Vec.from_numpy(cd.BOOL, rng.integers(0, 2, nl).astype(np.uint8))]
    elif shape == "two_cols":
        leq = req = [0, 1]
        right = [cd.Vec.from_numpy(cd.INT64, rng.integers(-40, 40, nr, dtype=np.int64)), cd.Vec.from_numpy(cd.INT32, rng.integers(0, 30, nr).astype(np.int32), rng.random(nr) > 0.02),
                 cd.Vec.from_numpy(cd.FLOAT64, rng.standard_normal(nr))]
        left = [cd.Vec.from_numpy(cd.INT16, rng.integers(-50, 50, nl).astype(np.int16)), cd.Vec.from_numpy(cd.INT64, rng.integers(-5, 40, nl, dtype=np.int64), rng.random(nl) > 0.02),
                cd.Vec.from_numpy(cd.INT64, np.arange(nl, dtype=np.int64))]
    elif shape == "bytes_payload":
        right = [cd.Vec.from_numpy(cd.INT64, rng.integers(0, 4000, nr, dtype=np.int64) * 977),
                 cd.Vec.from_values(cd.BYTES, [None if i % 11 == 0 else b"r%d" % i * (1 + i % 4) for i in range(nr)])]
        left = [cd.Vec.from_numpy(cd.INT64, rng.integers(0, 4400, nl, dtype=np.int64) * 977),
                cd.Vec.from_values(cd.BYTES, [None if i % 13 == 0 else b"l%d" % i for i in range(nl)])]
    elif shape == "empty_marker_key":  # -1 packs to the tables' EMPTY marker: it lives in the side slot, with its duplicates
        rk = rng.integers(-3, 3, nr, dtype=np.int64)
        right = [cd.Vec.from_numpy(cd.INT64, rk), cd.Vec.from_numpy(cd.INT64, np.arange(nr, dtype=np.int64))]
        nl = 300
        left = [cd.Vec.from_numpy(cd.INT64, rng.integers(-4, 4, nl, dtype=np.int64)), cd.Vec.from_numpy(cd.INT64, np.arange(nl, dtype=np.int64))]
    else:
        vals = np.array([0.0, np.nan, 1.5, -2.25, 1e300, 3.0, 4.0, 5.0])
        right = [cd.Vec.from_numpy(cd.FLOAT64, rng.choice(vals, nr)), cd.Vec.from_numpy(cd.INT64, np.arange(nr, dtype=np.int64))]
        nl = 200
        left = [cd.Vec.from_numpy(cd.FLOAT64, rng.choice(np.concatenate([vals, np.array([-0.0, 7.0])]), nl)), cd.Vec.from_numpy(cd.INT64, np.arange(nl, dtype=np.int64))]
    want = oracle.hash_join(H.JOIN_TYPE[jt], left, right, leq, req, False)
    got_probe, got_right = G.gpu_join_vecs(jt, left, right, leq, req, False)
    if any(v.type == cd.BYTES for v in want):
        H.assert_same_multiset(H.vecs_to_tuples(got_probe), H.vecs_to_tuples(want), f"{jt} {shape} {radix_env}")
    else:
        assert np.array_equal(G.rows_as_sorted_array(got_probe), G.rows_as_sorted_array(want)), f"{jt} {shape} {radix_env}"
    assert len(got_right[0]) == 0


def test_join_radix_duplicates_skewed_key_falls_back(oracle, radix_env):
    """One key with more rows than a sub-partition holds: the grouped table cannot be built, the chained table takes over."""
    rng = np.random.default_rng(5)
    rk = np.concatenate([np.full(6000, 7, dtype=np.int64), rng.integers(100, 400, 3000, dtype=np.int64)])
    right = [cd.Vec.from_numpy(cd.INT64, rk), cd.Vec.from_numpy(cd.INT64, np.arange(len(rk), dtype=np.int64))]
    left = [cd.Vec.from_numpy(cd.INT64, rng.integers(0, 420, 2000, dtype=np.int64)), cd.Vec.from_numpy(cd.INT64, np.arange(2000, dtype=np.int64))]
    for jt in ("INNER", "LEFT_OUTER"):
        want = oracle.hash_join(H.JOIN_TYPE[jt], left, right, [0], [0], False)
        got_probe, _ = G.gpu_join_vecs(jt, left, right, [0], [0], False)
        assert np.array_equal(G.rows_as_sorted_array(got_probe), G.rows_as_sorted_array(want)), jt


def test_join_radix_duplicates_large_checksums():
    """2^22 build rows (4 per key on average) x 2^23 probe rows through the grouped radix table at its real thresholds:
    output row count and per-column checksums against numpy (sum over probe rows of the per-key aggregates)."""
    from cockroach_amd import device as dev
    rng = np.random.default_rng(21)
    nb, npr, nkeys = 1 << 22, 1 << 23, 1 << 20
    bk = rng.integers(0, nkeys, nb, dtype=np.int64) * 5 - 77
    bv = np.arange(nb, dtype=np.int64)
    pk = rng.integers(0, nkeys + nkeys // 4, npr, dtype=np.int64) * 5 - 77
    pw = np.arange(npr, dtype=np.int64)
    cnt = np.bincount((bk + 77) // 5, minlength=nkeys + nkeys // 4).astype(np.int64)
    vsum = np.bincount((bk + 77) // 5, weights=bv.astype(np.float64), minlength=nkeys + nkeys // 4)
    kidx = (pk + 77) // 5
    for jt, name in ((0, "INNER"), (1, "LEFT_OUTER")):
        j = dev.Join(jt, [cd.INT64, cd.INT64], [cd.INT64, cd.INT64], [0], [0], False, out_cols=[1, 3])
        j.build([dev.upload(cd.Vec.from_numpy(cd.INT64, bk)), dev.upload(cd.Vec.from_numpy(cd.INT64, bv))], nb)
        res = j.probe([dev.upload(cd.Vec.from_numpy(cd.INT64, pk)), dev.upload(cd.Vec.from_numpy(cd.INT64, pw))], npr)
        want_rows = int(cnt[kidx].sum()) + (int((cnt[kidx] == 0).sum()) if jt == 1 else 0)
        assert int(res.n) == want_rows, name
        w_out, v_out = dev.result_to_vecs(res)
        mult = np.maximum(cnt[kidx], 1) if jt == 1 else cnt[kidx]
        assert int(w_out.data.sum()) == int((pw * mult).sum()), name
        pres = v_out.present()
        assert int(pres.sum()) == int(cnt[kidx].sum())
        assert float(v_out.data[pres].astype(np.float64).sum()) == pytest.approx(float(vsum[kidx].sum()), rel=1e-12)
        # every (w, v) pair is a real match: v's build key equals w's probe key
        sample = rng.integers(0, int(res.n), 200_000)
        sp = pres[sample]
        assert np.array_equal(bk[v_out.data[sample][sp]], pk[w_out.data[sample][sp]])
        j.close()
