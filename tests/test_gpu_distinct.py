"""GPU parity of the unordered DISTINCT (cockroach_amd/csrc/distinct.hip) through the C-ABI: the reference's literal
cases (tests/golden/distinct_cases.json, ordered comparison as in distinct_test.go) and seeded random inputs against
the oracle, fed in one call and in several (the key store and the index carry over), plus size-independent properties
at 2^26 rows."""
import numpy as np
import pytest

from cockroach_amd import coldata as cd
from tests import helpers as H
from tests.test_oracle_distinct import CASES, ordered_equal

pytestmark = pytest.mark.gpu


def gpu_distinct(cols, distinct_cols, nulls_are_distinct=False, chunks=1, error_on_dup=False):
    from cockroach_amd import device as dev
    d = dev.Distinct([v.type for v in cols], distinct_cols, nulls_are_distinct, error_on_dup)
    n = len(cols[0])
    bounds = np.linspace(0, n, chunks + 1).astype(np.int64)
    rows = []
    for s, e in zip(bounds[:-1], bounds[1:]):
        chunk = [v.slice(int(s), int(e)) for v in cols]
        out = dev.result_to_vecs(d.consume([dev.upload(v) for v in chunk], int(e - s)))
        rows += H.vecs_to_tuples(out)
    d.close()
    return rows


@pytest.mark.parametrize("case", CASES, ids=[c["id"] for c in CASES])
def test_distinct_reference_cases(case):
    from cockroach_amd import _lib
    cols = H.tuples_to_vecs(case["tuples"], case["types"])
    for chunks in (1, 2, len(case["tuples"])):
        if case["error_on_dup"] and not case["no_error"]:
            with pytest.raises(_lib.CrgpuError) as e:
                gpu_distinct(cols, case["distinct_cols"], case["nulls_are_distinct"], chunks, error_on_dup=True)
            assert e.value.code == _lib.ERR_DUPLICATE
            continue
        got = gpu_distinct(cols, case["distinct_cols"], case["nulls_are_distinct"], chunks, error_on_dup=bool(case["error_on_dup"]))
        ordered_equal(got, case["expected"], f"{case['id']} chunks={chunks}")


@pytest.mark.parametrize("nulls_are_distinct", [False, True])
@pytest.mark.parametrize("nkeys", [30, 5000, 400_000])
def test_distinct_random_vs_oracle(oracle, nkeys, nulls_are_distinct):
    rng = np.random.default_rng(nkeys + int(nulls_are_distinct))
    n = 300_000
    a = rng.integers(0, nkeys, size=n, dtype=np.int64)
    f = rng.choice(np.array([0.0, -0.0, 1.5, float("nan"), -2.0, 1e300]), size=n)
    b = rng.integers(0, 2, size=n).astype(np.uint8)
    i16 = rng.integers(-3, 3, size=n).astype(np.int16)
    s = [bytes([65 + int(x)]) * int(y) for x, y in zip(rng.integers(0, 3, size=n), rng.integers(0, 40, size=n))]
    cols = [cd.Vec.from_numpy(cd.INT64, a, rng.random(n) > 0.05), cd.Vec.from_numpy(cd.FLOAT64, f, rng.random(n) > 0.05),
            cd.Vec.from_values(cd.BYTES, s), cd.Vec.from_numpy(cd.BOOL, b), cd.Vec.from_numpy(cd.INT16, i16, rng.random(n) > 0.5),
            cd.Vec.from_numpy(cd.INT64, np.arange(n, dtype=np.int64))]
    for dc in ([0], [0, 1, 2], [2, 4, 3, 1]):
        want, _ = oracle.unordered_distinct(cols, dc, nulls_are_distinct)
        want = H.vecs_to_tuples(want)
        for chunks in (1, 7):
            got = gpu_distinct(cols, dc, nulls_are_distinct, chunks)
            ordered_equal(got, want, f"dc={dc} chunks={chunks}")


def test_distinct_reset_and_empty():
    from cockroach_amd import device as dev
    d = dev.Distinct([cd.INT64], [0])
    v = cd.Vec.from_values(cd.INT64, [3, 3, 4])
    assert dev.result_to_vecs(d.consume([dev.upload(v)], 3))[0].to_values() == [3, 4]
    assert int(d.consume([dev.upload(v)], 3).n) == 0
    assert int(d.consume([dev.upload(cd.Vec.from_values(cd.INT64, []))], 0).n) == 0
    d.reset()
    assert dev.result_to_vecs(d.consume([dev.upload(v)], 3))[0].to_values() == [3, 4]
    d.close()


@pytest.mark.parametrize("nkeys", [1000, 3_000_000, 1 << 26])
def test_distinct_full_size_invariants(nkeys):
    """2^26 rows in HBM: the output of the two halves fed one after the other holds every key exactly once, each with
    the payload of its FIRST occurrence (the payload is the row number, so first occurrence = per-key minimum)."""
    import torch
    from cockroach_amd import device as dev
    from tests.gpu_util import wrap_torch
    n = 1 << 26
    dv = torch.device("cuda", 0)
    g = torch.Generator(device=dv)
    g.manual_seed(nkeys)
    k = torch.randperm(n, generator=g, device=dv, dtype=torch.int64) * 7 - 5 if nkeys == n else \
        torch.randint(0, nkeys, (n,), generator=g, device=dv, dtype=torch.int64) * 11 - 3
    w = torch.arange(n, device=dv, dtype=torch.int64)
    torch.cuda.synchronize()
    d = dev.Distinct([cd.INT64, cd.INT64], [0])
    outs = []
    h = n // 2
    for lo, hi in ((0, h), (h, n)):
        res = d.consume([dev.from_torch(cd.INT64, k[lo:hi]), dev.from_torch(cd.INT64, w[lo:hi])], hi - lo)
        m = int(res.n)
        rc = dev.result_columns(res)
        ok, ow = wrap_torch(rc[0], m, torch.int64, dv), wrap_torch(rc[1], m, torch.int64, dv)
        outs.append((ok.clone(), ow.clone()))
        assert bool((ow[1:] > ow[:-1]).all()), "output is not in input order"
    d.close()
    ok = torch.cat([o[0] for o in outs])
    ow = torch.cat([o[1] for o in outs])
    uk, inv = torch.unique(k, return_inverse=True)
    assert ok.numel() == uk.numel()
    first = torch.full((uk.numel(),), n, device=dv, dtype=torch.int64).scatter_reduce(0, inv, w, reduce="amin")
    order = torch.argsort(ok)
    assert torch.equal(ok[order], uk) and torch.equal(ow[order], first)


def test_distinct_bulk_path_vs_oracle(oracle):
    """A fresh table fed >= 2^23 mostly distinct INT64 keys takes the bulk path (radix partition, first occurrence of every
    key found in LDS, no hash index); the second batch (old and new keys) forces the index to be rebuilt from the hash
    words.  Ordered comparison with the oracle, payload = row number, so a wrong 'first' row would show."""
    from cockroach_amd import device as dev
    rng = np.random.default_rng(41)
    n1, n2 = 1 << 23, 200_000
    base = rng.permutation(n1).astype(np.int64)[: n1 * 5 // 8] * 1_000_003 - 11
    base[:4] = [-1, 0, np.iinfo(np.int64).min, np.iinfo(np.int64).max]
    k1 = np.concatenate([base, rng.choice(base, n1 - base.size)])  # 3/8 of the rows repeat a key, some keys several times
    rng.shuffle(k1)
    k2 = np.concatenate([rng.choice(base, n2 // 2), np.arange(n2 // 2, dtype=np.int64) * 1_000_003 + 5])
    rng.shuffle(k2)
    keys = np.concatenate([k1, k2])
    cols = [cd.Vec.from_numpy(cd.INT64, keys), cd.Vec.from_numpy(cd.INT64, np.arange(n1 + n2, dtype=np.int64))]
    d = dev.Distinct([cd.INT64, cd.INT64], [0])
    dev.prof_enable(True)
    dev.prof_reset()
    out1 = dev.result_to_vecs(d.consume([dev.upload(v.slice(0, n1)) for v in cols], n1))
    assert dev.prof_get("distinct_bulk")[1] == 1, "the bulk path did not run"
    out2 = dev.result_to_vecs(d.consume([dev.upload(v.slice(n1, n1 + n2)) for v in cols], n2))
    assert dev.prof_get("distinct_reindex")[1] == 1
    dev.prof_enable(False)
    d.close()
    want, _ = oracle.unordered_distinct(cols, [0], False)
    for c in range(2):
        got = np.concatenate([out1[c].data, out2[c].data])
        assert np.array_equal(got, want[c].data)


@pytest.mark.parametrize("first", ["all_distinct", "few_repeats"])
def test_distinct_bulk_first_batch_handed_out_from_the_key_store(oracle, first):
    """A first batch whose rows are all emitted (or nearly: fewer than half repeat, so only the emitted rows join the
    store): the distinct column of the output is not a buffer of its own but the key store's new rows, and the hash
    words of the stored keys are only written when the second batch forces the index to be built.  The second batch
    repeats old keys and brings new ones; a third grows the store (its memory may move).  Ordered comparison with the
    oracle, every result copied before the next call."""
    from cockroach_amd import device as dev
    rng = np.random.default_rng(47)
    n1, n2, n3 = 1 << 23, 300_000, 1 << 23
    k1 = rng.permutation(n1).astype(np.int64) * 1_000_003 - 11
    if first == "few_repeats":
        k1[rng.choice(n1, 2000, replace=False)] = k1[:2000]
    k2 = np.concatenate([rng.choice(k1, n2 // 2), np.arange(n2 // 2, dtype=np.int64) * 1_000_003 + 5])
    rng.shuffle(k2)
    k3 = np.concatenate([rng.choice(k1, n3 // 2), np.arange(n3 // 2, dtype=np.int64) * 1_000_003 + 7])
    rng.shuffle(k3)
    keys = np.concatenate([k1, k2, k3])
    cols = [cd.Vec.from_numpy(cd.INT64, np.arange(keys.size, dtype=np.int64)), cd.Vec.from_numpy(cd.INT64, keys)]
    d = dev.Distinct([cd.INT64, cd.INT64], [1])
    dev.prof_enable(True)
    dev.prof_reset()
    outs, lo = [], 0
    for n in (n1, n2, n3):
        outs.append(dev.result_to_vecs(d.consume([dev.upload(v.slice(lo, lo + n)) for v in cols], n)))
        lo += n
    assert dev.prof_get("distinct_bulk")[1] == 1 and dev.prof_get("distinct_reindex")[1] >= 1
    dev.prof_enable(False)
    d.close()
    want, _ = oracle.unordered_distinct(cols, [1], False)
    for c in range(2):
        assert np.array_equal(np.concatenate([o[c].data for o in outs]), want[c].data)


def test_distinct_bulk_path_row_count_no_power_of_two(oracle):
    """10^7 rows: ten regions per radix partition (ceil(n / 2^20), any number: a multiply-shift of hash bits
    picks the region), three quarters of the keys distinct; ordered comparison with the oracle."""
    from cockroach_amd import device as dev
    rng = np.random.default_rng(43)
    n = 10_000_000
    base = rng.permutation(n).astype(np.int64)[: n * 3 // 4] * 1_000_003 - 11
    keys = np.concatenate([base, rng.choice(base, n - base.size)])
    rng.shuffle(keys)
    cols = [cd.Vec.from_numpy(cd.INT64, keys), cd.Vec.from_numpy(cd.INT64, np.arange(n, dtype=np.int64))]
    d = dev.Distinct([cd.INT64, cd.INT64], [0])
    dev.prof_enable(True)
    dev.prof_reset()
    out = dev.result_to_vecs(d.consume([dev.upload(v) for v in cols], n))
    assert dev.prof_get("distinct_bulk")[1] == 1, "the bulk path did not run"
    dev.prof_enable(False)
    d.close()
    want, _ = oracle.unordered_distinct(cols, [0], False)
    for c in range(2):
        assert np.array_equal(out[c].data, want[c].data)


@pytest.mark.parametrize("nkeys", [30_000, 150_000])
def test_distinct_bulk_part_form_vs_oracle(oracle, nkeys):
    """Mid cardinality (tens to thousands of keys per radix partition): the bulk path without the level-2 split; only the
    emitted rows join the key store, the second batch goes through the rebuilt index.  A third case: so many keys per
    partition that the LDS index overflows although the sample is read as mid cardinality -- cannot happen with uniform
    keys, so it is forced with keys that all fall into few partitions... not constructible from outside; the overflow
    branch is covered by the aggregator's twin (test_agg_bulk_part_form_skew_and_overflow)."""
    from cockroach_amd import device as dev
    rng = np.random.default_rng(nkeys)
    n1, n2 = 1 << 23, 100_000
    base = rng.permutation(nkeys * 4)[:nkeys].astype(np.int64) * 7_919 - 3
    base[:4] = [-1, 0, np.iinfo(np.int64).min, np.iinfo(np.int64).max]
    k1 = base[rng.integers(0, nkeys, size=n1)]
    k2 = np.concatenate([rng.choice(base, n2 // 2), np.arange(n2 // 2, dtype=np.int64) * 7_919 + 1])
    rng.shuffle(k2)
    cols = [cd.Vec.from_numpy(cd.INT64, np.concatenate([k1, k2])), cd.Vec.from_numpy(cd.INT64, np.arange(n1 + n2, dtype=np.int64))]
    d = dev.Distinct([cd.INT64, cd.INT64], [0])
    dev.prof_enable(True)
    dev.prof_reset()
    out1 = dev.result_to_vecs(d.consume([dev.upload(v.slice(0, n1)) for v in cols], n1))
    assert dev.prof_get("distinct_bulk_part")[1] == 1, "the part form of the bulk path did not run"
    out2 = dev.result_to_vecs(d.consume([dev.upload(v.slice(n1, n1 + n2)) for v in cols], n2))
    dev.prof_enable(False)
    d.close()
    want, _ = oracle.unordered_distinct(cols, [0], False)
    for c in range(2):
        assert np.array_equal(np.concatenate([out1[c].data, out2[c].data]), want[c].data)


def test_distinct_bulk_path_key_with_all_set_validity_bitmap():
    """A distinct column that carries a validity bitmap without a NULL (a join's or an aggregation's output) takes the
    bulk path like one without; a single NULL keeps the batch on the index path.  All three give the same rows for the
    non-NULL keys, in input order."""
    import torch
    from cockroach_amd import device as dev
    from tests.gpu_util import wrap_torch
    n = (1 << 23) + 3
    dv = torch.device("cuda", 0)
    g = torch.Generator(device=dv)
    g.manual_seed(5)
    k = torch.randint(0, n // 2, (n,), generator=g, device=dv, dtype=torch.int64) * 13 - 9
    w = torch.arange(n, device=dv, dtype=torch.int64)
    pres = np.ones(n, dtype=bool)
    all_set = torch.from_numpy(np.packbits(pres, bitorder="little")).to(dv)
    pres[n - 2] = False
    one_null = torch.from_numpy(np.packbits(pres, bitorder="little")).to(dv)
    torch.cuda.synchronize()
    outs = []
    dev.prof_enable(True)
    for valid, want_bulk in ((None, 1), (all_set, 1), (one_null, 0)):
        d = dev.Distinct([cd.INT64, cd.INT64], [0])
        dev.prof_reset()
        res = d.consume([dev.from_torch(cd.INT64, k, valid=valid), dev.from_torch(cd.INT64, w)], n)
        assert dev.prof_get("distinct_bulk")[1] == want_bulk
        m = int(res.n)
        rc = dev.result_columns(res)
        outs.append((wrap_torch(rc[0], m, torch.int64, dv).clone(), wrap_torch(rc[1], m, torch.int64, dv).clone()))
        d.close()
    dev.prof_enable(False)
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    uk, inv = torch.unique(k, return_inverse=True)
    first = torch.full((uk.numel(),), n, device=dv, dtype=torch.int64).scatter_reduce(0, inv, w, reduce="amin")
    assert torch.equal(torch.sort(outs[0][1]).values, torch.sort(first).values) and bool((outs[0][1][1:] > outs[0][1][:-1]).all())
    # the NULL row (position n - 2) is a key of its own on the index path: one more row unless its key was seen before
    keep = outs[2][1] != n - 2
    assert int((~keep).sum()) == 1
    seen_before = bool((k[: n - 2] == k[n - 2]).any())
    later = (k[n - 1] == k[n - 2]) and not seen_before  # the row after it would then be its key's first occurrence
    if not later:
        want_w = outs[0][1] if seen_before else outs[0][1][outs[0][1] != n - 2]
        assert torch.equal(outs[2][1][keep], want_w)


def test_distinct_bytes_keys_whole_batch_append_with_offset_base(oracle):
    """Mostly-new batches join the key store whole.  Here the BYTES key column's offsets do not start at 0 (a slice of
    a larger buffer, as a decode or join result can be) and the batches contain duplicates of keys from the same and
    from earlier batches; a later lookup must still find every stored key (the index is rebuilt on growth)."""
    import torch
    from cockroach_amd import device as dev
    rng = np.random.default_rng(123)
    dv = torch.device("cuda", 0)
    d = dev.Distinct([cd.BYTES, cd.INT64], [0])
    seen, want_rows, got_rows = set(), [], []
    serial = 0
    for b in range(6):
        n = 20_000
        vals = []
        for i in range(n):
            if rng.random() < 0.15 and serial > 10:
                k = int(rng.integers(0, serial))  # a key seen before (this or an earlier batch)
            else:
                k = serial
                serial += 1
            vals.append(b"key-%d-" % k + b"x" * (k % 9))
        payload = np.arange(n, dtype=np.int64) + b * n
        lens = np.array([len(v) for v in vals], dtype=np.int64)
        junk = 7 + b
        offs = (np.concatenate([[0], np.cumsum(lens)]) + junk).astype(np.int32)
        data = np.frombuffer(b"#" * junk + b"".join(vals) + b"#" * 16, dtype=np.uint8).copy()
        tk = torch.from_numpy(data).to(dv)
        to = torch.from_numpy(offs).to(dv)
        tp = torch.from_numpy(payload).to(dv)
        torch.cuda.synchronize()
        res = d.consume([dev.from_torch(cd.BYTES, tk, offsets=to), dev.from_torch(cd.INT64, tp)], n)
        out = dev.result_to_vecs(res)
        got_rows += list(zip(out[0].to_values(), out[1].to_values()))
        for v, p in zip(vals, payload.tolist()):
            if v not in seen:
                seen.add(v)
                want_rows.append((v, p))
    d.close()
    assert got_rows == want_rows


@pytest.mark.parametrize("nkeys", [30, 5000, 150_000])
def test_distinct_lds_filter_vs_oracle(oracle, nkeys):
    """One INT64 distinct column without NULLs, batches of >= 2^16 rows: the resolve kernel's LDS filter
    (distinct.hip, distinct_resolve_kernel<LDSF>) drops rows whose key the workgroup has already sent through the index
    in an earlier tile.  Output order and first occurrences against the oracle, in one call and in three (the later
    calls meet keys of the store); the filter's own "no key" value is among the keys."""
    rng = np.random.default_rng(nkeys)
    n = 400_000
    a = rng.integers(0, nkeys, size=n, dtype=np.int64) * 1_000_003 - 17
    a[rng.integers(0, n, 50)] = np.int64(-0x7FFFFFFFFFFFFFFF)  # = 0x8000000000000001: never remembered, still distinct-ed
    cols = [cd.Vec.from_numpy(cd.INT64, a), cd.Vec.from_numpy(cd.INT64, np.arange(n, dtype=np.int64))]
    want, _ = oracle.unordered_distinct(cols, [0], False)
    want = H.vecs_to_tuples(want)
    for chunks in (1, 3):
        ordered_equal(gpu_distinct(cols, [0], False, chunks), want, f"chunks={chunks}")
