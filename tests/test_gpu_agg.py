"""GPU parity: device hash aggregation (crgpu_dev_agg_*) against the reference's
case tables and the oracle.  Integer aggregates bit-exact; float SUM/AVG within
rel 1e-9 / abs 1e-6 (colexectestutils/utils.go:1416)."""
import numpy as np
import pytest

from cockroach_amd import coldata as cd
from tests import helpers as H
from tests import gpu_util as G

pytestmark = pytest.mark.gpu
CASES = H.agg_cases()
F = H.AGG_FUNC


@pytest.mark.parametrize("case", CASES, ids=[c["id"] for c in CASES])
def test_agg_golden_case(case):
    cols, group_cols, aggs, expected = H.prepare_agg_case(case)
    if not aggs:
        pytest.skip("no in-scope aggregation in this case")
    for chunks in (1, 2):
        out = G.gpu_agg(cols, group_cols, aggs, chunks=chunks if len(case["input"]) >= 2 else 1)
        H.assert_same_multiset(H.vecs_to_tuples(out), expected, case["id"])


def _canon(vecs):
    return G.rows_as_sorted_array(vecs)


def _bulk_launches(dev):
    """Launches of the bulk path's sub form: the all-distinct attempt and / or the general form."""
    return dev.prof_get("agg_bulk")[1] + dev.prof_get("agg_bulk_distinct")[1]


@pytest.mark.parametrize("ngroups", [1, 4, 50, 3000, 200_000])
def test_agg_random_int_exact_vs_oracle(oracle, ngroups):
    rng = np.random.default_rng(ngroups)
    n = 300_000
    k1 = rng.integers(0, ngroups, size=n, dtype=np.int64)
    k2 = rng.integers(0, 3, size=n).astype(np.int16)
    x = rng.integers(-10**12, 10**12, size=n, dtype=np.int64)
    y = rng.integers(-100, 100, size=n).astype(np.int32)
    b = rng.integers(0, 2, size=n).astype(np.uint8)
    cols = [cd.Vec.from_numpy(cd.INT64, k1, rng.random(n) > 0.02), cd.Vec.from_numpy(cd.INT16, k2),
            cd.Vec.from_numpy(cd.INT64, x, rng.random(n) > 0.1), cd.Vec.from_numpy(cd.INT32, y),
            cd.Vec.from_numpy(cd.BOOL, b, rng.random(n) > 0.3)]
    aggs = [(F["ANY_NOT_NULL"], 0, -1), (F["ANY_NOT_NULL"], 1, -1), (F["COUNT_ROWS"], -1, -1), (F["COUNT"], 2, -1),
            (F["SUM_INT"], 2, -1), (F["MIN"], 2, -1), (F["MAX"], 3, -1), (F["SUM_INT"], 3, 4), (F["BOOL_AND"], 4, -1),
            (F["BOOL_OR"], 4, -1), (F["ANY_NOT_NULL"], 2, -1), (F["COUNT_ROWS"], -1, 4), (F["MIN"], 4, -1)]
    want = oracle.hash_agg(cols, [0, 1], aggs)
    got = G.gpu_agg(cols, [0, 1], aggs, chunks=3)
    assert np.array_equal(_canon(got), _canon(want))


@pytest.mark.parametrize("ngroups", [4, 1000])
def test_agg_float_vs_oracle(oracle, ngroups):
    rng = np.random.default_rng(99 + ngroups)
    n = 200_000
    k = rng.integers(0, ngroups, size=n, dtype=np.int64)
    v = 2048 * (rng.random(n) - 0.5)
    v[:50] = np.nan if ngroups == 4 else v[:50]
    cols = [cd.Vec.from_numpy(cd.INT64, k), cd.Vec.from_numpy(cd.FLOAT64, v, rng.random(n) > 0.1)]
    aggs = [(F["ANY_NOT_NULL"], 0, -1), (F["SUM"], 1, -1), (F["AVG"], 1, -1), (F["MIN"], 1, -1), (F["MAX"], 1, -1),
            (F["COUNT"], 1, -1)]
    if ngroups == 4:
        cols[1] = cd.Vec.from_numpy(cd.FLOAT64, np.where(np.arange(n) < 50, np.nan, v))
        aggs = [(F["ANY_NOT_NULL"], 0, -1), (F["MIN"], 1, -1), (F["MAX"], 1, -1)]  # NaN sums are NaN; MIN => NaN smallest
    want = H.vecs_to_tuples(oracle.hash_agg(cols, [0], aggs))
    got = H.vecs_to_tuples(G.gpu_agg(cols, [0], aggs, chunks=2))
    H.assert_same_multiset(got, want)


def test_agg_bytes_and_float_group_keys_vs_oracle(oracle):
    rng = np.random.default_rng(3)
    n = 60_000
    words = [b"", b"A", b"N", b"R", b"foo", b"x" * 33, b"y" * 70]
    k1 = cd.Vec.from_values(cd.BYTES, [None if rng.random() < 0.02 else words[rng.integers(len(words))] for _ in range(n)])
    k2 = cd.Vec.from_numpy(cd.FLOAT64, rng.choice([0.0, -0.0, np.nan, 1.5], size=n))
    x = cd.Vec.from_numpy(cd.INT64, rng.integers(0, 1000, size=n, dtype=np.int64))
    s = cd.Vec.from_values(cd.BYTES, [b"v%d" % i for i in range(n)])
    aggs = [(F["ANY_NOT_NULL"], 0, -1), (F["SUM_INT"], 2, -1), (F["COUNT_ROWS"], -1, -1), (F["ANY_NOT_NULL"], 3, -1)]
    want = H.vecs_to_tuples(oracle.hash_agg([k1, k2, x, s], [0, 1], aggs))
    got = H.vecs_to_tuples(G.gpu_agg([k1, k2, x, s], [0, 1], aggs, chunks=2))
    H.assert_same_multiset(got, want)


def test_agg_q1_shape(oracle):
    """TPC-H Q1 shape: 2 one-byte BYTES group columns (4 groups), 4 SUM + 3 AVG + COUNT_ROWS over floats."""
    rng = np.random.default_rng(1)
    n = 400_000
    rf = cd.Vec.bytes_from_fixed(rng.choice(np.frombuffer(b"ANR", dtype=np.uint8), size=(n, 1)))
    ls = cd.Vec.bytes_from_fixed(rng.choice(np.frombuffer(b"FO", dtype=np.uint8), size=(n, 1)))
    fl = [cd.Vec.from_numpy(cd.FLOAT64, rng.random(n) * 100) for _ in range(4)]
    cols = [rf, ls] + fl
    aggs = [(F["ANY_NOT_NULL"], 0, -1), (F["ANY_NOT_NULL"], 1, -1)] + [(F["SUM"], 2 + i, -1) for i in range(4)] + \
           [(F["AVG"], 2 + i, -1) for i in range(3)] + [(F["COUNT_ROWS"], -1, -1)]
    want = H.vecs_to_tuples(oracle.hash_agg(cols, [0, 1], aggs))
    got = H.vecs_to_tuples(G.gpu_agg(cols, [0, 1], aggs))
    H.assert_same_multiset(got, want)


def test_agg_sum_int_overflow_and_empty(oracle):
    from cockroach_amd import _lib
    cols = [cd.Vec.from_values(cd.INT64, [1, 1]), cd.Vec.from_values(cd.INT64, [2**63 - 1, 1])]
    with pytest.raises(_lib.CrgpuError) as e:
        G.gpu_agg(cols, [0], [(F["SUM_INT"], 1, -1)])
    assert e.value.code == _lib.ERR_INT_OUT_OF_RANGE
    # large magnitudes that cancel exactly: fits, 128-bit accumulation is order independent
    cols = [cd.Vec.from_values(cd.INT64, [1] * 4), cd.Vec.from_values(cd.INT64, [2**62, 2**62 - 1, -(2**62), 5])]
    out = G.gpu_agg(cols, [0], [(F["SUM_INT"], 1, -1)])
    assert out[0].to_values() == [2**62 - 1 + 5]
    empty = [cd.Vec.from_values(cd.INT64, []), cd.Vec.from_values(cd.INT64, [])]
    out = G.gpu_agg(empty, [0], [(F["ANY_NOT_NULL"], 0, -1), (F["SUM_INT"], 1, -1)])
    assert len(out[0]) == 0  # hash_aggregator_tmpl.go:229-239


def test_agg_many_groups_outputs_in_place(oracle):
    """From 2^16 groups on, finish hands out the accumulator arrays themselves for outputs that need no finalisation
    (counts, INT64 MIN/MAX, SUM_INT low word, FLOAT64 SUM, 8-byte group keys): NULL results, the SUM_INT range check
    and a second finish after more input must behave as with the copying path."""
    from cockroach_amd import _lib, device as dev
    rng = np.random.default_rng(77)
    n, ng = 400_000, 100_000
    k = rng.integers(0, ng, size=n, dtype=np.int64)
    x = rng.integers(-10**15, 10**15, size=n, dtype=np.int64)
    xf = rng.integers(-1000, 1000, size=n).astype(np.float64)
    cols = [cd.Vec.from_numpy(cd.INT64, k, rng.random(n) > 0.01), cd.Vec.from_numpy(cd.INT64, x, rng.random(n) > 0.6),
            cd.Vec.from_numpy(cd.FLOAT64, xf, rng.random(n) > 0.6)]
    aggs = [(F["ANY_NOT_NULL"], 0, -1), (F["SUM_INT"], 1, -1), (F["MIN"], 1, -1), (F["MAX"], 1, -1), (F["COUNT"], 1, -1),
            (F["COUNT_ROWS"], -1, -1), (F["SUM"], 2, -1), (F["AVG"], 2, -1)]
    a = dev.Agg([v.type for v in cols], [0], aggs)
    half = n // 2
    a.consume([dev.upload(v.slice(0, half)) for v in cols], half)
    got = dev.result_to_vecs(a.finish())
    assert np.array_equal(_canon(got), _canon(oracle.hash_agg([v.slice(0, half) for v in cols], [0], aggs)))
    a.consume([dev.upload(v.slice(half, n)) for v in cols], n - half)
    got = dev.result_to_vecs(a.finish())
    a.close()
    want = oracle.hash_agg(cols, [0], aggs)
    assert any(not v.present().all() for v in want[1:4]), "the case should contain NULL results"
    assert np.array_equal(_canon(got), _canon(want))
    # one group out of 2^17 exceeds int64
    k2 = np.arange(1 << 17, dtype=np.int64)
    x2 = np.ones(1 << 17, dtype=np.int64)
    k2 = np.concatenate([k2, [99_999, 99_999]])
    x2 = np.concatenate([x2, [2**63 - 1, 2**63 - 1]]).astype(np.int64)
    with pytest.raises(_lib.CrgpuError) as e:
        G.gpu_agg([cd.Vec.from_numpy(cd.INT64, k2), cd.Vec.from_numpy(cd.INT64, x2)], [0], [(F["SUM_INT"], 1, -1), (F["COUNT_ROWS"], -1, -1)])
    assert e.value.code == _lib.ERR_INT_OUT_OF_RANGE


def test_agg_all_distinct_keys_growth():
    """Config G shape scaled down: every key distinct => the table grows repeatedly."""
    n = 1_500_000
    k = (np.arange(n, dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15)).view(np.int64)  # bijective mix: all distinct
    x = np.arange(n, dtype=np.int64)
    out = G.gpu_agg([cd.Vec.from_numpy(cd.INT64, k), cd.Vec.from_numpy(cd.INT64, x)], [0],
                    [(F["ANY_NOT_NULL"], 0, -1), (F["SUM_INT"], 1, -1), (F["COUNT_ROWS"], -1, -1)], chunks=3)
    assert len(out[0]) == n
    order = np.argsort(out[1].data)
    assert np.array_equal(out[1].data[order], x) and np.array_equal(out[0].data[order], k)
    assert (out[2].data == 1).all()


@pytest.mark.parametrize("ngroups", [6, 700, 2500, 20_000, 500_000])
def test_agg_accumulation_paths_vs_oracle(oracle, ngroups):
    """No ANY_NOT_NULL over value columns => the fused lookup+accumulate pass runs from the second chunk on.
    The group counts select each accumulation strategy (replicated LDS tables, one large LDS table, gid sweeps,
    HBM atomics); the key range widens chunk by chunk so fused passes keep meeting groups that do not exist yet."""
    rng = np.random.default_rng(7 + ngroups)
    n, chunks = 480_000, 4
    hi = np.repeat((np.arange(chunks) + 1) * ngroups // chunks, n // chunks)
    k = (rng.random(n) * hi).astype(np.int64)
    x = rng.integers(-10**12, 10**12, size=n, dtype=np.int64)
    f = rng.integers(-1000, 1000, size=n).astype(np.float64)  # integer-valued: float sums are exact in any order
    b = rng.integers(0, 2, size=n).astype(np.uint8)
    cols = [cd.Vec.from_numpy(cd.INT64, k, rng.random(n) > 0.01), cd.Vec.from_numpy(cd.INT64, x, rng.random(n) > 0.1),
            cd.Vec.from_numpy(cd.FLOAT64, f, rng.random(n) > 0.1), cd.Vec.from_numpy(cd.BOOL, b, rng.random(n) > 0.2)]
    aggs = [(F["ANY_NOT_NULL"], 0, -1), (F["COUNT_ROWS"], -1, -1), (F["COUNT"], 1, -1), (F["SUM_INT"], 1, -1),
            (F["MIN"], 1, -1), (F["MAX"], 1, -1), (F["SUM"], 2, -1), (F["AVG"], 2, -1), (F["MIN"], 2, -1), (F["MAX"], 2, -1),
            (F["BOOL_AND"], 3, -1), (F["BOOL_OR"], 3, -1), (F["SUM_INT"], 1, 3)]
    want = H.vecs_to_tuples(oracle.hash_agg(cols, [0], aggs))
    got = H.vecs_to_tuples(G.gpu_agg(cols, [0], aggs, chunks=chunks))
    H.assert_same_multiset(got, want)


def test_agg_reset_reuses_handle(oracle):
    from cockroach_amd import device as dev
    rng = np.random.default_rng(5)
    a = dev.Agg([cd.INT64, cd.INT64], [0], [(F["ANY_NOT_NULL"], 0, -1), (F["SUM_INT"], 1, -1), (F["COUNT_ROWS"], -1, -1)])
    for ngroups in (5000, 3, 100_000):
        n = 200_000
        cols = [cd.Vec.from_numpy(cd.INT64, rng.integers(0, ngroups, size=n, dtype=np.int64)),
                cd.Vec.from_numpy(cd.INT64, rng.integers(-50, 50, size=n, dtype=np.int64))]
        for s, e in ((0, n // 2), (n // 2, n)):
            a.consume([dev.upload(v.slice(s, e)) for v in cols], e - s)
        got = dev.result_to_vecs(a.finish())
        want = oracle.hash_agg(cols, [0], [(F["ANY_NOT_NULL"], 0, -1), (F["SUM_INT"], 1, -1), (F["COUNT_ROWS"], -1, -1)])
        assert np.array_equal(_canon(got), _canon(want))
        a.reset()
    a.close()


@pytest.mark.parametrize("ngroups", [5, 3000, 40_000, 200_000, 1_500_000, 5_000_000, 1 << 26])
def test_agg_full_size_invariants(ngroups):
    """2^26 rows resident in HBM (the bench shapes; too large for the oracle in the test budget): size-independent
    properties that any loss, duplication or mis-grouping of rows would break -- one output row per distinct key,
    COUNT_ROWS sums to the row count, SUM_INT/MIN/MAX per group equal an independent torch reduction, and
    two passes over halves give the same result as one pass."""
    import torch
    from cockroach_amd import device as dev
    n = 1 << 26
    dv = torch.device("cuda", 0)
    g = torch.Generator(device=dv)
    g.manual_seed(ngroups)
    if ngroups == n:
        k = torch.randperm(n, generator=g, device=dv, dtype=torch.int64) * 7 - 12345
    else:
        k = torch.randint(0, ngroups, (n,), generator=g, device=dv, dtype=torch.int64) * 7 - 12345
    x = torch.randint(-10**9, 10**9, (n,), generator=g, device=dv, dtype=torch.int64)
    torch.cuda.synchronize()
    aggs = [(F["ANY_NOT_NULL"], 0, -1), (F["SUM_INT"], 1, -1), (F["COUNT_ROWS"], -1, -1), (F["MIN"], 1, -1), (F["MAX"], 1, -1)]
    a = dev.Agg([cd.INT64, cd.INT64], [0], aggs)

    def result():
        res = a.finish()
        cols = dev.result_columns(res)
        m = int(res.n)
        return [G.wrap_torch(c, m, torch.int64, dv).clone() for c in cols]

    a.consume([dev.from_torch(cd.INT64, k), dev.from_torch(cd.INT64, x)], n)
    keys, sums, cnts, mins, maxs = result()
    uk, inv = torch.unique(k, return_inverse=True)
    assert keys.numel() == uk.numel()
    order = torch.argsort(keys)
    assert torch.equal(keys[order], uk)
    assert int(cnts.sum().item()) == n
    assert torch.equal(cnts[order], torch.bincount(inv, minlength=uk.numel()))
    ref_sum = torch.zeros(uk.numel(), dtype=torch.int64, device=dv).index_add_(0, inv, x)
    assert torch.equal(sums[order], ref_sum)
    ref_min = torch.full((uk.numel(),), 2**62, dtype=torch.int64, device=dv).scatter_reduce_(0, inv, x, "amin")
    ref_max = torch.full((uk.numel(),), -2**62, dtype=torch.int64, device=dv).scatter_reduce_(0, inv, x, "amax")
    assert torch.equal(mins[order], ref_min) and torch.equal(maxs[order], ref_max)
    # online: two consume calls over the halves
    a.reset()
    h = n // 2
    a.consume([dev.from_torch(cd.INT64, k[:h]), dev.from_torch(cd.INT64, x[:h])], h)
    a.consume([dev.from_torch(cd.INT64, k[h:]), dev.from_torch(cd.INT64, x[h:])], n - h)
    keys2, sums2, cnts2, mins2, maxs2 = result()
    o2 = torch.argsort(keys2)
    assert torch.equal(keys2[o2], uk) and torch.equal(sums2[o2], ref_sum) and torch.equal(cnts2[o2], cnts[order])
    assert torch.equal(mins2[o2], ref_min) and torch.equal(maxs2[o2], ref_max)
    a.close()


@pytest.mark.parametrize("ngroups", [1000, 3_000_000, 1 << 25])
def test_agg_full_size_invariants_bytes_keys(ngroups):
    """The BYTES-key variant of the high-cardinality GROUP BY (BASELINE config G) at 2^25 rows resident in HBM: 16-byte
    keys (the hex digits of an int64), so every property can be checked against an independent torch reduction over the
    integers the keys spell -- one output row per distinct key, the key bytes returned intact, COUNT_ROWS and SUM_INT per
    group, and two passes over halves equal to one pass."""
    import torch
    from cockroach_amd import device as dev
    n = 1 << 25
    dv = torch.device("cuda", 0)
    g = torch.Generator(device=dv)
    g.manual_seed(ngroups)
    mix = 0x9E3779B97F4A7C15 >> 2
    if ngroups == n:
        k = (torch.randperm(n, generator=g, device=dv, dtype=torch.int64) * mix) & (2**62 - 1)
    else:
        k = (torch.randint(0, ngroups, (n,), generator=g, device=dv, dtype=torch.int64) * mix) & (2**62 - 1)
    x = torch.randint(-10**9, 10**9, (n,), generator=g, device=dv, dtype=torch.int64)
    hexd = torch.tensor(list(b"0123456789abcdef"), device=dv, dtype=torch.uint8)
    data = torch.empty((n, 16), device=dv, dtype=torch.uint8)
    for d in range(16):
        data[:, d] = hexd[(k >> (60 - 4 * d)) & 15]
    offs = torch.arange(n + 1, device=dv, dtype=torch.int32) * 16
    torch.cuda.synchronize()
    aggs = [(F["ANY_NOT_NULL"], 0, -1), (F["SUM_INT"], 1, -1), (F["COUNT_ROWS"], -1, -1)]
    a = dev.Agg([cd.BYTES, cd.INT64], [0], aggs)

    def unhex(col, m):
        o = G.wrap_torch_ptr(col.offsets_ptr, m + 1, torch.int32, dv)
        assert torch.equal(o, torch.arange(m + 1, device=dv, dtype=torch.int32) * 16)  # every key is 16 bytes long
        c = G.wrap_torch(col, m * 16, torch.uint8, dv).view(m, 16).to(torch.int64)
        dig = torch.where(c >= 97, c - 87, c - 48)
        assert bool(((dig >= 0) & (dig < 16)).all())
        out = torch.zeros(m, dtype=torch.int64, device=dv)
        for d in range(16):
            out |= dig[:, d] << (60 - 4 * d)
        return out

    def result():
        res = a.finish()
        cols = dev.result_columns(res)
        m = int(res.n)
        return unhex(cols[0], m), G.wrap_torch(cols[1], m, torch.int64, dv).clone(), G.wrap_torch(cols[2], m, torch.int64, dv).clone()

    def rows(lo, hi):
        return [dev.from_torch(cd.BYTES, data[lo:hi].reshape(-1), offsets=offs[:hi - lo + 1]), dev.from_torch(cd.INT64, x[lo:hi])]

    a.consume(rows(0, n), n)
    keys, sums, cnts = result()
    uk, inv = torch.unique(k, return_inverse=True)
    assert keys.numel() == uk.numel()
    order = torch.argsort(keys)
    assert torch.equal(keys[order], uk)
    assert int(cnts.sum().item()) == n
    assert torch.equal(cnts[order], torch.bincount(inv, minlength=uk.numel()))
    ref_sum = torch.zeros(uk.numel(), dtype=torch.int64, device=dv).index_add_(0, inv, x)
    assert torch.equal(sums[order], ref_sum)
    a.reset()
    h = n // 2
    a.consume(rows(0, h), h)
    a.consume(rows(h, n), n - h)
    keys2, sums2, cnts2 = result()
    o2 = torch.argsort(keys2)
    assert torch.equal(keys2[o2], uk) and torch.equal(sums2[o2], ref_sum) and torch.equal(cnts2[o2], cnts[order])
    a.close()


def test_agg_bulk_path_then_more_input_vs_oracle(oracle):
    """A fresh table fed >= 2^23 mostly distinct INT64 keys takes the bulk path (radix partition, grouping in LDS, no
    hash index); the second, small input forces the index to be rebuilt from the groups' hash words and then goes
    through the incremental path.  Everything is compared with the oracle on the concatenated input."""
    from cockroach_amd import device as dev
    rng = np.random.default_rng(21)
    n1, n2 = 1 << 23, 150_000
    base = rng.permutation(n1 // 2).astype(np.int64) * 1_000_003 - 7
    base[:4] = [-1, 0, np.iinfo(np.int64).min, np.iinfo(np.int64).max]  # -1 is the radix kernels' empty-slot marker
    k1 = np.concatenate([base, base])  # every key twice
    rng.shuffle(k1)
    k2 = np.concatenate([rng.choice(base, n2 // 2), np.arange(n2 // 2, dtype=np.int64) * 1_000_003 + 1])  # old and new keys
    x = rng.integers(-10**9, 10**9, size=n1 + n2, dtype=np.int64)
    aggs = [(F["ANY_NOT_NULL"], 0, -1), (F["SUM_INT"], 1, -1), (F["COUNT_ROWS"], -1, -1), (F["MIN"], 1, -1), (F["MAX"], 1, -1),
            (F["COUNT"], 1, -1)]
    a = dev.Agg([cd.INT64, cd.INT64], [0], aggs)
    dev.prof_enable(True)
    dev.prof_reset()
    a.consume([dev.upload(cd.Vec.from_numpy(cd.INT64, k1)), dev.upload(cd.Vec.from_numpy(cd.INT64, x[:n1]))], n1)
    assert _bulk_launches(dev) >= 1, "the bulk path did not run"
    a.consume([dev.upload(cd.Vec.from_numpy(cd.INT64, k2)), dev.upload(cd.Vec.from_numpy(cd.INT64, x[n1:]))], n2)
    dev.prof_enable(False)
    got = dev.result_to_vecs(a.finish())
    a.close()
    want = oracle.hash_agg([cd.Vec.from_numpy(cd.INT64, np.concatenate([k1, k2])), cd.Vec.from_numpy(cd.INT64, x)], [0], aggs)
    assert np.array_equal(_canon(got), _canon(want))


def _bytes_vec(mat, lens):
    """uint8[n, w] + per-row lengths -> BYTES Vec holding the first lens[i] bytes of row i."""
    keep = np.arange(mat.shape[1])[None, :] < lens[:, None]
    offs = np.concatenate([[0], np.cumsum(lens, dtype=np.int64)]).astype(np.uint32)
    return cd.Vec(cd.BYTES, mat[keep], None, offs)


def _bytes_as_int_cols(v):
    """A BYTES column of values <= 24 bytes as four INT64 columns (three zero-padded words + the length), so that results
    with BYTES keys go through the fixed-width canonical form."""
    offs = v.offsets.astype(np.int64)
    lens = offs[1:] - offs[:-1]
    assert lens.max(initial=0) <= 24
    mat = np.zeros((v.n, 24), dtype=np.uint8)
    rows = np.repeat(np.arange(v.n), lens)
    cols = np.arange(int(lens.sum())) - np.repeat(offs[:-1] - offs[0], lens)
    mat[rows, cols] = v.data[offs[0]:offs[-1]]
    words = mat.view(np.int64)
    return [cd.Vec.from_numpy(cd.INT64, words[:, j].copy()) for j in range(3)] + [cd.Vec.from_numpy(cd.INT64, lens)]


def _wide_keys(rng, count):
    """`count` distinct keys of 0 ... 16 bytes: the bytes of i, then filler (often zero bytes), cut to a random length >= 3;
    for the first thousand i also the same bytes one longer (keys that differ by one trailing byte, zero or not), and the
    empty and one- and two-byte keys."""
    i = np.arange(count, dtype=np.int64)
    mat = np.zeros((count, 16), dtype=np.uint8)
    mat[:, :8] = i.view(np.uint8).reshape(count, 8)
    fill = rng.integers(0, 256, size=(count, 13), dtype=np.uint8)
    fill[rng.random(count) < 0.3] = 0
    mat[:, 3:] = fill
    lens = rng.choice([3, 5, 8, 9, 12, 15, 16], size=count, p=[.05, .05, .1, .1, .1, .1, .5]).astype(np.int64)
    extra = np.arange(1000)
    extra = extra[lens[extra] < 16]
    lens2 = lens[extra] + 1
    lens[count - extra.size:] = lens2          # the last keys are replaced by the one-longer twins of the first
    mat[count - extra.size:] = mat[extra]
    special = [b"", b"\x00", b"\x00\x00", b"a", b"ab"]
    for j, b in enumerate(special):
        r = 2000 + j
        mat[r] = 0
        mat[r, :len(b)] = np.frombuffer(b, dtype=np.uint8)
        lens[r] = len(b)
    return mat, lens


def test_agg_bulk_wide_bytes_keys_vs_oracle(oracle):
    """BYTES keys of at most 16 bytes take the wide form of the bulk path (32-byte rows {k0, k1, value, length}): keys of
    every length from 0 to 16, keys that differ only in their length or in a trailing zero byte, every key twice; the
    second input (old keys, new keys, keys longer than 16 bytes) goes through the index rebuilt from the groups' hash
    words and the key store the bulk path filled."""
    from cockroach_amd import device as dev
    rng = np.random.default_rng(31)
    n1, n2 = 1 << 23, 120_000
    mat, lens = _wide_keys(rng, n1 // 2)
    pick = np.concatenate([np.arange(n1 // 2), np.arange(n1 // 2)])
    rng.shuffle(pick)
    k1 = _bytes_vec(mat[pick], lens[pick])
    old = rng.integers(0, n1 // 2, size=n2 // 2)
    new = np.zeros((n2 // 2, 24), dtype=np.uint8)
    new[:, :8] = (np.arange(n2 // 2, dtype=np.int64) + (1 << 40)).view(np.uint8).reshape(-1, 8)
    new[:, 8:] = 7
    new_lens = rng.choice([8, 16, 17, 24], size=n2 // 2).astype(np.int64)
    mat2 = np.zeros((n2, 24), dtype=np.uint8)
    mat2[:n2 // 2, :16] = mat[old]
    mat2[n2 // 2:] = new
    k2 = _bytes_vec(mat2, np.concatenate([lens[old], new_lens]))
    x = rng.integers(-10**9, 10**9, size=n1 + n2, dtype=np.int64)
    aggs = [(F["ANY_NOT_NULL"], 0, -1), (F["SUM_INT"], 1, -1), (F["COUNT_ROWS"], -1, -1), (F["MIN"], 1, -1), (F["MAX"], 1, -1)]
    a = dev.Agg([cd.BYTES, cd.INT64], [0], aggs)
    dev.prof_enable(True)
    dev.prof_reset()
    a.consume([dev.upload(k1), dev.upload(cd.Vec.from_numpy(cd.INT64, x[:n1]))], n1)
    assert _bulk_launches(dev) >= 1, "the wide form of the bulk path did not run"
    a.consume([dev.upload(k2), dev.upload(cd.Vec.from_numpy(cd.INT64, x[n1:]))], n2)
    dev.prof_enable(False)
    got = dev.result_to_vecs(a.finish())
    a.close()
    kall = cd.Vec(cd.BYTES, np.concatenate([k1.data, k2.data]), None,
                  np.concatenate([k1.offsets, k2.offsets[1:].astype(np.int64) + int(k1.offsets[-1])]).astype(np.uint32))
    want = oracle.hash_agg([kall, cd.Vec.from_numpy(cd.INT64, x)], [0], aggs)
    assert got[0].n == want[0].n == n1 // 2 + n2 // 2
    assert np.array_equal(_canon(_bytes_as_int_cols(got[0]) + got[1:]), _canon(_bytes_as_int_cols(want[0]) + want[1:]))


@pytest.mark.parametrize("t2", ["INT64"])
def test_agg_bulk_wide_two_integer_keys_vs_oracle(oracle, t2):
    """Two INT64 group columns take the wide form too ({k0, k1} = the two values): negative values, the
    extremes of both types, key pairs that agree in one column only, every pair twice; then more input through the
    rebuilt index."""
    from cockroach_amd import device as dev
    typ = getattr(cd, t2)
    npt = {"INT64": np.int64, "INT32": np.int32, "INT16": np.int16}[t2]
    info = np.iinfo(npt)
    rng = np.random.default_rng(33)
    n1, n2 = 1 << 23, 100_000
    m = n1 // 2
    a = (rng.permutation(m).astype(np.int64) - m // 2) * 977           # distinct first components ...
    a[:m // 4] = a[m // 4:m // 2]                                        # ... a quarter of them used twice, with
    b = rng.integers(int(info.min), int(info.max), size=m, dtype=np.int64, endpoint=True)
    b[:m // 4] = b[m // 4:m // 2] + np.where(b[m // 4:m // 2] < info.max, 1, -1)  # a different second component
    a[:4] = [np.iinfo(np.int64).min, np.iinfo(np.int64).max, -1, 0]
    b[:4] = [info.min, info.max, -1, 0]
    pick = np.concatenate([np.arange(m), np.arange(m)])
    rng.shuffle(pick)
    old = rng.integers(0, m, size=n2 // 2)
    ka = np.concatenate([a[pick], a[old], np.arange(n2 // 2, dtype=np.int64) + (1 << 50)])
    kb = np.concatenate([b[pick], b[old], np.zeros(n2 // 2, dtype=np.int64)]).astype(npt)
    x = rng.integers(-10**9, 10**9, size=n1 + n2, dtype=np.int64)
    cols = [cd.Vec.from_numpy(cd.INT64, ka), cd.Vec.from_numpy(typ, kb), cd.Vec.from_numpy(cd.INT64, x)]
    aggs = [(F["ANY_NOT_NULL"], 0, -1), (F["ANY_NOT_NULL"], 1, -1), (F["SUM_INT"], 2, -1), (F["COUNT_ROWS"], -1, -1), (F["MAX"], 2, -1)]
    ag = dev.Agg([cd.INT64, typ, cd.INT64], [0, 1], aggs)
    dev.prof_enable(True)
    dev.prof_reset()
    ag.consume([dev.upload(v.slice(0, n1)) for v in cols], n1)
    assert _bulk_launches(dev) >= 1, "the wide form of the bulk path did not run"
    ag.consume([dev.upload(v.slice(n1, n1 + n2)) for v in cols], n2)
    dev.prof_enable(False)
    got = dev.result_to_vecs(ag.finish())
    ag.close()
    want = oracle.hash_agg(cols, [0, 1], aggs)
    assert got[0].n == want[0].n
    assert np.array_equal(_canon(got), _canon(want))


@pytest.mark.parametrize("tk", ["INT64"])
def test_agg_bulk_wide_one_key_two_value_columns_vs_oracle(oracle, tk):
    """One integer group column and TWO value columns (the final stage of a distributed GROUP BY: key, partial sum,
    partial count): the second value rides in the wide row's fourth word, which is then no part of the key.  Values are
    chosen so that rows of one key differ in that word."""
    from cockroach_amd import device as dev
    typ = getattr(cd, tk)
    npt = {"INT64": np.int64, "INT32": np.int32}[tk]
    rng = np.random.default_rng(35)
    n1, n2 = 1 << 23, 100_000
    m = n1 // 2
    base = (rng.permutation(m).astype(np.int64) - m // 2) * (977 if tk == "INT64" else 3)
    base[:4] = [np.iinfo(npt).min, np.iinfo(npt).max, -1, 0]
    pick = np.concatenate([np.arange(m), np.arange(m)])
    rng.shuffle(pick)
    k = np.concatenate([base[pick], rng.choice(base, n2 // 2), np.arange(n2 // 2, dtype=np.int64) * 2 + 10**9 + 1]).astype(npt)
    x = rng.integers(-10**9, 10**9, size=n1 + n2, dtype=np.int64)
    y = rng.integers(-1000, 1000, size=n1 + n2).astype(np.float64)
    cols = [cd.Vec.from_numpy(typ, k), cd.Vec.from_numpy(cd.INT64, x), cd.Vec.from_numpy(cd.FLOAT64, y)]
    aggs = [(F["ANY_NOT_NULL"], 0, -1), (F["SUM_INT"], 1, -1), (F["SUM"], 2, -1), (F["COUNT_ROWS"], -1, -1), (F["MIN"], 1, -1),
            (F["MAX"], 2, -1), (F["AVG"], 2, -1)]
    ag = dev.Agg([typ, cd.INT64, cd.FLOAT64], [0], aggs)
    dev.prof_enable(True)
    dev.prof_reset()
    ag.consume([dev.upload(v.slice(0, n1)) for v in cols], n1)
    assert _bulk_launches(dev) >= 1, "the wide form of the bulk path did not run"
    ag.consume([dev.upload(v.slice(n1, n1 + n2)) for v in cols], n2)
    dev.prof_enable(False)
    got = dev.result_to_vecs(ag.finish())
    ag.close()
    want = oracle.hash_agg(cols, [0], aggs)
    assert got[0].n == want[0].n
    assert np.array_equal(_canon(got), _canon(want))  # (integer-valued floats: sums and averages are exact in any order)


def test_agg_bulk_path_columns_with_validity_bitmaps(oracle):
    """Key and value columns that carry a validity bitmap without a NULL in it (the partial rows of a distributed GROUP
    BY: the local stage's outputs are nullable) still take the bulk path; one NULL anywhere keeps the input on the
    incremental path.  Results against the oracle either way."""
    from cockroach_amd import device as dev
    rng = np.random.default_rng(36)
    n = (1 << 23) + 5  # the bitmap's last byte is partly used
    k = rng.permutation(n).astype(np.int64) * 31 - 7
    x = rng.integers(-10**6, 10**6, size=n, dtype=np.int64)
    c = rng.integers(1, 5, size=n, dtype=np.int64)
    aggs = [(F["ANY_NOT_NULL"], 0, -1), (F["SUM_INT"], 1, -1), (F["SUM_INT"], 2, -1)]
    all_set = np.ones(n, dtype=bool)

    def run(pk, px, pc):
        # the bitmaps are attached on the device side: Vec.from_numpy drops a mask without a NULL
        import torch
        dv = torch.device("cuda", 0)
        dcols = [dev.from_torch(cd.INT64, torch.from_numpy(a).to(dv), valid=torch.from_numpy(np.packbits(pres, bitorder="little")).to(dv))
                 for a, pres in ((k, pk), (x, px), (c, pc))]
        torch.cuda.synchronize()
        ag = dev.Agg([cd.INT64] * 3, [0], aggs)
        dev.prof_reset()
        ag.consume(dcols, n)
        ran = _bulk_launches(dev)
        got = dev.result_to_vecs(ag.finish())
        ag.close()
        host = [cd.Vec.from_numpy(cd.INT64, k, pk), cd.Vec.from_numpy(cd.INT64, x, px), cd.Vec.from_numpy(cd.INT64, c, pc)]
        want = oracle.hash_agg(host, [0], aggs)
        assert np.array_equal(_canon(got), _canon(want))
        return ran

    dev.prof_enable(True)
    assert run(all_set, all_set, all_set) >= 1, "all-set bitmaps kept the input off the bulk path"
    one_null = all_set.copy()
    one_null[n - 2] = False  # in the partly used last byte
    assert run(all_set, all_set, one_null) == 0
    assert run(one_null, all_set, all_set) == 0
    dev.prof_enable(False)


@pytest.mark.parametrize("dups", [0, 3, 5_000])
def test_agg_bulk_all_distinct_attempt_and_its_fallback(oracle, dups):
    """The sub form's all-distinct attempt (agg_bulk_kernel<.., TRY>: group ids from the prefix sums of the regions' row
    counts, no accumulator planes): keys the 8192-row sample sees as distinct.  dups = 0: the attempt is the result
    (and SUM_INT skips its range check: one row per group).  dups > 0: a few keys repeat at rows the sample does not
    visit -- the attempt is void and the general form runs over every region; either way the oracle's rows, also after
    a second, small input."""
    from cockroach_amd import _lib, device as dev
    rng = np.random.default_rng(41 + dups)
    n = 1 << 23
    k = rng.permutation(n).astype(np.int64) * 1_000_003 - 11
    stride = n // 8192
    if dups:
        src = rng.choice(np.arange(n)[np.arange(n) % stride == 5], dups, replace=False)
        dst = src + 1  # never a sampled row (those are the multiples of the stride)
        k[dst] = k[src]
    x = rng.integers(-2**62, 2**62, size=n, dtype=np.int64)
    aggs = [(F["ANY_NOT_NULL"], 0, -1), (F["SUM_INT"], 1, -1), (F["COUNT_ROWS"], -1, -1)]
    a = dev.Agg([cd.INT64, cd.INT64], [0], aggs)
    dev.prof_enable(True)
    dev.prof_reset()
    a.consume([dev.upload(cd.Vec.from_numpy(cd.INT64, k)), dev.upload(cd.Vec.from_numpy(cd.INT64, x))], n)
    tried, general = dev.prof_get("agg_bulk_distinct")[1], dev.prof_get("agg_bulk")[1]
    dev.prof_enable(False)
    assert tried == 1 and general == (1 if dups else 0)
    got = dev.result_to_vecs(a.finish())
    want = oracle.hash_agg([cd.Vec.from_numpy(cd.INT64, k), cd.Vec.from_numpy(cd.INT64, x)], [0], aggs)
    assert np.array_equal(_canon(got), _canon(want))
    # more input behind the bulk groups: sums may now leave the int64 range, and the check must be back
    k2 = k[:1000].copy()
    x2 = np.where(x[:1000] >= 0, np.iinfo(np.int64).max, np.iinfo(np.int64).min)
    a.consume([dev.upload(cd.Vec.from_numpy(cd.INT64, k2)), dev.upload(cd.Vec.from_numpy(cd.INT64, x2))], 1000)
    with pytest.raises(_lib.CrgpuError) as e:
        a.finish()
    assert e.value.code == _lib.ERR_INT_OUT_OF_RANGE
    a.close()


@pytest.mark.parametrize("n,keys_per", [((1 << 23) + 1, 1), (10_000_000, 1), (3 * (1 << 22) + 777, 2), (11_111_111, 3)])
def test_agg_bulk_region_count_follows_the_row_count(oracle, n, keys_per):
    """The sub form splits every radix partition into ceil(n / 2^20) regions, any number of them (9, 10, 13 and 11 here),
    by a multiply-shift of hash bits (radix_split_kernel<.., ANY>): row counts that are no power of two, all keys
    distinct (the all-distinct attempt) and every key two or three times (the general form), against the oracle."""
    from cockroach_amd import device as dev
    rng = np.random.default_rng(n % 1000 + keys_per)
    g = (n + keys_per - 1) // keys_per
    k = np.tile(rng.permutation(g).astype(np.int64) * 1_000_003 - 11, keys_per)[:n]
    rng.shuffle(k)
    x = rng.integers(-2**40, 2**40, size=n, dtype=np.int64)
    aggs = [(F["ANY_NOT_NULL"], 0, -1), (F["SUM_INT"], 1, -1), (F["COUNT_ROWS"], -1, -1), (F["MIN"], 1, -1)]
    a = dev.Agg([cd.INT64, cd.INT64], [0], aggs)
    dev.prof_enable(True)
    dev.prof_reset()
    a.consume([dev.upload(cd.Vec.from_numpy(cd.INT64, k)), dev.upload(cd.Vec.from_numpy(cd.INT64, x))], n)
    launches, incremental = _bulk_launches(dev), dev.prof_get("agg_resolve")[1]
    dev.prof_enable(False)
    assert launches >= 1 and incremental == 0, "the bulk path did not take the input"
    got = dev.result_to_vecs(a.finish())
    a.close()
    want = oracle.hash_agg([cd.Vec.from_numpy(cd.INT64, k), cd.Vec.from_numpy(cd.INT64, x)], [0], aggs)
    assert np.array_equal(_canon(got), _canon(want))


@pytest.mark.parametrize("shape", ["distinct", "twice", "few_groups_in_a_piece"])
def test_agg_bulk_input_split_into_pieces_vs_oracle(oracle, shape, monkeypatch):
    """A first input beyond the bulk path's 2^27 rows is split by key hash into pieces that go through the bulk path one
    after the other, each appending its groups behind the previous pieces' (try_bulk_huge).  Here with the threshold
    lowered to 2^24 rows (two pieces of ~2^23): all keys distinct, every key twice, and an input whose keys are mostly
    one hot key apart from distinct ones -- the bulk path declines such pieces and they take the incremental path
    behind the bulk groups.  Against the oracle; then a second, small input that meets old and new keys."""
    from cockroach_amd import device as dev
    monkeypatch.setenv("CRGPU_BULK_HUGE_MIN", str(1 << 24))
    rng = np.random.default_rng({"distinct": 51, "twice": 52, "few_groups_in_a_piece": 53}[shape])
    n = (1 << 24) + 12345
    if shape == "distinct":
        k = rng.permutation(n).astype(np.int64) * 1_000_003 - 5
    elif shape == "twice":
        half = rng.permutation(n // 2 + 1).astype(np.int64) * 1_000_003 - 5
        k = np.concatenate([half, half])[:n]
        rng.shuffle(k)
    else:
        k = rng.permutation(n).astype(np.int64) * 1_000_003 - 5
        k[rng.random(n) < 0.3] = 424242  # a hot key: the sample sees it, its piece is declined
    x = rng.integers(-10**9, 10**9, size=n, dtype=np.int64)
    aggs = [(F["ANY_NOT_NULL"], 0, -1), (F["SUM_INT"], 1, -1), (F["COUNT_ROWS"], -1, -1), (F["MIN"], 1, -1)]
    a = dev.Agg([cd.INT64, cd.INT64], [0], aggs)
    dev.prof_enable(True)
    dev.prof_reset()
    a.consume([dev.upload(cd.Vec.from_numpy(cd.INT64, k)), dev.upload(cd.Vec.from_numpy(cd.INT64, x))], n)
    pieces_bulk = _bulk_launches(dev) + dev.prof_get("agg_bulk_part")[1]
    parts = dev.prof_get("partition")[1]
    dev.prof_enable(False)
    assert parts == 1, "the input was not split"
    if shape != "few_groups_in_a_piece":
        assert pieces_bulk >= 2, "the pieces did not take the bulk path"
    k2 = np.concatenate([k[:5000], np.arange(5000, dtype=np.int64) * 1_000_003 + 1])
    x2 = rng.integers(-10**9, 10**9, size=len(k2), dtype=np.int64)
    a.consume([dev.upload(cd.Vec.from_numpy(cd.INT64, k2)), dev.upload(cd.Vec.from_numpy(cd.INT64, x2))], len(k2))
    got = dev.result_to_vecs(a.finish())
    a.close()
    want = oracle.hash_agg([cd.Vec.from_numpy(cd.INT64, np.concatenate([k, k2])), cd.Vec.from_numpy(cd.INT64, np.concatenate([x, x2]))], [0], aggs)
    assert np.array_equal(_canon(got), _canon(want))


@pytest.mark.parametrize("value", ["float", "none"])
def test_agg_bulk_input_split_into_pieces_other_value_shapes(oracle, value, monkeypatch):
    """try_bulk_huge with a FLOAT64 value column (SUM / MIN / MAX / AVG) and with no value column at all (COUNT_ROWS
    only), unused input columns beside them (the pieces carry only the key and the value column)."""
    from cockroach_amd import device as dev
    monkeypatch.setenv("CRGPU_BULK_HUGE_MIN", str(1 << 24))
    rng = np.random.default_rng(61 if value == "float" else 62)
    n = (1 << 24) + 77
    half = rng.permutation(n // 2 + 1).astype(np.int64) * 7919 + 3
    k = np.concatenate([half, half])[:n]
    rng.shuffle(k)
    unused = cd.Vec.from_numpy(cd.INT32, np.zeros(n, dtype=np.int32))
    if value == "float":
        x = rng.integers(-2**20, 2**20, size=n).astype(np.float64) / 8.0  # exact in any order
        cols = [unused, cd.Vec.from_numpy(cd.INT64, k), cd.Vec.from_numpy(cd.FLOAT64, x)]
        aggs = [(F["ANY_NOT_NULL"], 1, -1), (F["SUM"], 2, -1), (F["MIN"], 2, -1), (F["MAX"], 2, -1), (F["AVG"], 2, -1), (F["COUNT_ROWS"], -1, -1)]
    else:
        cols = [unused, cd.Vec.from_numpy(cd.INT64, k)]
        aggs = [(F["ANY_NOT_NULL"], 1, -1), (F["COUNT_ROWS"], -1, -1)]
    a = dev.Agg([v.type for v in cols], [1], aggs)
    dev.prof_enable(True)
    dev.prof_reset()
    a.consume([dev.upload(v) for v in cols], n)
    assert dev.prof_get("partition")[1] == 1 and _bulk_launches(dev) >= 2
    dev.prof_enable(False)
    got = dev.result_to_vecs(a.finish())
    a.close()
    want = oracle.hash_agg(cols, [1], aggs)
    assert np.array_equal(_canon(got), _canon(want))


def test_agg_bulk_wide_abandoned_on_a_long_key(oracle):
    """One 17-byte key that the 8192-row sample does not see: the histogram pass of the wide form reports it, the
    attempt is abandoned before anything is committed, and the incremental path gives the oracle's result."""
    from cockroach_amd import device as dev
    rng = np.random.default_rng(32)
    n = 1 << 23
    mat = np.zeros((n, 17), dtype=np.uint8)
    mat[:, :8] = rng.permutation(n).astype(np.int64).view(np.uint8).reshape(n, 8)
    mat[:, 8:] = 0x41
    lens = np.full(n, 16, dtype=np.int64)
    lens[12345] = 17
    k = _bytes_vec(mat, lens)
    x = cd.Vec.from_numpy(cd.INT64, rng.integers(-1000, 1000, size=n, dtype=np.int64))
    aggs = [(F["ANY_NOT_NULL"], 0, -1), (F["SUM_INT"], 1, -1), (F["COUNT_ROWS"], -1, -1)]
    dev.prof_enable(True)
    dev.prof_reset()
    got = G.gpu_agg([k, x], [0], aggs)
    assert dev.prof_get("radix_build_hist")[1] == 1 and _bulk_launches(dev) == 0, "expected an abandoned wide attempt"
    dev.prof_enable(False)
    want = oracle.hash_agg([k, x], [0], aggs)
    assert got[0].n == n
    assert np.array_equal(_canon(_bytes_as_int_cols(got[0]) + got[1:]), _canon(_bytes_as_int_cols(want[0]) + want[1:]))


@pytest.mark.parametrize("ngroups", [20_000, 150_000])
def test_agg_bulk_part_form_vs_oracle(oracle, ngroups):
    """Mid cardinality (tens to ~2000 groups per radix partition): the bulk path without the level-2 split; a
    workgroup walks its variable-length partition in tiles and keeps the groups in LDS.  More input afterwards goes
    through the rebuilt index."""
    from cockroach_amd import device as dev
    rng = np.random.default_rng(ngroups)
    n1, n2 = 1 << 23, 100_000
    base = rng.permutation(ngroups * 4)[:ngroups].astype(np.int64) * 7_919 - 3
    base[:4] = [-1, 0, np.iinfo(np.int64).min, np.iinfo(np.int64).max]
    k1 = base[rng.integers(0, ngroups, size=n1)]
    k2 = np.concatenate([rng.choice(base, n2 // 2), np.arange(n2 // 2, dtype=np.int64) * 7_919 + 1])
    x = rng.integers(-10**9, 10**9, size=n1 + n2, dtype=np.int64)
    aggs = [(F["ANY_NOT_NULL"], 0, -1), (F["SUM_INT"], 1, -1), (F["COUNT_ROWS"], -1, -1), (F["MIN"], 1, -1), (F["MAX"], 1, -1)]
    a = dev.Agg([cd.INT64, cd.INT64], [0], aggs)
    dev.prof_enable(True)
    dev.prof_reset()
    a.consume([dev.upload(cd.Vec.from_numpy(cd.INT64, k1)), dev.upload(cd.Vec.from_numpy(cd.INT64, x[:n1]))], n1)
    assert dev.prof_get("agg_bulk_part")[1] == 1, "the part form of the bulk path did not run"
    a.consume([dev.upload(cd.Vec.from_numpy(cd.INT64, k2)), dev.upload(cd.Vec.from_numpy(cd.INT64, x[n1:]))], n2)
    dev.prof_enable(False)
    got = dev.result_to_vecs(a.finish())
    a.close()
    want = oracle.hash_agg([cd.Vec.from_numpy(cd.INT64, np.concatenate([k1, k2])), cd.Vec.from_numpy(cd.INT64, x)], [0], aggs)
    assert np.array_equal(_canon(got), _canon(want))


def test_agg_bulk_part_form_skew_and_overflow(oracle):
    """A hot key (10 % of the rows) keeps the input away from the part form; a group count the sample underestimates
    (half of the rows on 30 000 keys, the other half all distinct) overflows the LDS planes and the attempt is
    abandoned.  Either way the result is the oracle's."""
    from cockroach_amd import device as dev
    rng = np.random.default_rng(5)
    n = 1 << 23
    aggs = [(F["ANY_NOT_NULL"], 0, -1), (F["SUM_INT"], 1, -1), (F["COUNT_ROWS"], -1, -1)]
    x = rng.integers(-1000, 1000, size=n, dtype=np.int64)
    k = rng.integers(0, 200_000, size=n, dtype=np.int64)
    k[rng.random(n) < 0.1] = 77
    cols = [cd.Vec.from_numpy(cd.INT64, k), cd.Vec.from_numpy(cd.INT64, x)]
    dev.prof_enable(True)
    dev.prof_reset()
    got = G.gpu_agg(cols, [0], aggs)
    assert dev.prof_get("agg_bulk_part")[1] == 0 and _bulk_launches(dev) == 0
    assert np.array_equal(_canon(got), _canon(oracle.hash_agg(cols, [0], aggs)))
    k = np.where(np.arange(n) % 2 == 0, rng.integers(0, 30_000, size=n, dtype=np.int64), np.arange(n, dtype=np.int64) + 10**9)
    cols = [cd.Vec.from_numpy(cd.INT64, k), cd.Vec.from_numpy(cd.INT64, x)]
    dev.prof_reset()
    got = G.gpu_agg(cols, [0], aggs)
    dev.prof_enable(False)
    assert np.array_equal(_canon(got), _canon(oracle.hash_agg(cols, [0], aggs)))


def test_agg_bulk_path_float_values_and_fallback(oracle):
    from cockroach_amd import device as dev
    rng = np.random.default_rng(22)
    n = 1 << 23
    k = rng.permutation(n).astype(np.int64) - n // 2
    v = rng.integers(-1000, 1000, size=n).astype(np.float64)  # integer-valued: sums are exact in any order
    aggs = [(F["ANY_NOT_NULL"], 0, -1), (F["SUM"], 1, -1), (F["AVG"], 1, -1), (F["MIN"], 1, -1), (F["MAX"], 1, -1),
            (F["COUNT_ROWS"], -1, -1)]
    cols = [cd.Vec.from_numpy(cd.INT64, k), cd.Vec.from_numpy(cd.FLOAT64, v)]
    dev.prof_enable(True)
    dev.prof_reset()
    got = G.gpu_agg(cols, [0], aggs)
    assert _bulk_launches(dev) >= 1
    assert np.array_equal(_canon(got), _canon(oracle.hash_agg(cols, [0], aggs)))
    # a hot key hidden from the 8192-key sample (stride n / 8192) makes its sub-partition overflow after the sample
    # looked distinct: the bulk attempt is abandoned and the incremental path produces the result
    k2 = k.copy()
    hot = np.arange(n) % 1024 == 1
    k2[hot] = 123456789
    cols2 = [cd.Vec.from_numpy(cd.INT64, k2), cd.Vec.from_numpy(cd.FLOAT64, v)]
    dev.prof_reset()
    got2 = G.gpu_agg(cols2, [0], aggs)
    assert dev.prof_get("radix_build_split")[1] == 1 and _bulk_launches(dev) == 0, "expected an abandoned bulk attempt"
    dev.prof_enable(False)
    assert np.array_equal(_canon(got2), _canon(oracle.hash_agg(cols2, [0], aggs)))


@pytest.mark.parametrize("ngroups", [1, 7, 5000, 120_000])
def test_agg_min_max_bytes_vs_oracle(oracle, ngroups):
    """MIN / MAX over a BYTES column (hash_min_max_agg.eg.go:247-356: bytes.Compare; NULL inputs skipped, a group with no
    non-NULL input yields NULL), with FILTER, beside other aggregates, over several chunks so that a later chunk has to
    beat (or lose against) the stored value.  The values share long prefixes, differ only in length, contain 0x00 and
    0xff bytes and the empty string: the 8-byte prefix test alone decides none of those.  The reference's case tables
    hold no MIN/MAX over bytes (parity unpinned by literal vectors): the oracle restates bytes.Compare."""
    rng = np.random.default_rng(1000 + ngroups)
    n = 200_000
    stems = [b"", b"\x00", b"\x00\x00", b"a", b"ab", b"abcdefgh", b"abcdefgh\x00", b"abcdefghi", b"abcdefgh\xff", b"\xff" * 9,
             b"\xff" * 8, b"customer#00000", b"zz"]
    pick = rng.integers(0, len(stems), size=n)
    tail = rng.integers(0, 50, size=n)
    vals = [None if rng.random() < 0.1 else stems[pick[i]] + (b"%d" % tail[i] if tail[i] % 3 else b"") for i in range(n)]
    k = rng.integers(0, ngroups, size=n, dtype=np.int64)
    if ngroups > 1:  # some groups see NULLs only
        for i in np.nonzero(k == 1)[0]:
            vals[i] = None
    flt = rng.integers(0, 2, size=n).astype(np.uint8)
    cols = [cd.Vec.from_numpy(cd.INT64, k), cd.Vec.from_values(cd.BYTES, vals),
            cd.Vec.from_numpy(cd.BOOL, flt, rng.random(n) > 0.05), cd.Vec.from_numpy(cd.INT64, rng.integers(-9, 9, size=n, dtype=np.int64))]
    aggs = [(F["ANY_NOT_NULL"], 0, -1), (F["MIN"], 1, -1), (F["MAX"], 1, -1), (F["MIN"], 1, 2), (F["MAX"], 1, 2),
            (F["SUM_INT"], 3, -1), (F["COUNT"], 1, -1)]
    want = H.vecs_to_tuples(oracle.hash_agg(cols, [0], aggs))
    for chunks in (1, 4):
        got = H.vecs_to_tuples(G.gpu_agg(cols, [0], aggs, chunks=chunks))
        H.assert_same_multiset(got, want, f"chunks={chunks}")


def test_agg_min_max_bytes_group_key_and_empty(oracle):
    """BYTES group key + MIN/MAX over another BYTES column; an empty input; all-NULL input."""
    rng = np.random.default_rng(8)
    n = 30_000
    keys = [b"", b"k", b"key-%d"]
    k = cd.Vec.from_values(cd.BYTES, [None if i % 97 == 0 else (keys[i % 3] % (i % 211) if i % 3 == 2 else keys[i % 3]) for i in range(n)])
    v = cd.Vec.from_values(cd.BYTES, [bytes(rng.integers(0, 256, size=rng.integers(0, 20)).astype(np.uint8)) for _ in range(n)])
    aggs = [(F["ANY_NOT_NULL"], 0, -1), (F["MIN"], 1, -1), (F["MAX"], 1, -1), (F["COUNT_ROWS"], -1, -1)]
    want = H.vecs_to_tuples(oracle.hash_agg([k, v], [0], aggs))
    H.assert_same_multiset(H.vecs_to_tuples(G.gpu_agg([k, v], [0], aggs, chunks=3)), want)
    empty = [cd.Vec.from_values(cd.BYTES, []), cd.Vec.from_values(cd.BYTES, [])]
    assert len(G.gpu_agg(empty, [0], aggs)[0]) == 0
    nulls = [cd.Vec.from_values(cd.BYTES, [b"g"] * 5), cd.Vec.from_values(cd.BYTES, [None] * 5)]
    assert H.vecs_to_tuples(G.gpu_agg(nulls, [0], aggs)) == [(b"g", None, None, 5)]


def test_agg_dropped_reference_columns_are_exactly_the_unsupported_specs():
    """The (case, aggregation) pairs the harness leaves out of the reference's tables (helpers.dropped_agg_pairs, listed in
    DESIGN.md section 6) are the ones crgpu_dev_agg_create answers with CRGPU_ERR_UNSUPPORTED -- the boundary at which the
    planner keeps the CPU operator -- and nothing else is dropped."""
    from cockroach_amd import _lib
    from cockroach_amd import device as dev
    dropped = H.dropped_agg_pairs()
    assert len(dropped) == 5 and sum(1 for d in dropped if d[1] < 0) == 2
    reasons = {d[3].split(":")[0] for d in dropped if d[1] >= 0}
    assert reasons == {"function outside the GPU path"}
    with pytest.raises(_lib.CrgpuError) as e:  # DISTINCT on COUNT_ROWS has no input to be distinct over
        dev.Agg([cd.INT64, cd.INT64], [0], [(F["COUNT_ROWS"], -1, -1, True)])
    assert e.value.code == _lib.ERR_UNSUPPORTED
    with pytest.raises(_lib.CrgpuError) as e:  # CONCAT_AGG (execinfrapb.AggregatorSpec_CONCAT_AGG = 4)
        dev.Agg([cd.INT64, cd.BYTES], [0], [(4, 1, -1)])
    assert e.value.code == _lib.ERR_UNSUPPORTED
    with pytest.raises(_lib.CrgpuError) as e:  # AVG over INT => DECIMAL division: the shell asks for SUM and COUNT
        dev.Agg([cd.INT64, cd.INT64], [0], [(F["AVG"], 1, -1)])
    assert e.value.code == _lib.ERR_UNSUPPORTED
    for c in CASES:  # every surviving case keeps at least one aggregation
        assert H.prepare_agg_case(c)[2], c["id"]


def int128_of(b):
    return None if b is None else int.from_bytes(b, "little", signed=True)


@pytest.mark.parametrize("itype", [cd.INT16, cd.INT32, cd.INT64])
@pytest.mark.parametrize("ngroups", [7, 5000, 70000])  # one-launch finish, the general finish, the aliasing finish
def test_sum_over_int_is_the_exact_128_bit_total(oracle, itype, ngroups):
    """SUM over an integer column (DECIMAL in the reference, hash_sum_agg.eg.go:30-373): a BYTES column of 16-byte
    little-endian two's-complement totals.  No reference test holds a vector for it (theirs sum DECIMAL inputs), so the
    oracle's restatement is pinned by Python's integers here; the GPU against the oracle as everywhere."""
    rng = np.random.default_rng(ngroups + itype)
    n = 200_000
    info = np.iinfo({cd.INT16: np.int16, cd.INT32: np.int32, cd.INT64: np.int64}[itype])
    g = rng.integers(0, ngroups, n, dtype=np.int64)
    x = rng.integers(info.min, info.max, n, dtype=info.dtype, endpoint=True)
    x[: n // 4] = info.max  # totals beyond 64 bits for INT64
    valid = rng.random(n) > 0.1
    valid[g == 3] = False  # a group without a non-NULL input: NULL
    cols = [cd.Vec.from_numpy(cd.INT64, g), cd.Vec.from_numpy(itype, x, valid)]
    aggs = [(F["ANY_NOT_NULL"], 0, -1), (F["SUM"], 1, -1), (F["COUNT"], 1, -1)]
    want = {}
    for gi, xi, vi in zip(g.tolist(), x.tolist(), valid.tolist()):
        t = want.setdefault(gi, [None, 0])
        if vi:
            t[0] = (t[0] or 0) + xi
            t[1] += 1
    ora = {r[0]: (int128_of(r[1]), r[2]) for r in H.vecs_to_tuples(oracle.hash_agg(cols, [0], aggs))}
    assert ora == {k: tuple(v) for k, v in want.items()}
    got = {r[0]: (int128_of(r[1]), r[2]) for r in H.vecs_to_tuples(G.gpu_agg(cols, [0], aggs))}
    assert got == ora
    if itype == cd.INT64:
        assert any(v[0] is not None and abs(v[0]) >= 2 ** 63 for v in got.values())


def test_agg_two_to_the_thirty_rows_in_one_call():
    """One consume() call of more than 2^30 rows (17 GB of input): the settled low-cardinality path takes it in chunks of
    up to 2^30 rows, each chunk as one launch per 2^28 rows with the column bases moved along (agg.hip, try_sumfam), and
    rows of a group the first chunk has not seen reach the resolve pass with their ABSOLUTE row numbers.  Checked through
    size-independent properties: the counts add up to n, every group's count and sums are what its key's arithmetic
    progression gives."""
    import torch
    from cockroach_amd import _lib, device as dev
    _lib.check(_lib.load().crgpu_init(0))
    n = (1 << 30) + 12345
    NG = 6
    i = torch.arange(n, device="cuda", dtype=torch.int64)
    k = i % NG
    # group 6 appears only behind row 2^29 + 7 (later than the discovery chunk) and far apart: 9 rows in all
    late = torch.tensor([(1 << 29) + 7 + j * 60_000_001 for j in range(9)], device="cuda")
    k[late] = 6
    x = (i % 1000).to(torch.float64)
    del i
    torch.cuda.synchronize()
    a = dev.Agg([cd.INT64, cd.FLOAT64], [0], [(F["ANY_NOT_NULL"], 0, -1), (F["COUNT_ROWS"], -1, -1), (F["SUM"], 1, -1), (F["AVG"], 1, -1)])
    try:
        a.consume([dev.from_torch(cd.INT64, k), dev.from_torch(cd.FLOAT64, x)], n)
        got = {r[0]: r[1:] for r in H.vecs_to_tuples(dev.result_to_vecs(a.finish()))}
    finally:
        a.close()
    assert sorted(got) == list(range(7))
    assert sum(v[0] for v in got.values()) == n
    for g in range(7):
        m = k == g
        cnt = int(m.sum().item())
        s = float(x[m].sum().item())  # integers below 2^53: exact in any order
        assert got[g][0] == cnt and got[g][1] == s and got[g][2] == s / cnt
    assert got[6][0] == 9


def test_agg_two_to_the_thirty_rows_and_half_as_many_groups():
    """BASELINE.json configs[4] at the size ONE GPU holds: 2^30 rows over 2^29 distinct int64 keys (the most groups a
    table takes: 32-bit group ids, 2^30 index entries), every key twice at scattered rows, SUM_INT + COUNT_ROWS, in one
    consume() call -- split by key hash into 16 pieces that go through the bulk path one after the other
    (try_bulk_huge), no hash index built.  Size-independent checks: 2^29 groups, every count is 2, the sums of the keys and
    of their squares (mod 2^64) are those of the 2^29 distinct keys, the grand total of the values is the input's."""
    import time
    import torch
    from cockroach_amd import _lib, device as dev
    from tests import gpu_util as GU
    _lib.check(_lib.load().crgpu_init(0))
    n, ng = 1 << 30, 1 << 29
    MIX = 0x9E3779B97F4A7C15 - (1 << 64)  # odd: g -> g * MIX is a bijection of the int64 values
    j = (torch.arange(n, device="cuda", dtype=torch.int64) * 0x9E3779B1) & (n - 1)  # a permutation of the row numbers
    k = (j >> 1) * MIX
    x = j & 1023
    del j
    gk = torch.arange(ng, device="cuda", dtype=torch.int64) * MIX
    want_k, want_k2, want_x = int(gk.sum().item()), int((gk * gk).sum().item()), int(x.sum().item())
    del gk
    torch.cuda.synchronize()
    a = dev.Agg([cd.INT64, cd.INT64], [0], [(F["ANY_NOT_NULL"], 0, -1), (F["SUM_INT"], 1, -1), (F["COUNT_ROWS"], -1, -1)])
    try:
        for attempt in ("cold", "warm"):  # (the first pass pays ~100 GB of first-time device allocations)
            a.reset()
            dev.prof_enable(True)
            dev.prof_reset()
            t0 = time.perf_counter()
            a.consume([dev.from_torch(cd.INT64, k), dev.from_torch(cd.INT64, x)], n)
            res = a.finish()
            _lib.check(_lib.load().crgpu_dev_synchronize())
            dt = time.perf_counter() - t0
            pieces = _bulk_launches(dev)
            resolved = dev.prof_get("agg_resolve")[1]
            kern = {f: round(dev.prof_get(f)[0], 2) for f in ("partition", "radix_build_hist", "radix_build_scatter", "radix_build_split",
                                                               "agg_bulk", "agg_bulk_distinct")}
            dev.prof_enable(False)
            print(f"2^30 rows / 2^29 groups ({attempt}): {dt * 1e3:.1f} ms, {n / dt:.3g} rows/s, {pieces} bulk launches, kernels ms {kern}")
        assert int(res.n) == ng
        assert pieces >= 16 and resolved == 0, "the pieces did not all take the bulk path"
        d = torch.device("cuda", 0)
        ko = GU.wrap_torch_ptr(res.cols[0].data, ng, torch.int64, d)
        so = GU.wrap_torch_ptr(res.cols[1].data, ng, torch.int64, d)
        co = GU.wrap_torch_ptr(res.cols[2].data, ng, torch.int64, d)
        assert int(co.min().item()) == 2 and int(co.max().item()) == 2
        assert int(ko.sum().item()) == want_k and int((ko * ko).sum().item()) == want_k2
        assert int(so.sum().item()) == want_x
    finally:
        a.close()
        del k, x
        torch.cuda.empty_cache()
        _lib.load().crgpu_mem_trim()


@pytest.mark.parametrize("xtype", [cd.INT64, cd.FLOAT64, cd.BYTES])
@pytest.mark.parametrize("ngroups", [5, 3000])
def test_distinct_aggregations_vs_oracle(oracle, xtype, ngroups):
    """DISTINCT aggregations (aggregators_util.go:298-373; the reference's DistinctAggregation and
    DistinctFilteringAggregation cases run with the golden tables): COUNT / SUM_INT / SUM / AVG over the first row of
    every (group, input) tuple among the rows that pass the aggregation's FILTER, NULL inputs and NULL filter values
    included, beside plain aggregations of the same columns, fed in several consume() calls."""
    from cockroach_amd import device as dev
    rng = np.random.default_rng(ngroups * 7 + xtype)
    n = 60_000
    g = cd.Vec.from_numpy(cd.INT64, rng.integers(0, ngroups, n, dtype=np.int64), rng.random(n) > 0.02)
    xi = rng.integers(0, 12, n, dtype=np.int64)
    xvalid = rng.random(n) > 0.1
    if xtype == cd.INT64:
        x = cd.Vec.from_numpy(cd.INT64, xi * 1_000_003 - 5, xvalid)
        fns = [F["COUNT"], F["SUM_INT"], F["SUM"]]
    elif xtype == cd.FLOAT64:
        x = cd.Vec.from_numpy(cd.FLOAT64, xi.astype(np.float64) * 0.25, xvalid)
        fns = [F["COUNT"], F["SUM"], F["AVG"]]
    else:
        x = cd.Vec.from_values(cd.BYTES, [None if not v else b"v%d" % k * (1 + k % 3) for k, v in zip(xi.tolist(), xvalid.tolist())])
        fns = [F["COUNT"]]
    flt = cd.Vec.from_numpy(cd.BOOL, rng.random(n) > 0.4, rng.random(n) > 0.1)
    cols = [g, x, flt]
    aggs = [(F["ANY_NOT_NULL"], 0, -1, False), (F["COUNT_ROWS"], -1, -1, False)]
    for fn in fns:
        aggs += [(fn, 1, -1, True), (fn, 1, 2, True), (fn, 1, -1, False)]
    want = H.vecs_to_tuples(oracle.hash_agg(cols, [0], aggs))
    a = dev.Agg([v.type for v in cols], [0], aggs)
    try:
        for lo in range(0, n, 16_384):
            part = [v.take(np.arange(lo, min(n, lo + 16_384))) for v in cols]
            a.consume([dev.upload(v) for v in part], len(part[0]))
        got = H.vecs_to_tuples(dev.result_to_vecs(a.finish()))
        H.assert_same_multiset(got, want)
        a.reset()  # the DISTINCT tables are reset with the aggregator
        a.consume([dev.upload(v.take(np.arange(0, 5000))) for v in cols], 5000)
        got = H.vecs_to_tuples(dev.result_to_vecs(a.finish()))
        H.assert_same_multiset(got, H.vecs_to_tuples(oracle.hash_agg([v.take(np.arange(0, 5000)) for v in cols], [0], aggs)))
    finally:
        a.close()


def test_distinct_aggregations_three_million_rows(oracle):
    """The DISTINCT composition beyond the size of a capped launch grid: 3 M rows in one call, 1000 groups, COUNT and
    SUM_INT DISTINCT with a FILTER, against numpy."""
    from cockroach_amd import device as dev
    rng = np.random.default_rng(88)
    n = 3_000_017
    g = rng.integers(0, 1000, n, dtype=np.int64)
    x = rng.integers(0, 500, n, dtype=np.int64)
    f = rng.integers(0, 2, n).astype(np.uint8)
    aggs = [(F["ANY_NOT_NULL"], 0, -1), (F["COUNT"], 1, -1, True), (F["SUM_INT"], 1, 2, True), (F["COUNT_ROWS"], -1, -1)]
    a = dev.Agg([cd.INT64, cd.INT64, cd.BOOL], [0], aggs)
    a.consume([dev.upload(cd.Vec.from_numpy(cd.INT64, g)), dev.upload(cd.Vec.from_numpy(cd.INT64, x)), dev.upload(cd.Vec.from_numpy(cd.BOOL, f))], n)
    got = {r[0]: r[1:] for r in H.vecs_to_tuples(dev.result_to_vecs(a.finish()))}
    a.close()
    assert len(got) == 1000
    pairs = np.unique(np.stack([g, x], 1), axis=0)
    cnt = np.bincount(pairs[:, 0], minlength=1000)
    fp = np.unique(np.stack([g[f != 0], x[f != 0]], 1), axis=0)
    fsum = np.bincount(fp[:, 0], weights=fp[:, 1], minlength=1000)
    rows = np.bincount(g, minlength=1000)
    for k in range(1000):
        assert got[k] == (int(cnt[k]), int(fsum[k]), int(rows[k])), k

