"""Selection / projection kernels (cockroach_amd/csrc/expr.hip) through the C-ABI against the
oracle's restatement, the one literal case the reference holds for these overloads (selection_ops_test.go:28-41:
{0},{1},{2},{NULL} < 2 keeps {0},{1}), and the Q1 plan `sel -> projections -> hash aggregation` against the oracle
run the reference's way (selection first, then aggregation of the surviving rows).  The FLOAT64 +,-,* overloads have
no literal vectors in the reference (only Div and the integer Plus do): they are IEEE double operations, compared
bit for bit with numpy."""
import numpy as np
import pytest

from cockroach_amd import coldata as cd
from tests import helpers as H

pytestmark = pytest.mark.gpu


def test_select_reference_case_and_random(oracle):
    from cockroach_amd import device as dev
    v = cd.Vec.from_values(cd.INT64, [0, 1, 2, None])
    got = dev.select_i64_const("LT", dev.upload(v), 2)
    assert dev.download(got.data_ptr, np.uint8, 4).tolist() == [1, 1, 0, 0]
    assert oracle.select_i64_const("LT", v, 2).tolist() == [0, 1]
    rng = np.random.default_rng(4)
    n = 100_003
    x = cd.Vec.from_numpy(cd.INT64, rng.integers(-50, 50, size=n, dtype=np.int64), rng.random(n) > 0.1)
    for cmp in ("EQ", "NE", "LT", "LE", "GT", "GE"):
        col = dev.select_i64_const(cmp, dev.upload(x), 7)  # keep the column alive: it owns the device buffer
        got = dev.download(col.data_ptr, np.uint8, n)
        assert np.flatnonzero(got).tolist() == oracle.select_i64_const(cmp, x, 7).tolist(), cmp


@pytest.mark.parametrize("n", [1, 2, 7, 100_001])
def test_project_f64_bit_exact(oracle, n):
    from cockroach_amd import device as dev
    rng = np.random.default_rng(n)
    special = np.array([0.0, -0.0, np.inf, -np.inf, np.nan, 1e308, -1e308, 5e-324])
    a = np.where(rng.random(n) < 0.1, rng.choice(special, n), rng.normal(size=n) * 1e3)
    b = np.where(rng.random(n) < 0.1, rng.choice(special, n), rng.normal(size=n))
    va = cd.Vec.from_numpy(cd.FLOAT64, a, rng.random(n) > 0.2)
    vb = cd.Vec.from_numpy(cd.FLOAT64, b)
    da, db = dev.upload(va), dev.upload(vb)
    for op in ("+", "-", "*", "*(1-)", "*(1+)"):
        for x, y, dx, dy in ((va, vb, da, db), (1.0, vb, 1.0, db), (va, 0.25, da, 0.25), (vb, vb, db, db)):
            got = dev.project_f64(op, dx, dy)
            want = oracle.project_f64(op, x, y)
            g = dev.download(got.data_ptr, np.float64, n)
            pres = want.present()
            gp, wp = g[pres], want.data[pres]
            nan = np.isnan(wp)
            assert np.array_equal(np.isnan(gp), nan), (op, "NaN positions")
            assert np.array_equal(gp[~nan].view(np.uint64), wp[~nan].view(np.uint64)), (op, "bit patterns")  # incl. -0.0, inf
            if got.valid_ptr:
                bm = dev.download(got.valid_ptr, np.uint8, (n + 7) // 8)
                assert np.array_equal(np.unpackbits(bm, bitorder="little")[:n].astype(bool), pres)
            else:
                assert pres.all()


def test_q1_plan_select_project_aggregate_vs_oracle(oracle):
    """tpch_vec:20537-20546: WHERE l_shipdate <= c; disc_price = price * (1 - disc); charge = disc_price * (1 + tax);
    GROUP BY returnflag, linestatus with SUMs, AVGs and COUNT_ROWS.  A (returnflag, linestatus) pair whose rows all
    fail the WHERE must not appear."""
    from cockroach_amd import device as dev
    rng = np.random.default_rng(8)
    n = 200_000
    ship = rng.integers(8000, 10600, size=n, dtype=np.int64)
    rf = rng.integers(0, 3, size=n)
    ls = rng.integers(0, 2, size=n)
    ship[(rf == 2) & (ls == 1)] = 11_000  # this group never passes
    cols = [cd.Vec.from_numpy(cd.INT64, ship, rng.random(n) > 0.01),
            cd.Vec.from_values(cd.BYTES, [b"ANR"[i:i + 1] for i in rf]), cd.Vec.from_values(cd.BYTES, [b"FO"[i:i + 1] for i in ls]),
            cd.Vec.from_numpy(cd.FLOAT64, rng.integers(1, 51, size=n).astype(np.float64)),
            cd.Vec.from_numpy(cd.FLOAT64, np.round(rng.random(n) * 1e5, 2), rng.random(n) > 0.02),
            cd.Vec.from_numpy(cd.FLOAT64, rng.integers(0, 11, size=n) / 100.0),
            cd.Vec.from_numpy(cd.FLOAT64, rng.integers(0, 9, size=n) / 100.0)]
    cutoff = 10_471
    aggs = [(0, 1, -1), (0, 2, -1), (10, 3, -1), (10, 4, -1), (10, 7, -1), (10, 8, -1), (1, 3, -1), (1, 4, -1), (1, 5, -1), (14, -1, -1)]
    # the reference's order of operations, on the CPU
    keep = oracle.select_i64_const("LE", cols[0], cutoff)
    sel = [v.take(keep) for v in cols]
    disc_price = oracle.project_f64("*", sel[4], oracle.project_f64("-", 1.0, sel[5]))
    charge = oracle.project_f64("*", disc_price, oracle.project_f64("+", 1.0, sel[6]))
    want = oracle.hash_agg(sel + [disc_price, charge], [1, 2], aggs)
    # the device plan: nothing is moved, the selection is a FILTER
    d = [dev.upload(v) for v in cols]
    where = dev.select_i64_const("LE", d[0], cutoff)
    g_disc_price = dev.project_f64("*", d[4], dev.project_f64("-", 1.0, d[5]))
    g_charge = dev.project_f64("*", g_disc_price, dev.project_f64("+", 1.0, d[6]))
    fa = dev.FilteredAgg([v.type for v in cols] + [cd.FLOAT64, cd.FLOAT64], [1, 2], aggs)
    half = n // 2
    for lo, hi in ((0, half), (half, n)):
        part = [dev.upload(v.slice(lo, hi)) for v in cols]
        w = dev.select_i64_const("LE", part[0], cutoff)
        dp = dev.project_f64("*", part[4], dev.project_f64("-", 1.0, part[5]))
        ch = dev.project_f64("*", dp, dev.project_f64("+", 1.0, part[6]))
        fa.consume(part + [dp, ch], w, hi - lo)
    got = fa.finish_vecs()
    fa.close()
    assert len(got[0]) == 5, "the group whose rows all fail the WHERE must be dropped"
    H.assert_same_multiset(H.vecs_to_tuples(got), H.vecs_to_tuples(want), "Q1 plan")
    del where, g_charge


def _q1_inputs(rng, n, null_frac=0.02):
    ship = rng.integers(8000, 10600, size=n, dtype=np.int64)
    rf = rng.integers(0, 3, size=n)
    ls = rng.integers(0, 2, size=n)
    ship[(rf == 2) & (ls == 1)] = 11_000  # this group never passes
    nulls = (lambda: rng.random(n) > null_frac) if null_frac else (lambda: None)
    return [cd.Vec.from_numpy(cd.INT64, ship, nulls()),
            cd.Vec.from_values(cd.BYTES, [b"ANR"[i:i + 1] for i in rf]), cd.Vec.from_values(cd.BYTES, [b"FO"[i:i + 1] for i in ls]),
            cd.Vec.from_numpy(cd.FLOAT64, rng.integers(1, 51, size=n).astype(np.float64)),
            cd.Vec.from_numpy(cd.FLOAT64, np.round(rng.random(n) * 1e5, 2), nulls()),
            cd.Vec.from_numpy(cd.FLOAT64, rng.integers(0, 11, size=n) / 100.0, nulls()),
            cd.Vec.from_numpy(cd.FLOAT64, rng.integers(0, 9, size=n) / 100.0, nulls())]


@pytest.mark.parametrize("null_frac", [0.0, 0.03])
def test_q1_plan_fused_into_the_aggregation_vs_oracle(oracle, null_frac):
    """The same Q1 plan (tpch_vec:20537-20546) through crgpu_dev_agg_create_fused: the selection and both projections are
    evaluated inside the aggregation kernels, nothing is materialised.  Compared with the reference's order of
    operations on the CPU (select, project with separate roundings, aggregate), MIN / MAX / COUNT over the computed
    columns and NULL operands included; fed in three chunks so that the first one goes through find-or-insert (the
    claimants' first rows evaluate the expressions in the fixup kernel) and the later ones through the fused pass."""
    from cockroach_amd import device as dev
    rng = np.random.default_rng(80)
    n = 300_000
    cols = _q1_inputs(rng, n, null_frac)
    cutoff = 10_471
    # ordinals: 0 ship, 1 rf, 2 ls, 3 qty, 4 price, 5 disc, 6 tax | computed: 7 disc_price, 8 charge, 9 where
    aggs = [(0, 1, -1), (0, 2, -1), (10, 3, -1), (10, 4, -1), (10, 7, -1), (10, 8, -1), (1, 3, -1), (1, 4, -1), (1, 5, -1), (14, -1, -1),
            (7, 8, -1), (8, 7, -1), (5, 8, -1), (1, 8, -1)]
    keep = oracle.select_i64_const("LE", cols[0], cutoff)
    sel = [v.take(keep) for v in cols]
    disc_price = oracle.project_f64("*", sel[4], oracle.project_f64("-", 1.0, sel[5]))
    charge = oracle.project_f64("*", disc_price, oracle.project_f64("+", 1.0, sel[6]))
    want = oracle.hash_agg(sel + [disc_price, charge], [1, 2], aggs)
    vcols = [("project", "*(1-)", 4, 5), ("project", "*(1+)", 7, 6), ("select", "LE", 0, cutoff)]
    fa = dev.FusedAgg([v.type for v in cols], [1, 2], aggs, vcols, where=9)
    for lo, hi in ((0, 1000), (1000, 100_000), (100_000, n)):
        fa.consume([dev.upload(v.slice(lo, hi)) for v in cols], hi - lo)
    got = fa.finish_vecs()
    assert len(got[0]) == 5, "the group whose rows all fail the WHERE must be dropped"
    H.assert_same_multiset(H.vecs_to_tuples(got), H.vecs_to_tuples(want), "Q1 plan, fused")
    # Reset and once more in one piece
    fa.reset()
    fa.consume([dev.upload(v) for v in cols], n)
    H.assert_same_multiset(H.vecs_to_tuples(fa.finish_vecs()), H.vecs_to_tuples(want), "Q1 plan, fused, after Reset")
    fa.close()


def test_fused_agg_many_groups_and_per_aggregate_filters(oracle):
    """Computed inputs beyond the fused LDS pass: 50 000 groups (index lookup + HBM atomics), two different selections
    as FILTERs of different aggregates (no common filter), every projection operator."""
    from cockroach_amd import device as dev
    from cockroach_amd import _lib
    rng = np.random.default_rng(81)
    n = 400_000
    k = cd.Vec.from_numpy(cd.INT64, rng.integers(0, 50_000, size=n, dtype=np.int64))
    d = cd.Vec.from_numpy(cd.INT64, rng.integers(-50, 50, size=n, dtype=np.int64), rng.random(n) > 0.05)
    x = cd.Vec.from_numpy(cd.FLOAT64, rng.standard_normal(n) * 100)
    y = cd.Vec.from_numpy(cd.FLOAT64, rng.random(n), rng.random(n) > 0.05)
    cols = [k, d, x, y]
    # computed: 4 x+y, 5 x-y, 6 x*y, 7 (x+y)*y, 8 d < 0, 9 d >= 10
    vcols = [("project", "+", 2, 3), ("project", "-", 2, 3), ("project", "*", 2, 3), ("project", "*", 4, 3),
             ("select", "LT", 1, 0), ("select", "GE", 1, 10)]
    aggs = [(0, 0, -1), (10, 4, -1), (10, 5, 8), (10, 6, 9), (1, 7, -1), (14, -1, 8), (5, 6, 9), (8, 7, 8)]
    pf = oracle.project_f64
    host = cols + [pf("+", x, y), pf("-", x, y), pf("*", x, y), pf("*", pf("+", x, y), y)]
    for cmp, c in (("LT", 0), ("GE", 10)):
        m = np.zeros(n, dtype=np.uint8)
        m[oracle.select_i64_const(cmp, d, c)] = 1
        host.append(cd.Vec.from_numpy(cd.BOOL, m))
    want = oracle.hash_agg(host, [0], aggs)
    fa = dev.FusedAgg([v.type for v in cols], [0], aggs, vcols)
    for lo, hi in ((0, 150_000), (150_000, n)):
        fa.consume([dev.upload(v.slice(lo, hi)) for v in cols], hi - lo)
    H.assert_same_multiset(H.vecs_to_tuples(fa.finish_vecs()), H.vecs_to_tuples(want), "fused, many groups")
    fa.close()
    with pytest.raises(_lib.CrgpuError) as e:  # three chained projections
        dev.FusedAgg([cd.FLOAT64, cd.FLOAT64], [], [(10, 4, -1)], [("project", "+", 0, 1), ("project", "*", 2, 1), ("project", "-", 3, 1)])
    assert e.value.code == _lib.ERR_UNSUPPORTED
    with pytest.raises(_lib.CrgpuError) as e:  # grouping by a computed column
        dev.FusedAgg([cd.FLOAT64, cd.FLOAT64], [2], [(14, -1, -1)], [("project", "+", 0, 1)])
    assert e.value.code == _lib.ERR_UNSUPPORTED


def test_select_bytes_const_vs_oracle(oracle):
    """selEQBytesBytesConstOp / selNEBytesBytesConstOp / selContainsBytesBytesConstOp (Q3: c_mktsegment = 'BUILDING',
    Q9: p_name LIKE '%green%').  The reference holds no literal vectors for the BYTES overloads (its selection tests
    are int64 only, selection_ops_test.go:28-41): byte-string equality / containment are compared with Python's."""
    from cockroach_amd import device as dev
    rng = np.random.default_rng(11)
    segs = [b"AUTOMOBILE", b"BUILDING", b"FURNITURE", b"MACHINERY", b"HOUSEHOLD", b"", b"BUILDIN", b"BUILDINGS", b"xBUILDING"]
    n = 50_021
    vals = [segs[i] for i in rng.integers(0, len(segs), size=n)]
    for i in rng.integers(0, n, size=n // 20):
        vals[i] = None
    v = cd.Vec.from_values(cd.BYTES, vals)
    d = dev.upload(v)
    for cmp, cst in (("EQ", b"BUILDING"), ("NE", b"BUILDING"), ("CONTAINS", b"BUILD"), ("NOT CONTAINS", b"INGS"), ("EQ", b""), ("CONTAINS", b"")):
        col = dev.select_bytes_const(cmp, d, cst)
        got = dev.download(col.data_ptr, np.uint8, n)
        assert np.flatnonzero(got).tolist() == oracle.select_bytes_const(cmp, v, cst).tolist(), (cmp, cst)
    colors = [b"green", b"red", b"forest green", b"gree", b"reen", b"dark greenish lime", b"grEen", b"g r e e n"]
    names = cd.Vec.from_values(cd.BYTES, [b" ".join(colors[j] for j in rng.integers(0, len(colors), size=3)) for _ in range(5000)])
    col = dev.select_bytes_const("CONTAINS", dev.upload(names), b"green")
    assert np.flatnonzero(dev.download(col.data_ptr, np.uint8, 5000)).tolist() == oracle.select_bytes_const("CONTAINS", names, b"green").tolist()


def test_extract_year_vs_calendar(oracle):
    """EXTRACT(year FROM date): every day of 1968..2004 (leap years, century rule at 2000), far dates, negative days."""
    from cockroach_amd import device as dev
    days = np.concatenate([np.arange(-800, 13000, dtype=np.int64),
                           np.array([-719162, -719163, -1, 0, 365, 366, 10957, 11016, 11322, 11323, 47482, 2932896, -141427], dtype=np.int64)])
    v = cd.Vec.from_numpy(cd.INT64, days)
    got = dev.extract_year(dev.upload(v))
    g = dev.download(got.data_ptr, np.float64, len(days))
    want = oracle.extract_year(v)
    assert np.array_equal(g, want.data)
    # known answers: 1970-01-01 = day 0; 2000-02-29 = day 11016; 1992-01-01 = day 8035; 1998-12-01 = day 10561
    probe = cd.Vec.from_numpy(cd.INT64, np.array([0, 11016, 8035, 10561, -1], dtype=np.int64))
    col = dev.extract_year(dev.upload(probe))  # keep the column alive: it owns the device buffer
    g = dev.download(col.data_ptr, np.float64, 5)
    assert g.tolist() == [1970.0, 2000.0, 1992.0, 1998.0, 1969.0]


def test_deselect_vs_oracle(oracle):
    """deselectorOp: the rows a BOOL column keeps, all column families, NULLs preserved, input order kept."""
    from cockroach_amd import device as dev
    rng = np.random.default_rng(3)
    for n in (0, 1, 4097, 60_001):
        k = cd.Vec.from_numpy(cd.INT64, rng.integers(-9, 9, size=n, dtype=np.int64), rng.random(n) > 0.1)
        s = cd.Vec.from_values(cd.BYTES, [None if i % 13 == 5 else b"row%d" % i * (i % 3) for i in range(n)])
        f = cd.Vec.from_numpy(cd.FLOAT64, rng.standard_normal(n))
        b = cd.Vec.from_numpy(cd.BOOL, rng.random(n) > 0.5)
        i32 = cd.Vec.from_numpy(cd.INT32, rng.integers(-5, 5, size=n).astype(np.int32))
        cols = [k, s, f, b, i32]
        keep = (rng.random(n) < 0.3).astype(np.uint8)
        ds = dev.Deselector([c.type for c in cols])
        flags = dev.upload(cd.Vec.from_numpy(cd.BOOL, keep))
        res = ds.deselect([dev.upload(c) for c in cols], flags, n)
        got = dev.result_to_vecs(res)
        want = oracle.deselect(cols, np.flatnonzero(keep))
        assert int(res.n) == int(keep.sum())
        for gv, wv in zip(got, want):
            assert gv.to_values() == wv.to_values()
        ds.close()


@pytest.mark.parametrize("shift", [0, 1, 5, 16])
@pytest.mark.parametrize("density", [0.0, 0.3, 1.0])
def test_deselect_flag_array_at_any_alignment(oracle, shift, density):
    """The stream compaction behind the deselector reads sixteen flags per load when the flag array is 16-byte aligned
    and byte by byte otherwise (gather.hip: tile_sum_flags16_kernel / tile_sum_kernel): the same rows either way, for
    no row kept, some and all of them, and a row count that ends inside a tile and inside a 16-flag group."""
    from cockroach_amd import device as dev
    rng = np.random.default_rng(shift * 10 + int(density * 10))
    n = 3 * 4096 + 1234 + 7
    vals = cd.Vec.from_numpy(cd.INT64, rng.integers(-10**12, 10**12, size=n, dtype=np.int64))
    keep = (rng.random(n) < density).astype(np.uint8) if 0.0 < density < 1.0 else np.full(n, int(density), dtype=np.uint8)
    keep[keep != 0] = rng.integers(1, 256, size=int((keep != 0).sum()), dtype=np.uint8)  # any nonzero byte is "true"
    whole = dev.upload(cd.Vec.from_numpy(cd.BOOL, np.concatenate([np.zeros(shift, dtype=np.uint8), keep])))
    flags = dev.DeviceColumn(cd.BOOL, n, whole.data_ptr + shift, None, None, keep=whole)
    ds = dev.Deselector([cd.INT64])
    res = ds.deselect([dev.upload(vals)], flags, n)
    got = dev.result_to_vecs(res)
    assert int(res.n) == int((keep != 0).sum())
    assert np.array_equal(got[0].data, vals.data[keep != 0])
    ds.close()

