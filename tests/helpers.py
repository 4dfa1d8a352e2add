"""Shared helpers for the parity tests: fixture loading, tuple <-> column
conversion and the unordered (multiset) verifier the reference uses
(colexectestutils/utils.go:1411-1430,1522-1538: sort both sides, floats within
abs 1e-6, NaN == NaN)."""
from __future__ import annotations

import json
import math
import os

import numpy as np

from cockroach_amd import coldata as cd

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

JOIN_TYPE = {"INNER": 0, "LEFT_OUTER": 1, "RIGHT_OUTER": 2, "FULL_OUTER": 3, "LEFT_SEMI": 4, "LEFT_ANTI": 5,
             "INTERSECT_ALL": 6, "EXCEPT_ALL": 7, "RIGHT_SEMI": 8, "RIGHT_ANTI": 9}
AGG_FUNC = {"ANY_NOT_NULL": 0, "AVG": 1, "BOOL_AND": 2, "BOOL_OR": 3, "CONCAT_AGG": 4, "COUNT": 5, "MAX": 7,
            "MIN": 8, "SUM": 10, "SUM_INT": 11, "COUNT_ROWS": 14}
FLOAT_ABS_TOL = 1e-6      # colexectestutils/utils.go:1416
FLOAT_REL_TOL = 1e-9


def load_cases(name):
    with open(os.path.join(GOLDEN, name)) as f:
        return json.load(f)["cases"]


def type_id(name):
    # DECIMAL columns of the reference's aggregation tables are run as FLOAT64
    # (apd arithmetic is outside the GPU path); the expected
    # values are compared with the float tolerance.
    if name == "decimal":
        return cd.FLOAT64
    return cd.TYPE_BY_NAME[name]


def _coerce(typ, v):
    if v is None:
        return None
    if typ == cd.FLOAT64:
        return float(v)
    if typ == cd.BYTES:
        return v.encode() if isinstance(v, str) else bytes(v)
    if typ == cd.BOOL:
        return bool(v)
    return int(v)


def tuples_to_vecs(tuples, types):
    """list of rows -> list[Vec] (one per column)."""
    tids = [type_id(t) if isinstance(t, str) else t for t in types]
    cols = []
    for c, t in enumerate(tids):
        cols.append(cd.Vec.from_values(t, [_coerce(t, row[c]) for row in tuples]))
    return cols


def vecs_to_tuples(vecs, nrows=None):
    if not vecs:
        return [tuple() for _ in range(nrows or 0)]
    cols = [v.to_values() for v in vecs]
    return [tuple(col[i] for col in cols) for i in range(len(vecs[0]))]


def _sort_key(row):
    key = []
    for v in row:
        if v is None:
            key.append((0, 0))
        elif isinstance(v, float) and math.isnan(v):
            key.append((1, 0))
        elif isinstance(v, (bytes, str)):
            key.append((3, v if isinstance(v, bytes) else v.encode()))
        elif isinstance(v, bool):
            key.append((2, int(v)))
        else:
            # floats: tolerance-friendly primary key, exact value as the tie-break
            key.append((2, round(float(v), 5), v) if isinstance(v, float) else (2, v))
    return key


def _val_eq(a, b):
    if a is None or b is None:
        return a is None and b is None
    if isinstance(a, (bytes, str)) or isinstance(b, (bytes, str)):
        a = a.encode() if isinstance(a, str) else a
        b = b.encode() if isinstance(b, str) else b
        return a == b
    if isinstance(a, float) or isinstance(b, float):
        a, b = float(a), float(b)
        if math.isnan(a) or math.isnan(b):
            return math.isnan(a) and math.isnan(b)
        if math.isinf(a) or math.isinf(b):
            return a == b
        return abs(a - b) <= max(FLOAT_ABS_TOL, FLOAT_REL_TOL * max(abs(a), abs(b)))
    return a == b


def assert_same_multiset(actual, expected, msg=""):
    """UnorderedVerifier (colexectestutils/utils.go:1522-1538)."""
    actual = [tuple(r) for r in actual]
    expected = [tuple(r) for r in expected]
    assert len(actual) == len(expected), f"{msg}: row count {len(actual)} != {len(expected)}\nactual={actual[:20]}\nexpected={expected[:20]}"
    a = sorted(actual, key=_sort_key)
    e = sorted(expected, key=_sort_key)
    for ra, re_ in zip(a, e):
        assert len(ra) == len(re_), f"{msg}: arity {ra} vs {re_}"
        for x, y in zip(ra, re_):
            assert _val_eq(x, y), f"{msg}: {ra} != {re_}\nactual={a[:20]}\nexpected={e[:20]}"


# ---------------------------------------------------------------------------
# join cases
# ---------------------------------------------------------------------------
def left_in_output(jt):
    return jt not in ("RIGHT_SEMI", "RIGHT_ANTI")


def right_in_output(jt):
    return jt not in ("LEFT_SEMI", "LEFT_ANTI", "INTERSECT_ALL", "EXCEPT_ALL")


def mirror_case(c):
    """joinTestCase.mirror (joiner_utils_test.go:77-104): LEFT SEMI/ANTI -> RIGHT SEMI/ANTI."""
    if c["join_type"] not in ("LEFT_SEMI", "LEFT_ANTI"):
        return None
    m = dict(c)
    m["description"] = c["description"] + " (mirror)"
    m["join_type"] = "RIGHT_SEMI" if c["join_type"] == "LEFT_SEMI" else "RIGHT_ANTI"
    for a, b in (("left", "right"), ("left_types", "right_types"), ("left_out", "right_out"),
                 ("left_eq", "right_eq"), ("left_eq_cols_are_key", "right_eq_cols_are_key")):
        m[a], m[b] = c[b], c[a]
    return m


def join_cases():
    cases = []
    for f in ("hashjoiner_cases.json", "mergejoiner_cases.json"):
        for c in load_cases(f):
            if not c["supported"]:
                continue
            c = dict(c)
            c["id"] = f.split("_")[0] + ":" + c["description"]
            cases.append(c)
            m = mirror_case(c)
            if m is not None:
                m["id"] = c["id"] + " (mirror)"
                cases.append(m)
    return cases


def _eval_on_expr(expr, row):
    import re
    py = re.sub(r"@(\d+)", lambda m: f"r[{int(m.group(1)) - 1}]", expr).replace("AND", "and")
    try:
        return bool(eval(py, {"r": row}))  # noqa: S307 - fixture expressions only
    except TypeError:  # NULL operand -> not true
        return False


def project_join_output(case, out_rows):
    """Apply ON filter (planned as a filter above the joiner, execplan.go:1331)
    and the PostProcessSpec projection (hashjoiner_test.go:939-968)."""
    jt = case["join_type"]
    if case.get("on_expr"):
        out_rows = [r for r in out_rows if _eval_on_expr(case["on_expr"], r)]
    proj = list(case["left_out"]) if left_in_output(jt) else []
    roff = len(case["left_types"]) if left_in_output(jt) else 0
    if right_in_output(jt) or not left_in_output(jt):
        proj += [roff + c for c in case["right_out"]]
    return [tuple(r[i] for i in proj) for r in out_rows]


def expected_join_rows(case):
    types = []
    jt = case["join_type"]
    if left_in_output(jt):
        types += [case["left_types"][i] for i in case["left_out"]]
    if right_in_output(jt) or not left_in_output(jt):
        types += [case["right_types"][i] for i in case["right_out"]]
    rows = []
    for r in case["expected"]:
        rows.append(tuple(_coerce(type_id(t), v) for t, v in zip(types, r)))
    return rows


# ---------------------------------------------------------------------------
# aggregation cases
# ---------------------------------------------------------------------------
SUPPORTED_AGGS = {"ANY_NOT_NULL", "AVG", "BOOL_AND", "BOOL_OR", "COUNT", "MAX", "MIN", "SUM", "SUM_INT", "COUNT_ROWS"}


def agg_cases():
    out = []
    for f in ("aggregator_cases.json", "hash_aggregator_cases.json"):
        for c in load_cases(f):
            if c["has_unsupported_values"]:
                continue
            c = dict(c)
            c["id"] = f.split("_cases")[0] + ":" + c["name"]
            out.append(c)
    return out


def dropped_agg_pairs():
    """(case id, aggregation ordinal, function, reason) of everything the GPU harness leaves out of the reference's
    aggregation tables -- the list DESIGN.md section 6 prints; tests/test_oracle_agg.py pins it."""
    out = []
    for f in ("aggregator_cases.json", "hash_aggregator_cases.json"):
        for c in load_cases(f):
            cid = f.split("_cases")[0] + ":" + c["name"]
            if c["has_unsupported_values"]:
                out.append((cid, -1, "*", "value types outside the GPU path: " + ",".join(c["types"])))
                continue
            for i, fn in enumerate(c["agg_fns"]):
                col = c["agg_cols"][i][0] if c["agg_cols"][i] else -1
                t = c["types"][col] if col >= 0 else None
                if fn not in SUPPORTED_AGGS:
                    out.append((cid, i, fn, "function outside the GPU path"))
                elif fn in ("SUM", "AVG") and t != "float64" and type_id(t) != cd.FLOAT64:
                    out.append((cid, i, fn, f"{fn} over {t}: DECIMAL result (apd)"))
    return out


def prepare_agg_case(c):
    """Returns (cols, group_cols, aggs[(func, col, filter)], expected_rows) keeping
    only the aggregations in scope (no CONCAT_AGG, SUM/AVG over
    float only); expected columns are projected accordingly."""
    types = [type_id(t) for t in c["types"]]
    cols = tuples_to_vecs(c["input"], types)
    keep, aggs = [], []
    for i, fn in enumerate(c["agg_fns"]):
        if fn not in SUPPORTED_AGGS:
            continue
        col = c["agg_cols"][i][0] if c["agg_cols"][i] else -1
        if fn in ("SUM", "AVG") and types[col] != cd.FLOAT64:
            continue  # SUM/AVG over INT => DECIMAL result (apd), out of scope
        flt = c["agg_filter"][i] if c.get("agg_filter") else -1
        keep.append(i)
        aggs.append((AGG_FUNC[fn], col, flt, bool(c.get("agg_distinct") and c["agg_distinct"][i])))
    c.setdefault("_dropped", [(i, fn) for i, fn in enumerate(c["agg_fns"]) if i not in keep])
    expected = []
    for r in c["expected"]:
        row = []
        for i in keep:
            v = r[i]
            if isinstance(v, str) and c["agg_fns"][i] in ("AVG", "SUM", "MIN", "MAX") and \
                    c["types"][c["agg_cols"][i][0]] == "decimal":
                v = float(v)
            if isinstance(v, str):
                v = v.encode()
            row.append(v)
        expected.append(tuple(row))
    return cols, list(c["group_cols"]), aggs, expected


# ---------------------------------------------------------------------------
# KV fetch-decode cases
# ---------------------------------------------------------------------------
def kv_random_rows(rng, n, key_types, val_types, null_frac=0.15):
    """Random key / value columns of the given type ids (NULLs in both)."""
    def col(t, allow_null=True):
        pres = rng.random(n) >= (null_frac if allow_null else 0.0)
        if t == cd.BOOL:
            return cd.Vec.from_numpy(cd.BOOL, rng.integers(0, 2, n).astype(np.uint8), pres)
        if t in (cd.INT16, cd.INT32, cd.INT64):
            bits = {cd.INT16: 15, cd.INT32: 31, cd.INT64: 63}[t]
            mag = rng.integers(0, bits + 1, n)  # spread over every encoded length
            v = (rng.integers(0, 2**62, n, dtype=np.int64) >> (62 - np.minimum(mag, 62))) * rng.choice([-1, 1], n)
            if t == cd.INT64:
                v[: min(n, 4)] = np.array([np.iinfo(np.int64).min, np.iinfo(np.int64).max, 0, -1])[: min(n, 4)]
            return cd.Vec.from_numpy(t, v.astype(cd.NP_DTYPE[t]), pres)
        if t == cd.FLOAT64:
            v = rng.standard_normal(n) * 10.0 ** rng.integers(-5, 6, n)
            sp = np.array([0.0, -0.0, np.nan, np.inf, -np.inf, 5e-324, -1.7976931348623157e308])
            v[: min(n, len(sp))] = sp[: min(n, len(sp))]
            return cd.Vec.from_numpy(cd.FLOAT64, v, pres)
        words = [b"", b"a", b"\x00", b"a\x00b", b"\xff\x00\xff", b"hello world", b"x" * 40, bytes(range(256))]
        vals = [None if not pres[i] else words[rng.integers(len(words))] + (b"%d" % i if i % 3 else b"") for i in range(n)]
        return cd.Vec.from_values(cd.BYTES, vals)
    return [col(t) for t in key_types], [col(t) for t in val_types]


def assert_vecs_equal(got, want, msg=""):
    """Bit-exact column comparison (NULL slots compare by validity only)."""
    assert len(got) == len(want), f"{msg}: {len(got)} columns != {len(want)}"
    for c, (g, w) in enumerate(zip(got, want)):
        assert g.type == w.type and len(g) == len(w), f"{msg}: column {c} type/len {g.type}/{len(g)} vs {w.type}/{len(w)}"
        gp, wp = g.present(), w.present()
        assert np.array_equal(gp, wp), f"{msg}: column {c} validity differs at rows {np.nonzero(gp != wp)[0][:10]}"
        if g.type == cd.BYTES:
            assert np.array_equal(g.offsets, w.offsets), f"{msg}: column {c} offsets differ"
            assert np.array_equal(g.data, w.data), f"{msg}: column {c} bytes differ"
        else:
            gv = g.data.view(np.uint8).reshape(len(g), -1)[gp]
            wv = w.data.view(np.uint8).reshape(len(w), -1)[wp]
            assert np.array_equal(gv, wv), f"{msg}: column {c} values differ"


def kv_key_canonical(vec):
    """What a key-encoded column decodes to: float keys lose the sign of zero and the NaN payload
    (float.go:36-39 "encodes both positive and negative zero the same")."""
    if vec.type != cd.FLOAT64:
        return vec
    d = vec.data.copy()
    d[d == 0] = 0.0
    d[np.isnan(d)] = np.nan
    return cd.Vec(cd.FLOAT64, d, vec.valid)


# ---------------------------------------------------------------------------
# partially ordered hash aggregation
# ---------------------------------------------------------------------------
def ordered_agg_cases():
    """The cases of hash_aggregator_test.go:104-397 that set orderedCols."""
    return [c for c in agg_cases() if c.get("ordered_cols")]


def assert_partial_order(actual, expected, ordered_positions, msg=""):
    """PartialOrderedVerifier (colexectestutils/utils.go:1337-1400): both row lists are cut where the values in the
    ordered columns change; the runs must come in the same order and hold the same multiset of rows."""
    def runs(rows):
        out, key = [], object()
        for r in rows:
            k = tuple(_sort_key((r[p],)) for p in ordered_positions)
            if k != key:
                out.append([])
                key = k
            out[-1].append(tuple(r))
        return out
    ra, re_ = runs(actual), runs(expected)
    assert len(ra) == len(re_), f"{msg}: {len(ra)} runs of ordered values, expected {len(re_)}\nactual={actual[:30]}\nexpected={expected[:30]}"
    for i, (a, e) in enumerate(zip(ra, re_)):
        assert_same_multiset(a, e, f"{msg}: run {i}")
