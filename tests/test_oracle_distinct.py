"""Pins the oracle's UnorderedDistinct (oracle/crdb_oracle.c: ora_unordered_distinct, restating
pkg/sql/colexec/unordered_distinct.go:94-121 and colexechash/hashtable.go:628-716) against the reference's literal
case table (pkg/sql/colexec/distinct_test.go:45-330; tests/golden/distinct_cases.json).  The reference runs these
through an ORDERED verifier: the distinct operator emits first occurrences in input order, so the comparison here is
ordered too.  The JSON-typed case is outside the GPU type set and skipped."""
import numpy as np
import pytest

from cockroach_amd import coldata as cd
from tests import helpers as H

CASES = [dict(c, id=f"distinct{i}") for i, c in enumerate(H.load_cases("distinct_cases.json")) if c["supported"]]


def ordered_equal(actual, expected, msg=""):
    assert len(actual) == len(expected), f"{msg}: {actual} != {expected}"
    for ra, re_ in zip(actual, expected):
        assert len(ra) == len(re_) and all(H._val_eq(x, y) for x, y in zip(ra, re_)), f"{msg}: {actual} != {expected}"


@pytest.mark.parametrize("case", CASES, ids=[c["id"] for c in CASES])
def test_distinct_case(oracle, case):
    cols = H.tuples_to_vecs(case["tuples"], case["types"])
    for bs in (1, 2, 3, 1024):
        for nb in (1, 2, 128):
            out, dropped = oracle.unordered_distinct(cols, case["distinct_cols"], case["nulls_are_distinct"], batch_size=bs,
                                                     initial_buckets=nb)
            if case["error_on_dup"]:
                # distinct_test.go:336-372: the error must (not) be raised
                assert dropped == (not case["no_error"]), f"{case['id']} bs={bs}"
                if not case["no_error"]:
                    continue
            ordered_equal(H.vecs_to_tuples(out), case["expected"], f"{case['id']} bs={bs} nb={nb}")


@pytest.mark.parametrize("nulls_are_distinct", [False, True])
def test_distinct_random_vs_python_model(oracle, nulls_are_distinct):
    """Random input with NULLs, NaN and -0.0 against a dict-based model of 'first occurrence in input order'."""
    rng = np.random.default_rng(9 + int(nulls_are_distinct))
    n = 20_000
    a = rng.integers(0, 40, size=n, dtype=np.int64)
    f = rng.choice(np.array([0.0, -0.0, 1.5, float("nan"), -2.0]), size=n)
    s = [bytes([65 + int(x)]) * int(y) for x, y in zip(rng.integers(0, 3, size=n), rng.integers(0, 4, size=n))]
    w = np.arange(n, dtype=np.int64)
    cols = [cd.Vec.from_numpy(cd.INT64, a, rng.random(n) > 0.1), cd.Vec.from_numpy(cd.FLOAT64, f, rng.random(n) > 0.1),
            cd.Vec.from_values(cd.BYTES, s), cd.Vec.from_numpy(cd.INT64, w)]
    out, _ = oracle.unordered_distinct(cols, [0, 1, 2], nulls_are_distinct, batch_size=777)
    seen, want = set(), []
    av, fv = cols[0].to_values(), cols[1].to_values()
    for i in range(n):
        fk = None if fv[i] is None else ("nan" if fv[i] != fv[i] else fv[i] + 0.0)
        key = (av[i], fk, s[i])
        if nulls_are_distinct and (av[i] is None or fv[i] is None):
            want.append(i)
        elif key not in seen:
            seen.add(key)
            want.append(i)
    assert out[3].to_values() == want
