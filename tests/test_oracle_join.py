"""Pins the oracle's hashJoiner against the reference's literal case tables
(pkg/sql/colexec/hashjoiner_test.go:50-925 + the merge-join table that
TestHashJoiner also runs, :1022) under the runner matrix of
colexectestutils.RunTestsWithTyps (utils.go:656-698): batch sizes
{1, 2, 1024}, shuffled inputs, unordered verifier."""
import random

import pytest

from tests import helpers as H

CASES = H.join_cases()


def run_oracle_join(oracle, case, batch_size, shuffle_seed=None):
    left_rows, right_rows = list(case["left"]), list(case["right"])
    if shuffle_seed is not None:
        rng = random.Random(shuffle_seed)
        rng.shuffle(left_rows)
        rng.shuffle(right_rows)
    left = H.tuples_to_vecs(left_rows, case["left_types"])
    right = H.tuples_to_vecs(right_rows, case["right_types"])
    out = oracle.hash_join(H.JOIN_TYPE[case["join_type"]], left, right, case["left_eq"], case["right_eq"],
                           case["right_eq_cols_are_key"], batch_size=batch_size)
    return H.project_join_output(case, H.vecs_to_tuples(out))


@pytest.mark.parametrize("case", CASES, ids=[c["id"] for c in CASES])
def test_join_case(oracle, case):
    expected = H.expected_join_rows(case)
    for bs in (1, 2, 1024):
        for seed in (None, 7):
            got = run_oracle_join(oracle, case, bs, seed)
            H.assert_same_multiset(got, expected, f"{case['id']} bs={bs} seed={seed}")


def test_counts():
    # 27 hash-join cases (one DECIMAL case out of scope) + 110 merge-join cases + SEMI/ANTI mirrors
    assert len([c for c in CASES if c["id"].startswith("hashjoiner")]) >= 26 + 2
    assert len([c for c in CASES if c["id"].startswith("mergejoiner")]) >= 110
