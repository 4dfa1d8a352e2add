"""CPU: the Q3 / Q9 plans (workloads/tpch_plans.py) run by the oracle's executor against an independent evaluation
(tests/tpch_check.py: dense-key lookups, no hash table), and the synthetic tables' key structure."""
import numpy as np
import torch

from tests import tpch_check as K
from workloads import tpch_plans as P
from workloads import tpch_synth as T


def _tables(sf, seed=42):
    t, rows = T.generate(sf, "cpu", seed)
    return t, rows, rows["_aux"]


def test_synthetic_tables_key_structure():
    t, rows, aux = _tables(0.02)
    ps = t["partsupp"]
    key = ps["ps_partkey"] * (1 << 32) + ps["ps_suppkey"]
    assert key.unique().numel() == key.numel()  # (ps_partkey, ps_suppkey) is a key: Q9's two-column join is on a key
    li = t["lineitem"]
    assert torch.isin(li["l_partkey"] * (1 << 32) + li["l_suppkey"], key).all()
    o = t["orders"]
    assert o["o_orderkey"].unique().numel() == rows["orders"]
    assert (aux["order_index_of_key"](o["o_orderkey"]) == torch.arange(rows["orders"])).all()
    assert int((o["o_custkey"] % 3 == 0).sum()) == 0 and int(o["o_custkey"].max()) <= rows["customer"]
    assert 3.5 < rows["lineitem"] / rows["orders"] < 4.5
    names = t["part"]["p_name"]
    d, off = bytes(names.data.numpy()), names.offsets.numpy()
    flags = np.array([b"green" in d[off[i]:off[i + 1]] for i in range(len(names))])
    assert np.array_equal(flags, aux["part_is_green"].numpy())
    assert 0.03 < flags.mean() < 0.08  # 5 of 92 colour words


def test_q3_plan_on_the_oracle_matches_independent_evaluation(oracle):
    from oracle.plan_exec import OracleExec
    t, rows, aux = _tables(0.02)
    ex = OracleExec()
    tabs = {k: ex.table(v) for k, v in t.items()}
    for alt in (False, True):
        res = P.q3(ex, tabs, build_orders_by_custkey=alt)
        keys, rev, odate = K.q3_expected(t, aux)
        assert len(res[0]) == keys.numel() > 50
        order = np.argsort(res[0].data)
        assert np.array_equal(res[0].data[order], keys.numpy())
        assert np.array_equal(res[1].data[order], odate.numpy())
        assert np.all(res[2].data == 0)
        assert np.allclose(res[3].data[order], rev.numpy(), rtol=1e-12, atol=1e-6)


def test_q9_plan_on_the_oracle_matches_independent_evaluation(oracle):
    from oracle.plan_exec import OracleExec
    t, rows, aux = _tables(0.02)
    ex = OracleExec()
    tabs = {k: ex.table(v) for k, v in t.items()}
    res = P.q9(ex, tabs)
    nat, year, profit = K.q9_expected(t, aux)
    got = sorted(zip(res[0].to_values(), res[1].to_values(), res[2].to_values()))
    want = sorted(zip([T.NATIONS[i] for i in nat.tolist()], [float(y) for y in year.tolist()], profit.tolist()))
    assert len(got) == len(want) and len(got) > 100
    for g, w in zip(got, want):
        assert g[0] == w[0] and g[1] == w[1]
        assert abs(g[2] - w[2]) <= 1e-9 * abs(w[2]) + 1e-6
