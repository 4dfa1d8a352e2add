"""Independent evaluation of Q3 / Q9 on the synthetic TPC-H tables (tests only): no hash table anywhere -- dense key
lookups with torch, from the generator's ground-truth flags -- so that it checks the PLAN (workloads/tpch_plans.py)
and, at full size, the GPU chain, without going through the oracle."""
import torch

from workloads import tpch_synth as T


def q3_expected(t, aux):
    """Returns (number of groups, sum over groups of revenue, per-order revenue dict for small inputs)."""
    o, l = t["orders"], t["lineitem"]
    building = aux["cust_is_building"]
    order_ok = (o["o_orderdate"] < T.DAY_1995_03_15) & building[o["o_custkey"] - 1]
    oi = aux["order_index_of_key"](l["l_orderkey"])
    line_ok = (l["l_shipdate"] > T.DAY_1995_03_15) & order_ok[oi]
    rev = (l["l_extendedprice"] * (1.0 - l["l_discount"]))[line_ok]
    keys = l["l_orderkey"][line_ok]
    uniq, inv = torch.unique(keys, return_inverse=True)
    per = torch.zeros(uniq.numel(), dtype=torch.float64, device=rev.device).index_add_(0, inv, rev)
    return uniq, per, o["o_orderdate"][aux["order_index_of_key"](uniq)]


def q9_expected(t, aux):
    """Returns (nation id, year, sum_profit) per group, sorted by (nation id, year)."""
    l, ps, s, o = t["lineitem"], t["partsupp"], t["supplier"], t["orders"]
    S = s["s_suppkey"].numel()
    green = aux["part_is_green"][l["l_partkey"] - 1]
    lp, lsup = l["l_partkey"][green], l["l_suppkey"][green]
    i4 = torch.arange(4, device=lp.device, dtype=torch.int64)
    cand = (lp[:, None] + i4[None, :] * (S // 4 + (lp[:, None] - 1) // S)) % S + 1
    j = (cand == lsup[:, None]).to(torch.int64).argmax(1)
    cost = ps["ps_supplycost"][(lp - 1) * 4 + j]
    amount = l["l_extendedprice"][green] * (1.0 - l["l_discount"][green]) - cost * l["l_quantity"][green]
    nation = s["s_nationkey"][lsup - 1]
    days = o["o_orderdate"][aux["order_index_of_key"](l["l_orderkey"][green])]
    # civil year of a day number in 1992..1998
    year = torch.zeros_like(days)
    for y in range(1992, 1999):
        start = (torch.tensor(0).new_tensor(0) + _days_of(y))
        year = torch.where(days >= int(start), torch.full_like(days, y), year)
    code = nation * 10000 + year
    uniq, inv = torch.unique(code, return_inverse=True)
    per = torch.zeros(uniq.numel(), dtype=torch.float64, device=amount.device).index_add_(0, inv, amount)
    return uniq // 10000, uniq % 10000, per


def _days_of(year):
    import datetime
    return (datetime.date(year, 1, 1) - datetime.date(1970, 1, 1)).days
