"""Synthetic TPC-H-shaped tables for the Q3 / Q9 pipelines (BASELINE.json configs[2], SURVEY.md 8(d) "Config Q3/Q9").

There is no dbgen and no dataset here (no network): the tables are generated with the TPC-H specification's
cardinalities, key structure and value distributions for exactly the columns Q3 and Q9 touch (schema:
/root/reference/pkg/workload/tpch/tpch.go:403-486; cardinalities per scale factor: supplier 10k, part 200k,
partsupp 800k, customer 150k, orders 1.5M, lineitem ~6M rows).  Text columns carry the spec's vocabularies
(25 nation names, 5 market segments, p_name = 5 of 92 colour words) so the BYTES selections and the BYTES group key
see realistic lengths; money columns are FLOAT64 as in the reference's tpch workload (tpch.go:465-468); DATE columns
are int64 days since 1970-01-01 (coldata's canonical family for DATE, typeconv.go:46-47).

Generated with torch on any device (CPU for the parity tests, the GPU for SF10) from one seed.
"""
from __future__ import annotations

import torch

NATIONS = [b"ALGERIA", b"ARGENTINA", b"BRAZIL", b"CANADA", b"EGYPT", b"ETHIOPIA", b"FRANCE", b"GERMANY", b"INDIA",
           b"INDONESIA", b"IRAN", b"IRAQ", b"JAPAN", b"JORDAN", b"KENYA", b"MOROCCO", b"MOZAMBIQUE", b"PERU", b"CHINA",
           b"ROMANIA", b"SAUDI ARABIA", b"VIETNAM", b"RUSSIA", b"UNITED KINGDOM", b"UNITED STATES"]
SEGMENTS = [b"AUTOMOBILE", b"BUILDING", b"FURNITURE", b"MACHINERY", b"HOUSEHOLD"]
COLORS = [c.encode() for c in (
    "almond antique aquamarine azure beige bisque black blanched blue blush brown burlywood burnished chartreuse "
    "chiffon chocolate coral cornflower cornsilk cream cyan dark deep dim dodger drab firebrick floral forest frosted "
    "gainsboro ghost goldenrod green grey honeydew hot indian ivory khaki lace lavender lawn lemon light lime linen "
    "magenta maroon medium metallic midnight mint misty moccasin navajo navy olive orange orchid pale papaya peach "
    "peru pink plum powder puff purple red rose rosy royal saddle salmon sandy seashell sienna sky slate smoke snow "
    "spring steel tan thistle tomato turquoise violet wheat white yellow").split()]
assert len(COLORS) == 92

DAY_1992_01_01 = 8035
DAY_1998_08_02 = 10440
DAY_1995_03_15 = 9204  # Q3's date constant


class BytesCol:
    """A BYTES column as (flat uint8 payload, uint32 offsets[n+1]) tensors."""

    def __init__(self, data, offsets):
        self.data, self.offsets = data, offsets

    def __len__(self):
        return self.offsets.numel() - 1


def _vocab_col(ids, vocab, device):
    """ids[n] -> BYTES column of vocab[ids]."""
    lens = torch.tensor([len(v) for v in vocab], dtype=torch.int64, device=device)
    starts = torch.zeros(len(vocab) + 1, dtype=torch.int64, device=device)
    starts[1:] = torch.cumsum(lens, 0)
    flat = torch.tensor(list(b"".join(vocab)), dtype=torch.uint8, device=device)
    rl = lens[ids]
    offs = torch.zeros(ids.numel() + 1, dtype=torch.int64, device=device)
    offs[1:] = torch.cumsum(rl, 0)
    total = int(offs[-1].item())
    row = torch.repeat_interleave(torch.arange(ids.numel(), device=device), rl, output_size=total)
    pos = torch.arange(total, device=device) - offs[row]
    data = flat[starts[ids][row] + pos]
    return BytesCol(data.contiguous(), offs.to(torch.uint32) if total < 2 ** 32 else offs)


def _names_col(word_ids, vocab, device):
    """word_ids[n, k] -> BYTES column: the k words joined by single spaces (p_name)."""
    n, k = word_ids.shape
    lens = torch.tensor([len(v) for v in vocab], dtype=torch.int64, device=device)
    starts = torch.zeros(len(vocab) + 1, dtype=torch.int64, device=device)
    starts[1:] = torch.cumsum(lens, 0)
    flat = torch.tensor(list(b"".join(vocab)), dtype=torch.uint8, device=device)
    wl = lens[word_ids]  # [n, k]
    seg = wl + 1         # every word is followed by a separator, the last one's is dropped below
    row_len = seg.sum(1) - 1
    offs = torch.zeros(n + 1, dtype=torch.int64, device=device)
    offs[1:] = torch.cumsum(row_len, 0)
    total = int(offs[-1].item())
    data = torch.full((total,), 32, dtype=torch.uint8, device=device)  # spaces
    word_off = offs[:-1, None] + torch.cumsum(seg, 1) - seg  # start of word j in row i
    wl_f, wo_f, ws_f = wl.reshape(-1), word_off.reshape(-1), starts[word_ids].reshape(-1)
    tot_w = int(wl_f.sum().item())
    idx = torch.repeat_interleave(torch.arange(n * k, device=device), wl_f, output_size=tot_w)
    first = torch.zeros(n * k + 1, dtype=torch.int64, device=device)
    first[1:] = torch.cumsum(wl_f, 0)
    pos = torch.arange(tot_w, device=device) - first[idx]
    data[wo_f[idx] + pos] = flat[ws_f[idx] + pos]
    return BytesCol(data.contiguous(), offs.to(torch.uint32))


def generate(sf: float, device="cpu", seed=42):
    """Returns {table: {column: int64/float64 tensor | BytesCol}} and the row counts."""
    dev = torch.device(device)
    g = torch.Generator(device=dev)
    g.manual_seed(seed)

    def randint(lo, hi, n):  # inclusive bounds
        return torch.randint(lo, hi + 1, (n,), generator=g, device=dev, dtype=torch.int64)

    S = max(int(10_000 * sf), 8)
    P = max(int(200_000 * sf), 16)
    C = max(int(150_000 * sf), 16)
    O = max(int(1_500_000 * sf), 32)
    t = {}
    t["nation"] = {"n_nationkey": torch.arange(25, device=dev, dtype=torch.int64),
                   "n_name": _vocab_col(torch.arange(25, device=dev), NATIONS, dev)}
    t["supplier"] = {"s_suppkey": torch.arange(1, S + 1, device=dev, dtype=torch.int64), "s_nationkey": randint(0, 24, S)}
    # p_name: 5 distinct colour words (a random permutation's first five)
    pk = torch.arange(1, P + 1, device=dev, dtype=torch.int64)
    words = torch.cat([torch.rand(min(1 << 20, P - a), len(COLORS), generator=g, device=dev).argsort(1)[:, :5]
                       for a in range(0, P, 1 << 20)])
    t["part"] = {"p_partkey": pk, "p_name": _names_col(words, COLORS, dev)}
    aux = {"part_is_green": (words == COLORS.index(b"green")).any(1)}  # ground truth for the checks, not a table column
    # partsupp: 4 suppliers per part, TPC-H 4.2.3: ps_suppkey = (partkey + i * (S/4 + (partkey-1)/S)) mod S + 1
    i4 = torch.arange(4, device=dev, dtype=torch.int64)
    ps_part = pk.repeat_interleave(4)
    ps_supp = ((pk[:, None] + i4[None, :] * (S // 4 + (pk[:, None] - 1) // S)) % S + 1).reshape(-1)
    t["partsupp"] = {"ps_partkey": ps_part, "ps_suppkey": ps_supp,
                     "ps_supplycost": randint(100, 100_000, 4 * P).to(torch.float64) / 100.0}
    t["customer"] = {"c_custkey": torch.arange(1, C + 1, device=dev, dtype=torch.int64),
                     "c_mktsegment": None}
    seg = randint(0, 4, C)
    t["customer"]["c_mktsegment"] = _vocab_col(seg, SEGMENTS, dev)
    aux["cust_is_building"] = seg == SEGMENTS.index(b"BUILDING")
    # orders: sparse keys (8 used of every 32), customers whose key is not a multiple of 3
    oi = torch.arange(O, device=dev, dtype=torch.int64)
    okey = (oi // 8) * 32 + oi % 8 + 1
    ocust = randint(1, max(C - C // 3, 1), O)
    ocust = ocust + (ocust - 1) // 2  # skips the multiples of 3: 1,2,4,5,7,8,...
    ocust = torch.clamp(ocust, max=C)
    odate = randint(DAY_1992_01_01, DAY_1998_08_02, O)
    t["orders"] = {"o_orderkey": okey, "o_custkey": ocust, "o_orderdate": odate,
                   "o_shippriority": torch.zeros(O, device=dev, dtype=torch.int64)}
    # lineitem: 1..7 lines per order
    nl = randint(1, 7, O)
    L = int(nl.sum().item())
    oidx = torch.repeat_interleave(torch.arange(O, device=dev), nl, output_size=L)
    lpart = randint(1, P, L)
    j = randint(0, 3, L)
    lsupp = (lpart + j * (S // 4 + (lpart - 1) // S)) % S + 1
    qty = randint(1, 50, L).to(torch.float64)
    retail = (90000 + (lpart // 10) % 20001 + 100 * (lpart % 1000)).to(torch.float64) / 100.0
    t["lineitem"] = {"l_orderkey": okey[oidx], "l_partkey": lpart, "l_suppkey": lsupp, "l_quantity": qty,
                     "l_extendedprice": qty * retail, "l_discount": randint(0, 10, L).to(torch.float64) / 100.0,
                     "l_shipdate": odate[oidx] + randint(1, 121, L)}
    aux["order_index_of_key"] = lambda k: (k - 1) // 32 * 8 + (k - 1) % 32
    rows = {"nation": 25, "supplier": S, "part": P, "partsupp": 4 * P, "customer": C, "orders": O, "lineitem": L}
    rows["_aux"] = aux
    return t, rows
