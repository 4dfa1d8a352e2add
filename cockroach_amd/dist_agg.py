"""Distributed GROUP BY: local pre-aggregation -> BY_HASH exchange of the partial rows on the group columns -> final
aggregation -> final rendering, the plan the reference builds in distsql_physical_planner.go:2790-2844 with the
local/final decomposition of physicalplan.DistAggregationTable (pkg/sql/physicalplan/aggregator_funcs.go:83-306):

    ANY_NOT_NULL, BOOL_AND, BOOL_OR, MIN, MAX, SUM, SUM_INT : same function in both stages
    COUNT, COUNT_ROWS                                        : final SUM_INT of the local counts (:164-172, 298-306)
    AVG                                                      : local SUM + COUNT, final SUM + SUM_INT, rendered as
                                                               sum / float(count) (:94-137)

One process per GPU; only the PARTIAL rows cross xGMI (Q1: a handful; config G: every group once).  Both stages run
through an executor: `GpuExecutor` drives libcrgpu (crgpu_dev_agg_*, crgpu_dev_partition) on device tensors;
tests/test_dist_agg_gloo.py plugs in the CPU oracle to cover the N > 1 bookkeeping without a GPU.  Columns travel
between the steps as `TCol`s (torch tensors on the executor's device: values, one validity byte per row, one length per
row for BYTES), the form in which regions can be cut at arbitrary row boundaries for the all-to-all.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import List, Optional

import numpy as np
import torch
import torch.distributed as dist

from . import coldata as cd

ANY_NOT_NULL, AVG, BOOL_AND, BOOL_OR, COUNT, MAX, MIN, SUM, SUM_INT, COUNT_ROWS = 0, 1, 2, 3, 5, 7, 8, 10, 11, 14

# func -> (local stage functions, [(final function, index into the local ones)], rendering)
DIST_TABLE = {
    ANY_NOT_NULL: ([ANY_NOT_NULL], [(ANY_NOT_NULL, 0)], None),
    AVG: ([SUM, COUNT], [(SUM, 0), (SUM_INT, 1)], "div"),
    BOOL_AND: ([BOOL_AND], [(BOOL_AND, 0)], None),
    BOOL_OR: ([BOOL_OR], [(BOOL_OR, 0)], None),
    COUNT: ([COUNT], [(SUM_INT, 0)], None),
    COUNT_ROWS: ([COUNT_ROWS], [(SUM_INT, 0)], None),
    MAX: ([MAX], [(MAX, 0)], None),
    MIN: ([MIN], [(MIN, 0)], None),
    SUM: ([SUM], [(SUM, 0)], None),
    SUM_INT: ([SUM_INT], [(SUM_INT, 0)], None),
}


def _out_type(func, in_type):
    if func in (COUNT, COUNT_ROWS, SUM_INT):
        return cd.INT64
    if func in (AVG, SUM):
        return cd.FLOAT64
    if func in (BOOL_AND, BOOL_OR):
        return cd.BOOL
    return in_type


@dataclass
class StagePlan:
    local_aggs: list          # (func, col, filter_col) over the input columns
    inter_types: list         # column types of the partial rows
    final_group_cols: list    # positions of the group columns among the partial columns
    final_aggs: list          # (func, col, -1) over the partial columns
    render: list              # per requested aggregation: ("col", i) or ("div", sum_i, count_i) over the final output
    out_types: list


def plan_stages(types, group_cols, aggs) -> StagePlan:
    """Decompose `aggs` = [(func, col, filter_col)] into the two stages.  Identical local aggregations are shared
    (the planner dedups them too, distsql_physical_planner.go:2655-2700); every group column travels as a local
    ANY_NOT_NULL so the final stage can group on it."""
    local, inter_types = [], []

    def local_idx(func, col, flt):
        key = (func, col, flt)
        if key not in local:
            local.append(key)
            inter_types.append(_out_type(func, types[col] if col >= 0 else cd.INT64))
        return local.index(key)

    final_aggs, render, out_types = [], [], []
    for a in aggs:
        func, col = a[0], a[1]
        flt = a[2] if len(a) > 2 else -1
        if func not in DIST_TABLE:
            raise ValueError(f"aggregate function {func} has no local stage (DistAggregationTable)")
        lstage, fstage, rend = DIST_TABLE[func]
        lidx = [local_idx(lf, -1 if lf == COUNT_ROWS else col, flt) for lf in lstage]
        fidx = []
        for ff, li in fstage:
            final_aggs.append((ff, lidx[li], -1))
            fidx.append(len(final_aggs) - 1)
        render.append(("div", fidx[0], fidx[1]) if rend == "div" else ("col", fidx[0]))
        out_types.append(_out_type(func, types[col] if col >= 0 else cd.INT64))
    final_group_cols = [local_idx(ANY_NOT_NULL, g, -1) for g in group_cols]
    return StagePlan(list(local), inter_types, final_group_cols, final_aggs, render, out_types)


# ------------------------------------------------------------------------------------------------ columns in flight
@dataclass
class TCol:
    type: int
    data: torch.Tensor                    # fixed width: one element per row; BYTES: the concatenated payload (uint8)
    valid: Optional[torch.Tensor] = None  # uint8 per row (1 = present); None = no NULLs
    lens: Optional[torch.Tensor] = None   # BYTES: int64 length per row

    @property
    def rows(self):
        return int(self.lens.numel() if self.type == cd.BYTES else self.data.numel())


_TORCH = {cd.BOOL: torch.uint8, cd.INT16: torch.int16, cd.INT32: torch.int32, cd.INT64: torch.int64, cd.FLOAT64: torch.float64}


def vec_to_tcol(v: cd.Vec, device="cpu") -> TCol:
    valid = None
    if v.valid is not None:
        valid = torch.from_numpy(np.unpackbits(v.valid, bitorder="little")[:len(v)].copy()).to(device)
    if v.type == cd.BYTES:
        lens = torch.from_numpy(np.diff(v.offsets.astype(np.int64))).to(device)
        return TCol(cd.BYTES, torch.from_numpy(np.ascontiguousarray(v.data[:int(v.offsets[-1])])).to(device), valid, lens)
    return TCol(v.type, torch.from_numpy(np.ascontiguousarray(v.data)).to(device), valid)


def tcol_to_vec(c: TCol) -> cd.Vec:
    valid = None
    if c.valid is not None:
        valid = np.packbits(c.valid.cpu().numpy().astype(np.uint8), bitorder="little")
    if c.type == cd.BYTES:
        offs = np.concatenate([[0], np.cumsum(c.lens.cpu().numpy())]).astype(np.uint32)
        return cd.Vec(cd.BYTES, c.data.cpu().numpy(), valid, offs)
    return cd.Vec(c.type, c.data.cpu().numpy(), valid)


def cut_tcol(c: TCol, lo, hi) -> TCol:
    """Rows [lo, hi) of a TCol."""
    lo, hi = int(lo), int(hi)
    valid = c.valid[lo:hi] if c.valid is not None else None
    if c.type == cd.BYTES:
        cs = torch.cat([torch.zeros(1, dtype=torch.int64, device=c.lens.device), torch.cumsum(c.lens, 0)])
        return TCol(cd.BYTES, c.data[int(cs[lo]):int(cs[hi])], valid, c.lens[lo:hi])
    return TCol(c.type, c.data[lo:hi], valid)


def cat_tcols(parts: List[TCol]) -> TCol:
    """The rows of `parts` one after the other (validity present on all parts or on none)."""
    t = parts[0].type
    valid = torch.cat([p.valid for p in parts]) if parts[0].valid is not None else None
    lens = torch.cat([p.lens for p in parts]) if t == cd.BYTES else None
    return TCol(t, torch.cat([p.data for p in parts]), valid, lens)


def exchange_tcols(cols: List[TCol], send_counts, group=None):
    """Regions of `cols` (rows grouped by destination, `send_counts` rows each) swapped with one all-to-all per tensor.
    The only collective on the data path."""
    send_counts = [int(x) for x in send_counts]
    dev = cols[0].data.device
    sc = torch.tensor(send_counts, dtype=torch.int64, device=dev)
    rc = torch.empty_like(sc)
    dist.all_to_all_single(rc, sc, group=group)
    recv_counts = [int(x) for x in rc.tolist()]

    def a2a(t, s_split, r_split):
        out = torch.empty(sum(r_split), dtype=t.dtype, device=t.device)
        dist.all_to_all_single(out, t.contiguous(), output_split_sizes=r_split, input_split_sizes=s_split, group=group)
        return out

    out = []
    for c in cols:
        if c.rows != sum(send_counts):
            raise ValueError(f"column of {c.rows} rows, partition counts sum to {sum(send_counts)}")
        valid = a2a(c.valid, send_counts, recv_counts) if c.valid is not None else None
        if c.type == cd.BYTES:
            lens = a2a(c.lens, send_counts, recv_counts)
            bounds = np.concatenate([[0], np.cumsum(send_counts)])
            csum = torch.cat([torch.zeros(1, dtype=torch.int64, device=dev), torch.cumsum(c.lens, 0)])
            sbytes = [int(csum[bounds[p + 1]] - csum[bounds[p]]) for p in range(len(send_counts))]
            sb = torch.tensor(sbytes, dtype=torch.int64, device=dev)
            rb = torch.empty_like(sb)
            dist.all_to_all_single(rb, sb, group=group)
            out.append(TCol(cd.BYTES, a2a(c.data, sbytes, [int(x) for x in rb.tolist()]), valid, lens))
        else:
            out.append(TCol(c.type, a2a(c.data, send_counts, recv_counts), valid))
    # validity may be absent on some ranks only: every rank must agree on which tensors exist, so ranks send explicit
    # validity for every nullable partial column (see local_stage)
    return out, recv_counts


# ------------------------------------------------------------------------------------------------ executors
class GpuExecutor:
    """Both stages on the device through the C-ABI (crgpu_dev_agg_*, crgpu_dev_partition)."""

    def __init__(self, device_index=0):
        from . import _lib
        self.device = torch.device("cuda", device_index)
        _lib.check(_lib.load().crgpu_init(device_index))

    def _to_dev(self, c: TCol):
        from . import device as dev
        valid = None
        if c.valid is not None:
            n = c.valid.numel()
            pad = torch.zeros((-n) % 8 + 8, dtype=torch.uint8, device=c.valid.device)  # whole bytes + one spare
            bits = torch.cat([c.valid.to(torch.uint8), pad]).reshape(-1, 8).to(torch.int32)
            w = torch.tensor([1, 2, 4, 8, 16, 32, 64, 128], dtype=torch.int32, device=bits.device)
            valid = (bits * w).sum(1).to(torch.uint8)
        if c.type == cd.BYTES:
            offs = torch.cat([torch.zeros(1, dtype=torch.int64, device=c.lens.device), torch.cumsum(c.lens, 0)]).to(torch.int32)
            data = c.data if c.data.numel() else torch.zeros(16, dtype=torch.uint8, device=offs.device)
            return dev.from_torch(cd.BYTES, data, valid, offs)
        data = c.data if c.data.numel() else torch.zeros(2, dtype=c.data.dtype, device=c.data.device)
        d = dev.from_torch(c.type, data, valid)
        d.n = c.rows
        return d

    def _from_result(self, res) -> List[TCol]:
        from . import device as dev
        return self.cols_to_tcols(dev.result_columns(res))

    def cols_to_tcols(self, cols) -> List[TCol]:
        """device.DeviceColumns -> TCols (copies: the TCols own their tensors)."""
        n = int(cols[0].n) if cols else 0
        out = []

        def view(ptr, count, dtype, typestr):
            if count == 0:
                return torch.zeros(0, dtype=dtype, device=self.device)

            class H:
                pass
            h = H()
            h.__cuda_array_interface__ = {"shape": (int(count),), "typestr": typestr, "data": (int(ptr), False), "version": 3}
            return torch.as_tensor(h, device=self.device).clone()

        ts = {cd.BOOL: "|u1", cd.INT16: "<i2", cd.INT32: "<i4", cd.INT64: "<i8", cd.FLOAT64: "<f8"}
        for c in cols:
            valid = None
            if c.valid_ptr:
                bm = view(c.valid_ptr, (n + 7) // 8, torch.uint8, "|u1")
                i = torch.arange(n, device=self.device)
                valid = ((bm[i >> 3].to(torch.int32) >> (i & 7).to(torch.int32)) & 1).to(torch.uint8)
            if c.type == cd.BYTES:
                offs = view(c.offsets_ptr, n + 1, torch.int32, "<i4").to(torch.int64)
                total = int(offs[-1]) if n else 0
                out.append(TCol(cd.BYTES, view(c.data_ptr, total, torch.uint8, "|u1"), valid, offs[1:] - offs[:-1]))
            else:
                out.append(TCol(c.type, view(c.data_ptr, n, _TORCH[c.type], ts[c.type]), valid))
        torch.cuda.synchronize(self.device)
        return out

    def aggregate(self, types, group_cols, aggs, cols: List[TCol]) -> List[TCol]:
        from . import device as dev
        a = dev.Agg(types, group_cols, aggs)
        try:
            n = cols[0].rows if cols else 0
            if n:
                torch.cuda.synchronize(self.device)  # the inputs were produced on torch's stream
                dcols = [self._to_dev(c) for c in cols]
                torch.cuda.synchronize(self.device)
                a.consume(dcols, n)
            return self._from_result(a.finish())
        finally:
            a.close()

    def partition(self, types, hash_cols, world, cols: List[TCol]):
        from . import device as dev
        p = dev.Partitioner(types, hash_cols, world)
        try:
            n = cols[0].rows if cols else 0
            dcols = [self._to_dev(c) for c in cols]
            torch.cuda.synchronize(self.device)
            res, counts = p.partition(dcols, n)
            return self._from_result(res), [int(x) for x in counts]
        finally:
            p.close()


# ------------------------------------------------------------------------------------------------ the two stages
def _explicit_validity(c: TCol) -> TCol:
    if c.valid is None:
        dev = c.lens.device if c.type == cd.BYTES else c.data.device
        return TCol(c.type, c.data, torch.ones(c.rows, dtype=torch.uint8, device=dev), c.lens)
    return c


def local_stage(ex, plan: StagePlan, types, group_cols, cols: List[TCol], world):
    """Pre-aggregate this rank's rows, then regroup the partial rows by destination = hash(group columns) % world
    (the reference's router hash, so equal groups meet on one rank)."""
    partial = ex.aggregate(types, group_cols, plan.local_aggs, cols)
    regions, counts = ex.partition(plan.inter_types, plan.final_group_cols, world, partial)
    return [_explicit_validity(c) for c in regions], counts


def final_stage(ex, plan: StagePlan, received: List[TCol]) -> List[cd.Vec]:
    """Aggregate the partial rows this rank received and render the requested aggregations (AVG = sum / count)."""
    fin = ex.aggregate(plan.inter_types, plan.final_group_cols, plan.final_aggs, received)
    vecs = [tcol_to_vec(c) for c in fin]
    out = []
    for r in plan.render:
        if r[0] == "col":
            out.append(vecs[r[1]])
            continue
        s, c = vecs[r[1]], vecs[r[2]]
        cnt = c.data.astype(np.float64)
        pres = s.present() & c.present() & (cnt != 0)
        q = np.divide(s.data, cnt, out=np.zeros(len(s), dtype=np.float64), where=pres)
        out.append(cd.Vec.from_numpy(cd.FLOAT64, q, pres))
    return out


def distributed_agg(ex, types, group_cols, aggs, cols: List[TCol], group=None) -> List[cd.Vec]:
    """This rank's share of the result of GROUP BY over the union of all ranks' `cols`."""
    world = dist.get_world_size(group)
    plan = plan_stages(types, group_cols, aggs)
    regions, counts = local_stage(ex, plan, types, group_cols, cols, world)
    received, _ = exchange_tcols(regions, counts, group)
    return final_stage(ex, plan, received)


# ------------------------------------------------------------------------------------------------ device-resident form
def distributed_agg_device(xch, types, group_cols, aggs, dcols, n) -> List[cd.Vec]:
    """The same two-stage plan with every step behind the C-ABI and every intermediate in HBM: crgpu_dev_agg_* (local) ->
    crgpu_dev_partition of the partial rows -> crgpu_exchange_all_to_all (RCCL) -> crgpu_dev_agg_* (final); only the
    final groups of this rank come back to the host, where AVG is rendered.  `xch`: a device.Exchange; `dcols`: this
    rank's input as device columns.  This is the form bench.py --gpus N and a Go host run; the TCol form above exists
    for the gloo tests."""
    from . import device as dev
    plan = plan_stages(types, group_cols, aggs)
    local = dev.Agg(types, group_cols, plan.local_aggs)
    part = dev.Partitioner(plan.inter_types, plan.final_group_cols, xch.world)
    final = dev.Agg(plan.inter_types, plan.final_group_cols, plan.final_aggs)
    try:
        if n:
            local.consume(dcols, n)
        partial = dev.result_columns(local.finish())
        regions, counts = part.partition(partial, partial[0].n if partial else 0)
        got, rc = xch.all_to_all(dev.result_columns(regions), counts)
        m = int(rc.sum())
        if m:
            final.consume(dev.result_columns(got), m)
        vecs = dev.result_to_vecs(final.finish())
    finally:
        local.close()
        part.close()
        final.close()
    out = []
    for r in plan.render:
        if r[0] == "col":
            out.append(vecs[r[1]])
            continue
        s, c = vecs[r[1]], vecs[r[2]]
        cnt = c.data.astype(np.float64)
        pres = s.present() & c.present() & (cnt != 0)
        q = np.divide(s.data, cnt, out=np.zeros(len(s), dtype=np.float64), where=pres)
        out.append(cd.Vec.from_numpy(cd.FLOAT64, q, pres))
    return out
