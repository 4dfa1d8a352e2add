"""The one exchange step of the path: after the hash repartition
(crgpu_dev_partition) every rank holds, per column, one contiguous region per
destination; the regions are swapped with one all-to-all per column (RCCL over
xGMI on GPUs; gloo in the CPU tests).  This is the flow the reference builds
with BY_HASH routers on both join inputs (physical_plan.go:1120-1157,
colflow/routers.go:592) -- and the only collective on the data path.
"""
from __future__ import annotations

import torch
import torch.distributed as dist


def exchange_counts(send_counts, device, group=None):
    """send_counts[p] = rows this rank sends to rank p -> recv_counts[p] = rows rank p sends here."""
    sc = torch.as_tensor(list(send_counts), dtype=torch.int64, device=device)
    rc = torch.empty_like(sc)
    dist.all_to_all_single(rc, sc, group=group)
    return [int(x) for x in rc.tolist()]


def all_to_all_regions(cols, send_counts, group=None):
    """cols: 1-D tensors of equal length sum(send_counts), rows grouped by destination in rank order.
    Returns (received tensors, recv_counts)."""
    send_counts = [int(x) for x in send_counts]
    if not cols:
        return [], exchange_counts(send_counts, torch.device("cpu"), group)
    n = sum(send_counts)
    for c in cols:
        if c.dim() != 1 or c.numel() != n:
            raise ValueError(f"column of {c.numel()} rows, partition counts sum to {n}")
    recv_counts = exchange_counts(send_counts, cols[0].device, group)
    total = sum(recv_counts)
    out = []
    for c in cols:
        dst = torch.empty(total, dtype=c.dtype, device=c.device)
        dist.all_to_all_single(dst, c.contiguous(), output_split_sizes=recv_counts, input_split_sizes=send_counts,
                               group=group)
        out.append(dst)
    return out, recv_counts
