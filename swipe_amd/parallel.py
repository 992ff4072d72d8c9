"""Multi-GPU search: one process per GPU, the database sharded read-only by residue count.

The reference's only distributed exchange that matters is tag_search_report: each MPI worker
sends its top-K hit tuples to the master, which re-enters them through hits_enter
(swipe.cc:1951-1974, 2320).  Here every rank keeps a contiguous seqno range balanced by
RESIDUES, searches it on its own GPU with no data-path collective, and the
per-rank top-K lists are exchanged by ONE all_gather of K x 2 int64 (plus 3 counters) - over
RCCL/xGMI when the tensors live on the GPU ("nccl" backend), over gloo in the CPU tests.  The
merged list equals the single-list result because the reference's list is exactly the global
top-K under the total order (score desc, seqno desc) (hits.cc:188-219).
"""
from __future__ import annotations

from typing import List, Sequence, Tuple

import numpy as np

from .api import merge_frame_hits, merge_hits


def shard_bounds(offsets: np.ndarray, world_size: int) -> List[Tuple[int, int]]:
    """Contiguous [first, last) sequence ranges with near-equal residue counts.

    offsets: int64 [nseq+1] prefix sums of sequence lengths (offsets[0] may be non-zero)."""
    off = np.asarray(offsets, dtype=np.int64)
    nseq = len(off) - 1
    total = int(off[-1] - off[0])
    cuts = [0]
    for r in range(1, world_size):
        target = off[0] + total * r // world_size
        c = int(np.searchsorted(off, target, side="left"))
        cuts.append(min(max(c, cuts[-1]), nseq))
    cuts.append(nseq)
    return [(cuts[r], cuts[r + 1]) for r in range(world_size)]


def gather_topk(local_hits: Sequence[Tuple[int, ...]], keep: int, totalhits: int = 0, obvious: int = 0,
                group=None, device=None, width: int = 2):
    """all_gather the per-rank ordered hit lists and merge them with the reference comparator.

    Hits are (seqno, score) pairs, or the 6-tuples of Database.search_frames_topk (translated searches); pass
    width=6 for those.  Returns (hits, totalhits_sum, obvious_sum) - identical on every rank."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    W = width
    buf = torch.zeros(keep * W + 3, dtype=torch.int64)
    n = min(len(local_hits), keep)
    if n:
        buf[: W * n] = torch.tensor([v for h in local_hits[:n] for v in h], dtype=torch.int64)
    buf[W * keep] = n
    buf[W * keep + 1] = totalhits
    buf[W * keep + 2] = obvious
    if device is not None:
        buf = buf.to(device)
    out = [torch.empty_like(buf) for _ in range(world)]
    dist.all_gather(out, buf, group=group)
    lists, tot, obv = [], 0, 0
    for t in out:
        t = t.cpu()
        k = int(t[W * keep])
        lists.append([tuple(int(t[W * i + j]) for j in range(W)) for i in range(k)])
        tot += int(t[W * keep + 1])
        obv += int(t[W * keep + 2])
    return (merge_hits(lists, keep) if W == 2 else merge_frame_hits(lists, keep)), tot, obv


def gather_topk_array(local_hits: np.ndarray, keep: int, totalhits: int = 0, obvious: int = 0, group=None, device=None):
    """gather_topk for (seqno, score) hits held as an int64 [n, 2] array (Database.search_topk_array): one
    all_gather of keep x 2 + 3 int64 per rank, merged by swa_hits_merge straight from the gathered buffer - no
    per-hit Python objects anywhere.  Returns (int64 [m, 2], totalhits_sum, obvious_sum), identical on every rank."""
    import torch
    import torch.distributed as dist
    from .api import merge_hit_arrays
    world = dist.get_world_size(group)
    n = min(len(local_hits), keep)
    row = np.zeros(2 * keep + 3, dtype=np.int64)
    row[: 2 * n] = np.asarray(local_hits[:n], dtype=np.int64).reshape(-1)
    row[2 * keep:] = (n, totalhits, obvious)
    buf = torch.from_numpy(row)
    if device is not None:
        buf = buf.to(device, non_blocking=True)
    out = torch.empty(world * (2 * keep + 3), dtype=torch.int64, device=buf.device)
    dist.all_gather_into_tensor(out, buf, group=group)
    g = out.cpu().numpy().reshape(world, 2 * keep + 3)
    merged = merge_hit_arrays(g[:, : 2 * keep].reshape(world, keep, 2), g[:, 2 * keep], keep)
    return merged, int(g[:, 2 * keep + 1].sum()), int(g[:, 2 * keep + 2].sum())


def align_sharded(db, query, hits: Sequence[Tuple[int, int]], first: int, last: int, dstrands=None, dframes=None,
                  group=None):
    """Alignment phase over shards: every rank holds the same merged hit list (gather_topk); the rank whose
    shard [first, last) contains a hit's sequence aligns it on its own GPU (db.align: end points on the
    device, traceback on its host) and ONE all_gather_object hands every rank the full list - the role of
    the alignment messages the MPI master collects from its workers (swipe.cc:2060-2140).
    Returns one alignment dict per hit, in hit order."""
    import torch.distributed as dist
    idx = [i for i, h in enumerate(hits) if first <= h[0] < last]
    mine = {}
    if idx:
        pick = lambda a: None if a is None else [a[i] for i in idx]
        for i, a in zip(idx, db.align(query, [hits[i][0] for i in idx], pick(dstrands), pick(dframes))):
            mine[i] = a
    parts = [None] * dist.get_world_size(group)
    dist.all_gather_object(parts, mine, group=group)
    merged = {}
    for part in parts:
        merged.update(part)
    if len(merged) != len(hits):
        raise RuntimeError("a hit's sequence lies in no shard: shard bounds do not cover the database")
    return [merged[i] for i in range(len(hits))]
