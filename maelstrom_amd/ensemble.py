"""Multi-GPU ensemble plumbing: instances are independent, so the hot path shards with no
data-path collective — rank r simulates a contiguous block of global instance ids.  The only exchange is
the end-of-batch, variable-length gather of the emitted histories to a root rank.

Two transports, one layout (every rank's part of every slab kind in rank order, `gather_layout`):
  * "cabi"  — `msim_gather` behind the C-ABI (csrc/gather.cpp): device-side compaction, ncclAllGather of the byte counts,
              one grouped ncclSend / ncclRecv per slab kind to the root over RCCL / xGMI.  What a JVM host would call; what
              bench.py uses on GPUs.
  * "torch" — the same exchange on torch.distributed tensors (`gather_to_root`): size all-gather + isend / irecv to the root.
              Runs over gloo on CPU tensors (tests/test_ensemble_gloo.py), which is how the N > 1 path is covered without GPUs.
"""
import torch


def shard(n_total, rank, world):
    """Contiguous block of global instance ids for `rank`: (first, count).  Blocks differ by at most one."""
    base, rem = divmod(n_total, world)
    count = base + (1 if rank < rem else 0)
    first = rank * base + min(rank, rem)
    return first, count


def compact(rows, payload, meta):
    """Drops the unused tail of every instance's slab.
    rows [n, max_rows, 4] i32, payload [n, max_words] i32, meta [n, 8] i32 (n_rows, n_words, flags, rounds, n_events, ...)
    -> (rows [sum n_rows, 4], payload [sum n_words], n_rows [n] i64, n_words [n] i64)."""
    nr = meta[:, 0].long()
    nw = meta[:, 1].long()
    rmask = torch.arange(rows.shape[1], device=rows.device)[None, :] < nr[:, None]
    wmask = torch.arange(payload.shape[1], device=payload.device)[None, :] < nw[:, None]
    return rows[rmask], payload[wmask], nr, nw


def gather_histories(crow, cpay, nr, nw, dist=None, world=1):
    """All-gathers compacted histories of every rank.  Returns per-rank lists
    (rows_r [R_r, 4], payload_r [W_r], n_rows_r [n_r], n_words_r [n_r]) in rank order, and the byte count moved."""
    if dist is None or world == 1:
        return [(crow, cpay, nr, nw)], int(crow.numel() * 4 + cpay.numel() * 4)
    dev = crow.device
    sizes = torch.tensor([crow.shape[0], cpay.shape[0], nr.shape[0]], dtype=torch.int64, device=dev)
    all_sizes = [torch.zeros_like(sizes) for _ in range(world)]
    dist.all_gather(all_sizes, sizes)
    mr = max(int(s[0]) for s in all_sizes)
    mw = max(int(s[1]) for s in all_sizes)
    mi = max(int(s[2]) for s in all_sizes)

    def padded(t, n, shape_tail=()):
        out = torch.zeros((n,) + shape_tail, dtype=t.dtype, device=dev)
        out[: t.shape[0]] = t
        return out
    bufs = []
    for t, n, tail in ((crow, mr, (4,)), (cpay, mw, ()), (nr, mi, ()), (nw, mi, ())):
        mine = padded(t, n, tail)
        out = torch.empty((world * mine.numel(),), dtype=t.dtype, device=dev)
        dist.all_gather_into_tensor(out, mine.reshape(-1))  # one flat collective per buffer
        bufs.append(out.view((world,) + tuple(mine.shape)))
    res = []
    for r in range(world):
        a, b, c = (int(x) for x in all_sizes[r])
        res.append((bufs[0][r, :a], bufs[1][r, :b], bufs[2][r, :c], bufs[3][r, :c]))
    nbytes = sum(int(s[0]) * 16 + int(s[1]) * 4 for s in all_sizes)
    return res, nbytes


def gather_layout(sizes):
    """sizes[r] = (rows bytes, payload bytes, meta bytes, stats bytes) of rank r -> (offsets[kind][rank], totals[kind]):
    where every rank's part lands in the root's buffers.  Mirrors msim_gather_layout (csrc/gather.cpp)."""
    world = len(sizes)
    offs, totals = [], []
    for k in range(len(sizes[0])):
        o, col = 0, []
        for r in range(world):
            col.append(o)
            o += int(sizes[r][k])
        offs.append(col)
        totals.append(o)
    return offs, totals


def gather_to_root(parts, dist, world, rank, root=0):
    """The exchange of msim_gather on torch tensors: `parts` = this rank's flat uint8 tensors, one per slab kind (compacted
    rows, compacted payload, meta, stats).  All-gathers the byte counts, then every peer sends each part ONCE to the root,
    which receives it at its place (gather_layout).  Returns (root: list of gathered uint8 tensors | None, bytes received)."""
    dev = parts[0].device
    mine = torch.tensor([p.numel() for p in parts], dtype=torch.int64, device=dev)
    if dist is None or world == 1:
        return [p.clone() for p in parts], 0
    all_sizes = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(all_sizes, mine)
    sizes = [[int(x) for x in s] for s in all_sizes]
    offs, totals = gather_layout(sizes)
    received = 0
    if rank == root:
        out = [torch.empty(t, dtype=torch.uint8, device=dev) for t in totals]
        reqs = []
        for k in range(len(parts)):
            out[k][offs[k][root]: offs[k][root] + sizes[root][k]] = parts[k]
            for p in range(world):
                if p != root and sizes[p][k]:
                    reqs.append(dist.irecv(out[k][offs[k][p]: offs[k][p] + sizes[p][k]], src=p))
                    received += sizes[p][k]
        for r in reqs:
            r.wait()
        return out, received
    reqs = [dist.isend(parts[k], dst=root) for k in range(len(parts)) if parts[k].numel()]
    for r in reqs:
        r.wait()
    return None, 0
