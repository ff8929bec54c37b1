"""File sharding across ranks (one process per GPU).  The codec path has no exchange step: every item is independent,
so rank r simply owns a subset of the items.  Longest-processing-time greedy keeps the per-rank work (frames / blocks)
balanced for mixed batches (BASELINE configs[4]); results are independent of the world size by construction."""
import heapq


def lpt_assign(weights, world_size):
    """Return rank_of[i] for each item: heaviest items first, each to the currently lightest rank (ties -> lowest rank)."""
    order = sorted(range(len(weights)), key=lambda i: (-weights[i], i))
    heap = [(0, r) for r in range(world_size)]
    heapq.heapify(heap)
    rank_of = [0] * len(weights)
    for i in order:
        load, r = heapq.heappop(heap)
        rank_of[i] = r
        heapq.heappush(heap, (load + weights[i], r))
    return rank_of


def my_items(weights, rank, world_size):
    rank_of = lpt_assign(weights, world_size)
    return [i for i, r in enumerate(rank_of) if r == rank]


def hca_weight(hca: bytes) -> int:
    """Work estimate of an HCA stream = its frame count (header field at 0x10, big endian)."""
    return int.from_bytes(hca[16:20], "big") if len(hca) >= 20 else 0


def adx_weight(adx: bytes) -> int:
    """Work estimate of an ADX file = blocks = ceil(samples / samples_per_block) * channels."""
    if len(adx) < 20 or adx[5] < 3 or adx[6] == 0:
        return 0
    spb = (adx[5] - 2) * 8 // adx[6]
    n = int.from_bytes(adx[12:16], "big")
    return ((n + spb - 1) // max(spb, 1)) * adx[7]


def gather_bytes_to_root(t, root=0, group=None):
    """The one exchange step a sharded batch may want (BASELINE configs[4]: decoded PCM of every rank on rank 0): a
    variable-length gather of 1-D uint8 tensors.  Sizes travel by all_gather; the payloads by point-to-point send/recv posted
    as ONE batch, so on a node the N-1 transfers into the root run concurrently over its N-1 direct xGMI links (there is
    no ring, hence no per-link ring bound).  Works with the nccl (= RCCL) backend on device tensors and with gloo on CPU.
    Returns (concatenated tensor, offsets list of length world+1) on the root, (None, None) elsewhere."""
    import torch
    import torch.distributed as dist
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    n = torch.tensor([t.numel()], dtype=torch.int64, device=t.device)
    sizes = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(sizes, n, group=group)
    sizes = [int(s.item()) for s in sizes]
    offs = [0]
    for s in sizes:
        offs.append(offs[-1] + s)
    if world == 1:
        return t, offs
    if rank == root:
        out = torch.empty(offs[-1], dtype=torch.uint8, device=t.device)
        out[offs[root]:offs[root + 1]].copy_(t)
        ops = [dist.P2POp(dist.irecv, out[offs[r]:offs[r + 1]], r, group) for r in range(world) if r != root and sizes[r]]
    else:
        out = None
        ops = [dist.P2POp(dist.isend, t, root, group)] if t.numel() else []
    if ops:
        for req in dist.batch_isend_irecv(ops):
            req.wait()
    return (out, offs) if rank == root else (None, None)
