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
