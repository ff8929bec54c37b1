"""The N > 1 path on CPU: world_size 2 over gloo.  Each rank plans its own shard (LPT by frame count); the union of the
shards is the whole batch, work is balanced, and the max-over-ranks reduction bench.py uses works."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import oracle_lib as O
    from pycricodecs_amd import shard, synth
    items = [O.hca_encode(synth.wav(300 + i, 1024 * (1 + (i * 7) % 5), 2, 48000), 1) for i in range(12)]
    weights = [shard.hca_weight(h) for h in items]
    mine = shard.my_items(weights, rank, world)
    load = torch.tensor([float(sum(weights[i] for i in mine))], dtype=torch.float64)
    owned = torch.zeros(len(items), dtype=torch.int32)
    owned[mine] = 1
    dist.all_reduce(owned)                                  # every item owned exactly once
    t = load.clone()
    dist.all_reduce(t, op=dist.ReduceOp.MAX)                # the reduction bench.py applies to the elapsed time
    tot = load.clone()
    dist.all_reduce(tot)
    # decode my shard with the oracle (stand-in for the device) and checksum it; the checksum of checksums must equal the
    # single-process one regardless of the world size
    import hashlib
    digest = [int.from_bytes(hashlib.sha256(O.hca_decode(items[i])).digest()[:4], "little") for i in mine]
    s = torch.tensor([sum(digest) % (1 << 31)], dtype=torch.int64)
    dist.all_reduce(s)
    full = sum(int.from_bytes(hashlib.sha256(O.hca_decode(h)).digest()[:4], "little") for h in items)
    # the optional exchange step: every rank's decoded bytes on rank 0, in rank order
    mine_pcm = b"".join(O.hca_decode(items[i]) for i in mine)
    got, offs = shard.gather_bytes_to_root(torch.frombuffer(bytearray(mine_pcm), dtype=torch.uint8))
    gathered_ok = None
    if rank == 0:
        expect = b"".join(b"".join(O.hca_decode(items[i]) for i in shard.my_items(weights, r, world)) for r in range(world))
        gathered_ok = bytes(got.numpy()) == expect and offs[-1] == len(expect)
    if rank == 0:
        q.put((owned.tolist(), float(t.item()), float(tot.item()), sum(weights), int(s.item()), full, gathered_ok))
    dist.destroy_process_group()


def test_two_rank_sharding_gloo():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    owned, tmax, tot, total_w, s, full, gathered_ok = q.get(timeout=180)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert owned == [1] * 12
    assert gathered_ok is True                                 # variable-length gather to rank 0 (send/recv batch)
    assert tot == total_w
    assert tmax <= 0.6 * total_w                              # LPT keeps the heavier rank within 60 % of the total
    assert s % (1 << 31) == full % (1 << 31) or s == full     # checksum of checksums is world-size independent


def test_lpt_is_deterministic_and_balanced():
    sys.path.insert(0, ROOT)
    from pycricodecs_amd import shard
    w = [469] * 100 + [47, 94, 12, 1407, 3, 800]
    for world in (1, 2, 4, 8):
        a = shard.lpt_assign(w, world)
        assert a == shard.lpt_assign(w, world)
        loads = [sum(x for x, r in zip(w, a) if r == k) for k in range(world)]
        assert max(loads) - min(loads) <= max(w)
