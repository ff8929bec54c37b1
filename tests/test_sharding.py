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


# ------------------------------------------------------------------------------------------------ configs[4], eight ranks, no hardware
def _awb_worker(rank, world, port, n_total, q):
    """One rank of `bench.py --gpus 8 --workload awb_mixed --scaling strong`, with the oracle standing in for the device: the rank
    builds its AFS2 bank from its LPT share of the fixed batch (bench.build_awb_bank), indexes it with the library's host-side reader,
    "decodes" every item with the oracle into the job's output layout (items in bank order, each placed so that the samples behind its
    WAV header start a 128-byte line: cri_capi.cpp, wav_item_start -- the GPU tests hold the device planner to the same bytes) and sends the PCM to rank 0
    through the path's one exchange step (shard.gather_bytes_to_root)."""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import bench as B
    import oracle_lib as O
    from pycricodecs_amd import awb, shard
    uniq, order_all, mine, subkey = B.awb_clip_plan(n_total, rank, world)
    bank, uniq_b, order, subkey_b = B.build_awb_bank(n_total, rank, world)
    assert order == [int(order_all[i]) for i in mine] and subkey == subkey_b
    offs, kinds, sk = awb.awb_index(bank)                      # the library's own AFS2 reader (host side: runs without a device)
    assert sk == subkey and len(kinds) == len(mine)
    refs = {}
    pcm = bytearray()
    frames = 0
    for i, u in enumerate(order):
        kind, data = uniq[u]
        item = bank[int(offs[i]):int(offs[i]) + len(data)]
        assert item == data and kinds[i] == (awb.KIND_HCA if kind == "hca" else awb.KIND_ADX)
        if u not in refs:
            refs[u] = O.hca_decode(data, B.KEY, subkey) if kind == "hca" else O.adx_decode(data)
        hdr = 0x70 if refs[u][0x24:0x28] == b"smpl" else 0x2C
        pcm += bytes(-(len(pcm) + hdr) % 128) + refs[u]
        frames += shard.hca_weight(data) if kind == "hca" else shard.adx_weight(data) // 2
    pcm += bytes(-len(pcm) % 64)
    t = torch.frombuffer(pcm, dtype=torch.uint8) if pcm else torch.zeros(0, dtype=torch.uint8)
    got, goffs = shard.gather_bytes_to_root(t)
    load = torch.tensor([float(frames)], dtype=torch.float64)
    mx, tot = load.clone(), load.clone()
    dist.all_reduce(mx, op=dist.ReduceOp.MAX)
    dist.all_reduce(tot)
    owned = torch.zeros(n_total, dtype=torch.int32)
    owned[mine] = 1
    dist.all_reduce(owned)
    if rank == 0:
        # what the unsharded job would hold: every clip of the global list once; the root's buffer = the ranks' outputs in rank order,
        # each the rank's items in its bank order
        ok = True
        all_refs = {}
        wts = [shard.hca_weight(d) if k == "hca" else shard.adx_weight(d) // 2 for k, d in uniq]
        weights_all = [wts[int(k)] for k in order_all]
        for r in range(world):
            part = bytes(got[goffs[r]:goffs[r + 1]].numpy())
            pos = 0
            for i in shard.my_items(weights_all, r, world):
                u = int(order_all[i])
                if u not in all_refs:
                    kind, data = uniq[u]
                    all_refs[u] = O.hca_decode(data, B.KEY, subkey) if kind == "hca" else O.adx_decode(data)
                w = all_refs[u]
                pos += -(pos + (0x70 if w[0x24:0x28] == b"smpl" else 0x2C)) % 128
                ok = ok and part[pos:pos + len(w)] == w
                pos += len(w)
            ok = ok and pos + (-pos % 64) == len(part)
        q.put((owned.tolist(), float(mx.item()), float(tot.item()), ok, [int(x) for x in goffs]))
    dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_eight_rank_awb_plan_gloo():
    """BASELINE configs[4] at eight ranks without hardware: the fixed batch is dealt out once (every clip owned by exactly one rank),
    the shares are balanced by frames (LPT: the heaviest rank within 2 % of the mean for 800 clips), every rank's bank parses, and the
    gathered buffer on rank 0 is the ranks' outputs in rank order -- byte for byte the oracle's decode of every clip of the unsharded
    list.  The day an 8-GPU node exists the only unknown is bandwidth."""
    sys.path.insert(0, ROOT)
    world, n_total = 8, 800
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_awb_worker, args=(r, world, port, n_total, q)) for r in range(world)]
    for p in procs:
        p.start()
    owned, mx, tot, ok, goffs = q.get(timeout=300)
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert owned == [1] * n_total
    assert ok is True
    assert len(goffs) == world + 1 and all(b > a for a, b in zip(goffs, goffs[1:]))
    assert mx <= 1.02 * tot / world
