"""GPU parity, through the C ABI, against the pinned CPU oracle and the committed golden vectors.
The drop-in boundary (row b): the C ABI's single-file calls, batch jobs on streams / graphs / threads, the host-memory paths, the CPython module, error domains, sharded jobs."""
import ctypes as C
import os
import threading

import numpy as np
import pytest

import golden_util as G
import hca_forge
import oracle_lib as O
from gpu_common import KEY, MAN, cc, diff, run_job, run_job_floats  # noqa: F401
from pycricodecs_amd import synth

pytestmark = pytest.mark.gpu


# ------------------------------------------------------------------------------------------------ malformed inputs
def _both(gpu_call, ora_call):
    """Run the device path and the oracle on the same input: same accept/reject decision, same bytes when accepted."""
    try:
        ref = ora_call()
    except O.OracleError:
        ref = None
    from pycricodecs_amd._capi import CriCodecsError
    try:
        got = gpu_call()
    except CriCodecsError as e:
        if e.code == -304:                                               # documented "valid but not on the device path" (e.g. > 64 ADX channels)
            return "unsupported", ref
        got = None
    except (ValueError, NotImplementedError, RuntimeError):
        got = None
    return got, ref


# ------------------------------------------------------------------------------------------------ host path (b)
def _ragged_hca_batch():
    """HCA streams of ragged lengths (one of them shorter than a frame's delay, i.e. no samples), one rejected header in the
    middle, repeats of the same bytes object."""
    rng = np.random.default_rng(5)
    uniq = [O.hca_crypt(O.hca_encode(synth.wav(300 + k, int(rng.integers(200, 30000)), 2, 48000), 1), 1, 56, KEY) for k in range(9)]
    items = [uniq[int(k)] for k in rng.integers(0, len(uniq), 70)]
    items[17] = b"HCA\0" + bytes(200)                           # rejected on the host
    items[40] = uniq[0][:96]                                    # a header without any frame
    return uniq, items


def test_hca_decode_errors(cc):
    hca = G.load("s0_3008_2_48000_q1.hca")
    bad = bytearray(hca)
    bad[300] ^= 0x55
    with pytest.raises(ValueError, match="Decoding error"):
        cc.HcaDecode(bytes(bad), 96, 0, 0)
    enc = O.hca_crypt(hca, 1, 56, KEY)
    with pytest.raises(ValueError, match="Decoding error"):
        cc.HcaDecode(enc, 96, KEY + 2, 0)
    with pytest.raises(ValueError, match="not a valid HCA header"):
        cc.HcaDecode(b"HCA\x00" + bytes(200), 96, 0, 0)
    with pytest.raises(ValueError, match="copyright"):
        cc.AdxDecode(bytes([0x80, 0, 0, 0x2C, 3, 18, 4, 2, 0, 0, 0xBB, 0x80, 0, 0, 0, 64, 1, 0xF4, 4, 0]) + bytes(200))
    with pytest.raises(ValueError, match="Bitdepth"):
        cc.AdxEncode(synth.wav(0, 320, 2), 1, 18, 3, 500, 0, 4, False)
    adx = bytearray(O.adx_encode(synth.wav(0, 320, 2)))
    for bs in (1, 2):                                          # no samples per block (found by the long header fuzz: the oracle crashed on it)
        adx[5] = bs
        with pytest.raises(ValueError):
            cc.AdxDecode(bytes(adx))
        with pytest.raises(O.OracleError):
            O.adx_decode(bytes(adx))


# ------------------------------------------------------------------------------------------------ batch jobs
def test_batch_mixed_formats(cc):
    from pycricodecs_amd.batch import Job
    items, keys, refs = [], [], []
    for i, (n, ch, sr, q) in enumerate([(3000, 2, 48000, 1), (5000, 1, 44100, 1), (2048, 2, 48000, 2), (7000, 2, 48000, 3),
                                        (1024, 2, 48000, 1), (300, 2, 22050, 0), (4000, 2, 48000, 1)]):
        h = O.hca_encode(synth.wav(100 + i, n, ch, sr), q)
        if i % 2:
            h = O.hca_crypt(h, 1, 56, KEY + i)
            keys.append(KEY + i)
        else:
            keys.append(0)
        items.append(h)
        refs.append(O.hca_decode(h, keys[-1]))
    items.insert(3, b"garbage" * 30)
    keys.insert(3, 0)
    refs.insert(3, b"")
    job = Job.hca_decode(items, keys=keys)
    outs, status = job.run_host()
    assert job.host_status[3] == -201 and status[3] == -201
    for i, (o, r) in enumerate(zip(outs, refs)):
        assert diff(o, r) is None, i
    assert job.units == sum(int.from_bytes(h[16:20], "big") for h in items if h[:3] == b"HCA" or h[:1] == b"\xc8")


def test_drop_in_extension_module(cc):
    """The CPython module `CriCodecs` built from csrc/pyext gives the same bytes as the ctypes binding."""
    import importlib.util
    import os
    import sysconfig
    from pycricodecs_amd import build
    path = os.path.join(build.LIBDIR, "CriCodecs" + sysconfig.get_config_var("EXT_SUFFIX"))
    spec = importlib.util.spec_from_file_location("CriCodecs", path)
    ext = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ext)
    w = synth.wav(21, 4000, 2, 48000)
    adx = ext.AdxEncode(w, 4, 18, 3, 500, 0, 4, False)
    assert adx == cc.AdxEncode(w, 4, 18, 3, 500, 0, 4, False) == O.adx_encode(w)
    assert ext.AdxDecode(adx) == O.adx_decode(adx)
    hca = ext.HcaEncode(w, 0, 1)
    assert hca == O.hca_encode(w, 1)
    enc = ext.HcaCrypt(hca, 1, 96, 56, KEY, 0)
    assert enc == O.hca_crypt(hca, 1, 56, KEY) and hca == O.hca_encode(w, 1)      # input not mutated
    assert ext.HcaDecode(enc, 96, KEY, 0) == O.hca_decode(enc, KEY)
    with pytest.raises(ValueError, match="Decoding error"):
        ext.HcaDecode(enc, 96, KEY + 2, 0)
    with pytest.raises(ValueError, match="Bitdepth"):
        ext.AdxEncode(w, 1, 18, 3, 500, 0, 4, False)


@pytest.mark.parametrize("kind", ["hca", "adx", "wav_adx", "wav_hca"])
def test_header_mutation_fuzz(cc, kind):
    """Random byte edits and truncations in the header region: the host planners must take the oracle's accept/reject
    decision and produce its bytes (and, above all, must not read or write out of bounds doing so)."""
    import os
    rng = np.random.default_rng({"hca": 1, "adx": 2, "wav_adx": 3, "wav_hca": 4}[kind] + 10 * int(os.environ.get("CRI_FUZZ_SEED", "0")))
    w = synth.wav(77, 3008, 2, 48000)
    base = {"hca": O.hca_encode(w, quality=2), "adx": O.adx_encode(w), "wav_adx": w, "wav_hca": w}[kind]
    region = {"hca": 96, "adx": 40, "wav_adx": 44, "wav_hca": 44}[kind]
    agree_ok = 0
    import os
    for it in range(int(os.environ.get("CRI_FUZZ_ITERS", "600"))):
        b = bytearray(base)
        if it % 8 == 7:
            b = b[:int(rng.integers(0, len(b)))]                        # truncation
        else:
            for _ in range(int(rng.integers(1, 4))):
                p = int(rng.integers(0, min(region, len(b))))
                b[p] = int(rng.integers(0, 256)) if rng.random() < 0.5 else b[p] ^ (1 << int(rng.integers(0, 8)))
            if kind == "hca" and it % 2 == 0:                            # half of the edits keep a valid header checksum
                hs0 = int.from_bytes(base[6:8], "big")
                b[6:8] = base[6:8]
                b[hs0 - 2:hs0] = hca_forge.crc16(bytes(b[:hs0 - 2])).to_bytes(2, "big")
        data = bytes(b)
        if kind == "hca":
            hs = int.from_bytes(data[6:8], "big") if len(data) >= 8 else 0
            got, ref = _both(lambda: cc.HcaDecode(data, hs, 0, 0), lambda: O.hca_decode(data))
        elif kind == "adx":
            got, ref = _both(lambda: cc.AdxDecode(data), lambda: O.adx_decode(data))
        elif kind == "wav_adx":
            got, ref = _both(lambda: cc.AdxEncode(data, 4, 18, 3, 500, 0, 4, False), lambda: O.adx_encode(data))
        else:
            got, ref = _both(lambda: cc.HcaEncode(data, False, 1), lambda: O.hca_encode(data, quality=1))
        if got == "unsupported":
            continue
        assert (got is None) == (ref is None), (kind, it, "device %s, oracle %s" % ("rejects" if got is None else "accepts", "rejects" if ref is None else "accepts"))
        if ref is not None:
            assert diff(got, ref) is None, (kind, it)
            agree_ok += 1
    assert agree_ok > 5


# ------------------------------------------------------------------------------------------------ shards
@pytest.mark.parametrize("world", [2, 3, 8])
def test_shard_jobs_equal_the_unsharded_batch(cc, world):
    """pycricodecs_amd.shard (what bench.py --gpus N and a multi-GPU caller use): every rank's shard, decoded by its own
    Job, byte-equals the items of the unsharded batch -- the results do not depend on the world size."""
    from pycricodecs_amd import shard
    from pycricodecs_amd.batch import Job
    items, keys = [], []
    for i in range(41):
        w = synth.wav(120 + i, 1024 * (1 + (i * 5) % 9) + 32 * (i % 4), 2, 48000)
        items.append(O.hca_crypt(O.hca_encode(w, 1 + i % 3), 1, 56, KEY))
        keys.append(KEY)
    whole, status, _ = run_job_floats(Job.hca_decode(items, keys=keys))
    assert not status.any()
    weights = [shard.hca_weight(h) for h in items]
    seen = set()
    for r in range(world):
        mine = shard.my_items(weights, r, world)
        seen.update(mine)
        if not mine:
            continue
        outs, status, _ = run_job_floats(Job.hca_decode([items[i] for i in mine], keys=[keys[i] for i in mine]))
        assert not status.any()
        for o, i in zip(outs, mine):
            assert bytes(o) == bytes(whole[i]), (r, i)
    assert seen == set(range(len(items)))


# ------------------------------------------------------------------------------------------------ threads / devices
def test_two_threads_through_the_extension_module(cc):
    """The CPython module releases the GIL around the library calls: two threads decode / encode concurrently and both get
    the oracle's bytes (the library keeps no mutable global state besides the one-time device probe)."""
    import importlib.util
    import os
    import sysconfig
    from pycricodecs_amd import build
    path = os.path.join(build.LIBDIR, "CriCodecs" + sysconfig.get_config_var("EXT_SUFFIX"))
    spec = importlib.util.spec_from_file_location("CriCodecs", path)
    ext = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ext)
    wavs = [synth.wav(200 + i, 9000 + 1000 * i, 2, 48000) for i in range(4)]
    hcas = [O.hca_crypt(O.hca_encode(w, 1), 1, 56, KEY) for w in wavs]
    exp_dec = [O.hca_decode(h, KEY) for h in hcas]
    exp_adx = [O.adx_encode(w) for w in wavs]
    exp_hca = [O.hca_encode(w, 2) for w in wavs]
    errors = []

    def worker(tid):
        try:
            for rep in range(6):
                for i in range(len(wavs)):
                    k = (i + tid) % len(wavs)
                    if ext.HcaDecode(hcas[k], 96, KEY, 0) != exp_dec[k]:
                        errors.append(("dec", tid, rep, k))
                    if ext.AdxEncode(wavs[k], 4, 18, 3, 500, 0, 4, False) != exp_adx[k]:
                        errors.append(("adx", tid, rep, k))
                    if ext.HcaEncode(wavs[k], 0, 2) != exp_hca[k]:
                        errors.append(("enc", tid, rep, k))
        except Exception as e:  # noqa: BLE001
            errors.append((tid, repr(e)))

    ts = [threading.Thread(target=worker, args=(t,)) for t in range(2)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errors, errors[:4]


def test_device_selection_entry_points(cc):
    from pycricodecs_amd import _capi
    from pycricodecs_amd.batch import Job
    L = _capi.lib()
    n = L.cri_device_count()
    assert n >= 1 and L.cri_get_device() == 0
    assert L.cri_set_device(0) == 0
    assert L.cri_set_device(n) == -301 and L.cri_set_device(-1) == -301
    job = Job.hca_decode([G.load("s0_3008_2_48000_q1.hca")])
    assert L.cri_job_device(job._h) == 0
    # a job created in one thread runs (host wrapper) from another
    res = {}

    def other():
        outs, st = job.run_host()
        res["out"] = bytes(outs[0]); res["st"] = int(st[0])
    t = threading.Thread(target=other)
    t.start(); t.join()
    assert res["st"] == 0 and res["out"] == G.load("s0_3008_2_48000_q1.decoded.wav")


@pytest.mark.parametrize("order", ["default", "pipelined", "pipelined-small-pieces", "one-piece"])
def test_run_host_equals_device_resident_run(cc, knobs, order):
    """cri_job_run_host_items / _into (host buffers in and out, the arena's private streams) give the bytes and statuses of the
    device-resident cri_job_run, and those are the oracle's -- in one piece and pipelined (CRICODECS_HOST_SLICE_MIN=0: tile slices,
    uploads pulled by k_pull_host from page-locked or staged memory, downloads beside them), from separate items (staged; with
    CRICODECS_HOST_STAGE_PIECE=1000 every item straddles several staging pieces), from a pageable blob (locked for the call), from
    a page-locked blob, into pageable and into page-locked memory."""
    import ctypes as C
    from pycricodecs_amd import _capi
    from pycricodecs_amd.batch import Job, pinned_array
    if order != "default":
        knobs(host_slice_min=(1 << 62) if order == "one-piece" else 0)
    if order == "pipelined-small-pieces":
        knobs(host_stage_piece=1000)
    uniq, items = _ragged_hca_batch()
    job = Job.hca_decode(items, keys=[KEY] * len(items))
    assert job.host_status[17] != 0
    want, st_dev = run_job(job)
    st_want = np.where(job.host_status != 0, job.host_status, st_dev)
    for rep in range(2):                                        # (the second call runs on the cached arena)
        outs, st = job.run_host()
        assert (st == st_want).all()
        for i, (a, b) in enumerate(zip(outs, want)):
            assert bytes(a) == bytes(b), (rep, i)
    refs = {id(u): O.hca_decode(u, KEY) for u in uniq}
    for i, it in enumerate(items):
        if id(it) in refs:
            assert bytes(want[i]) == refs[id(it)], i
    # into page-locked memory
    buf = pinned_array(job.output_bytes)
    outs, st = job.run_host(out=buf)
    for i, (a, b) in enumerate(zip(outs, want)):
        assert bytes(a) == bytes(b), i
    # the blob form: a pageable blob, then the same bytes in page-locked memory at an odd address
    blob = job.blob
    status = (C.c_int32 * job.n)()
    buf[:] = 0xEE
    assert _capi.lib().cri_job_run_host_into(job._h, blob, buf.ctypes.data, status) == 0
    assert (np.array(status[:job.n]) == st_want).all()
    for i, (a, b) in enumerate(zip(job.split(memoryview(buf)), want)):
        assert bytes(a) == bytes(b), i
    pin_in = pinned_array(len(blob))
    pin_in[:] = np.frombuffer(blob, dtype=np.uint8)
    page_out = np.full(job.output_bytes, 0xEE, dtype=np.uint8)
    assert _capi.lib().cri_job_run_host_into(job._h, pin_in.ctypes.data, page_out.ctypes.data, status) == 0
    for i, (a, b) in enumerate(zip(job.split(memoryview(page_out)), want)):
        assert bytes(a) == bytes(b), i
    # bytes no kernel writes are zero, in every order (alignment gaps between the items)
    o = job.output_offsets
    for i in range(job.n - 1):
        end = int(o[i]) + len(want[i])
        assert not page_out[end:int(o[i + 1])].any(), i
    del outs
    del buf, pin_in


@pytest.mark.parametrize("order", ["pipelined", "pipelined-small-pieces", "one-piece"])
def test_run_host_items_at_caller_offsets(cc, knobs, order):
    """HCA streams placed at caller offsets with gaps between them (the gaps are zero on the device whatever the staging slots
    held before), decoded from the items' own buffers."""
    from pycricodecs_amd.batch import Job
    knobs(host_slice_min=(1 << 62) if order == "one-piece" else 0)
    if order == "pipelined-small-pieces":
        knobs(host_stage_piece=777)
    uniq, items = _ragged_hca_batch()
    items = [it for k, it in enumerate(items) if k != 17][:40]
    offs = np.zeros(len(items) + 1, dtype=np.uint64)
    for i, it in enumerate(items):
        offs[i + 1] = (int(offs[i]) + len(it) + 1000 + 37 * i) // 64 * 64
    big = Job.hca_decode([uniq[0]] * 3, keys=[KEY] * 3)         # leaves non-zero bytes in the arena's input buffer and staging slots
    big.run_host()
    job = Job.hca_decode(items, keys=[KEY] * len(items), offsets=offs)
    want, st_dev = run_job(job)
    outs, st = job.run_host()
    assert (st == np.where(job.host_status != 0, job.host_status, st_dev)).all()
    for i, (a, b) in enumerate(zip(outs, want)):
        assert bytes(a) == bytes(b), i
    refs = {id(u): O.hca_decode(u, KEY) for u in uniq}
    for i, it in enumerate(items):
        if id(it) in refs:
            assert bytes(outs[i]) == refs[id(it)], i


def test_run_host_blob_form_and_items_with_offsets(cc):
    """The blob form (cri_job_run_host_into) on a job made from one blob; a job made from items placed at caller offsets has no
    blob form (CRI_ERR_INVALID_ARG) and runs through cri_job_run_host_items."""
    import ctypes as C
    from pycricodecs_amd import _capi
    from pycricodecs_amd.batch import Job, pack
    adx = [O.adx_encode(synth.wav(610 + k, 3200 + 640 * k, 1 + k % 2, 48000)) for k in range(5)]
    refs = [O.adx_decode(a) for a in adx]
    blob, offs = pack(adx)
    h = C.c_void_p()
    rc = _capi.lib().cri_job_create_adx_decode(blob, offs.ctypes.data_as(C.POINTER(C.c_uint64)), len(adx), C.byref(h))
    assert rc == 0
    job = Job(h, blob, offs)
    outs, st = job.run_host()
    assert not st.any() and [bytes(o) for o in outs] == refs
    # items at 256-byte aligned device offsets
    aligned = np.zeros(len(adx) + 1, dtype=np.uint64)
    for i, a in enumerate(adx):
        aligned[i + 1] = (int(aligned[i]) + len(a) + 255) // 256 * 256
    job2 = Job.adx_decode(adx, offsets=aligned)
    outs, st = job2.run_host()
    assert not st.any() and [bytes(o) for o in outs] == refs
    out = np.empty(max(job2.output_bytes, 1), dtype=np.uint8)
    status = (C.c_int32 * len(adx))()
    assert _capi.lib().cri_job_run_host_into(job2._h, blob, out.ctypes.data, status) == -301


def test_single_file_calls_reuse_the_arena(cc):
    """Back-to-back single-file calls of different sizes and kinds (the arena grows, is reused, and cri_release_cache drops it)."""
    from pycricodecs_amd import _capi
    for rep in range(3):
        for n in (320, 48000, 4800, 96000):
            w = synth.wav(900 + n % 7, n, 2, 48000)
            a = cc.AdxEncode(w, 4, 18, 3, 500, 0, 4, False)
            assert a == O.adx_encode(w) and cc.AdxDecode(a) == O.adx_decode(a)
            h = cc.HcaEncode(w, False, 1)
            assert h == O.hca_encode(w, 1) and cc.HcaDecode(h, 96, 0, 0) == O.hca_decode(h)
        _capi.lib().cri_release_cache()


def test_run_host_from_two_threads_at_once(cc, knobs):
    """Two threads in the pipelined host path at the same time (one works on the device's arena, the other on buffers, streams
    and staging slots of its own for the call), both from the same items and into their own buffers, several times over."""
    import threading
    from pycricodecs_amd.batch import Job
    knobs(host_slice_min=0)
    knobs(host_stage_piece=4096)
    uniq, items = _ragged_hca_batch()
    items = [it for k, it in enumerate(items) if k != 17]
    refs = {id(u): O.hca_decode(u, KEY) for u in uniq}
    errors = []

    def work(seed):
        try:
            order = list(np.random.default_rng(seed).permutation(len(items)))
            mine = [items[i] for i in order]
            job = Job.hca_decode(mine, keys=[KEY] * len(mine))
            for rep in range(4):
                outs, st = job.run_host()
                for i, it in enumerate(mine):
                    if id(it) in refs and bytes(outs[i]) != refs[id(it)]:
                        errors.append((seed, rep, i)); return
        except Exception as e:                                  # noqa: BLE001
            errors.append((seed, repr(e)))

    ts = [threading.Thread(target=work, args=(s,)) for s in (1, 2, 3)]
    for t in ts: t.start()
    for t in ts: t.join()
    assert not errors, errors[:3]


# ------------------------------------------------------------------------------------------------ b: cri_job_run inside a hipGraph
@pytest.mark.parametrize("kind", ["hca_decode", "hca_encode", "adx_decode", "adx_encode", "hca_crypt"])
def test_job_run_captured_in_a_hip_graph(cc, kind):
    """include/cricodecs_hip.h: "cri_job_run only enqueues kernels on the caller's stream and can be captured in a hipGraph".  One
    capture, three replays over zeroed output buffers, every replay's bytes are the oracle's (small banks are launch-bound: this is
    how a caller amortises the launches)."""
    import torch
    from pycricodecs_amd.batch import Job
    wavs = [synth.wav(4100 + k, 32 * (40 + 300 * k), 1 + k % 2, 48000) for k in range(6)]
    if kind == "hca_decode":
        items = [O.hca_crypt(O.hca_encode(w, 1 + k % 3), 1, 56, KEY) for k, w in enumerate(wavs)]
        job, refs = Job.hca_decode(items, keys=[KEY] * len(items)), [O.hca_decode(h, KEY) for h in items]
    elif kind == "hca_encode":
        job, refs = Job.hca_encode(wavs, quality=2), [O.hca_encode(w, 2) for w in wavs]
    elif kind == "adx_decode":
        items = [O.adx_encode(w) for w in wavs]
        job, refs = Job.adx_decode(items), [O.adx_decode(a) for a in items]
    elif kind == "adx_encode":
        job, refs = Job.adx_encode(wavs), [O.adx_encode(w) for w in wavs]
    else:
        items = [O.hca_encode(w, 1) for w in wavs]
        job, refs = Job.hca_crypt(items, True, 56, keys=[KEY] * len(items)), [O.hca_crypt(h, 1, 56, KEY) for h in items]
    bufs = job.alloc("cuda:0")
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        job.run(*bufs)                                             # (uncaptured once: modules loaded, arena-free path)
    s.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        job.run(*bufs)
    for _ in range(3):
        bufs[1].zero_(); bufs[3].fill_(-1)
        torch.cuda.synchronize()
        g.replay()
        torch.cuda.synchronize()
        assert int(bufs[3][:job.n].abs().sum().item()) == 0
        outs = job.split(bytes(bufs[1].cpu().numpy()))
        for i, (o, r) in enumerate(zip(outs, refs)):
            assert bytes(o) == r, (kind, i)
    del g


# ------------------------------------------------------------------------------------------------ b: a job may die while its kernels run
def test_job_destroyed_with_work_in_flight(cc):
    """cri_job_destroy waits for the job's last enqueued run before its metadata allocation is recycled into the next job
    (ADVICE r3): a job is run on a side stream and dropped at once, a second job of the same size is planned and run right
    behind it; both outputs are the oracle's."""
    import gc
    import torch
    from pycricodecs_amd.batch import Job
    a_items = [O.hca_crypt(O.hca_encode(synth.wav(4300 + k, 48000 * 2, 2, 48000), 1), 1, 56, KEY) for k in range(4)] * 60
    b_items = [O.hca_encode(synth.wav(4400 + k, 48000 * 2, 1, 48000), 3) for k in range(4)] * 60
    a_refs = [O.hca_decode(h, KEY) for h in a_items[:4]]
    b_refs = [O.hca_decode(h) for h in b_items[:4]]
    side = torch.cuda.Stream()
    for _ in range(3):
        ja = Job.hca_decode(a_items, keys=[KEY] * len(a_items))
        bufs_a = ja.alloc("cuda:0")
        offs_a = ja.output_offsets.copy()
        torch.cuda.synchronize()
        ja.run(*bufs_a, stream=side)
        del ja
        gc.collect()                                               # (the handle is destroyed here, kernels possibly still running)
        jb = Job.hca_decode(b_items)
        outs_b, st_b = run_job(jb)
        torch.cuda.synchronize()
        assert not st_b.any()
        for i in range(len(b_items)):
            assert bytes(outs_b[i]) == b_refs[i % 4], i
        assert int(bufs_a[3].abs().sum().item()) == 0
        blob = bytes(bufs_a[1].cpu().numpy())
        for i in range(len(a_items)):
            o = int(offs_a[i])
            assert blob[o:o + len(a_refs[i % 4])] == a_refs[i % 4], i


# ------------------------------------------------------------------------------------------------ the pipelined host path, ADX decode
@pytest.mark.parametrize("order", ["pipelined", "pipelined-small-pieces", "one-piece"])
def test_adx_decode_run_host_in_parts(cc, knobs, order):
    """An ADX decode job cannot be cut inside (its lanes are laid out by length, not by item), so its pipelined host path plans the
    items again as a few jobs over consecutive item ranges (run_host_core, host_parts_ready) that write where the whole job would:
    bytes and statuses equal the device-resident run's and the oracle's -- from separate items (staged in small pieces too), from a
    pageable blob and from a page-locked one, into pageable and page-locked memory, with items the host rejects among them."""
    from pycricodecs_amd import _capi
    from pycricodecs_amd.batch import Job, pinned_array
    knobs(host_slice_min=(1 << 62) if order == "one-piece" else 0)
    if order == "pipelined-small-pieces":
        knobs(host_stage_piece=1000)
    rng = np.random.default_rng(91)
    wavs = [synth.wav(700 + k, int(rng.integers(40, 60000)), 1 + k % 2, 48000) for k in range(10)]
    uniq = [O.adx_encode(w) for w in wavs] + [O.adx_encode(wavs[2], bitdepth=8), O.adx_encode(wavs[3], mode=4)]
    items = [uniq[int(k)] for k in rng.integers(0, len(uniq), 90)]
    items[11] = b"\x80\x00" + bytes(64)                          # rejected on the host
    items[57] = uniq[1][:40]                                    # a header cut short
    job = Job.adx_decode(items)
    assert job.host_status[11] != 0 and job.host_status[57] != 0
    want, st_dev = run_job(job)
    st_want = np.where(job.host_status != 0, job.host_status, st_dev)
    refs = {id(u): O.adx_decode(u) for u in uniq}
    for i, it in enumerate(items):
        if id(it) in refs:
            assert bytes(want[i]) == refs[id(it)], i
    for rep in range(2):                                        # (the second call finds the parts made by the first)
        outs, st = job.run_host()
        assert (st == st_want).all()
        for i, (a, b) in enumerate(zip(outs, want)):
            assert bytes(a) == bytes(b), (rep, i)
    buf = pinned_array(job.output_bytes)
    outs, st = job.run_host(out=buf)
    for i, (a, b) in enumerate(zip(outs, want)):
        assert bytes(a) == bytes(b), i
    blob = job.blob
    status = (C.c_int32 * job.n)()
    buf[:] = 0xEE
    assert _capi.lib().cri_job_run_host_into(job._h, blob, buf.ctypes.data, status) == 0
    assert (np.array(status[:job.n]) == st_want).all()
    for i, (a, b) in enumerate(zip(job.split(memoryview(buf)), want)):
        assert bytes(a) == bytes(b), i
    pin_in = pinned_array(len(blob))
    pin_in[:] = np.frombuffer(blob, dtype=np.uint8)
    page_out = np.full(job.output_bytes, 0xEE, dtype=np.uint8)
    assert _capi.lib().cri_job_run_host_into(job._h, pin_in.ctypes.data, page_out.ctypes.data, status) == 0
    for i, (a, b) in enumerate(zip(job.split(memoryview(page_out)), want)):
        assert bytes(a) == bytes(b), i
    o = job.output_offsets                                      # bytes no kernel writes are zero
    for i in range(job.n - 1):
        end = int(o[i]) + len(want[i])
        assert not page_out[end:int(o[i + 1])].any(), i
    del outs
    del buf, pin_in


@pytest.mark.parametrize("order", ["pipelined", "pipelined-small-pieces"])
def test_hca_decode_of_mixed_formats_run_host_in_parts(cc, knobs, order):
    """An HCA decode job of several format groups (channel counts, qualities, keys) runs its groups one after the other, each over
    items from anywhere in the batch: the pipelined host path plans it again as parts over item ranges, like an ADX decode job."""
    from pycricodecs_amd.batch import Job
    knobs(host_slice_min=0)
    if order == "pipelined-small-pieces":
        knobs(host_stage_piece=1500)
    rng = np.random.default_rng(17)
    uniq, keys = [], []
    for k, (ch, q, key) in enumerate([(2, 1, KEY), (1, 1, 0), (2, 4, KEY), (6, 1, 0), (2, 2, 12345), (4, 3, KEY), (2, 1, 0)]):
        h = O.hca_encode(synth.wav(900 + k, int(rng.integers(3000, 40000)), ch, 48000), q)
        uniq.append(O.hca_crypt(h, 1, 56, key) if key else h); keys.append(key)
    pick = rng.integers(0, len(uniq), 60)
    items = [uniq[int(k)] for k in pick]
    item_keys = [keys[int(k)] for k in pick]
    items[9] = b"HCA\0" + bytes(100)
    job = Job.hca_decode(items, keys=item_keys)
    assert len(job.transform_forms()) > 3
    want, st_dev = run_job(job)
    st_want = np.where(job.host_status != 0, job.host_status, st_dev)
    refs = [O.hca_decode(u, k) for u, k in zip(uniq, keys)]
    for i, k in enumerate(pick):
        if i != 9:
            assert bytes(want[i]) == refs[int(k)], i
    for rep in range(2):
        outs, st = job.run_host()
        assert (st == st_want).all()
        for i, (a, b) in enumerate(zip(outs, want)):
            assert bytes(a) == bytes(b), (rep, i)


# ------------------------------------------------------------------------------------------------ b: a launch that fails is reported
@pytest.mark.parametrize("captured", [False, True])
def test_a_failed_launch_is_reported_also_while_capturing(cc, knobs, captured):
    """cri_job_run's verdict is the launches' own: a kernel the runtime refuses (test knob `bad_launch`: more LDS than a compute unit
    has) makes the run return CRI_ERR_HIP -- also while the stream is being captured into a hipGraph, where the bookkeeping behind
    the launches used to clear the error before it was read."""
    import torch
    from pycricodecs_amd import _capi
    from pycricodecs_amd.batch import Job
    wavs = [synth.wav(5100 + k, 32 * 200, 2, 48000) for k in range(3)]
    job = Job.adx_encode(wavs)
    bufs = job.alloc("cuda:0")
    job.run(*bufs)
    torch.cuda.synchronize()
    knobs(bad_launch=1)
    s = torch.cuda.Stream()
    if captured:
        g = torch.cuda.CUDAGraph()
        with pytest.raises(_capi.CriCodecsError) as e:
            with torch.cuda.graph(g, stream=s):
                job.run(*bufs)
        del g
    else:
        with pytest.raises(_capi.CriCodecsError) as e:
            job.run(*bufs, stream=s)
    assert e.value.code == -303                                   # CRI_ERR_HIP
    torch.cuda.synchronize()
    knobs(bad_launch=0)
    outs, st = run_job(job)                                        # and the job is still good
    assert not st.any()
    for o, w in zip(outs, wavs):
        assert bytes(o) == O.adx_encode(w)


# ------------------------------------------------------------------------------------------------ b: one job, several streams, destroyed in flight
def test_job_run_on_several_streams_then_destroyed(cc):
    """cri_job_destroy waits for the last run on EVERY stream the job was enqueued on (one event per stream, made under a lock): a
    job run on three streams and dropped at once leaves three correct outputs, and the next job -- which takes over the recycled
    metadata allocation -- is right as well."""
    import torch
    from pycricodecs_amd.batch import Job
    items = [O.hca_crypt(O.hca_encode(synth.wav(5200 + k, 1024 * 40 + 77 * k, 2, 48000), 1), 1, 56, KEY) for k in range(12)]
    refs = [O.hca_decode(h, KEY) for h in items]
    job = Job.hca_decode(items, keys=[KEY] * len(items))
    streams = [torch.cuda.Stream() for _ in range(3)]
    sets = [job.alloc("cuda:0") for _ in streams]
    for s, bufs in zip(streams, sets):
        job.run(*bufs, stream=s)
    offs = [int(job.output_offsets[i]) for i in range(job.n)]
    del job                                                        # destroyed with work in flight on three streams
    other = Job.adx_encode([synth.wav(5300 + k, 32 * 500, 2, 48000) for k in range(4)])
    outs2, st2 = run_job(other)
    torch.cuda.synchronize()
    for bufs in sets:
        assert int(bufs[3].abs().sum().item()) == 0
        blob = bytes(bufs[1].cpu().numpy())
        for o, r in zip(offs, refs):
            assert blob[o:o + len(r)] == r
    assert not st2.any()
    for k, o in enumerate(outs2):
        assert bytes(o) == O.adx_encode(synth.wav(5300 + k, 32 * 500, 2, 48000))


# ------------------------------------------------------------------------------------------------ c: the randomised soak, briefly
def test_randomised_parity_soak_for_a_few_seconds(cc):
    """tools/parity_soak.py (random banks through every batch job and single-file call, each output against the oracle; the long runs are
    under profiles/) for ten seconds with a seed of its own, one bank of 1000-4000 items among the rounds: no mismatch."""
    import subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "parity_soak.py"), "10", "20260929", "3"], cwd=root, capture_output=True, text=True, timeout=600)
    tail = "\n".join(r.stdout.splitlines()[-20:])
    assert r.returncode == 0, tail + r.stderr[-2000:]
    assert "TOTAL" in tail and " 0 mismatches" in tail.splitlines()[-1], tail
