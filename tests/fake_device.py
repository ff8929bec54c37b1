"""A test double of the device layer, for DRY RUNS of bench.py's own logic in this container (no GPU): `FakeJob` has the interface of
pycricodecs_amd.batch.Job that bench.py uses and produces every output with the CPU oracle; `install(bench_module)` puts it, a `Dist`
without a device and no-op stand-ins for the torch.cuda calls in place.

TEST INFRASTRUCTURE ONLY.  Nothing outside tests/ imports this; bench.py has no switch that reaches it (its `value` can only come from
the HIP library), and the timings a dry run prints are meaningless.  What a dry run proves is that the script's control flow, its
verification, its gather over ranks and its output contract (one JSON line <= 4 KB + bench_detail.json) work for every workload --
the parts of round 6 that could not be run on a GPU box while the pool was closed."""
import os

import numpy as np

import oracle_lib as O


def _align(x, a=64):
    return (x + a - 1) // a * a


class FakeJob:
    KMS = {"hca_decode": {"k_hca_parse": 0.9, "k_hca_transform": 1.1}, "hca_encode": {"k_hca_encode": 2.0},
           "adx_encode": {"k_adx_lane_encode": 1.3}, "adx_decode": {"k_adx_seg_decode": 0.7}, "hca_crypt": {"k_hca_crypt_wpf": 0.1}}

    def __init__(self, kind, items, fn, units_of, alg_of, skip=None):
        self.kind, self.items, self._fn = kind, list(items), fn
        self.n = len(self.items)
        self._skip = skip or [False] * self.n
        self._outs = {}
        sizes, units, alg = [], 0, 0
        for i, b in enumerate(self.items):
            if self._skip[i]:
                sizes.append(0)
                continue
            o = self._out(b)
            sizes.append(len(o))
            units += units_of(b, o)
            alg += alg_of(b, o)
        self.item_sizes = np.array(sizes, dtype=np.uint64)
        offs, pos = [0], 0
        for sz in sizes:
            pos += _align(sz)
            offs.append(pos)
        self.output_offsets = np.array(offs, dtype=np.uint64)
        ioffs, pos = [0], 0
        for b in self.items:
            pos += _align(len(b), 32)
            ioffs.append(pos)
        self.offsets = np.array(ioffs, dtype=np.uint64)
        self.input_bytes, self.output_bytes, self.scratch_bytes = int(ioffs[-1]), int(offs[-1]), 64
        self.units, self.units2, self.algorithmic_bytes = int(units), int(units) * 2, int(alg)
        self.dominant_kernel = max(self.KMS[kind], key=self.KMS[kind].get)
        self.host_status = np.zeros(self.n, dtype=np.int32)

    def _out(self, b):
        k = id(b)
        if k not in self._outs:
            self._outs[k] = self._fn(b)
        return self._outs[k]

    # ---- constructors (the subset bench.py uses)
    @staticmethod
    def _hca_frames(h):
        return int.from_bytes(h[16:20], "big")

    @classmethod
    def hca_decode(cls, items, keys=None, subkeys=None, offsets=None):
        keys = keys or [0] * len(items)
        subkeys = subkeys or [0] * len(items)
        kmap = {id(b): (k, s) for b, k, s in zip(items, keys, subkeys)}
        return cls("hca_decode", items, lambda h: O.hca_decode(h, *kmap[id(h)]), lambda h, o: cls._hca_frames(h),
                   lambda h, o: cls._hca_frames(h) * (int.from_bytes(h[0x1C:0x1E], "big") + 2048 * h[12]))

    @classmethod
    def hca_encode(cls, items, quality=1, force_no_loop=False, offsets=None):
        return cls("hca_encode", items, lambda w: O.hca_encode(w, quality, force_no_loop), lambda w, o: cls._hca_frames(o),
                   lambda w, o: cls._hca_frames(o) * (int.from_bytes(o[0x1C:0x1E], "big") + 2048 * o[12]))

    @classmethod
    def hca_crypt(cls, items, encrypt, ctype, keys=None, subkeys=None, offsets=None):
        keys = keys or [0] * len(items)
        kmap = {id(b): k for b, k in zip(items, keys)}
        return cls("hca_crypt", items, lambda h: O.hca_crypt(h, int(encrypt), ctype, kmap[id(h)]), lambda h, o: cls._hca_frames(h), lambda h, o: 2 * len(h))

    @staticmethod
    def _adx_rows(a):
        spb = (a[5] - 2) * 8 // a[6]
        return -(-int.from_bytes(a[12:16], "big") // spb)

    @classmethod
    def adx_encode(cls, items, bitdepth=4, blocksize=18, mode=3, highpass=500, filt=0, version=4, force_no_loop=False, offsets=None):
        return cls("adx_encode", items, lambda w: O.adx_encode(w, bitdepth, blocksize, mode, highpass, filt, version, force_no_loop),
                   lambda w, o: cls._adx_rows(o), lambda w, o: cls._adx_rows(o) * o[7] * 82)

    @classmethod
    def adx_decode(cls, items, offsets=None):
        return cls("adx_decode", items, O.adx_decode, lambda a, o: cls._adx_rows(a), lambda a, o: cls._adx_rows(a) * a[7] * 82)

    @classmethod
    def awb_decode(cls, awb, key=0):
        """(HCA job, ADX job) over one AFS2 bank, as batch.Job.awb_decode: items of the other kind are skipped (status 1)."""
        import struct
        magic, _ver, osz, _idsz, n, align, subkey = struct.unpack_from("<4sBBHIHH", awb, 0)
        assert magic == b"AFS2" and osz == 4
        offs = np.frombuffer(awb, dtype="<u4", count=n + 1, offset=16 + 2 * n).astype(np.int64)
        items = []
        for i in range(n):
            o = int(_align(int(offs[i]), align))
            items.append(bytes(awb[o:int(offs[i + 1])]))
        is_hca = [(b[0] & 0x7F) == 0x48 and (b[1] & 0x7F) == 0x43 for b in items]      # "HCA" with or without the cipher's mask
        hj = cls("hca_decode", items, lambda h: O.hca_decode(h, key, subkey), lambda h, o: cls._hca_frames(h),
                 lambda h, o: cls._hca_frames(h) * (int.from_bytes(h[0x1C:0x1E], "big") + 2048 * h[12]), skip=[not x for x in is_hca])
        aj = cls("adx_decode", items, O.adx_decode, lambda a, o: cls._adx_rows(a), lambda a, o: cls._adx_rows(a) * a[7] * 82, skip=is_hca)
        hj._awb_bytes = aj._awb_bytes = len(awb)
        hj.input_bytes = aj.input_bytes = len(awb)
        return hj, aj

    # ---- "device" execution
    def alloc(self, device="cpu", upload=True):
        import torch
        return (torch.zeros(max(self.input_bytes, 1), dtype=torch.uint8), torch.zeros(max(self.output_bytes, 1), dtype=torch.uint8),
                torch.zeros(max(self.scratch_bytes, 1), dtype=torch.uint8), torch.zeros(max(self.n, 1), dtype=torch.int32))

    def run(self, d_in, d_out, d_scratch, d_status, stream=None):
        import torch
        for i, b in enumerate(self.items):
            if self._skip[i]:
                d_status[i] = 1
                continue
            o = int(self.output_offsets[i])
            out = self._out(b)
            d_out[o:o + len(out)] = torch.frombuffer(bytearray(out), dtype=torch.uint8)
            d_status[i] = 0

    def enable_events(self, on=True):
        pass

    def event_ms(self):
        return dict(self.KMS[self.kind])

    def record_census(self, d_scratch):
        return {"frames": self.units, "narrow": self.units}

    def transform_forms(self):
        return [2]

    def split(self, blob):
        return [blob[int(self.output_offsets[i]):int(self.output_offsets[i]) + int(self.item_sizes[i])] for i in range(self.n)]


class _Stream:
    def __init__(self, *a, **k):
        pass

    def wait_stream(self, other):
        pass


def make_dist(bench):
    class FakeDist(bench.Dist):
        """bench.Dist without a device: ranks (if any) rendezvous over gloo, tensors live on the CPU."""

        def __init__(self):
            self.rank = int(os.environ.get("RANK", "0"))
            self.world = int(os.environ.get("WORLD_SIZE", "1"))
            self.local = int(os.environ.get("LOCAL_RANK", "0"))
            self.shared, self.dev, self.numa = True, "cpu", None
            self.affinity0 = os.sched_getaffinity(0)
            if self.world > 1:
                import torch.distributed as dist
                dist.init_process_group("gloo")
    return FakeDist


def install(bench, monkeypatch=None):
    """Route `bench` (the imported bench.py module) through the fakes.  With a pytest monkeypatch everything is undone after the test."""
    import torch
    import pycricodecs_amd.batch as batch

    def setattr_(obj, name, value):
        if monkeypatch is not None:
            monkeypatch.setattr(obj, name, value, raising=False)
        else:
            setattr(obj, name, value)
    setattr_(batch, "Job", FakeJob)
    setattr_(bench, "Dist", make_dist(bench))
    setattr_(torch.cuda, "synchronize", lambda *a, **k: None)
    setattr_(torch.cuda, "empty_cache", lambda *a, **k: None)
    setattr_(torch.cuda, "Stream", _Stream)
    setattr_(torch.cuda, "current_stream", lambda *a, **k: _Stream())
    # parts of the default run that are the library's own host paths (no stand-in for those here)
    setattr_(bench, "sec_host_paths", lambda args, D, out: out.update(hca_decode_host={"skipped": "dry run"}, adx_decode_host={"skipped": "dry run"}))
    setattr_(bench, "sec_usm", lambda args, D, out: out.update(sfa_pack={"skipped": "dry run"}, usm_demux={"skipped": "dry run"}))
    setattr_(bench, "single_call_latency", lambda seconds: {"skipped": "dry run"})


def install_emulated(bench, monkeypatch=None):
    """The other kind of dry run (CRI_TEST_HOSTWAVE=1, tests/test_hostwave.py): bench.py on the REAL batch.Job over the EMULATED build of the
    library (tests/hostwave: the product's kernel sources on the CPU) -- every call of the C ABI the script makes, the library's own host
    paths, the event read-out and the job's own figures (units, algorithmic bytes, dominant kernel) are the real ones; only the device
    layer of torch is replaced.  Timings mean nothing here either."""
    import sys
    import torch
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "hostwave"))
    import mode
    assert mode.enable(), "CRI_TEST_HOSTWAVE=1 and CRICODECS_LIB_DIR=tests/hostwave/lib are the caller's to set"

    def setattr_(obj, name, value):
        if monkeypatch is not None:
            monkeypatch.setattr(obj, name, value, raising=False)
        else:
            setattr(obj, name, value)
    setattr_(bench, "Dist", make_dist(bench))
    setattr_(torch.cuda, "empty_cache", lambda *a, **k: None)
