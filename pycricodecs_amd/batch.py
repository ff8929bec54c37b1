"""Batch jobs on device-resident buffers (AFS2-style blob + offsets), the path bench.py measures.

PyTorch is used only as the owner of device memory and streams; pointers are handed to the C ABI as integers.
"""
import ctypes as C
import weakref

import numpy as np

from . import _capi

KIND_NAMES = {1: "adx_decode", 2: "adx_encode", 3: "hca_decode", 4: "hca_encode", 5: "hca_crypt", 6: "usm_demux", 7: "sfa_pack"}


def pack(items):
    """list of bytes -> (blob bytes, offsets uint64[n+1])."""
    offs = np.zeros(len(items) + 1, dtype=np.uint64)
    if items:
        offs[1:] = np.cumsum([len(b) for b in items], dtype=np.uint64)
    return b"".join(items), offs


class CriItems(C.Structure):
    _fields_ = [("ptrs", C.POINTER(C.c_void_p)), ("lens", C.POINTER(C.c_uint64)), ("offsets", C.POINTER(C.c_uint64)), ("n", C.c_uint32)]


def items_struct(items, offsets=None):
    """cri_items over a list of bytes objects (no copy: the library reads the host bytes only while it plans the job).
    offsets: device offsets uint64[n+1] (e.g. another job's output offsets), or None for packed.  Returns (struct, keepalive)."""
    n = len(items)
    ptrs = (C.c_void_p * max(n, 1))()
    lens = (C.c_uint64 * max(n, 1))()
    keep = []
    for i, b in enumerate(items):
        cp = C.c_char_p(b)
        keep.append(cp)
        ptrs[i] = C.cast(cp, C.c_void_p).value
        lens[i] = len(b)
    offs = None
    if offsets is not None:
        offs = np.ascontiguousarray(offsets, dtype=np.uint64)
        assert offs.shape == (n + 1,)
    st = CriItems(C.cast(ptrs, C.POINTER(C.c_void_p)), C.cast(lens, C.POINTER(C.c_uint64)),
                  offs.ctypes.data_as(C.POINTER(C.c_uint64)) if offs is not None else None, n)
    return st, (ptrs, lens, keep, offs, items)


def pinned_array(nbytes):
    """uint8 numpy array over page-locked host memory (cri_pinned_alloc): PCIe copies to / from it are asynchronous DMA.
    The block lives exactly as long as the array or anything derived from it (slices, memoryviews) does."""
    n = max(int(nbytes), 1)
    L = _capi.lib()
    ptr = L.cri_pinned_alloc(n)
    if not ptr:
        raise MemoryError("cri_pinned_alloc(%d)" % n)
    base = (C.c_uint8 * n).from_address(ptr)                   # numpy keeps this object alive for as long as a view of it exists
    weakref.finalize(base, L.cri_pinned_free, ptr)
    return np.ctypeslib.as_array(base)


class Job:
    """Owns a cri_job.  Create with one of the classmethods; `run()` enqueues it on torch's current stream.
    Batches are handed to the library item by item (cri_items): a list that repeats the same bytes object costs no host copy."""

    def __init__(self, handle, blob, offsets, items=None):
        self._h = handle
        self._blob, self.items = blob, items
        L = self._L = _capi.lib()                              # the library that made the handle serves it to the end
        self.n = L.cri_job_items(handle)
        self.kind = KIND_NAMES[L.cri_job_kind(handle)]
        self.input_bytes = L.cri_job_input_bytes(handle)
        self.output_bytes = L.cri_job_output_bytes(handle)
        self.scratch_bytes = L.cri_job_scratch_bytes(handle)
        self.units = L.cri_job_units(handle)
        self.units2 = L.cri_job_units2(handle)
        self.algorithmic_bytes = L.cri_job_algorithmic_bytes(handle)
        self.dominant_kernel = L.cri_job_dominant_kernel(handle).decode()
        self.output_offsets = np.ctypeslib.as_array(L.cri_job_output_offsets(handle), shape=(self.n + 1,)).copy()
        self.offsets = np.ctypeslib.as_array(L.cri_job_input_offsets(handle), shape=(self.n + 1,)).copy() if offsets is None else offsets
        self.host_status = np.ctypeslib.as_array(L.cri_job_host_status(handle), shape=(max(self.n, 1),)).copy()[:self.n]
        tags = L.cri_job_item_tags(handle)
        self.item_tags = np.ctypeslib.as_array(tags, shape=(self.n,)).copy() if tags else None
        sizes = L.cri_job_item_sizes(handle)
        self.item_sizes = np.ctypeslib.as_array(sizes, shape=(self.n,)).copy() if sizes else None

    def __del__(self):
        try:
            if self._h:
                self._L.cri_job_destroy(self._h)
                self._h = None
        except Exception:
            pass

    @property
    def blob(self):
        """The device input as one host bytes object (built on first use for jobs created from an item list)."""
        if self._blob is None:
            buf = bytearray(int(self.input_bytes))
            for i, b in enumerate(self.items):
                o = int(self.offsets[i])
                buf[o:o + len(b)] = b
            self._blob = bytes(buf)
        return self._blob

    # ---- constructors
    @staticmethod
    def _blob_args(items):
        blob, offs = pack(items)
        return blob, offs, (blob if blob else b"\0")      # bytes are passed by pointer, no copy

    @classmethod
    def _finish(cls, rc, h, blob, offs, items=None):
        if rc:
            _capi.raise_for(rc)
        return cls(h, blob, offs, items)

    @staticmethod
    def _as_bytes(items):
        return [b if isinstance(b, bytes) else bytes(b) for b in items]

    @classmethod
    def hca_decode(cls, items, keys=None, subkeys=None, offsets=None):
        items = cls._as_bytes(items)
        st, keep = items_struct(items, offsets)
        k = None if keys is None else (C.c_uint64 * max(len(items), 1))(*[x & 0xFFFFFFFFFFFFFFFF for x in keys])
        s = None if subkeys is None else (C.c_uint16 * max(len(items), 1))(*subkeys)
        h = C.c_void_p()
        rc = _capi.lib().cri_job_create_hca_decode_items(C.byref(st), k, s, C.byref(h))
        return cls._finish(rc, h, None, None, items)

    @classmethod
    def awb_decode(cls, awb, key=0):
        """(HCA decode job, ADX decode job) over one AFS2 / AWB bank: both read the SAME blob (upload it once, e.g. with
        hca_job.alloc(); pass that d_in to both run() calls).  Items of the other kind have status CRI_ITEM_SKIPPED (1)."""
        from .awb import awb_index
        offs, _kinds, _subkey = awb_index(awb)
        hh, ha = C.c_void_p(), C.c_void_p()
        rc = _capi.lib().cri_job_create_awb_decode(bytes(awb) if not isinstance(awb, bytes) else awb, len(awb), key & 0xFFFFFFFFFFFFFFFF, C.byref(hh), C.byref(ha))
        if rc:
            _capi.raise_for(rc)
        return cls(hh, awb, offs), cls(ha, awb, offs)

    @classmethod
    def usm_audio_demux(cls, usm, key=0, decrypt=False):
        """One item per @SFA channel of a USM container (ascending channel number): its audio stream as the reference's
        USM.demux() returns it.  item_tags = channel | codec << 16 (2 ADX, 4 HCA).  The input blob is the container."""
        buf = usm if isinstance(usm, bytes) else bytes(usm)
        h = C.c_void_p()
        rc = _capi.lib().cri_job_create_usm_audio_demux(buf, len(buf), key & 0xFFFFFFFFFFFFFFFF, int(bool(decrypt)), C.byref(h))
        if rc:
            _capi.raise_for(rc)
        return cls(h, buf, np.array([0, len(buf)], dtype=np.uint64))

    @classmethod
    def sfa_pack(cls, items, codec, key=0, encrypt_audio=False):
        """ADX (codec 2) or HCA (codec 4) files -> their @SFA chunk streams (USMBuilder.get_data, usm.py:578-716).
        item_tags = chunks per item."""
        blob, offs, buf = cls._blob_args(items)
        h = C.c_void_p()
        rc = _capi.lib().cri_job_create_sfa_pack(buf, offs.ctypes.data_as(C.POINTER(C.c_uint64)), len(items), codec,
                                                 key & 0xFFFFFFFFFFFFFFFF, int(bool(encrypt_audio)), C.byref(h))
        return cls._finish(rc, h, blob, offs)

    @classmethod
    def adx_decode(cls, items, offsets=None):
        items = cls._as_bytes(items)
        st, keep = items_struct(items, offsets)
        h = C.c_void_p()
        rc = _capi.lib().cri_job_create_adx_decode_items(C.byref(st), C.byref(h))
        return cls._finish(rc, h, None, None, items)

    @classmethod
    def adx_encode(cls, items, bitdepth=4, blocksize=18, mode=3, highpass=500, filt=0, version=4, force_no_loop=False, offsets=None):
        items = cls._as_bytes(items)
        st, keep = items_struct(items, offsets)
        p = _capi.AdxEncodeParams(bitdepth, blocksize, mode, highpass, filt, version, int(force_no_loop))
        h = C.c_void_p()
        rc = _capi.lib().cri_job_create_adx_encode_items(C.byref(st), C.byref(p), C.byref(h))
        return cls._finish(rc, h, None, None, items)

    @classmethod
    def hca_encode(cls, items, quality=1, force_no_loop=False, offsets=None):
        items = cls._as_bytes(items)
        st, keep = items_struct(items, offsets)
        h = C.c_void_p()
        rc = _capi.lib().cri_job_create_hca_encode_items(C.byref(st), int(force_no_loop), quality, C.byref(h))
        return cls._finish(rc, h, None, None, items)

    @classmethod
    def hca_crypt(cls, items, encrypt, ctype, keys=None, subkeys=None, offsets=None):
        items = cls._as_bytes(items)
        st, keep = items_struct(items, offsets)
        k = None if keys is None else (C.c_uint64 * max(len(items), 1))(*[x & 0xFFFFFFFFFFFFFFFF for x in keys])
        s = None if subkeys is None else (C.c_uint16 * max(len(items), 1))(*subkeys)
        h = C.c_void_p()
        rc = _capi.lib().cri_job_create_hca_crypt_items(C.byref(st), int(encrypt), ctype, k, s, C.byref(h))
        return cls._finish(rc, h, None, None, items)

    # ---- device execution (torch tensors are only buffers here)
    def alloc(self, device="cuda:0", upload=True):
        import torch
        d_in = torch.empty(max(self.input_bytes, 1), dtype=torch.uint8, device=device)
        if upload and self.input_bytes:
            self.upload(d_in)
        d_out = torch.zeros(max(self.output_bytes, 1), dtype=torch.uint8, device=device)
        d_scratch = torch.empty(max(self.scratch_bytes, 1), dtype=torch.uint8, device=device)
        d_status = torch.zeros(max(self.n, 1), dtype=torch.int32, device=device)
        return d_in, d_out, d_scratch, d_status

    def upload(self, d_in):
        """Fill the device input.  Item lists go up one distinct bytes object at a time; repeats are device-to-device copies."""
        import warnings
        import torch
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")                    # (read-only buffers: they are only read)
            if self.items is None:
                d_in[:self.input_bytes].copy_(torch.frombuffer(memoryview(self._blob), dtype=torch.uint8)[:self.input_bytes])
                return
            if self.n and int(self.offsets[self.n]) != sum(len(b) for b in self.items):
                d_in.zero_()                                   # gaps between items (aligned layouts) are defined
            first = {}
            for i, b in enumerate(self.items):
                if not len(b):
                    continue
                o = int(self.offsets[i])
                src = first.get(id(b))
                if src is None:
                    d_in[o:o + len(b)].copy_(torch.frombuffer(memoryview(b), dtype=torch.uint8))
                    first[id(b)] = o
                else:
                    d_in[o:o + len(b)].copy_(d_in[src:src + len(b)])

    def run(self, d_in, d_out, d_scratch, d_status, stream=None):
        import torch
        st = (stream or torch.cuda.current_stream()).cuda_stream
        rc = self._L.cri_job_run(self._h, d_in.data_ptr(), d_out.data_ptr(), d_scratch.data_ptr(),
                                     d_status.data_ptr() if d_status is not None else None, st)
        if rc:
            _capi.raise_for(rc)

    def run_floats(self, d_in, d_out, d_scratch, d_status, stream=None):
        """Validation run of an HCA decode job: run() plus the samples before the int16 conversion.  Returns
        (float32 tensor, offsets uint64[n+1] in floats): item i = [frame][1024][channels] at offsets[i]."""
        import torch
        L = self._L
        n = int(L.cri_job_float_count(self._h))
        offs = np.ctypeslib.as_array(L.cri_job_float_offsets(self._h), shape=(self.n + 1,)).copy()
        d_f = torch.zeros(max(n, 1), dtype=torch.float32, device=d_in.device)
        st = (stream or torch.cuda.current_stream()).cuda_stream
        rc = L.cri_job_run_floats(self._h, d_in.data_ptr(), d_out.data_ptr(), d_scratch.data_ptr(),
                                  d_status.data_ptr() if d_status is not None else None, d_f.data_ptr(), st)
        if rc:
            _capi.raise_for(rc)
        return d_f, offs

    def record_census(self, d_scratch):
        """HCA decode diagnostics after a run: {"frames": n, "narrow": frames whose quantised lines crossed scratch as int8}."""
        import torch
        L = self._L
        arr = (_capi.HcaGroupInfo * 64)()
        n = L.cri_job_hca_groups(self._h, arr, 64)
        frames = narrow = 0
        words = d_scratch.view(torch.int32)
        for g in arr[:n]:
            if not g.frames:
                continue
            base = (g.first_record_offset + g.flags_offset) // 4
            fl = words[base:base + (g.frames - 1) * (g.record_bytes // 4) + 1:g.record_bytes // 4]
            frames += g.frames
            narrow += int(((fl & g.narrow_flag) != 0).sum().item())
        return {"frames": frames, "narrow": narrow}

    def transform_forms(self):
        """HCA decode: the transform kernel of every format group (cri_hca_group_info.transform_form: 0 generic, 1 general,
        2 / 3 / 4 in-lane plain / joint / noise fill, | 8 wide)."""
        arr = (_capi.HcaGroupInfo * 64)()
        n = self._L.cri_job_hca_groups(self._h, arr, 64)
        return [int(g.transform_form) for g in arr[:n]]

    def enable_events(self, on=True):
        self._L.cri_job_enable_events(self._h, 1 if on else 0)

    def event_ms(self):
        """{kernel class name: milliseconds of the last run} (waits for that run)."""
        ms = (C.c_float * 4)()
        names = (C.c_char_p * 4)()
        n = self._L.cri_job_event_ms(self._h, ms, names, 4)
        return {names[i].decode(): float(ms[i]) for i in range(n)}

    def run_host(self, out=None, joined=False):
        """Upload, run, download: returns (list of outputs per item, status int32[n]).  The outputs are memoryviews into one
        buffer owned by this Job and reused by its next run_host() call (copy what must outlive it with bytes(...)), or into
        `out` (a uint8 numpy array of at least output_bytes, e.g. pinned_array()).  A job made from an item list uploads every
        item from its own bytes object (cri_job_run_host_items): no joined copy of the batch is made on the host -- unless
        `joined` asks for it (one upload of Job.blob, built once and kept: faster per call for batches of many small items)."""
        n = max(self.output_bytes, 1)
        if out is None:
            if getattr(self, "_host_out", None) is None or self._host_out.size < n:
                self._host_out = np.empty(n, dtype=np.uint8)
            out = self._host_out
        assert out.dtype == np.uint8 and out.size >= n and out.flags["C_CONTIGUOUS"]
        status = (C.c_int32 * max(self.n, 1))()
        use_blob = self.items is None or (joined and int(self.offsets[self.n]) == sum(len(b) for b in self.items))
        if not use_blob:
            if getattr(self, "_run_items", None) is None:      # (built once: the items list is this Job's for its lifetime)
                self._run_items = items_struct(self.items)
            st, keep = self._run_items
            rc = self._L.cri_job_run_host_items(self._h, C.byref(st), out.ctypes.data, status)
        else:
            blob = self.blob
            rc = self._L.cri_job_run_host_into(self._h, blob if blob else b"\0", out.ctypes.data, status)
        if rc:
            _capi.raise_for(rc)
        return self.split(memoryview(out)), np.array(status[:self.n], dtype=np.int32)

    def item_length(self, blob, i):
        """True byte length of output item i (items are placed with gaps between them -- a decoded WAV so that its samples start a
        128-byte line -- so the item carries its own size)."""
        o = int(self.output_offsets[i])
        if self.kind in ("adx_decode", "hca_decode"):
            return int.from_bytes(blob[o + 4:o + 8], "little") + 8 if blob[o:o + 4] == b"RIFF" else 0
        if self.kind == "hca_crypt":
            return len(self.items[i]) if self.items is not None else int(self.offsets[i + 1] - self.offsets[i])
        if self.item_sizes is not None:
            return int(self.item_sizes[i])
        return None

    def split(self, blob):
        if self.kind in ("adx_decode", "hca_decode") and self.n > 64:
            return self._split_wavs(blob)
        outs = []
        for i in range(self.n):
            o = int(self.output_offsets[i])
            if self.host_status[i]:
                outs.append(b"")
                continue
            n = self.item_length(blob, i)
            if n is None:    # encoders: length is implied by the format
                n = self._encoded_length(blob, i)
            outs.append(blob[o:o + n])
        return outs

    def _split_wavs(self, blob):
        """split() for the decoders on large batches: the RIFF lengths of all items in one numpy gather instead of a Python loop."""
        b = np.frombuffer(blob, dtype=np.uint8)
        o = np.asarray(self.output_offsets[:self.n], dtype=np.int64)
        ok = np.asarray(self.host_status[:self.n]) == 0
        safe = np.where(ok & (o + 8 <= b.size), o, 0)
        if b.size < 8:
            return [b""] * self.n
        head = b[safe[:, None] + np.arange(8)]
        riff = (head[:, 0] == 0x52) & (head[:, 1] == 0x49) & (head[:, 2] == 0x46) & (head[:, 3] == 0x46)
        size = head[:, 4].astype(np.int64) | head[:, 5].astype(np.int64) << 8 | head[:, 6].astype(np.int64) << 16 | head[:, 7].astype(np.int64) << 24
        n = np.where(ok & riff & (o + 8 <= b.size), size + 8, 0)
        ol, nl, okl = o.tolist(), n.tolist(), ok.tolist()
        return [blob[ol[i]:ol[i] + nl[i]] if okl[i] else b"" for i in range(self.n)]

    def _encoded_length(self, blob, i):
        o = int(self.output_offsets[i])
        if self.kind == "adx_encode":
            hs = int.from_bytes(blob[o + 2:o + 4], "big") + 4
            bs, bd, ch = blob[o + 5], blob[o + 6], blob[o + 7]
            e = int(self.output_offsets[i + 1])
            # the file ends with the 80 01 trailer block; find it from the aligned end
            k = e
            while k - bs >= o + hs and not (blob[k - bs] == 0x80 and blob[k - bs + 1] == 0x01 and ((k - bs - o - hs) % bs) == 0):
                k -= 1
            return k - o
        if self.kind == "hca_encode":
            hs = int.from_bytes(blob[o + 6:o + 8], "big")
            fc = int.from_bytes(blob[o + 16:o + 20], "big")
            fs = int.from_bytes(blob[o + 28:o + 30], "big")
            return hs + fc * fs
        raise AssertionError
