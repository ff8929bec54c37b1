"""AFS2 / AWB wave banks as the batch front door of the device codecs (mirror of PyCriCodecs/awb.py `AWB`).

The header parse is the library's host-side `cri_awb_index` (awb.py:32-52 of the reference); `AWB.extract(decode=True)`
decodes every HCA item -- and, beyond the reference, every ADX item -- with ONE batch job per codec over the bank as it
sits in HBM instead of one `HCA(i).decode()` call per file (awb.py:54-81).
"""
import ctypes as C
from io import FileIO

import numpy as np

from . import _capi

KIND_OTHER, KIND_HCA, KIND_ADX = 0, 1, 2


def awb_index(data):
    """(offsets uint64[n+1], kinds uint8[n], subkey) of an AFS2 bank; raises ValueError like the reference on a bad header."""
    L = _capi.lib()
    buf = data if isinstance(data, bytes) else bytes(data)
    n, align, subkey, hs = C.c_uint32(), C.c_uint32(), C.c_uint16(), C.c_uint32()
    rc = L.cri_awb_index(buf, len(buf), C.byref(n), C.byref(align), C.byref(subkey), C.byref(hs), None, None, 0)
    if rc:
        raise ValueError(_capi.strerror(rc))
    offs = np.zeros(n.value + 1, dtype=np.uint64)
    kinds = np.zeros(max(n.value, 1), dtype=np.uint8)
    rc = L.cri_awb_index(buf, len(buf), C.byref(n), C.byref(align), C.byref(subkey), C.byref(hs),
                         offs.ctypes.data_as(C.POINTER(C.c_uint64)), kinds.ctypes.data_as(C.POINTER(C.c_uint8)), n.value)
    if rc:
        raise ValueError(_capi.strerror(rc))
    return offs, kinds[:n.value], subkey.value


class AWB:
    """Reads an AFS2 bank (path or bytes).  Attributes as in the reference: numfiles, align, subkey, version, ofs, headersize."""

    def __init__(self, stream):
        if isinstance(stream, str):
            self.filename = stream
            with FileIO(stream) as f:
                self.data = f.readall()
        else:
            self.filename = ""
            self.data = bytes(stream)
        self.readheader()

    def readheader(self):
        d = self.data
        self.offsets, self.kinds, self.subkey = awb_index(d)
        self.version = d[4]
        self.numfiles = int.from_bytes(d[8:12], "little")
        self.align = int.from_bytes(d[12:14], "little")
        self.ofs = [int(x) for x in self.offsets]
        self.headersize = self.ofs[0]

    def getfiles(self):
        """Generator over the items' bytes (awb.py:83-88)."""
        for i in range(self.numfiles):
            yield self.data[self.ofs[i]:self.ofs[i + 1]]

    def getfile_atindex(self, index):
        return self.data[self.ofs[index]:self.ofs[index + 1]]

    def decode_all(self, key=0, device="cuda:0"):
        """[WAV bytes or None per item]: HCA items (key mixed with the bank's subkey) and ADX items decoded by two batch jobs."""
        import torch
        from .batch import Job
        hca_job, adx_job = Job.awb_decode(self.data, key)
        d_in, d_out, d_scr, d_st = hca_job.alloc(device)
        outs = [None] * self.numfiles
        # the two jobs are independent: the ADX one (one wave per file, as long as its longest clip) goes to a stream of its own and
        # the HCA kernels fill the rest of the chip meanwhile
        main, side = torch.cuda.current_stream(device), torch.cuda.Stream(device)
        ran = []
        for job, kind, bufs, stream in ((adx_job, KIND_ADX, None, side), (hca_job, KIND_HCA, (d_out, d_scr, d_st), main)):
            if not (self.kinds == kind).any():
                continue
            if bufs is None:
                _, o2, s2, st2 = job.alloc(device, upload=False)
                bufs = (o2, s2, st2)
            if stream is side:
                side.wait_stream(main)                     # (the bank's upload)
            job.run(d_in, *bufs, stream=stream)
            ran.append((job, kind, bufs))
        torch.cuda.synchronize(device)
        for job, kind, bufs in ran:
            blob = bytes(bufs[0][:max(job.output_bytes, 1)].cpu().numpy())
            status = bufs[2].cpu().numpy()
            items = job.split(blob)
            for i in range(self.numfiles):
                if self.kinds[i] == kind:
                    if job.host_status[i] or status[i]:
                        _capi.raise_for(int(job.host_status[i] or status[i]))
                    outs[i] = items[i]
        return outs

    def extract(self, decode=False, key=0):
        """Writes the items next to the bank like the reference (awb.py:54-81): <name>_<n>.hca / .wav / .dat, or <n>.* for
        an in-memory bank.  With decode=True HCA items become WAVs -- one batch decode, not one call per file."""
        wavs = self.decode_all(key) if decode else None
        base = self.filename.rsplit(".", 1)[0] + "_" if self.filename else ""
        for count, item in enumerate(self.getfiles()):
            if self.kinds[count] == KIND_HCA:
                name, payload = (base + str(count) + ".wav", wavs[count]) if decode else (base + str(count) + ".hca", item)
            else:                                              # "Probably ADX." (awb.py:65): the reference leaves these as .dat
                name, payload = base + str(count) + ".dat", item
            with open(name, "wb") as f:
                f.write(payload)
