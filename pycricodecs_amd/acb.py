"""The audio side of an ACB: what the reference's `ACB.extract` does with the bank an ACB carries or points to
(/root/reference/PyCriCodecs/acb.py:141-154) -- every waveform of the bank written out, HCA ones optionally decoded -- with the
decode as ONE batch job over all HCA waveforms instead of one `HCA(...).decode()` per file (binding A).

The `@UTF` tables of the ACB are NOT parsed here (containers are out of scope): the caller -- the reference's own `ACB` class -- hands
over the bank (`self.awb` or its bytes) and the `EncodeType` column of its `WaveformTable`:

    acb = PyCriCodecs.ACB("bgm.acb")                                       # the reference parses the tables
    types = [w["EncodeType"][1] for w in acb.payload[0]["WaveformTable"]]
    bank = acb.payload[0]["AwbFile"][1] or "bgm.awb"                        # the bank inside the ACB, or the file beside it (acb.py:33-43)
    AcbAudio(bank, types).extract(decode=True, key=key, dirname="out")
"""
import os

from .awb import AWB

# acb.py:156-180: the file extension of a waveform by its EncodeType
_EXTENSIONS = {0: ".adx", 3: ".adx", 2: ".hca", 6: ".hca", 7: ".vag", 10: ".vag", 8: ".at3", 9: ".bcwav", 11: ".at9", 18: ".at9", 12: ".xma",
               13: ".dsp", 4: ".dsp", 5: ".dsp", 19: ".m4a"}


def get_extension(encode_type):
    """acb.py:156-180 (`ACB.get_extension`); anything it does not know is "" there too."""
    return _EXTENSIONS.get(int(encode_type), "")


class AcbAudio:
    """The bank of an ACB + the EncodeType of each of its waveforms (bank order, as `ACB.extract` indexes them)."""

    def __init__(self, awb, encode_types):
        self.awb = awb if isinstance(awb, AWB) else AWB(awb)
        self.encode_types = [int(t) for t in encode_types]
        if len(self.encode_types) < self.awb.numfiles:
            raise IndexError("WaveformTable has %d rows for a bank of %d files" % (len(self.encode_types), self.awb.numfiles))   # (the reference's list index fails the same way)

    def names(self, decode=False):
        """File names `ACB.extract` writes, in bank order: "<n>.wav" for a decoded HCA waveform, "<n><extension>" otherwise."""
        out = []
        for n in range(self.awb.numfiles):
            ext = get_extension(self.encode_types[n])
            out.append("%d.wav" % n if decode and ext == ".hca" else "%d%s" % (n, ext))
        return out

    def files(self, decode=False, key=0, device="cuda:0"):
        """[(name, bytes)] in bank order.  decode=True: every waveform whose EncodeType says HCA is decoded (key mixed with the bank's
        subkey, acb.py:149) -- all of them by one batch job on the device; a waveform that fails raises what `HCA.decode` raises."""
        items = list(self.awb.getfiles())
        names = self.names(decode)
        if not decode:
            return list(zip(names, items))
        which = [n for n in range(len(items)) if get_extension(self.encode_types[n]) == ".hca"]
        outs = list(items)
        if which:
            import torch
            from . import _capi
            from .batch import Job
            job = Job.hca_decode([items[n] for n in which], keys=[key] * len(which), subkeys=[self.awb.subkey] * len(which))
            bufs = job.alloc(device)
            job.run(*bufs)
            torch.cuda.synchronize(device)
            status = bufs[3].cpu().numpy()
            wavs = job.split(bytes(bufs[1][:max(job.output_bytes, 1)].cpu().numpy()))
            for j, n in enumerate(which):
                rc = int(job.host_status[j] or status[j])
                if rc:
                    _capi.raise_for(rc)
                outs[n] = bytes(wavs[j])
        return list(zip(names, outs))

    def extract(self, decode=False, key=0, dirname=""):
        """acb.py:141-154: writes the files into `dirname` (created if needed), without preserving cue names."""
        if dirname:
            os.makedirs(dirname, exist_ok=True)
        for name, payload in self.files(decode, key):
            with open(os.path.join(dirname, name), "wb") as f:
                f.write(payload)
