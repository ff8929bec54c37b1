"""Developer tool (GPU box): per-phase cycle shares of k_hca_encode.  Needs a library built with
CRI_HIPCC_EXTRA=-DCRI_ENC_PROFILE (python -m pycricodecs_amd.build --force)."""
import ctypes as C
import sys
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import torch
from pycricodecs_amd import synth, _capi
from pycricodecs_amd.batch import Job
lib = _capi.lib()
ws = [synth.wav(i, 480000, 2, 48000) for i in range(8)] * 25
job = Job.hca_encode(ws, quality=1)
bufs = job.alloc("cuda:0")
job.run(*bufs); torch.cuda.synchronize()
out = (C.c_ulonglong * 24)()
lib.cri_debug_enc_profile(out, 1)
for _ in range(3):
    job.run(*bufs)
torch.cuda.synchronize()
lib.cri_debug_enc_profile(out, 0)
names = ["mdct", "intensity", "hfr+scalefactors+scale", "header+noise search", "boundary search", "resolutions+header pack", "spectra pack", "crc+store"]
names += ["  mdct: lane constants", "  mdct: first fetch issue", "  mdct: window/fold (waits for samples)", "  mdct: next fetch issue", "  mdct: butterflies", "  mdct: spectrum store"]
tot = sum(out[:8]) + out[14] + out[15]
names += ["  rate: header length", "  rate: band registers"]
for n, v in zip(names, list(out[:8]) + list(out[8:16])):
    print("%-28s %5.1f %%  %8.0f cycles/frame" % (n, 100.0 * v / tot, v / (3.0 * job.units)))
print("rate loop per frame: %.2f binary-search steps, %.2f of them exact evaluations, %.0f cycles per exact evaluation" % (
    out[16] / (3.0 * job.units), out[17] / (3.0 * job.units), out[18] / max(out[17], 1)))
