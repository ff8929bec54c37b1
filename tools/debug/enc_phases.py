"""Developer tool (GPU box): per-phase wave-cycle shares of k_hca_encode.  Needs a library built with
CRI_HIPCC_EXTRA=-DCRI_ENC_PROFILE (python -m pycricodecs_amd.build --force).  The shares are of ALL the cycles the waves
spent between their first and last instruction (they sum to 100 %)."""
import ctypes as C
import sys
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import torch
from pycricodecs_amd import synth, _capi
from pycricodecs_amd.batch import Job
lib = _capi.lib()
ch = int(sys.argv[1]) if len(sys.argv) > 1 else 2
quality = int(sys.argv[2]) if len(sys.argv) > 2 else 1
ws = [synth.wav(i, 480000, ch, 48000) for i in range(8)] * (25 if ch <= 2 else 6)
job = Job.hca_encode(ws, quality=quality)
bufs = job.alloc("cuda:0")
job.run(*bufs); torch.cuda.synchronize()
out = (C.c_ulonglong * 16)()
lib.cri_debug_enc_profile(out, 1)
for _ in range(3):
    job.run(*bufs)
torch.cuda.synchronize()
lib.cri_debug_enc_profile(out, 0)
names = ["tables + sample staging", "  barrier behind it", "mdct", "intensity stereo (+ its barriers)", "hfr + scalefactors + scale",
         "header length + noise search (8 steps, a barrier each)", "boundary search", "resolutions + header pack", "spectra: quantise + codes + scans",
         "spectra: row exchange + bit writes", "  barrier before the checksum", "crc + store (channel 0's wave)"]
tot = sum(out[:12])
for n, v in zip(names, list(out[:12])):
    print("%-56s %5.1f %%  %8.0f wave-cycles/frame" % (n, 100.0 * v / tot, v / (3.0 * job.units)))
print("total %.0f wave-cycles per frame (%d channels, quality %d)" % (tot / (3.0 * job.units), ch, quality))
