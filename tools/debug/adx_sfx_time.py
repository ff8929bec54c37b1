"""Developer tool (GPU box): ADX round trip (configs[1] shape) per data family, kernel times by class."""
import sys
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import torch
import bench as B
from pycricodecs_amd.batch import Job
import oracle_lib as O
fams = sys.argv[1:] or ["tonal", "sfx", "sparse"]
for fam in fams:
    for seconds, n in ((10.0, 1000), (1.0, 12500)):
        uniq = [B.family_wav(3000 + u, seconds, fam) for u in range(16)]
        adx_u = [O.adx_encode(w) for w in uniq]
        enc = Job.adx_encode(B.tile(uniq, n)); dec = Job.adx_decode(B.tile(adx_u, n))
        eb = enc.alloc("cuda:0"); db = dec.alloc("cuda:0")
        enc.enable_events(True); dec.enable_events(True)
        for _ in range(2):
            enc.run(*eb); dec.run(*db)
        torch.cuda.synchronize()
        te = td = 0.0
        for _ in range(3):
            enc.run(*eb); dec.run(*db); torch.cuda.synchronize()
            te += sum(enc.event_ms().values()) / 3; td += sum(dec.event_ms().values()) / 3
        outs = dec.split(memoryview(db[1].cpu().numpy()))
        for i in (0, 5, n - 1):
            assert bytes(outs[i]) == O.adx_decode(adx_u[i % 16]), (fam, i)
        print("%-7s %5d x %4.0f s: encode %7.3f ms (%s)  decode %7.3f ms (%s)  round trip %.2f G frames/s" % (fam, n, seconds, te, enc.dominant_kernel, td, dec.dominant_kernel, (enc.units + dec.units) / (te + td) / 1e6), flush=True)
        del eb, db
        torch.cuda.empty_cache()
