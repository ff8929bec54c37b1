"""Developer tool (GPU box): latency of the drop-in single-file calls (host buffers in and out)."""
import sys, time
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import oracle_lib as O
from pycricodecs_amd import synth, CriCodecs as cc
KEY = 0xCF222F1FE0748978
w = synth.wav(1, 480000, 2, 48000)
hca = O.hca_crypt(O.hca_encode(w, 1), 1, 56, KEY)
adx = O.adx_encode(w)
def t(f, n=10):
    f(); t0 = time.perf_counter()
    for _ in range(n): f()
    return (time.perf_counter() - t0) / n * 1e3
print("HcaDecode 10 s stereo: %.2f ms" % t(lambda: cc.HcaDecode(hca, 96, KEY, 0)))
print("HcaEncode 10 s stereo: %.2f ms" % t(lambda: cc.HcaEncode(w, False, 1)))
print("AdxDecode 10 s stereo: %.2f ms" % t(lambda: cc.AdxDecode(adx)))
print("AdxEncode 10 s stereo: %.2f ms" % t(lambda: cc.AdxEncode(w, 4, 18, 3, 500, 0, 4, False)))
print("HcaCrypt  10 s stereo: %.2f ms" % t(lambda: cc.HcaCrypt(hca, 0, 96, 0, KEY, 0)))
