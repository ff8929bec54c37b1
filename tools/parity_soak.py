"""GPU box: a time-bounded randomised parity soak of the whole path against the pinned oracle (the CPU restatement of adx.cpp / hca.cpp,
oracle/), through the batch jobs of the C ABI -- the kernels the bench measures, with the mixes of formats, lengths and signals that the
fixed parity tests only sample.  Every round draws a bank of WAVs (1-8 channels, six sample rates, lengths from one sample to 8 s, loop
points, seven signal kinds incl. full-scale noise, digital silence and values on the quantisers' clamps; one in twelve as u8 / s24 / s32 /
f32 / f64 samples) and takes it through

    HCA encode (one random quality per round)      -> bytes == oracle's encoder, file by file
    HCA crypt of those files (random keys / subkeys, type 56 or 1), and back -> bytes == oracle
    HCA decode of the enciphered files (with keys)  -> bytes == oracle's decoder; a tenth of the files with frames overwritten by random
                                                      bytes (half of those with the frame checksum renewed): same bytes or both refuse
    ADX encode (random bit depth / block size / mode / high-pass / version) -> bytes == oracle
    ADX decode of those files, a third with random block bytes -> bytes == oracle

every third round the plain HCA files again, half of them re-headed as v3.0 (noise fill) or v1.x, some with random frame payloads: the
decoder's floats before the clamp bit for bit, and the PCM of the same run;
and a few of the items through the five drop-in single-file calls as well.  The oracle's side runs on the box's host cores (threads).
Prints one line per round and a summary; exit code 1 on any mismatch.
    python tools/parity_soak.py [seconds [seed [a bank of 1000-4000 items every n-th round (default 10, 0 = never)]]]"""
import sys, time
from concurrent.futures import ThreadPoolExecutor
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np
import oracle_lib as O
from pycricodecs_amd import synth, CriCodecs as cc, _capi
from pycricodecs_amd.batch import Job
sys.path.insert(0, "tests/hostwave")
import mode                                    # (CRI_TEST_HOSTWAVE=1: the same soak on the emulated build of the library, tests/test_hostwave.py)
mode.enable()

BUDGET = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
SEED = int(sys.argv[2]) if len(sys.argv) > 2 else 1
BIG_EVERY = int(sys.argv[3]) if len(sys.argv) > 3 else 10
rng = np.random.default_rng(SEED)
pool = ThreadPoolExecutor(32)
RATES = [8000, 11025, 22050, 32000, 44100, 48000]
counts = {}
bad = []


def note(kind, ok, detail, refused=False):
    c = counts.setdefault(kind, [0, 0, 0]); c[0] += 1; c[2] += 1 if refused else 0
    if not ok:
        c[1] += 1; bad.append((kind,) + tuple(detail)); print("MISMATCH", kind, *detail, flush=True)


def oracle(fn, *a):
    try:
        return fn(*a)
    except O.OracleError:
        return None


def rand_pcm(n, ch):
    kind = int(rng.integers(0, 7))
    if kind == 0:
        x = rng.integers(-32768, 32768, (n, ch))
    elif kind == 1:
        x = rng.integers(0, 2, (n, ch)) * 65535 - 32768                               # square extremes: every value on a clamp
    elif kind == 2:
        x = np.zeros((n, ch), np.int64); x[rng.integers(0, n, max(1, n // 50))] = rng.integers(-32768, 32768)
    elif kind == 3:
        x = np.zeros((n, ch), np.int64)                                               # digital silence
    elif kind == 4:
        x = np.full((n, ch), int(rng.integers(-32768, 32768)))                         # DC
    elif kind == 5:
        t = np.arange(n)[:, None]
        x = np.sin(t * rng.uniform(0.0005, 3.1, (1, ch)) + rng.uniform(0, 6.28, (1, ch))) * rng.uniform(1, 32767) + rng.normal(0, rng.uniform(0, 300), (n, ch))
    else:
        return synth.pcm16(int(rng.integers(0, 1 << 30)), n, ch, 48000), kind
    return np.clip(np.rint(x), -32768, 32767).astype("<i2"), kind


def rand_wav(max_ch, whole_blocks=0.0, big=False):
    ch = int(rng.choice([1, 2, 2, 2, 3, 4, 5, 6, 7, 8])) if max_ch > 2 else int(rng.integers(1, max_ch + 1))
    r = rng.random()
    if big:          # a bank of thousands: mostly short clips, a few long files (segmented chains, long transform runs)
        n = int(rng.integers(1, 4000)) if r < 0.7 else (int(rng.integers(4000, 48000)) if r < 0.97 else int(rng.integers(48000, 30 * 48000 // max(1, ch // 2))))
    else:
        n = int(rng.integers(1, 2100)) if r < 0.25 else (int(rng.integers(2100, 60000)) if r < 0.85 else int(rng.integers(60000, 8 * 48000 // max(1, ch // 2))))
    if rng.random() < whole_blocks:
        n = (n + 31) // 32 * 32
    sr = int(rng.choice(RATES))
    if rng.random() < 0.08:                                      # other sample formats (the converters): the bench's signal as u8 / s24 / s32 / f32 / f64
        t = str(rng.choice(["u8", "s24", "s32", "f32", "f64"]))
        return synth.wav_typed(int(rng.integers(0, 1 << 30)), n, ch, sr, t), (ch, n, sr, t, None)
    pcm, kind = rand_pcm(n, ch)
    loop = None
    if rng.random() < 0.15 and n > 64:
        a = int(rng.integers(0, n - 1)); b = int(rng.integers(a + 1, n + 1)); loop = (a, b)
    return synth.wav_bytes(pcm, sr, loop), (ch, n, sr, kind, loop)


def run_job(job):
    outs, st = job.run_host()
    st = np.array(st[:job.n]); hs = np.asarray(job.host_status[:job.n])
    return [bytes(o) for o in outs], np.where(hs != 0, hs, st)                   # (an item the planner refused never reaches a kernel)


def crc16(b):
    return O.crc16(bytes(b))


def corrupt_hca(h):
    """Frames of a plain HCA file overwritten with random bytes at a random density; half of them with the frame checksum renewed."""
    a = bytearray(h)
    hs = int.from_bytes(a[6:8], "big")
    i = a.find(b"comp", 0, hs)
    i = a.find(b"dec\0", 0, hs) if i < 0 else i
    if i < 0:
        return bytes(a)
    fs = int.from_bytes(a[i + 4:i + 6], "big")
    nf = (len(a) - hs) // fs if fs >= 8 else 0
    if nf < 1:
        return bytes(a)
    for f in rng.integers(0, nf, max(1, nf // 6)):
        p = hs + int(f) * fs
        for j in rng.integers(2, fs - 2, int(rng.integers(1, fs - 4))):
            a[p + int(j)] = int(rng.integers(0, 256))
        if rng.random() < 0.5:
            a[p + fs - 2:p + fs] = crc16(a[p:p + fs - 2]).to_bytes(2, "big")
    return bytes(a)


def same_or_both_refuse(kind, outs, st, refs, info, inputs=None):
    for i, (o, s, r) in enumerate(zip(outs, st, refs)):
        ok = (s != 0) if r is None else (s == 0 and o == r)
        note(kind, ok, (info[i], "status %d" % s, None if r is None else len(r), len(o)), r is None)
        if not ok and inputs is not None:                            # keep the case: input, what the library made, what the oracle made
            import os
            os.makedirs("gpurun_out/soak", exist_ok=True)
            stem = "gpurun_out/soak/mismatch_%d_%s" % (len(bad), kind.replace(" ", "_"))
            open(stem + ".in", "wb").write(inputs[i]); open(stem + ".got", "wb").write(o); open(stem + ".ref", "wb").write(r or b"")


t0 = time.time(); rounds = 0
assert _capi.lib().cri_device_available() == 1, "no HIP device"
print("build %s, seed %d, %.0f s" % (_capi.build_id(), SEED, BUDGET), flush=True)
while time.time() - t0 < BUDGET:
    rounds += 1
    jk = ["-", "-", "-"]
    big = BIG_EVERY > 0 and rounds % BIG_EVERY == 0             # every so often a bank of thousands (the planner's other regimes)
    n_items = int(rng.integers(1000, 4001)) if big else int(rng.integers(8, 49))
    # ---- HCA: encode -> crypt -> decode
    bank = [rand_wav(8, 0.0, big) for _ in range(n_items)]
    wavs = [w for w, _ in bank]; info = [m for _, m in bank]
    q = int(rng.integers(0, 5))
    refs = list(pool.map(lambda w: oracle(O.hca_encode, w, q), wavs))
    outs, st = run_job(Job.hca_encode(wavs, quality=q))
    same_or_both_refuse("hca_encode q%d" % q, outs, st, refs, info, wavs)
    good = [(r, m) for r, m in zip(refs, info) if r is not None]
    if good:
        files = [r for r, _ in good]; finfo = [m for _, m in good]
        ctype = int(rng.choice([56, 56, 56, 1]))
        keys = [int(rng.integers(1, 1 << 62)) for _ in files]
        subs = [int(rng.integers(0, 1 << 16)) if rng.random() < 0.3 else 0 for _ in files]
        erefs = list(pool.map(lambda t: oracle(O.hca_crypt, t[0], 1, ctype, t[1], t[2]), zip(files, keys, subs)))
        eouts, est = run_job(Job.hca_crypt(files, 1, ctype, keys=keys, subkeys=subs))
        same_or_both_refuse("hca_crypt (encipher, type %d)" % ctype, eouts, est, erefs, finfo, files)
        enc = [e if e is not None else f for e, f in zip(erefs, files)]
        dkeys = [k if e is not None else 0 for e, k in zip(erefs, keys)]
        dsubs = [k if e is not None else 0 for e, k in zip(erefs, subs)]
        # and back (HcaCrypt's other direction)
        prefs = list(pool.map(lambda t: oracle(O.hca_crypt, t[0], 0, ctype, t[1], t[2]), zip(enc, dkeys, dsubs)))
        pouts, pst = run_job(Job.hca_crypt(enc, 0, ctype, keys=dkeys, subkeys=dsubs))
        same_or_both_refuse("hca_crypt (decipher, type %d)" % ctype, pouts, pst, prefs, finfo, enc)
        for i in range(len(enc)):
            if rng.random() < 0.1:
                plain = corrupt_hca(files[i])                     # corrupt the plain file, then encipher it with the oracle (so the checksums hold or not as drawn)
                e = oracle(O.hca_crypt, plain, 1, ctype, keys[i], subs[i])
                if e is not None:
                    enc[i] = e; dkeys[i] = keys[i]; dsubs[i] = subs[i]; finfo[i] = finfo[i] + ("corrupted",)
        drefs = list(pool.map(lambda t: oracle(O.hca_decode, t[0], t[1], t[2]), zip(enc, dkeys, dsubs)))
        jd = Job.hca_decode(enc, keys=dkeys, subkeys=dsubs); jk[0] = jd.dominant_kernel
        douts, dst = run_job(jd)
        same_or_both_refuse("hca_decode", douts, dst, drefs, finfo, enc)
    # ---- every third round: the plain files again, half of them re-headed (v3.0 with min_resolution 0: noise fill; v1.x: the other ATH default), a
    # few with every frame's payload random: the decoder's floats before the clamp, bit for bit, and the PCM of the same run
    if good and rounds % 3 == 0:
        import torch, hca_forge
        fitems = []
        for i, f in enumerate(files):
            r = rng.random()
            try:
                g = hca_forge.forge_v3(f, 0) if r < 0.35 else (hca_forge.forge_v1(f) if r < 0.5 else (hca_forge.random_frames(f, int(rng.integers(0, 1 << 30)), float(rng.choice([1.0, 0.3, 0.05]))) if r < 0.6 and len(f) < 200000 else f))
            except AssertionError:
                g = f
            fitems.append(g)
        frefs = list(pool.map(lambda h: oracle(O.hca_decode_float, h), fitems))
        wrefs = list(pool.map(lambda h: oracle(O.hca_decode, h, 0), fitems))
        jf = Job.hca_decode(fitems)
        bufs = jf.alloc("cuda:0")
        d_f, offs = jf.run_floats(*bufs)
        torch.cuda.synchronize()
        fl = d_f.cpu().numpy(); fouts = jf.split(bytes(bufs[1].cpu().numpy()))
        fst = bufs[3].cpu().numpy()[:jf.n]; fhs = np.asarray(jf.host_status[:jf.n]); fst = np.where(fhs != 0, fhs, fst)
        for i in range(len(fitems)):
            if frefs[i] is None or wrefs[i] is None:
                ok = fst[i] != 0
            else:
                mine = fl[int(offs[i]):int(offs[i + 1])]
                ok = fst[i] == 0 and mine.size == frefs[i].size and np.array_equal(mine.view(np.uint32), frefs[i].view(np.uint32)) and bytes(fouts[i]) == wrefs[i]
            note("hca_decode floats + PCM (v2.0 / forged v3.0, v1.x, random frames)", ok, (finfo[i], "status %d" % fst[i]), frefs[i] is None)
            if not ok:
                import os
                os.makedirs("gpurun_out/soak", exist_ok=True)
                open("gpurun_out/soak/mismatch_%d_floats.in" % len(bad), "wb").write(fitems[i])
    # ---- ADX: encode -> decode
    abank = [rand_wav(2, 0.6, big) for _ in range(n_items)]
    # (whole blocks only where the reference's decoder is to be run: it writes past its buffer otherwise, adx.cpp:392-415)
    awavs = [w for w, _ in abank]; ainfo = [m for _, m in abank]
    bd, bs = [(4, 18), (4, 18), (4, 18), (8, 34), (2, 10), (6, 26), (12, 20), (15, 32)][int(rng.integers(0, 8))]
    mode = int(rng.choice([2, 3, 3, 4])); ver = int(rng.choice([3, 4, 4, 5])) if mode != 2 else 3
    filt = int(rng.integers(0, 4)) if mode == 2 else 0
    hp = int(rng.choice([0, 500, 2000]))
    arefs = list(pool.map(lambda w: oracle(O.adx_encode, w, bd, bs, mode, hp, filt, ver), awavs))
    try:
        ja = Job.adx_encode(awavs, bd, bs, mode, hp, filt, ver); jk[1] = ja.dominant_kernel
        aouts, ast = run_job(ja)
        same_or_both_refuse("adx_encode", aouts, ast, arefs, [m + (bd, bs, mode, ver, filt, hp) for m in ainfo], awavs)
    except (ValueError, _capi.CriCodecsError) as e:                # parameters the library refuses for the whole job: the oracle must refuse them too
        note("adx_encode params refused", all(r is None for r in arefs), ((bd, bs, mode, ver, filt, hp), str(e)))
        arefs = []
    # (the reference's decoder writes past its buffer when the sample count is no multiple of 32, adx.cpp:392-415: whole-block files only)
    keep = [(r, m) for r, m in zip(arefs, ainfo) if r is not None and m[1] % 32 == 0]
    if keep:
        afiles = []
        for r, _ in keep:
            a = bytearray(r)
            if rng.random() < 0.33:
                hs = int.from_bytes(a[2:4], "big") + 4
                if len(a) > hs:
                    for i in rng.integers(hs, len(a), max(1, (len(a) - hs) // 3) if rng.random() < 0.5 else int(rng.integers(1, 4))):      # (dense, or a byte or three)
                        a[int(i)] = int(rng.integers(0, 256))
            afiles.append(bytes(a))
        adrefs = list(pool.map(lambda f: oracle(O.adx_decode, f), afiles))
        jad = Job.adx_decode(afiles); jk[2] = jad.dominant_kernel
        adouts, adst = run_job(jad)
        same_or_both_refuse("adx_decode", adouts, adst, adrefs, [m + (bd, bs, mode, ver) for _, m in keep], afiles)
    # ---- the drop-in single-file calls on a few of the round's items
    for i in rng.integers(0, n_items, 3):
        i = int(i)
        def single(fn, *a):
            try:
                return fn(*a)
            except (ValueError, NotImplementedError):
                return None
        note("HcaEncode (single)", single(cc.HcaEncode, wavs[i], False, q) == refs[i], (info[i], q))
        if refs[i] is not None:
            note("HcaDecode (single)", single(cc.HcaDecode, refs[i], int.from_bytes(refs[i][6:8], "big"), 0, 0) == oracle(O.hca_decode, refs[i], 0), (info[i], q))
        if arefs:
            note("AdxEncode (single)", single(cc.AdxEncode, awavs[i], bd, bs, mode, hp, filt, ver, False) == arefs[i], (ainfo[i], bd, bs, mode, ver))
            if arefs[i] is not None and ainfo[i][1] % 32 == 0:
                note("AdxDecode (single)", single(cc.AdxDecode, arefs[i]) == oracle(O.adx_decode, arefs[i]), (ainfo[i], bd, bs, mode, ver))
    print("round %d (%.0f s): %d items%s, quality %d, adx %d/%d mode %d v%d; comparisons so far %d, mismatches %d" %
          (rounds, time.time() - t0, n_items, " (%s / %s / %s)" % (jk[0], jk[1], jk[2]) if big else "", q, bd, bs, mode, ver, sum(c[0] for c in counts.values()), len(bad)), flush=True)

print("---- %d rounds in %.0f s" % (rounds, time.time() - t0))
for k in sorted(counts):
    print("%-72s %7d comparisons (%d of them: both sides refuse the input), %d mismatches" % (k, counts[k][0], counts[k][2], counts[k][1]))
print("TOTAL %d comparisons, %d mismatches" % (sum(c[0] for c in counts.values()), len(bad)))
sys.exit(1 if bad else 0)
