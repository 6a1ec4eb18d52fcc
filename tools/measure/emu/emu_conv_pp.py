"""CPU emulation of conv_pp_kernel's index arithmetic and ring schedule (conv_pp.hip: 256 x 256 tiles, eight waves in two alternating
groups, both operands through LDS by LDS-DMA) -- lane by lane against a direct convolution, plus a symbolic walk of the half-tile ring
that checks every read against the counted waits (RAW) and every re-stage against the last read of the slot (WAR).  Run before the
kernel's first GPU contact, kept by tests/test_tile_index_emulation.py."""
import numpy as np

PP_HALF = 128 * 128
PP_A, PP_B = 0, 4 * PP_HALF
PP_LDE = 68
OOB = 0x7ffff000


def mfma_wave(Aop, Bop, C):
    """v_mfma_f32_32x32x16_f16 for one wave: Aop / Bop [64 lanes][8], C [64][16] (in place)"""
    A = np.zeros((32, 16), np.float32); Bm = np.zeros((16, 32), np.float32)
    for l in range(64):
        A[l % 32, 8 * (l // 32): 8 * (l // 32) + 8] = Aop[l]
        Bm[8 * (l // 32): 8 * (l // 32) + 8, l % 32] = Bop[l]
    D = A @ Bm
    for l in range(64):
        for r in range(16):
            C[l, r] += D[(r & 3) + 8 * (r >> 2) + 4 * (l // 32), l % 32]


def run(B=2, Hs=13, Ci=64, Cs=72, cin_off=8, Cout=256, k=3, stride=1, pad=1, dil=1, relu=True, seed=0):
    rng = np.random.default_rng(seed)
    Ws = Hs
    Ho = (Hs + 2 * pad - dil * (k - 1) - 1) // stride + 1; Wo = Ho
    M = B * Ho * Wo
    K = k * k * Ci; Kpad = (K + 127) // 128 * 128
    Npad = (Cout + 127) // 128 * 128
    X = rng.standard_normal((B, Hs, Ws, Cs)).astype(np.float16)
    W = np.zeros((Npad, Kpad), np.float16)
    Wr = (rng.standard_normal((Cout, k, k, Ci)) / np.sqrt(K)).astype(np.float16)
    W[:Cout, :K] = Wr.reshape(Cout, K)
    bias = rng.standard_normal(Npad).astype(np.float32)
    Cos = Cout + 8; cout_off = 8
    OUT = np.full((M, Cos), np.nan, np.float32)
    Xf, Wf = X.reshape(-1), W.reshape(-1)
    in_bytes, w_bytes = Xf.size * 2, Wf.size * 2
    ci_shift = int(np.log2(Ci)); assert 1 << ci_shift == Ci and Ci >= 64
    kw_magic = (65536 + k - 1) // k

    def gload(flat, nbytes, off):
        if off < 0 or off + 16 > nbytes: return np.zeros(8, np.float16)
        assert off % 16 == 0
        return flat[off // 2: off // 2 + 8].copy()

    tilesN = (Cout + 255) // 256
    nk = Kpad // 64
    for tile in range(((M + 255) // 256) * tilesN):
        tm, tn = divmod(tile, tilesN)
        m0, n0 = tm * 256, tn * 256
        acc = np.zeros((8, 2, 2, 2, 64, 16), np.float32)          # [wave][jm][mb][jn][lane][r]
        for kt in range(nk):
            lds = np.full(8 * PP_HALF // 2, np.nan, np.float16)   # one ring buffer is enough for the functional walk
            # ---- staging: every thread two pieces per half tile ----
            for tid in range(512):
                wave = tid >> 6; lane = tid & 63
                srcchunk = (tid & 7) ^ ((tid >> 4) & 7)
                k0 = kt << 6
                tap = k0 >> ci_shift; c = k0 & (Ci - 1)
                kh_i = (tap * kw_magic) >> 16; kw_i = tap - kh_i * k
                dy, dx = kh_i * dil, kw_i * dil
                tapoff = ((dy * Ws + dx) * Cs + c) * 2
                for u in range(2):
                    for j in range(2):
                        m = m0 + u * 128 + j * 64 + (tid >> 3)
                        valid = m < M
                        mm = m if valid else 0
                        b, rem = divmod(mm, Ho * Wo); oy, ox = divmod(rem, Wo)
                        ly0, lx0 = oy * stride - pad, ox * stride - pad
                        a_ly0 = ly0 if valid else -0x4000
                        a_base = (((b * Hs + ly0) * Ws + lx0) * Cs + cin_off) * 2 + srcchunk * 16
                        ok = kh_i < k and 0 <= a_ly0 + dy < Hs and 0 <= lx0 + dx < Ws
                        off = a_base + tapoff if ok else OOB
                        dst = PP_A + j * PP_HALF + u * 8192 + wave * 1024 + lane * 16
                        lds[dst // 2: dst // 2 + 8] = gload(Xf, in_bytes, off)
                        b_voff = ((n0 + (tid >> 8) * 64 + ((tid >> 3) & 31)) * Kpad + srcchunk * 8) * 2
                        so = (kt << 7) + (u * 128 + j * 32) * Kpad * 2
                        dst = PP_B + j * PP_HALF + u * 8192 + wave * 1024 + lane * 16
                        lds[dst // 2: dst // 2 + 8] = gload(Wf, w_bytes, b_voff + so)
            # ---- fragments + MFMAs, phase order (m0,n0) (m0,n1) (m1,n1) (m1,n0) ----
            for wave in range(8):
                wr, wc = wave >> 2, wave & 3
                def rd(region, rowbase, extra, s):
                    out = np.zeros((64, 8), np.float16)
                    for lane in range(64):
                        frow, fhalf = lane & 31, lane >> 5; fsw = (frow >> 1) & 7
                        a = region + (rowbase + frow) * 128 + (((2 * s + fhalf) ^ fsw) << 4) + extra
                        out[lane] = lds[a // 2: a // 2 + 8]
                    return out
                fa = {}; fb = {}
                for jm in range(2):
                    for mb in range(2):
                        for s in range(4): fa[jm, mb, s] = rd(PP_A, wr * 64, jm * PP_HALF + mb * 4096, s)
                for jn in range(2):
                    for s in range(4): fb[jn, s] = rd(PP_B, wc * 32, jn * PP_HALF, s)
                for jm, jn in ((0, 0), (0, 1), (1, 1), (1, 0)):
                    for s in range(4):
                        for mb in range(2):
                            mfma_wave(fa[jm, mb, s].astype(np.float32), fb[jn, s].astype(np.float32), acc[wave, jm, mb, jn])
        # ---- epilogue: per wave, 64 x 64 at a time through LDS ----
        for wave in range(8):
            wr, wc = wave >> 2, wave & 3
            for jm in range(2):
                e = np.full(64 * PP_LDE, np.nan, np.float32)
                for lane in range(64):
                    frow, fhalf = lane & 31, lane >> 5
                    for mb in range(2):
                        for jn in range(2):
                            for r in range(16):
                                e[(mb * 32 + (r & 3) + 8 * (r >> 2) + 4 * fhalf) * PP_LDE + jn * 32 + frow] = acc[wave, jm, mb, jn, lane, r]
                for lane in range(64):
                    erow, c8 = lane >> 3, (lane & 7) * 8
                    n = n0 + wc * 64 + c8
                    if n >= Cout: continue
                    for ps in range(8):
                        row = ps * 8 + erow; m = m0 + wr * 128 + jm * 64 + row
                        if m >= M: continue
                        v = e[row * PP_LDE + c8: row * PP_LDE + c8 + 8] + bias[n: n + 8]
                        if relu: v = np.maximum(v, 0)
                        OUT[m, cout_off + n: cout_off + n + 8] = v
    # ---- reference: direct convolution on the same fp16 values ----
    Xp = np.zeros((B, Hs + 2 * pad, Ws + 2 * pad, Ci), np.float64)
    Xp[:, pad: pad + Hs, pad: pad + Ws] = X[..., cin_off: cin_off + Ci]
    ref = np.zeros((B, Ho, Wo, Cout))
    for ky in range(k):
        for kx in range(k):
            patch = Xp[:, ky * dil: ky * dil + (Ho - 1) * stride + 1: stride, kx * dil: kx * dil + (Wo - 1) * stride + 1: stride]
            ref += patch @ Wr[:, ky, kx].astype(np.float64).T
    ref = ref.reshape(M, Cout) + bias[:Cout]
    if relu: ref = np.maximum(ref, 0)
    got = OUT[:, cout_off: cout_off + Cout]
    assert not np.isnan(got).any(), "rows / channels not written"
    assert np.isnan(OUT[:, :cout_off]).all(), "store outside the channel slice"
    return float(np.abs(got - ref).max() / np.abs(ref).max())


def ring_schedule(nk, lead=6, inflight=4, read_at=(-1, 0, 1, 2)):
    """walk the half-tile ring of conv_pp_kernel: half tile h = 4 t + x (x = 0 B0, 1 A0, 2 B1, 3 A1) is staged in fetch(h - lead) (the first
    `lead` in the prologue), every fetch ends with vmcnt(2 * inflight) (vmcnt(0) in the tail), h is read in fetch(4 t + read_at[x]) -- K tile
    0's B0 in the prologue, behind its wait + barrier.  Returns the smallest RAW and WAR margins in barrier intervals (both must be >= 0)."""
    H = 4 * nk
    r = lambda h: 4 * (h // 4) + read_at[h % 4]                     # phase that reads half tile h (-1: the prologue)
    slot = lambda h: ((h // 4) & 1, h % 4)
    fetch_iv = lambda g, p: 2 * p + g                                # interval of fetch(p) for group g; the prologue ends where interval 0 starts
    visible = {}                                                     # first interval in which EVERY wave may read h
    for h in range(H):
        if h <= lead - 1 - inflight:                                 # covered by the prologue's vmcnt(2 * inflight) + barrier
            visible[h] = -1
            continue
        p = 0
        while True:                                                  # first phase whose wait covers h
            issued_last = min(p + lead, H - 1)
            landed_upto = issued_last if p + lead >= H else issued_last - inflight
            if landed_upto >= h: break
            p += 1
        visible[h] = fetch_iv(1, p) + 1                              # group 1's wait is the later one; readable behind the barrier that ends it
    raw = min((fetch_iv(0, r(h)) if r(h) >= 0 else -1) - visible[h] for h in range(H))
    war = 10 ** 9
    for h in range(8, H):
        assert slot(h) == slot(h - 8)
        retired = fetch_iv(1, r(h - 8)) + 1 if r(h - 8) >= 0 else 0  # group 1's lgkmcnt(0) opens the interval behind its fetch
        issue = fetch_iv(0, h - lead) if h >= lead else -1
        war = min(war, issue - retired)
    return raw, war


if __name__ == "__main__":
    for nk in (2, 4, 6, 36, 72):
        raw, war = ring_schedule(nk)
        print("ring nk=%d: RAW margin %d, WAR margin %d intervals" % (nk, raw, war))
        assert raw >= 0 and war >= 2, (nk, raw, war)      # (WAR 0 = staged in the very interval whose first instruction retires the reads: a race)
    # the schedule must FAIL when the lead grows beyond the ring, the wait shrinks, or B0 is not the first half tile staged (the walk is a real check)
    assert ring_schedule(36, lead=8)[1] < 2 and ring_schedule(36, lead=6, inflight=6)[0] < 0 and ring_schedule(36, read_at=(0, -1, 1, 2))[0] < 0
    cases = [dict(), dict(Hs=11, stride=2, pad=0, Ci=128, Cs=128, cin_off=0, Cout=256, relu=False),
             dict(Hs=9, dil=2, pad=2, Ci=64, Cs=64, cin_off=0, Cout=512, B=4), dict(Hs=12, k=1, pad=0, Ci=256, Cs=264, B=3, Cout=248)]
    for c in cases:
        err = run(**c)
        print("conv_pp emu %s: err %.2e" % (c, err))
        assert err < 2e-3, (c, err)
