"""CPU emulation of c3c1s_tile.inc's index arithmetic (the fused pair on 64-row tiles by ONE workgroup: conv3 in two channel halves
into the full Y image, the second convolution over the whole K, the XOR-swizzled output tile over the dead activation rows) -- lane by
lane against a plain matrix product; run before the routine's first GPU contact, kept by tests/test_tile_index_emulation.py."""
import numpy as np


def run(K3, N3, N1, relu1=True, valid_rows=64, seed=0):
    rng = np.random.default_rng(seed)
    NW, MF = 8, 2
    N3H = N3 // 2
    N3F = N3H // (32 * NW); KS3 = K3 // 16; NB1 = N1 // 32; KSPL = NW // NB1; KS1 = N3 // 16 // KSPL
    YP, AP, TP, YPH = N3 * 2, K3 * 2, N1 * 2, N3H * 2
    NA = 64 * AP // 1024 // NW; NY = 64 * YPH // 1024 // NW
    Y_OFF = 0; A_OFF = 64 * YP; P_OFF = A_OFF + max(64 * AP, 64 * TP)
    Cs = K3 + 16; cin_off = 8; res_Cs = N3 + 8; res_coff = 8; Cos3 = N3 + 24; co3 = 16; Cos1 = N1 + 8; co1 = 8
    m0 = 64; m_end = m0 + valid_rows; m_all = m0 + 128
    X = rng.standard_normal((m_all, Cs)).astype(np.float32)
    RES = rng.standard_normal((m_all, res_Cs)).astype(np.float32)
    W3 = (rng.standard_normal((N3, K3)) / np.sqrt(K3)).astype(np.float32); B3 = rng.standard_normal(N3).astype(np.float32)
    W1 = (rng.standard_normal((N1, N3)) / np.sqrt(N3)).astype(np.float32); B1 = rng.standard_normal(N1).astype(np.float32)
    OUT3 = np.full((m_all, Cos3), np.nan, np.float32); OUT1 = np.full((m_all, Cos1), np.nan, np.float32)

    def frag(W, Kp):
        N = W.shape[0]
        f = np.zeros((N // 32, Kp // 16, 64, 8), np.float32)
        for lane in range(64):
            f[:, :, lane, :] = W.reshape(N // 32, 32, Kp // 16, 2, 8)[:, lane % 32, :, lane // 32, :]
        return f.reshape(-1)
    F3 = frag(W3, K3); F1 = frag(W1, N3)
    OOB = 0x7ffff000

    def gload(buf_flat, byte_off):
        if byte_off >= buf_flat.size * 2: return np.zeros(8, np.float32)
        assert byte_off % 16 == 0
        return buf_flat[byte_off // 2: byte_off // 2 + 8].copy()
    Xf, RESf = X.reshape(-1), RES.reshape(-1)

    def mfma(Aop, Bop, C):
        A = np.zeros((32, 16), np.float32); Bm = np.zeros((16, 32), np.float32)
        for l in range(64):
            A[l % 32, 8 * (l // 32): 8 * (l // 32) + 8] = Aop[l]
            Bm[8 * (l // 32): 8 * (l // 32) + 8, l % 32] = Bop[l]
        D = A @ Bm
        for l in range(64):
            for r in range(16):
                C[l, r] += D[(r & 3) + 8 * (r >> 2) + 4 * (l // 32), l % 32]

    lds = np.full((P_OFF + (NB1 * 2 * 16 * 64 * 4 if KSPL > 1 else 0)) // 2, np.nan, np.float32)
    assert lds.size * 2 <= 160 * 1024
    P = np.full((NB1 * MF * 16 * 64,), np.nan, np.float32)
    LPR = AP // 16; CPRh = YPH // 16
    for w in range(NW):
        for lane in range(64):
            for j in range(NA):
                i = w * NA + j; row = i * (64 // LPR) + lane // LPR; m = m0 + row
                off = (((m * Cs + cin_off) << 1) + (lane % LPR) * 16) if m < m_end else OOB
                a = A_OFF + row * AP + (((lane % LPR) ^ (row & 15)) << 4)
                lds[a // 2: a // 2 + 8] = gload(Xf, off)
    for hh in (0, 1):
        rres = {}
        for w in range(NW):
            for lane in range(64):
                for j in range(NY):
                    g = (w * NY + j) * 64 + lane; row = g // CPRh; c = g % CPRh; m = m0 + row
                    assert row < 64
                    off = (((m * res_Cs + res_coff + hh * N3H) << 1) + c * 16) if m < m_end else OOB
                    rres[(w, lane, j)] = gload(RESf, off)
        acc3 = np.zeros((NW, N3F, MF, 64, 16), np.float32)
        for w in range(NW):
            lb3 = w * N3F; nb = hh * (N3H // 32) + lb3
            for lane in range(64):
                fh = lane >> 5
                for j in range(N3F):
                    for r in range(16):
                        acc3[w, j, :, lane, r] = B3[(nb + j) * 32 + 8 * (r >> 2) + 4 * fh + (r & 3)]
            for s in range(KS3):
                xa = np.zeros((MF, 64, 8), np.float32)
                for f in range(MF):
                    for lane in range(64):
                        fm, fh = lane & 31, lane >> 5; msw = fm & 15
                        a = A_OFF + (f * 32 + fm) * AP + (((2 * s + fh) ^ msw) << 4)
                        xa[f, lane] = lds[a // 2: a // 2 + 8]
                for j in range(N3F):
                    wf = np.zeros((64, 8), np.float32)
                    for lane in range(64):
                        off = ((nb + j) * KS3 * 64 + lane) * 16 + s * 1024
                        wf[lane] = F3[off // 2: off // 2 + 8]
                    for f in range(MF):
                        mfma(wf, xa[f], acc3[w, j, f])
        for w in range(NW):
            for lane in range(64):
                for j in range(NY):
                    g = (w * NY + j) * 64 + lane; row = g // CPRh; c = hh * CPRh + g % CPRh
                    a = Y_OFF + row * YP + ((c ^ (row & 15)) << 4)
                    assert np.isnan(lds[a // 2: a // 2 + 8]).all()
                    lds[a // 2: a // 2 + 8] = rres[(w, lane, j)]
        for w in range(NW):
            lb3 = w * N3F
            for lane in range(64):
                fm, fh = lane & 31, lane >> 5; msw = fm & 15
                for j in range(N3F):
                    for f in range(MF):
                        for q in range(4):
                            a = Y_OFF + (f * 32 + fm) * YP + ((((hh * (N3H // 32) + lb3 + j) * 4 + q) ^ msw) << 4) + fh * 8
                            r = lds[a // 2: a // 2 + 4].copy()
                            lds[a // 2: a // 2 + 4] = np.maximum(acc3[w, j, f, lane, 4 * q: 4 * q + 4] + r, 0).astype(np.float16).astype(np.float32)
    CPR = YP // 16; NST = 64 * CPR // 512
    OUT3f = OUT3.reshape(-1)
    for tid in range(512):
        for it in range(NST):
            g = it * 512 + tid; row = g // CPR; c = g % CPR; m = m0 + row
            a = Y_OFF + row * YP + ((c ^ (row & 15)) << 4)
            if m < m_end:
                so = ((m * Cos3 + co3) << 1) + c * 16
                assert np.isnan(OUT3f[so // 2: so // 2 + 8]).all()
                OUT3f[so // 2: so // 2 + 8] = lds[a // 2: a // 2 + 8]
    acc1 = np.zeros((NW, MF, 64, 16), np.float32)
    for w in range(NW):
        nb1, kq = w % NB1, w // NB1
        for lane in range(64):
            fh = lane >> 5
            for r in range(16):
                acc1[w, :, lane, r] = B1[nb1 * 32 + 8 * (r >> 2) + 4 * fh + (r & 3)] if kq == 0 else 0.0
        for s in range(KS1):
            wf = np.zeros((64, 8), np.float32)
            for lane in range(64):
                off = ((nb1 * (N3 // 16) + kq * KS1) * 64 + lane) * 16 + s * 1024
                wf[lane] = F1[off // 2: off // 2 + 8]
            for f in range(MF):
                ya = np.zeros((64, 8), np.float32)
                for lane in range(64):
                    fm, fh = lane & 31, lane >> 5; msw = fm & 15
                    a = Y_OFF + (f * 32 + fm) * YP + (((2 * (kq * KS1 + s) + fh) ^ msw) << 4)
                    ya[lane] = lds[a // 2: a // 2 + 8]
                mfma(wf, ya, acc1[w, f])
    if KSPL == 2:
        for w in range(NW):
            nb1, kq = w % NB1, w // NB1
            if kq == 1:
                for f in range(MF):
                    for lane in range(64):
                        for r in range(16): P[nb1 * (MF * 16 * 64) + (f * 16 + r) * 64 + lane] = acc1[w, f, lane, r]
        for w in range(NW):
            nb1, kq = w % NB1, w // NB1
            if kq == 0:
                for f in range(MF):
                    for lane in range(64):
                        for r in range(16): acc1[w, f, lane, r] += P[nb1 * (MF * 16 * 64) + (f * 16 + r) * 64 + lane]
    tile = np.full((64 * TP // 2,), np.nan, np.float32)          # the output tile over the dead activation rows
    for w in range(NW):
        nb1, kq = w % NB1, w // NB1
        if kq == 0:
            for f in range(MF):
                for q in range(4):
                    for lane in range(64):
                        fm, fh = lane & 31, lane >> 5; msw = fm & 15
                        v = acc1[w, f, lane, 4 * q: 4 * q + 4].copy()
                        if relu1: v = np.maximum(v, 0)
                        a = (f * 32 + fm) * TP + (((nb1 * 4 + q) ^ msw) << 4) + fh * 8
                        assert np.isnan(tile[a // 2: a // 2 + 4]).all()
                        tile[a // 2: a // 2 + 4] = v.astype(np.float16).astype(np.float32)
    CPRo = TP // 16; NSTo = 64 * CPRo // 512
    OUT1f = OUT1.reshape(-1)
    for tid in range(512):
        for it in range(NSTo):
            g = it * 512 + tid; row = g // CPRo; c = g % CPRo; m = m0 + row
            if m < m_end:
                a = row * TP + ((c ^ (row & 15)) << 4)
                so = ((m * Cos1 + co1) << 1) + c * 16
                OUT1f[so // 2: so // 2 + 8] = tile[a // 2: a // 2 + 8]
    xs = X[m0:m_end, cin_off:cin_off + K3]; rs = RES[m0:m_end, res_coff:res_coff + N3]
    Yref = np.maximum(xs @ W3.T + B3 + rs, 0).astype(np.float16).astype(np.float32)
    O1 = Yref @ W1.T + B1
    if relu1: O1 = np.maximum(O1, 0)
    e3 = np.abs(OUT3[m0:m_end, co3:co3 + N3] - Yref).max()
    e1 = np.abs(OUT1[m0:m_end, co1:co1 + N1] - O1).max()
    untouched = (np.isnan(OUT3[m_end:]).all() and np.isnan(OUT1[m_end:]).all() and np.isnan(OUT3[:m0]).all() and np.isnan(OUT1[:m0]).all() and
                 np.isnan(OUT3[m0:m_end, :co3]).all() and np.isnan(OUT3[m0:m_end, co3 + N3:]).all() and np.isnan(OUT1[m0:m_end, :co1]).all() and
                 np.isnan(OUT1[m0:m_end, co1 + N1:]).all())
    print("64-row solo K3 %d N3 %d N1 %d rows %d: conv3 err %.2e  second conv err %.2e  untouched elsewhere %s" % (K3, N3, N1, valid_rows, e3, e1, untouched))
    assert e3 < 5e-3 and e1 < 5e-3 and untouched


run(128, 512, 128, True, 64)
run(128, 512, 128, False, 33)
run(256, 1024, 256, True, 61)
run(256, 1024, 256, False, 1)
