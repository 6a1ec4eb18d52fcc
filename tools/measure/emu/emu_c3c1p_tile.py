"""CPU emulation of c3c1p_tile.inc's index arithmetic for ONE 64-row tile and BOTH CUs of its pair: LDS swizzles, fragment
layouts, wave / lane ownership, the channel / K halves, the slab exchange (layout, which waves give and which keep, where the
bias is added) and the global addresses -- lane by lane against a plain matrix product.  Run before the routine's first GPU
contact (tests/test_tile_index_emulation.py keeps it).  Not modelled: the K-loop stagger (a rotation of the k-step order), the
pair counter (timing), load order."""
import numpy as np


def run(K3, N3, N1, relu1=True, valid_rows=61, seed=0):
    rng = np.random.default_rng(seed)
    NW, MF = 8, 2
    N3H, N1H = N3 // 2, N1 // 2
    N3F = N3H // (32 * NW); KS3 = K3 // 16; NB1 = N1 // 32; KSPL = NW // NB1; KS1 = N3H // 16 // KSPL
    YP, AP, TP = N3H * 2, K3 * 2, N1H * 2 + 16
    NA = 64 * AP // 1024 // NW; NY = 64 * YP // 1024 // NW
    Y_OFF = 0; A_OFF = 64 * YP; T_OFF = A_OFF; P_OFF = A_OFF + max(64 * AP, 64 * TP)
    SLAB = (N1H // 32) * 2 * 4 * 64 * 16
    Cs = K3 + 16; cin_off = 8; res_Cs = N3 + 8; res_coff = 8; Cos3 = N3 + 24; co3 = 16; Cos1 = N1 + 8; co1 = 8
    m0 = 64; m_end = m0 + valid_rows; m_all = m0 + 128
    X = rng.standard_normal((m_all, Cs)).astype(np.float32)
    RES = rng.standard_normal((m_all, res_Cs)).astype(np.float32)
    W3 = (rng.standard_normal((N3, K3)) / np.sqrt(K3)).astype(np.float32); B3 = rng.standard_normal(N3).astype(np.float32)
    W1 = (rng.standard_normal((N1, N3)) / np.sqrt(N3)).astype(np.float32); B1 = rng.standard_normal(N1).astype(np.float32)
    OUT3 = np.full((m_all, Cos3), np.nan, np.float32); OUT1 = np.full((m_all, Cos1), np.nan, np.float32)

    def frag(W, Kp):   # [N/32][K/16][64][8]
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

    slabs = np.full((2, 2, SLAB // 4), np.nan, np.float32)       # [set][dest][floats]
    xn = 3                                                       # some exchange number: set xn & 1
    state = {}
    for h in (0, 1):                                             # ---- phase 1 of both CUs: up to the slab store
        lds = np.full((P_OFF + (NB1 * 2 * 16 * 64 * 4 if KSPL > 1 else 0)) // 2, np.nan, np.float32)
        P = np.full((NB1 * MF * 16 * 64,), np.nan, np.float32)
        LPR = AP // 16; CPRy = YP // 16
        rres = {}
        for w in range(NW):
            for lane in range(64):
                for j in range(NA):
                    i = w * NA + j; row = i * (64 // LPR) + lane // LPR; m = m0 + row
                    off = (((m * Cs + cin_off) << 1) + (lane % LPR) * 16) if m < m_end else OOB
                    a = A_OFF + row * AP + (((lane % LPR) ^ (row & 15)) << 4)
                    lds[a // 2: a // 2 + 8] = gload(Xf, off)
                for j in range(NY):
                    g = (w * NY + j) * 64 + lane; row = g // CPRy; c = g % CPRy; m = m0 + row
                    assert row < 64
                    off = (((m * res_Cs + res_coff + h * N3H) << 1) + c * 16) if m < m_end else OOB
                    rres[(w, lane, j)] = gload(RESf, off)
        acc3 = np.zeros((NW, N3F, MF, 64, 16), np.float32)
        for w in range(NW):
            lb3 = w * N3F; nb3 = h * (N3H // 32) + lb3
            for lane in range(64):
                fh = lane >> 5
                for j in range(N3F):
                    for r in range(16):
                        acc3[w, j, :, lane, r] = B3[(nb3 + j) * 32 + 8 * (r >> 2) + 4 * fh + (r & 3)]
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
                        wv3 = (nb3 * KS3) * 1024 + lane * 16
                        off = wv3 + j * (KS3 * 1024) + s * 1024
                        wf[lane] = F3[off // 2: off // 2 + 8]
                    for f in range(MF):
                        mfma(wf, xa[f], acc3[w, j, f])
        for w in range(NW):                                       # residual -> Y
            for lane in range(64):
                for j in range(NY):
                    g = (w * NY + j) * 64 + lane; row = g // CPRy; c = g % CPRy
                    a = Y_OFF + row * YP + ((c ^ (row & 15)) << 4)
                    lds[a // 2: a // 2 + 8] = rres[(w, lane, j)]
        for w in range(NW):                                       # epilogue in place
            lb3 = w * N3F
            for lane in range(64):
                fm, fh = lane & 31, lane >> 5; msw = fm & 15
                for j in range(N3F):
                    for f in range(MF):
                        for q in range(4):
                            a = Y_OFF + (f * 32 + fm) * YP + ((((lb3 + j) * 4 + q) ^ msw) << 4) + fh * 8
                            r = lds[a // 2: a // 2 + 4].copy()
                            lds[a // 2: a // 2 + 4] = np.maximum(acc3[w, j, f, lane, 4 * q: 4 * q + 4] + r, 0).astype(np.float16).astype(np.float32)
        NST = 64 * CPRy // 512                                    # Y half -> memory
        OUT3f = OUT3.reshape(-1)
        for tid in range(512):
            for it in range(NST):
                g = it * 512 + tid; row = g // CPRy; c = g % CPRy; m = m0 + row
                a = Y_OFF + row * YP + ((c ^ (row & 15)) << 4)
                if m < m_end:
                    so = ((m * Cos3 + co3 + h * N3H) << 1) + c * 16
                    assert so < m_all * Cos3 * 2 and np.isnan(OUT3f[so // 2: so // 2 + 8]).all()
                    OUT3f[so // 2: so // 2 + 8] = lds[a // 2: a // 2 + 8]
        acc1 = np.zeros((NW, MF, 64, 16), np.float32)              # second conv over this CU's K half
        for w in range(NW):
            nb1, kq = w % NB1, w // NB1
            keep_cu = nb1 // (NB1 // 2)
            for lane in range(64):
                fh = lane >> 5
                for r in range(16):
                    acc1[w, :, lane, r] = B1[nb1 * 32 + 8 * (r >> 2) + 4 * fh + (r & 3)] if (kq == 0 and keep_cu == h) else 0.0
            for s in range(KS1):
                wf = np.zeros((64, 8), np.float32)
                for lane in range(64):
                    wv1 = (nb1 * (N3 // 16) + h * (N3H // 16) + kq * KS1) * 1024 + lane * 16
                    off = wv1 + s * 1024
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
        for w in range(NW):                                       # give: the partner's blocks go to its slab
            nb1, kq = w % NB1, w // NB1
            keep_cu = nb1 // (NB1 // 2); bl = nb1 % (NB1 // 2)
            if kq == 0 and keep_cu != h:
                for f in range(MF):
                    for q in range(4):
                        for lane in range(64):
                            i = (((bl * MF + f) * 4 + q) * 64 + lane) << 2
                            assert np.isnan(slabs[xn & 1, h ^ 1, i: i + 4]).all()
                            slabs[xn & 1, h ^ 1, i: i + 4] = acc1[w, f, lane, 4 * q: 4 * q + 4]
        state[h] = (lds, acc1)
    for h in (0, 1):                                             # ---- phase 2: after both arrived
        lds, acc1 = state[h]
        for w in range(NW):
            nb1, kq = w % NB1, w // NB1
            keep_cu = nb1 // (NB1 // 2); bl = nb1 % (NB1 // 2)
            if kq == 0 and keep_cu == h:
                for f in range(MF):
                    for q in range(4):
                        for lane in range(64):
                            fm, fh = lane & 31, lane >> 5
                            bo = ((((bl * MF + f) * 4 + q) * 64 + lane) << 4)
                            assert bo + 16 <= SLAB
                            v = acc1[w, f, lane, 4 * q: 4 * q + 4] + slabs[xn & 1, h, bo // 4: bo // 4 + 4]
                            if relu1: v = np.maximum(v, 0)
                            a = T_OFF + (f * 32 + fm) * TP + (bl * 32 + 8 * q + 4 * fh) * 2
                            lds[a // 2: a // 2 + 4] = v.astype(np.float16).astype(np.float32)
        CPR = N1H * 2 // 16; NST = (64 * CPR + 511) // 512
        OUT1f = OUT1.reshape(-1)
        for tid in range(512):
            for it in range(NST):
                g = it * 512 + tid; row = g // CPR; c = g % CPR; m = m0 + row
                if g < 64 * CPR and m < m_end:
                    a = T_OFF + row * TP + c * 16
                    so = ((m * Cos1 + co1 + h * N1H) << 1) + c * 16
                    assert np.isnan(OUT1f[so // 2: so // 2 + 8]).all()
                    OUT1f[so // 2: so // 2 + 8] = lds[a // 2: a // 2 + 8]
    assert not np.isnan(slabs[xn & 1]).any() and np.isnan(slabs[1 - (xn & 1)]).all()
    xs = X[m0:m_end, cin_off:cin_off + K3]; rs = RES[m0:m_end, res_coff:res_coff + N3]
    Yref = np.maximum(xs @ W3.T + B3 + rs, 0).astype(np.float16).astype(np.float32)
    O1 = Yref @ W1.T + B1
    if relu1: O1 = np.maximum(O1, 0)
    e3 = np.abs(OUT3[m0:m_end, co3:co3 + N3] - Yref).max()
    e1 = np.abs(OUT1[m0:m_end, co1:co1 + N1] - O1).max()
    untouched = (np.isnan(OUT3[m_end:]).all() and np.isnan(OUT1[m_end:]).all() and np.isnan(OUT3[:m0]).all() and np.isnan(OUT1[:m0]).all() and
                 np.isnan(OUT3[m0:m_end, :co3]).all() and np.isnan(OUT3[m0:m_end, co3 + N3:]).all() and np.isnan(OUT1[m0:m_end, :co1]).all() and
                 np.isnan(OUT1[m0:m_end, co1 + N1:]).all())
    print("pair split K3 %d N3 %d N1 %d rows %d: conv3 err %.2e  second conv err %.2e  untouched elsewhere %s" % (K3, N3, N1, valid_rows, e3, e1, untouched))
    assert e3 < 5e-3 and e1 < 5e-3 and untouched


run(128, 512, 128, True, 61)
run(128, 512, 128, False, 33)
run(256, 1024, 256, True, 64)
run(256, 1024, 256, False, 1)
