"""CPU emulation of c3c1_tile.inc's index arithmetic (LDS swizzles, fragment layouts, wave/lane ownership) for one tile.

A restatement of the ROUTINE'S address formulas in Python, lane by lane, against a plain matrix product: what was run before the
first GPU contact of the routine (the kernel itself is held by tests/test_gpu_seq.py).  The K-loop stagger (a rotation of the
k-step order) and the order of the loads (residual before the team wait) are not modelled: neither changes an address."""
import numpy as np, sys

def run(K3, N3, N1, relu1=True, valid_rows=32, seed=0):
    rng = np.random.default_rng(seed)
    NW = 8; N3F = N3 // (32 * NW); KS3 = K3 // 16; NB1 = N1 // 32; KSPL = NW // NB1; KS1 = N3 // 16 // KSPL
    YP, AP, TP = N3 * 2, K3 * 2, N1 * 2 + 16
    NA = 32 * AP // 1024 // NW; NY = 32 * YP // 1024 // NW
    Y_OFF = 0; A_OFF = 32 * N3 * 2; T_OFF = A_OFF + 32 * K3 * 2; P_OFF = T_OFF + 32 * TP
    Cs = K3 + 16; cin_off = 8; res_Cs = N3 + 8; res_coff = 8; Cos3 = N3 + 24; co3 = 16; Cos1 = N1 + 8; co1 = 8
    m0 = 64; m_end = m0 + valid_rows; m_all = m0 + 64
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
        return f.reshape(-1)          # flat halves; byte address = idx * 2
    F3 = frag(W3, K3); F1 = frag(W1, N3)
    lds = np.full((P_OFF + 4 * 16 * 64 * 4) // 2, np.nan, np.float32)     # half-granular (floats of the P region modelled separately)
    P = np.full((4 * 16 * 64,), np.nan, np.float32)
    OOB = 0x7ffff000
    def gload(buf_flat, byte_off):     # 16-byte load of halves
        if byte_off >= buf_flat.size * 2: return np.zeros(8, np.float32)
        assert byte_off % 16 == 0
        return buf_flat[byte_off // 2: byte_off // 2 + 8].copy()
    Xf, RESf = X.reshape(-1), RES.reshape(-1)
    lanes = np.arange(64)
    # ---- activation rows + residual loads
    LPR = AP // 16; IPR = YP // 1024
    rres = {}
    for w in range(NW):
        for lane in range(64):
            for j in range(NA):
                i = w * NA + j; row = i * (64 // LPR) + lane // LPR; m = m0 + row
                off = (((m * Cs + cin_off) << 1) + (lane % LPR) * 16) if m < m_end else OOB
                v = gload(Xf, off)
                a = A_OFF + row * AP + (((lane % LPR) ^ (row & 15)) << 4)
                lds[a // 2: a // 2 + 8] = v
            for j in range(NY):
                i = w * NY + j; m = m0 + i // IPR
                off = (((m * res_Cs + res_coff) << 1) + ((i % IPR) * 64 + lane) * 16) if m < m_end else OOB
                rres[(w, lane, j)] = gload(RESf, off)
    # ---- conv3
    acc3 = np.zeros((NW, N3F, 64, 16), np.float32)
    for w in range(NW):
        nb3 = w * N3F
        for lane in range(64):
            fm, fh = lane & 31, lane >> 5
            for j in range(N3F):
                for r in range(16):
                    acc3[w, j, lane, r] = B3[(nb3 + j) * 32 + 8 * (r >> 2) + 4 * fh + (r & 3)]
    def mfma(Aop, Bop, C):   # Aop, Bop: [64][8]; C: [64][16]
        A = np.zeros((32, 16), np.float32); Bm = np.zeros((16, 32), np.float32)
        for l in range(64):
            A[l % 32, 8 * (l // 32): 8 * (l // 32) + 8] = Aop[l]
            Bm[8 * (l // 32): 8 * (l // 32) + 8, l % 32] = Bop[l]
        D = A @ Bm
        for l in range(64):
            for r in range(16):
                C[l, r] += D[(r & 3) + 8 * (r >> 2) + 4 * (l // 32), l % 32]
    for w in range(NW):
        nb3 = w * N3F
        for s in range(KS3):
            xa = np.zeros((64, 8), np.float32)
            for lane in range(64):
                fm, fh = lane & 31, lane >> 5; msw = fm & 15
                a = A_OFF + fm * AP + (((2 * s + fh) ^ msw) << 4)
                xa[lane] = lds[a // 2: a // 2 + 8]
            for j in range(N3F):
                wf = np.zeros((64, 8), np.float32)
                for lane in range(64):
                    wv3 = (nb3 * KS3) * 1024 + lane * 16
                    off = wv3 + j * (KS3 * 1024) + s * 1024
                    wf[lane] = F3[off // 2: off // 2 + 8]
                mfma(wf, xa, acc3[w, j])
    # ---- residual -> Y
    for w in range(NW):
        for lane in range(64):
            for j in range(NY):
                i = w * NY + j; row = i // IPR; c = (i % IPR) * 64 + lane
                a = Y_OFF + row * YP + ((c ^ (row & 15)) << 4)
                lds[a // 2: a // 2 + 8] = rres[(w, lane, j)]
    # ---- epilogue in place
    for w in range(NW):
        nb3 = w * N3F
        for lane in range(64):
            fm, fh = lane & 31, lane >> 5; msw = fm & 15
            for j in range(N3F):
                for q in range(4):
                    a = Y_OFF + fm * YP + ((((nb3 + j) * 4 + q) ^ msw) << 4) + fh * 8
                    r = lds[a // 2: a // 2 + 4].copy()
                    o = np.maximum(acc3[w, j, lane, 4 * q: 4 * q + 4] + r, 0).astype(np.float16).astype(np.float32)
                    lds[a // 2: a // 2 + 4] = o
    # ---- Y -> memory
    CPR = YP // 16; NST = 32 * CPR // 512
    OUT3f = OUT3.reshape(-1)
    for tid in range(512):
        for it in range(NST):
            g = it * 512 + tid; row = g // CPR; c = g % CPR; m = m0 + row
            a = Y_OFF + row * YP + ((c ^ (row & 15)) << 4)
            v = lds[a // 2: a // 2 + 8]
            if m < m_end:
                so = ((m * Cos3 + co3) << 1) + c * 16
                assert so < m_all * Cos3 * 2
                OUT3f[so // 2: so // 2 + 8] = v
    # ---- second conv
    acc1 = np.zeros((NW, 64, 16), np.float32)
    for w in range(NW):
        nb1, kq = w % NB1, w // NB1
        for lane in range(64):
            fh = lane >> 5
            for r in range(16):
                acc1[w, lane, r] = B1[nb1 * 32 + 8 * (r >> 2) + 4 * fh + (r & 3)] if kq == 0 else 0.0
        for s in range(KS1):
            ya = np.zeros((64, 8), np.float32); wf = np.zeros((64, 8), np.float32)
            for lane in range(64):
                fm, fh = lane & 31, lane >> 5; msw = fm & 15
                a = Y_OFF + fm * YP + (((2 * (kq * KS1 + s) + fh) ^ msw) << 4)
                ya[lane] = lds[a // 2: a // 2 + 8]
                wv1 = (nb1 * (N3 // 16) + kq * KS1) * 1024 + lane * 16
                off = wv1 + s * 1024
                wf[lane] = F1[off // 2: off // 2 + 8]
            mfma(wf, ya, acc1[w])
    if KSPL == 2:
        for w in range(NW):
            nb1, kq = w % NB1, w // NB1
            if kq == 1:
                for lane in range(64):
                    for r in range(16): P[nb1 * 1024 + r * 64 + lane] = acc1[w, lane, r]
        for w in range(NW):
            nb1, kq = w % NB1, w // NB1
            if kq == 0:
                for lane in range(64):
                    for r in range(16): acc1[w, lane, r] += P[nb1 * 1024 + r * 64 + lane]
    for w in range(NW):
        nb1, kq = w % NB1, w // NB1
        if kq == 0:
            for lane in range(64):
                fm, fh = lane & 31, lane >> 5
                for q in range(4):
                    v = acc1[w, lane, 4 * q: 4 * q + 4].copy()
                    if relu1: v = np.maximum(v, 0)
                    a = T_OFF + fm * TP + (nb1 * 32 + 8 * q + 4 * fh) * 2
                    lds[a // 2: a // 2 + 4] = v.astype(np.float16).astype(np.float32)
    CPR = N1 * 2 // 16; NST = (32 * CPR + 511) // 512
    OUT1f = OUT1.reshape(-1)
    for tid in range(512):
        for it in range(NST):
            g = it * 512 + tid; row = g // CPR; c = g % CPR; m = m0 + row
            if g < 32 * CPR and m < m_end:
                a = T_OFF + row * TP + c * 16
                so = ((m * Cos1 + co1) << 1) + c * 16
                OUT1f[so // 2: so // 2 + 8] = lds[a // 2: a // 2 + 8]
    # ---- reference
    xs = X[m0:m_end, cin_off:cin_off + K3]; rs = RES[m0:m_end, res_coff:res_coff + N3]
    Yref = np.maximum(xs @ W3.T + B3 + rs, 0).astype(np.float16).astype(np.float32)
    O1 = Yref @ W1.T + B1
    if relu1: O1 = np.maximum(O1, 0)
    e3 = np.abs(OUT3[m0:m_end, co3:co3 + N3] - Yref).max()
    e1 = np.abs(OUT1[m0:m_end, co1:co1 + N1] - O1).max()
    untouched = np.isnan(OUT3[m_end:]).all() and np.isnan(OUT1[m_end:]).all() and np.isnan(OUT3[:m0]).all() and np.isnan(OUT3[m0:m_end, :co3]).all() and np.isnan(OUT3[m0:m_end, co3 + N3:]).all() and np.isnan(OUT1[m0:m_end, :co1]).all()
    print("K3 %d N3 %d N1 %d rows %d: conv3 err %.2e  second conv err %.2e  untouched elsewhere %s" % (K3, N3, N1, valid_rows, e3, e1, untouched))
    assert e3 < 5e-3 and e1 < 5e-3 and untouched

run(128, 512, 128, True, 32)
run(128, 512, 128, False, 17)
run(256, 1024, 256, True, 1)
run(256, 1024, 256, False, 32)
