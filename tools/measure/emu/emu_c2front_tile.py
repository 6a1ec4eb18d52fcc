"""CPU emulation of the index arithmetic of c3c1_tile.inc's FRONT = 1 arm (a Bottleneck's 3x3 convolution in front of the fused conv3 +
next 1x1, on image-row tiles): the patch loads (three input rows, zero padded, chunk-swizzled), the tap / channel-step walk with the
rotated start tap, the weight fragment addresses in the (kh, kw, cin)-ordered pack, the channel-half split of layer2's shape and the
hand-over into conv3's activation rows -- lane by lane against a plain 3x3 convolution.  Run before the routine's first GPU contact, kept
by tests/test_tile_index_emulation.py."""
import numpy as np


def run(K3, ws, d, ty, img=1, slot=5, nslots=32, seed=0):
    rng = np.random.default_rng(seed)
    NW = 8
    C2S = K3 // 16; NB2 = K3 // 32; N2F = NB2 // 4; KSPL2 = 2; C2W = C2S // KSPL2; KS2W = 9 * C2W; D2 = 8 if N2F == 2 else 12
    XP = K3 * 2; AP = K3 * 2
    X_PIX = 3 * 35 + 1
    X_OFF = 0; A_OFF = X_PIX * XP; P_OFF = A_OFF + 32 * AP
    assert KS2W % D2 == 0
    hs = ws; B = 2
    Cs = K3 + 16; cin_off = 8
    X = rng.standard_normal((B, hs, ws, Cs)).astype(np.float32)
    W = (rng.standard_normal((K3, 3, 3, K3)) / np.sqrt(9 * K3)).astype(np.float32)       # [n][kh][kw][c]
    Bias = rng.standard_normal(K3).astype(np.float32)
    Kp = 9 * K3
    Wm = W.reshape(K3, Kp)

    def frag(Wmat, Kp_):
        N = Wmat.shape[0]
        f = np.zeros((N // 32, Kp_ // 16, 64, 8), np.float32)
        for lane in range(64):
            f[:, :, lane, :] = Wmat.reshape(N // 32, 32, Kp_ // 16, 2, 8)[:, lane % 32, :, lane // 32, :]
        return f.reshape(-1)
    F2 = frag(Wm, Kp)
    OOB = 0x7ffff000
    Xf = X.reshape(-1)

    def gload(buf_flat, byte_off):
        if byte_off >= buf_flat.size * 2 or byte_off < 0: return np.zeros(8, np.float32)
        assert byte_off % 16 == 0
        return buf_flat[byte_off // 2: byte_off // 2 + 8].copy()

    def mfma(Aop, Bop, C):
        A = np.zeros((32, 16), np.float32); Bm = np.zeros((16, 32), np.float32)
        for l in range(64):
            A[l % 32, 8 * (l // 32): 8 * (l // 32) + 8] = Aop[l]
            Bm[8 * (l // 32): 8 * (l // 32) + 8, l % 32] = Bop[l]
        D = A @ Bm
        for l in range(64):
            for r in range(16):
                C[l, r] += D[(r & 3) + 8 * (r >> 2) + 4 * (l // 32), l % 32]

    lds = np.full((P_OFF + 4 * 16 * 64 * 4) // 2, np.nan, np.float32)
    # ---- patch loads
    wp = ws + 2 * d; npix = 3 * wp + 1
    assert wp <= 35 and ws <= 32
    CPP = XP // 16; NXL = (X_PIX * CPP + 511) // 512
    for tid in range(512):
        for it in range(NXL):
            g = it * 512 + tid; pp = g // CPP; c = g % CPP
            r = 2 if pp >= 2 * wp else (1 if pp >= wp else 0)
            iy = ty + (r - 1) * d; ix = pp - r * wp - d
            ok = pp < npix - 1 and 0 <= iy < hs and 0 <= ix < ws
            off = ((((img * hs + iy) * ws + ix) * Cs + cin_off) << 1) + c * 16
            v = gload(Xf, off if ok else OOB)
            if pp < X_PIX:
                a = X_OFF + pp * XP + ((c ^ (pp & 15)) << 4)
                lds[a // 2: a // 2 + 8] = v
    tap0 = (slot * 9) // nslots
    acc = np.zeros((NW, N2F, 64, 16), np.float32)
    for w in range(NW):
        ng2 = w & 3; kq2 = w >> 2
        for f in range(N2F):
            for l in range(64):
                fh = l // 32
                for r in range(16):
                    acc[w, f, l, r] = Bias[(ng2 * N2F + f) * 32 + 8 * (r >> 2) + 4 * fh + (r & 3)] if kq2 == 0 else 0.0
        wv2 = ((ng2 * N2F) * (9 * C2S)) * 1024
        for j in range(KS2W):
            ti, cs = j // C2W, j % C2W
            tap = ti + tap0
            tap = tap - 9 if tap >= 9 else tap
            k16 = tap * C2S + kq2 * C2W + cs
            kh = (tap * 11) >> 5; kw = tap - 3 * kh
            Bop = np.zeros((64, 8), np.float32)
            for l in range(64):
                fm, fh = l % 32, l // 32
                pp = kh * wp + fm + kw * d
                psw = (pp & 15) ^ fh
                a = X_OFF + pp * XP + ((((kq2 * C2W + cs) * 2) ^ psw) << 4)
                Bop[l] = lds[a // 2: a // 2 + 8]
            assert not np.isnan(Bop).any(), (w, j)
            for f in range(N2F):
                Aop = np.zeros((64, 8), np.float32)
                for l in range(64):
                    o = wv2 + f * (9 * C2S * 1024) + l * 16 + k16 * 1024
                    Aop[l] = F2[o // 2: o // 2 + 8]
                mfma(Aop, Bop, acc[w, f])
    for w in range(4, NW):
        acc[w & 3] += acc[w]
    out = np.full((32, K3), np.nan, np.float32)
    for w in range(4):
        ng2 = w & 3
        for f in range(N2F):
            for l in range(64):
                fm, fh = l % 32, l // 32
                msw = fm & 15
                for q in range(4):
                    a = A_OFF + fm * AP + ((((ng2 * N2F + f) * 4 + q) ^ msw) << 4) + fh * 8
                    lds[a // 2: a // 2 + 4] = np.maximum(acc[w, f, l, 4 * q: 4 * q + 4], 0.0)
    # conv3's fragment read of the A rows (read_x): row fm, chunk 2 s + fh
    for fm in range(32):
        for ch in range(K3 // 8):
            a = A_OFF + fm * AP + ((ch ^ (fm & 15)) << 4)
            out[fm, ch * 8: ch * 8 + 8] = lds[a // 2: a // 2 + 8]
    # reference
    Xp = np.zeros((hs + 2 * d, ws + 2 * d, K3), np.float32)
    Xp[d: d + hs, d: d + ws] = X[img, :, :, cin_off: cin_off + K3]
    ref = np.zeros((ws, K3), np.float32)
    for x in range(ws):
        for kh in range(3):
            for kw in range(3):
                ref[x] += W[:, kh, kw, :] @ Xp[ty + kh * d, x + kw * d]
        ref[x] = np.maximum(ref[x] + Bias, 0.0)
    err = np.abs(out[:ws] - ref).max()
    assert err < 1e-3, err
    assert np.isfinite(out).all()
    return err


if __name__ == "__main__":
    for K3, ws, d, ty in ((256, 31, 2, 0), (256, 31, 2, 17), (256, 31, 2, 30), (256, 31, 1, 1), (128, 31, 1, 30), (128, 31, 1, 0), (128, 29, 2, 5)):
        e = run(K3, ws, d, ty)
        print("K3=%d W=%d d=%d row %d: max err %.2e" % (K3, ws, d, ty, e))
    print("ok")
