"""CPU emulation of wreg_halo_tile.inc's index arithmetic: patch fill by the producers (144-byte pitch, piece 8 = padding, zero
fill outside the image), fragment reads by the consumers (pixel base + tap offset), chunk-major fragment-order weights, chunk
stagger, K-group sum of the epilogue -- lane by lane against a direct dilated 3x3 convolution.  LDS starts as NaN: a read of a
byte no producer wrote shows up.  A restatement of the routine's formulas (the kernel itself is held by tests/test_gpu_seq.py)."""
import numpy as np

def run(FM, H, dil, Ci, ty, img=1, B=2, slot=5, nslots=32, seed=0, kstag=1):
    rng = np.random.default_rng(seed)
    W = H; Wo = Ho = H; Hl = Wl = Hs = Ws = H
    BM = 32 * FM; PBh = 10 * 256 * 16 // 2            # halves per patch buffer
    Cs = Ci + 8; ci0 = 8
    N = 64; Kpad = 9 * Ci; KS16 = Kpad // 16
    x = rng.standard_normal((B, H, W, Cs)).astype(np.float32)
    w = (rng.standard_normal((N, Ci, 3, 3)) / np.sqrt(9 * Ci)).astype(np.float32)
    # engine packs: rows [N][Kpad] with k = tap*Ci + ci; halo: k' = ((ci/64)*9 + tap)*64 + ci%64; fragment order of that
    rows = np.zeros((N, Kpad), np.float32)
    for tap in range(9):
        rows[:, tap * Ci:(tap + 1) * Ci] = w[:, :, tap // 3, tap % 3]
    hp = np.zeros_like(rows)
    for tap in range(9):
        for ci in range(Ci):
            hp[:, ((ci // 64) * 9 + tap) * 64 + ci % 64] = rows[:, tap * Ci + ci]
    frag = np.zeros(N * Kpad, np.float32)
    for n in range(N):
        blk = (n // 32) * KS16
        for k in range(Kpad):
            lane = (n % 32) + 32 * ((k % 16) // 8)
            frag[((blk + k // 16) * 64 + lane) * 8 + k % 8] = hp[n, k]
    xf = x.reshape(-1)
    OOB = 0x7ffff000
    RPT = BM // Wo; oy0 = ty * RPT; vr = min(RPT, Ho - oy0); npx = vr * Wo
    PW = Wl + 2 * dil; PR = (RPT + 2 * dil) * PW; NR = (PR * 9 + 255) >> 8
    assert NR <= 10, NR
    nch = Ci >> 6; c0 = (slot * nch) // nslots if kstag else 0
    chunk_of = lambda qc: (qc + c0) % nch
    lds = np.full(2 * PBh, np.nan, np.float32)
    def issue_patch(chunk, buf):
        for ptid in range(256):
            pw = ptid >> 6; lane = ptid & 63
            for j in range(NR):
                g = j * 256 + ptid
                prow, piece = divmod(g, 9)
                py, px = divmod(prow, PW)
                iy, ix = oy0 - dil + py, px - dil
                ok = piece < 8 and prow < PR and 0 <= iy < Hl and 0 <= ix < Wl
                off = ((((img * Hs + iy) * Ws + ix) * Cs + ci0) << 1) + piece * 16 if ok else OOB
                off += chunk * 128
                v = xf[off // 2: off // 2 + 8] if off < xf.size * 2 else np.zeros(8, np.float32)
                dst = buf * PBh * 2 + pw * 1024 + j * 4096 + lane * 16
                lds[dst // 2: dst // 2 + 8] = v
    acc = np.zeros((4, FM, 2, 64, 16), np.float32)
    def mfma(Aop, Bop, C):
        A = np.zeros((32, 16), np.float32); Bm = np.zeros((16, 32), np.float32)
        for l in range(64):
            A[l % 32, 8 * (l // 32): 8 * (l // 32) + 8] = Aop[l]
            Bm[8 * (l // 32): 8 * (l // 32) + 8, l % 32] = Bop[l]
        D = A @ Bm
        for l in range(64):
            for r in range(16):
                C[l, r] += D[(r & 3) + 8 * (r >> 2) + 4 * (l // 32), l % 32]
    issue_patch(chunk_of(0), 0)
    for qc in range(nch):
        if qc + 1 < nch: issue_patch(chunk_of(qc + 1), (qc + 1) & 1)     # (emulation: no timing, buffers alternate)
        buf = qc & 1
        for tap in range(9):
            kh, kw = divmod(tap, 3)
            for wk in range(4):
                so = ((chunk_of(qc) * 9 + tap) * 4 + wk) * 1024
                fbv = []
                for h in range(2):
                    wf = np.zeros((64, 8), np.float32)
                    for lane in range(64):
                        wv = ((0 >> 5) + h) * KS16 * 1024 + lane * 16
                        wf[lane] = frag[(wv + so) // 2: (wv + so) // 2 + 8]
                    fbv.append(wf)
                for i in range(FM):
                    fa = np.zeros((64, 8), np.float32)
                    for lane in range(64):
                        frow, fhalf = lane & 31, lane >> 5
                        r = i * 32 + frow
                        ry, rx = (r // Wo, r % Wo) if r < npx else (0, 0)
                        arow = (ry * PW + rx) * 144 + (wk * 2 + fhalf) * 16
                        a = arow + buf * PBh * 2 + kh * PW * dil * 144 + kw * dil * 144
                        fa[lane] = lds[a // 2: a // 2 + 8]
                    for h in range(2):
                        mfma(fa, fbv[h], acc[wk, i, h])
    # epilogue mapping: e[row][j*32 + frow] per wave, summed over the 4 K-group waves
    out = np.zeros((BM, 64), np.float32)
    for wk in range(4):
        for i in range(FM):
            for j in range(2):
                for lane in range(64):
                    frow, fhalf = lane & 31, lane >> 5
                    for r in range(16):
                        row = i * 32 + (r & 3) + 8 * (r >> 2) + 4 * fhalf
                        out[row, j * 32 + frow] += acc[wk, i, j, lane, r]
    # reference: conv of image img, output rows oy0..oy0+vr
    xp = np.zeros((H + 2 * dil, W + 2 * dil, Ci), np.float32)
    xp[dil:dil + H, dil:dil + W] = x[img, :, :, ci0:ci0 + Ci]
    ref = np.zeros((npx, N), np.float32)
    for p in range(npx):
        oy, ox = oy0 + p // Wo, p % Wo
        for kh in range(3):
            for kw in range(3):
                ref[p] += xp[oy + kh * dil, ox + kw * dil] @ w[:, :, kh, kw].T
    err = np.abs(out[:npx] - ref).max()
    print("FM %d H %d dil %d Ci %d ty %d: rows %d patch rows %d (NR %d) c0 %d: max err %.2e (nan %d)" % (FM, H, dil, Ci, ty, npx, PR, NR, c0, err, int(np.isnan(out[:npx]).sum())))
    assert err < 1e-3

run(4, 31, 2, 128, 0)
run(4, 31, 2, 128, 7, slot=20)
run(2, 31, 1, 128, 15, slot=31)
run(4, 31, 1, 256, 3, slot=9)
run(2, 15, 2, 128, 3)
