"""Functional emulation (numpy, host) of scripts/ubench/bhq32_probe.hip's data path for small problems: the LDS images are built
exactly as the DMA roles build them (per lane: source offset incl. out-of-range -> zero fill, destination byte), the fragments are
gathered with the kernel's read addresses, products are summed over (chunk, tap, k), and the accumulators are scattered with the
epilogue's (lane, register) -> (pixel, channel) map; the result must equal a direct 3x3 SAME convolution.  Covers the geometry
(patches, halo, borders, swizzles, chunk / tap order, filter indexing); not the hardware ordering (waits, barriers).
    python scripts/ubench/emulate_bhq32.py"""
import numpy as np
RB = 64; HP = 20; NPX = 200; NPIECE = 13; WR_B = NPIECE * 16 * RB; HSLOT = 4 * WR_B; B_OFF = 2 * HSLOT; BSL = 128 * RB
PAD_OFF = B_OFF + 4 * BSL; SMEM = PAD_OFF + 1024


def run(N, H, W, C, K=128, seed=0, taps=(-1, 1, -1, 1, 0, 3, 1)):
    oy0, oys, ox0, oxs, w0, wa, wb = taps      # the library's affine tap family: forward (-1, 1, -1, 1; 0, 3, 1), stride-1 dgrad (1, -1, 1, -1; 0, 3, 1)
    rng = np.random.RandomState(seed)
    x = rng.randint(-4, 5, size=(N, H, W, C)).astype(np.float64)
    w = rng.randint(-3, 4, size=(9, K, C)).astype(np.float64)          # [tap][column][channel]: the transposed shadow
    xf, wf = x.reshape(-1), w.reshape(-1)
    y = np.full((N, H, W, K), np.nan)
    tiles_x, tiles_y = W // 16, H // 32
    ntiles = (K + 127) // 128
    nch = C // 32

    def gather(flat, elem_off):          # 8 consecutive elements at an element offset; None = out of range -> zeros
        return np.zeros(8) if elem_off is None else flat[elem_off:elem_off + 8]

    for tile in range(N * tiles_x * tiles_y * ntiles):
        mt, nt = divmod(tile, ntiles)
        n0 = nt * 128
        img, trem = divmod(mt, tiles_x * tiles_y)
        tyi, txi = divmod(trem, tiles_x)
        y0, x0 = tyi * 32, txi * 16
        lds = {}                                                         # 16-byte granule address -> 8 values
        acc = np.zeros((8, 64, 4, 2, 16))                                 # [wave][lane][mb][nb][r]

        def issueH(t, wave, chunk):
            idp = 8 * t + wave
            live = idp < 4 * NPIECE
            j = idp // NPIECE if live else 0
            q = idp - j * NPIECE if live else 0
            for lane in range(64):
                hp = 16 * q + (lane >> 2)
                hy, hx = divmod(hp, HP)
                yy, xx = y0 + 8 * j - 1 + hy, x0 - 1 + hx
                ok = live and hp < NPX and hx < 18 and 0 <= yy < H and 0 <= xx < W and chunk < nch
                g = (lane & 3) ^ ((hx >> 2) & 3)
                off = (((img * H + yy) * W + xx) * C + g * 8 + chunk * 32) if ok else None
                dst = ((chunk & 1) * HSLOT + j * WR_B + q * 1024) if live else PAD_OFF
                lds[dst + lane * 16] = gather(xf, off)

        def issueB(t, wave):
            c, tap = divmod(t, 9)
            for lane in range(64):
                n = 16 * wave + (lane >> 2)
                g = (lane & 3) ^ ((n >> 2) & 3)
                ok = n0 + n < K and t < 9 * nch
                slab = w0 + (tap // 3) * wa + (tap % 3) * wb
                off = ((slab * K + n0 + n) * C + c * 32 + g * 8) if ok else None
                lds[B_OFF + (t & 3) * BSL + wave * 1024 + lane * 16] = gather(wf, off)

        for t in range(7):
            for wave in range(8):
                issueH(t, wave, 0)
        for t in range(3):
            for wave in range(8):
                issueB(t, wave)
        t = 0
        for c in range(nch):
            for tap in range(9):
                ta, tb = divmod(tap, 3)
                frA = {}; frB = {}
                for wave in range(8):                                    # all reads of this k-tile first (both groups), then the DMA
                    wr, wc = wave >> 1, wave & 1
                    for lane in range(64):
                        l31, half = lane & 31, lane >> 5
                        tx, tyl = l31 & 15, l31 >> 4
                        dyy, dxx = 1 + oy0 + ta * oys, 1 + ox0 + tb * oxs
                        hx = tx + dxx
                        sw = (hx >> 2) & 3
                        base = (c & 1) * HSLOT + wr * WR_B + ((dyy + tyl) * HP + hx) * RB
                        for mb in range(4):
                            for ks in range(2):
                                frA[wave, lane, mb, ks] = lds[base + (2 * mb) * HP * RB + (((2 * ks + half) ^ sw) << 4)]
                        for nb in range(2):
                            n = wc * 64 + 32 * nb + l31
                            for ks in range(2):
                                frB[wave, lane, nb, ks] = lds[B_OFF + (t & 3) * BSL + n * RB + (((2 * ks + half) ^ ((n >> 2) & 3)) << 4)]
                for wave in range(8):
                    issueB(t + 3, wave)
                    if tap < 7:
                        issueH(tap, wave, c + 1)
                # v_mfma_f32_32x32x16_bf16(filter fragment, pixel fragment): D[i][j] += sum_k Bf[i][k] * Af[j][k]; lane l holds
                # column j = l % 32, rows i = 8 (r / 4) + 4 (l / 32) + r % 4; operand lane (i or j = l % 32, k = 8 (l / 32) .. + 7)
                for wave in range(8):
                    for mb in range(4):
                        for nb in range(2):
                            for ks in range(2):
                                Af = np.zeros((32, 16)); Bf = np.zeros((32, 16))
                                for lane in range(64):
                                    Af[lane & 31, 8 * (lane >> 5):8 * (lane >> 5) + 8] = frA[wave, lane, mb, ks]
                                    Bf[lane & 31, 8 * (lane >> 5):8 * (lane >> 5) + 8] = frB[wave, lane, nb, ks]
                                D = Bf @ Af.T
                                for lane in range(64):
                                    for r in range(16):
                                        acc[wave, lane, mb, nb, r] += D[8 * (r >> 2) + 4 * (lane >> 5) + (r & 3), lane & 31]
                t += 1
        for wave in range(8):
            wr, wc = wave >> 1, wave & 1
            for lane in range(64):
                l31, half = lane & 31, lane >> 5
                for mb in range(4):
                    yy, xx = y0 + 8 * wr + 2 * mb + (l31 >> 4), x0 + (l31 & 15)
                    for nb in range(2):
                        for q in range(4):
                            ch = n0 + wc * 64 + nb * 32 + 8 * q + 4 * half
                            for e in range(4):
                                if ch + e < K:
                                    assert np.isnan(y[img, yy, xx, ch + e])
                                    y[img, yy, xx, ch + e] = acc[wave, lane, mb, nb, 4 * q + e]
    ref = np.zeros((N, H, W, K))
    xp = np.pad(x, ((0, 0), (1, 1), (1, 1), (0, 0)))
    for ta in range(3):
        for tb in range(3):
            sy, sx = 1 + oy0 + ta * oys, 1 + ox0 + tb * oxs
            ref += xp[:, sy:sy + H, sx:sx + W, :] @ w[w0 + ta * wa + tb * wb].T
    assert not np.isnan(y).any(), "unwritten outputs"
    err = np.abs(y - ref).max()
    print("N %d H %d W %d C %d taps %s: max |err| %.3g %s" % (N, H, W, C, taps, err, "OK" if err == 0 else "MISMATCH"))
    return err == 0


if __name__ == "__main__":
    ok = run(1, 32, 16, 32) and run(1, 64, 32, 64, seed=1) and run(2, 32, 32, 96, seed=2) and run(1, 64, 32, 64, seed=3, taps=(1, -1, 1, -1, 0, 3, 1))
    raise SystemExit(0 if ok else 1)
