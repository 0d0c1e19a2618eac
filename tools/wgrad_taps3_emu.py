"""CPU emulation of the INDEX LOGIC of wgrad_taps3_kernel (bayesian_torch_amd/csrc/btx_wgrad_taps.h): the staging pieces, the x
ring, the transpose-read lane addressing (ds_read_b64_tr_b16 as probed by tools/ubench/tr_probe.hip), the per-lane tap validity,
the MFMA operand / accumulator layouts and the slab indices — thread by thread, against the definition of the weight gradient.
No GPU: it exists so that a layout mistake is found here and not on a leased MI355X.

    python tools/wgrad_taps3_emu.py            # a few small geometries, prints max |diff| (exact integers: must be 0)
"""
import sys

import numpy as np

T3_R = 256
T3_XBLK = T3_R * 32 + 128
T3_YBLK = 64 * 32 + 128
T3_XRING, T3_YTILE = 4 * T3_XBLK, 4 * T3_YBLK
NK = 2


def lds_x(k):
    return k * T3_XRING


def lds_y(k, b):
    return NK * T3_XRING + (k * 2 + b) * T3_YTILE


LDS_ZERO = NK * T3_XRING + NK * 2 * T3_YTILE
LDS_TOTAL = LDS_ZERO + 64


def mix32(x):
    x &= 0xFFFFFFFF
    x ^= x >> 16
    x = (x * 0x7FEB352D) & 0xFFFFFFFF
    x ^= x >> 15
    x = (x * 0x846CA68B) & 0xFFFFFFFF
    x ^= x >> 16
    return x


def sign_word(wi, ka, kb):
    return mix32((mix32((wi ^ ka) & 0xFFFFFFFF) + kb) & 0xFFFFFFFF)


def sign_bitpos(e):
    return (31 if (e & 1) else 15) - (((e >> 3) << 2) + ((e & 7) >> 1))


def sign_of(flat, ka, kb):
    """the definition: element `flat` of a tensor is negated when its bit of the hashed word is set"""
    return -1.0 if (sign_word(flat >> 5, ka, kb) >> sign_bitpos(flat & 31)) & 1 else 1.0


def piece_geom(pc):
    l, g8 = pc & 63, (pc & 63) >> 3
    half, q = l & 1, g8 & 3
    row = 8 * (pc >> 6) + 4 * (g8 >> 2) + ((l >> 1) & 3)
    return row, q, half


def emulate(NB, H, W, C, N, chunk_px, seed=0):
    rng = np.random.default_rng(seed)
    M = NB * H * W
    x = rng.integers(-3, 4, size=(M, C)).astype(np.float32)    # small integers: every product and sum is exact
    dy = rng.integers(-3, 4, size=(M, N)).astype(np.float32)
    kin, kout = (0x12345678, 0x9ABCDEF0), (0x0F1E2D3C, 0x4B5A6978)
    chunks = (M + chunk_px - 1) // chunk_px
    ntiles, ctiles = N // 64, C // 64
    E = N * 9 * C
    slab = np.full((chunks, NK, E), np.nan, np.float32)

    for blk in range(ntiles * ctiles * chunks):
        ct = blk % ctiles
        u = blk // ctiles
        nt, chunk = u % ntiles, u // ntiles
        m_begin = chunk * chunk_px
        m_end = min(M, m_begin + chunk_px)
        lds = np.full(LDS_TOTAL // 2, np.nan, np.float32)  # one float per 2-byte slot (values are small integers)
        lds[LDS_ZERO // 2:LDS_ZERO // 2 + 32] = 0.0

        def stash_x(r0):
            for pc in range(512):
                row, q, half = piece_geom(pc)
                r = r0 + row
                ok = 0 <= r < M
                for e in range(8):
                    c = ct * 64 + 16 * q + 8 * half + e
                    v = x[r, c] if ok else 0.0
                    flat = (r if ok else 0) * C + c
                    off = q * T3_XBLK + ((r0 + row) & (T3_R - 1)) * 32 + half * 16 + 2 * e
                    lds[(lds_x(0) + off) // 2] = v
                    # the kernel: word of the piece's first element shifted by its pair base, pair d of the piece at bits 15-d / 31-d
                    f0 = flat - e
                    w = (sign_word(f0 >> 5, *kin) << ((f0 & 31) >> 1)) & 0xFFFFFFFF
                    d, hi = e >> 1, e & 1
                    neg = ((w << d) & 0xFFFFFFFF) & (0x80000000 if hi else 0x8000)
                    lds[(lds_x(1) + off) // 2] = -v if neg else v

        def stash_dy(m0, buf):
            for pc in range(512):
                row, q, half = piece_geom(pc)
                m = m0 + row
                ok = m < m_end
                for e in range(8):
                    n = nt * 64 + 16 * q + 8 * half + e
                    v = dy[m, n] if ok else 0.0
                    flat = (m if ok else 0) * N + n
                    off = q * T3_YBLK + row * 32 + half * 16 + 2 * e
                    lds[(lds_y(0, buf) + off) // 2] = v
                    f0 = flat - e
                    w = (sign_word(f0 >> 5, *kout) << ((f0 & 31) >> 1)) & 0xFFFFFFFF
                    d, hi = e >> 1, e & 1
                    neg = ((w << d) & 0xFFFFFFFF) & (0x80000000 if hi else 0x8000)
                    lds[(lds_y(1, buf) + off) // 2] = -v if neg else v

        def tr_read(addrs):
            """ds_read_b64_tr_b16 of one wave: addrs[64] byte addresses -> [64][4] values"""
            out = np.zeros((64, 4), np.float32)
            for g in range(4):
                for c in range(16):
                    for rr in range(4):
                        a = addrs[16 * g + 4 * rr + c // 4] + 2 * (c % 4)
                        out[16 * g + c, rr] = lds[a // 2]
            return out

        stash_x(m_begin - 64)
        stash_x(m_begin)
        stash_x(m_begin + 64)
        stash_dy(m_begin, 0)
        acc = np.zeros((12, 2, 3, 64, 16), np.float32)
        buf = 0
        for m0 in range(m_begin, m_end, 64):
            more = m0 + 64 < m_end
            for wave in range(12):
                kind, wr = wave & 1, wave >> 1
                j, kh = wr & 1, wr >> 1
                tap_shift = (kh - 1) * W - 1
                oh_lo, oh_hi = (1 if kh == 0 else 0), (H - 2 if kh == 2 else H - 1)
                lanes = np.arange(64)
                for ks in range(4):
                    a = []
                    for i in range(2):
                        parts = []
                        for hoff in (0, 128):
                            addrs = []
                            for lane in lanes:
                                hk, c16, g16 = lane >> 5, lane & 15, (lane >> 4) & 1
                                rsel, csel = c16 >> 2, (c16 & 3) * 8
                                ytile = lds_y(kind, buf) + g16 * T3_YBLK + (8 * hk + rsel) * 32 + csel
                                addrs.append(ytile + 2 * i * T3_YBLK + ks * 512 + hoff)
                            parts.append(tr_read(addrs))
                        a.append(np.concatenate(parts, axis=1))  # [64 lanes][8]
                    b = []
                    for kw in range(3):
                        parts = []
                        for h in range(2):
                            addrs = []
                            for lane in lanes:
                                hk, c16, g16 = lane >> 5, lane & 15, (lane >> 4) & 1
                                rsel, csel = c16 >> 2, (c16 & 3) * 8
                                xring = lds_x(kind) + (2 * j + g16) * T3_XBLK + csel
                                pp = m0 + 16 * ks + 8 * hk + 4 * h + rsel
                                t = pp // W
                                ow, oh = pp - t * W, t % H
                                rv = pp < M and oh_lo <= oh <= oh_hi
                                r5 = (pp + tap_shift) << 5
                                if kw == 0:
                                    ad = xring + (r5 & ((T3_R - 1) << 5)) if (rv and ow >= 1) else LDS_ZERO + csel
                                elif kw == 1:
                                    ad = xring + ((r5 + 32) & ((T3_R - 1) << 5)) if rv else LDS_ZERO + csel
                                else:
                                    ad = xring + ((r5 + 64) & ((T3_R - 1) << 5)) if (rv and ow <= W - 2) else LDS_ZERO + csel
                                addrs.append(ad)
                            parts.append(tr_read(addrs))
                        b.append(np.concatenate(parts, axis=1))
                    # v_mfma_f32_32x32x16_bf16: A[row = l31][k = 8 hk + e], B[k][col = l31]; D reg r of lane = [(r&3)+8(r>>2)+4hk][l31]
                    for i in range(2):
                        A = np.zeros((32, 16), np.float32)
                        for lane in lanes:
                            A[lane & 31, 8 * (lane >> 5):8 * (lane >> 5) + 8] = a[i][lane]
                        for kw in range(3):
                            B = np.zeros((16, 32), np.float32)
                            for lane in lanes:
                                B[8 * (lane >> 5):8 * (lane >> 5) + 8, lane & 31] = b[kw][lane]
                            D = A @ B
                            for lane in lanes:
                                for r in range(16):
                                    acc[wave, i, kw, lane, r] += D[(r & 3) + 8 * (r >> 2) + 4 * (lane >> 5), lane & 31]
            if more:
                stash_dy(m0 + 64, buf ^ 1)
                stash_x(m0 + 128)
            buf ^= 1
        for wave in range(12):
            kind, wr = wave & 1, wave >> 1
            j, kh = wr & 1, wr >> 1
            for i in range(2):
                for kw in range(3):
                    for lane in range(64):
                        for r in range(16):
                            n = nt * 64 + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
                            c = ct * 64 + 32 * j + (lane & 31)
                            slab[chunk, kind, (n * 9 + kh * 3 + kw) * C + c] = acc[wave, i, kw, lane, r]

    assert not np.isnan(slab).any(), "slab elements never written"
    got = slab.sum(0).reshape(NK, N, 9, C)

    # the definition (btx_wgrad.hip header): dW_mu[n][tap][c] = sum_p dy[p][n] x[p @ tap][c], dW_delta with both operands signed
    sx = np.array([[sign_of(r * C + c, *kin) for c in range(C)] for r in range(M)], np.float32)
    sy = np.array([[sign_of(m * N + n, *kout) for n in range(N)] for m in range(M)], np.float32)
    want = np.zeros((NK, N, 9, C), np.float32)
    x4, xs4 = x.reshape(NB, H, W, C), (x * sx).reshape(NB, H, W, C)
    dy4, dys4 = dy.reshape(NB, H, W, N), (dy * sy).reshape(NB, H, W, N)
    for kh in range(3):
        for kw in range(3):
            for (k, xa, ya) in ((0, x4, dy4), (1, xs4, dys4)):
                xp = np.zeros_like(xa)
                h0, h1 = max(0, 1 - kh), min(H, H + 1 - kh)
                w0, w1 = max(0, 1 - kw), min(W, W + 1 - kw)
                xp[:, h0:h1, w0:w1] = xa[:, h0 + kh - 1:h1 + kh - 1, w0 + kw - 1:w1 + kw - 1]
                want[k, :, kh * 3 + kw, :] = np.einsum("bhwn,bhwc->nc", ya, xp)
    return float(np.abs(got - want).max())


if __name__ == "__main__":
    cases = [(2, 5, 6, 64, 64, 64), (3, 7, 9, 64, 64, 128), (1, 4, 63, 128, 64, 192)]
    if len(sys.argv) > 1:
        cases = cases[:int(sys.argv[1])]
    for cs in cases:
        d = emulate(*cs)
        print("NB %d H %d W %d C %d N %d chunk_px %d: max |diff| = %g" % (cs + (d,)))
        assert d == 0.0
