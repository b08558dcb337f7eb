"""CPU: lane-level numpy emulation of the index arithmetic of the two MFMA kernels (mnc_amd/csrc/conv.hip
conv3x3_c8_kernel, mnc_amd/csrc/gemm.hip fc_mfma_kernel) against torch.  It transliterates the kernels' LDS offsets,
fragment reads and accumulator->address mapping using the documented v_mfma_f32_32x32x2_f32 layout
(A[i=lane&31][k=lane>>5], B[k=lane>>5][j=lane&31], D col=lane&31, row=(reg&3)+8*(reg>>2)+4*(lane>>5)).
It cannot see compiler or hardware behaviour -- the -m gpu parity tests do that -- but it pins the design."""
import numpy as np
import torch

LANES = np.arange(64)
J, KK = LANES & 31, LANES >> 5


def mfma_32x32x2(a, b, acc):
    """a, b: [64] per-lane operands; acc: [64][16] per-lane accumulators."""
    A = np.zeros((32, 2), np.float64)
    B = np.zeros((2, 32), np.float64)
    A[J, KK] = a
    B[KK, J] = b
    D = A @ B
    for reg in range(16):
        rows = (reg & 3) + 8 * (reg >> 2) + 4 * KK
        acc[:, reg] += D[rows, J]


def pack_conv(w):                      # pack_conv3x3_kernel: [Cin/8][Cout][76]
    Cout, Cin = w.shape[:2]
    out = np.zeros((Cin // 8, Cout, 76), np.float32)
    for cb in range(Cin // 8):
        for tap in range(9):
            out[cb, :, tap * 8:tap * 8 + 8] = w[:, cb * 8:cb * 8 + 8, tap // 3, tap % 3]
    return out


def to_c8(x):                          # [C][H][W] -> [C/8][H][W][8]
    C, H, W = x.shape
    return np.ascontiguousarray(x.reshape(C // 8, 8, H, W).transpose(0, 2, 3, 1))


def from_c8(x):
    CB, H, W, _ = x.shape
    return x.transpose(0, 3, 1, 2).reshape(CB * 8, H, W)


def emulate_conv_block(inp, wpk, bias, out, H, W, Cin, Cout, bx, by, bz, CO_T):
    NCO = 32 * CO_T
    w0, h0, co0 = bx * 32, by * 4, bz * NCO
    acc = np.zeros((4, CO_T, 64, 16), np.float64)
    for c in range(Cin // 8):
        halo = np.zeros(6 * 34 * 12, np.float32)
        for q in range(6 * 34 * 2):
            pix, half = q >> 1, q & 1
            r, cc = divmod(pix, 34)
            gh, gw = h0 - 1 + r, w0 - 1 + cc
            if 0 <= gh < H and 0 <= gw < W:
                halo[pix * 12 + half * 4: pix * 12 + half * 4 + 4] = inp[c, gh, gw, half * 4: half * 4 + 4]
        sw = wpk[c, co0:co0 + NCO].reshape(-1).copy()            # linear copy incl. pad, row pitch 76
        for wave in range(4):
            p_base = (wave * 34 + J) * 12 + KK * 4
            w_base = J * 76 + KK * 4
            for tap in range(9):
                kh, kw = divmod(tap, 3)
                po = p_base + (kh * 34 + kw) * 12
                for t in range(CO_T):
                    wo = w_base + t * 32 * 76 + tap * 8
                    for s in range(4):
                        mfma_32x32x2(sw[wo + s], halo[po + s], acc[wave, t])
    for wave in range(4):
        oh = h0 + wave
        for lane in range(64):
            j, kk = lane & 31, lane >> 5
            ow = w0 + j
            if oh < H and ow < W:
                for t in range(CO_T):
                    for g in range(4):
                        co = co0 + t * 32 + g * 8 + kk * 4
                        v = acc[wave, t, lane, 4 * g:4 * g + 4] + bias[co:co + 4]
                        out[co >> 3, oh, ow, kk * 4: kk * 4 + 4] = np.maximum(v, 0)


def test_conv3x3_index_math():
    rng = np.random.default_rng(0)
    H, W, Cin, Cout, CO_T = 6, 37, 16, 64, 2
    x = rng.normal(size=(Cin, H, W)).astype(np.float32)
    w = rng.normal(size=(Cout, Cin, 3, 3)).astype(np.float32)
    b = rng.normal(size=Cout).astype(np.float32)
    out = np.full((Cout // 8, H, W, 8), np.nan, np.float32)
    for bz in range(Cout // (32 * CO_T)):
        for by in range((H + 3) // 4):
            for bx in range((W + 31) // 32):
                emulate_conv_block(to_c8(x), pack_conv(w), b, out, H, W, Cin, Cout, bx, by, bz, CO_T)
    ref = torch.relu(torch.nn.functional.conv2d(torch.from_numpy(x)[None], torch.from_numpy(w), torch.from_numpy(b),
                                                padding=1))[0].numpy()
    got = from_c8(out)
    assert not np.isnan(got).any()
    assert np.abs(got - ref).max() < 1e-3


def test_fc_index_math():
    rng = np.random.default_rng(1)
    M, N, K = 45, 150, 64            # 2 row tiles (one partial), 2 column blocks (one partial), 2 stages
    A = rng.normal(size=(M, K)).astype(np.float32)
    Wt = rng.normal(size=(N, K)).astype(np.float32)
    out = np.full((M, N), np.nan, np.float64)
    P = 36
    mtiles = (M + 31) // 32
    for bx in range((N + 127) // 128):
        n0 = bx * 128
        acc = np.zeros((4, 10, 64, 16), np.float64)
        for s in range(K // 32):
            sA = np.zeros(320 * P, np.float32)
            sB = np.zeros(128 * P, np.float32)
            for u in range(10):
                if u < mtiles:
                    for tid in range(256):
                        q = tid + u * 256
                        r, c4 = q >> 3, q & 7
                        gr = min(r, M - 1)
                        sA[r * P + c4 * 4: r * P + c4 * 4 + 4] = A[gr, s * 32 + c4 * 4: s * 32 + c4 * 4 + 4]
            for u in range(4):
                for tid in range(256):
                    q = tid + u * 256
                    r, c4 = q >> 3, q & 7
                    gr = min(n0 + r, N - 1)
                    sB[r * P + c4 * 4: r * P + c4 * 4 + 4] = Wt[gr, s * 32 + c4 * 4: s * 32 + c4 * 4 + 4]
            for wave in range(4):
                a_base = J * P + KK * 4
                b_base = (wave * 32 + J) * P + KK * 4
                for kc in range(4):
                    for t in range(mtiles):
                        for e in range(4):
                            mfma_32x32x2(sA[a_base + t * 32 * P + kc * 8 + e], sB[b_base + kc * 8 + e], acc[wave, t])
        for wave in range(4):
            for lane in range(64):
                j, kk = lane & 31, lane >> 5
                n = n0 + wave * 32 + j
                if n < N:
                    for t in range(mtiles):
                        for e in range(16):
                            m = t * 32 + (e & 3) + 8 * (e >> 2) + 4 * kk
                            if m < M:
                                out[m, n] = acc[wave, t, lane, e]
    assert not np.isnan(out).any()
    assert np.abs(out - A.astype(np.float64) @ Wt.astype(np.float64).T).max() < 1e-4


def test_lds_pitches_are_conflict_free():
    """ds_read_b128 is serviced in four 16-lane groups; a group is conflict-free when its 16 four-dword slots hit
    distinct banks (bank = dword address mod 64).  MI355X_MICROARCH.md, LDS table."""
    groups = [[0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27],
              [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31]]
    for pitch in (12, 76, 36):         # halo pixel pitch, weight row pitch, fc row pitch
        for g in groups:
            banks = set()
            for lane in g:
                for d in range(4):
                    banks.add((lane * pitch + d) % 64)
            assert len(banks) == 64, (pitch, len(banks))


B128_GROUPS = [[0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27],
               [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31]]       # + the same two for lanes 32-63


def _b128_conflict_free(dword_addr):
    """dword_addr[64]: start address of each lane's 16-byte read -> True if every 16-lane group hits 64 distinct banks."""
    for base in (0, 32):
        for g in B128_GROUPS:
            banks = set()
            for lane in g:
                for d in range(4):
                    banks.add(int(dword_addr[base + lane] + d) % 64)
            if len(banks) != 64:
                return False
    return True


def test_winograd_halo_half_swap():
    """conv_wino.hip, flat / rotated builds: a lane's halo read is pixel (4 rg + 2 ty + row, 2 tx + c), channel half kk, of a
    [10][34] pixel halo with a 12-dword pixel pitch.  Tile columns are 24 dwords apart, so without the swap every lane starts
    on a multiple of 8 banks and a 16-lane group covers 32 banks (2-way conflicts); with the two 16-byte halves of a pixel
    swapped in halo rows whose (row >> 1) is odd, the two tile rows of a group are 4 banks apart: conflict-free.  Writer
    (store_chunk: h_off) and reader (row_off) must agree on where a half lives."""
    pitch, cols = 12, 34
    ty, tx = (LANES & 31) >> 4, LANES & 15
    kk = LANES >> 5

    def writer(r, c, half, swap):
        return (r * cols + c) * pitch + ((half ^ ((r >> 1) & 1)) if swap else half) * 4

    def reader(rg, row, c, swap):
        base = ((4 * rg + 2 * ty + row) * cols + 2 * tx + c) * pitch
        return base + ((kk ^ ((ty + (row >> 1)) & 1)) if swap else kk) * 4

    # every (pixel, half) has its own 4 dwords inside the pixel's 8 data dwords
    for swap in (False, True):
        seen = set()
        for r in range(10):
            for c in range(cols):
                for half in (0, 1):
                    a = writer(r, c, half, swap)
                    assert a not in seen and (r * cols + c) * pitch <= a < (r * cols + c) * pitch + 8
                    seen.add(a)
    for rg in (0, 1):
        for row in range(4):
            for c in range(4):
                got = reader(rg, row, c, True)
                want = np.array([writer(4 * rg + 2 * int(ty[l]) + row, 2 * int(tx[l]) + c, int(kk[l]), True) for l in LANES])
                assert np.array_equal(got, want)
                assert _b128_conflict_free(reader(rg, row, c, True))
                assert not _b128_conflict_free(reader(rg, row, c, False))     # what SQ_LDS_BANK_CONFLICT showed (39 % of LDS cycles)


def test_fc_dma_swizzle():
    """gemm.hip fc_mfma_dma_kernel: a stage is 448 rows (320 activation + 128 weight) of 128 bytes = 8 chunks, copied by 56
    DMA instructions whose 64 lanes land contiguously; lane L of piece p fills slot 64 p + L = (row slot >> 3, chunk slot & 7)
    with k-chunk (slot & 7) ^ ((row >> 1) & 7) of that row.  A fragment read of K-group kc by lane (j, kk) goes to chunk slot
    (2 kc + kk) ^ ((j >> 1) & 7) of row (tile base + j).  Checks: the copy covers every (row, k-chunk) once; the reads find
    their chunk; every ds_read_b128 lane group is conflict-free on the unpadded rows; one XOR switches buffers."""
    rows, kBM = 448, 320
    where = {}
    for wave in range(4):
        for i in range(14):
            p = wave + 4 * i
            for lane in range(64):
                slot = p * 64 + lane
                r, c = slot >> 3, (slot & 7) ^ ((slot >> 4) & 7)          # (r >> 1) & 7 with r = slot >> 3
                assert (r, c) not in where
                where[(r, c)] = slot * 16                                  # byte offset in the buffer
    assert len(where) == rows * 8
    j, kk = LANES & 31, LANES >> 5
    for kc in range(4):
        c = (2 * kc + kk) ^ ((j >> 1) & 7)
        a_off = (j * 32 + c * 4) * 4
        for t in range(10):                                               # activation tiles: + t * 4096 bytes
            got = a_off + t * 4096
            want = np.array([where[(t * 32 + int(j[l]), 2 * kc + int(kk[l]))] for l in LANES])
            assert np.array_equal(got, want)
            assert _b128_conflict_free(got // 4)
        for wave in range(4):                                             # weight rows kBM + 32 wave + j
            got = ((kBM + wave * 32 + j) * 32 + c * 4) * 4
            want = np.array([where[(kBM + wave * 32 + int(j[l]), 2 * kc + int(kk[l]))] for l in LANES])
            assert np.array_equal(got, want)
            assert _b128_conflict_free(got // 4)
            assert got.max() + 16 <= rows * 128 <= 65536 and np.array_equal(got ^ 65536, got + 65536)
    # the plain row-major placement (chunk c in slot c) would put all 16 lanes of a group on 2 x 4 banks per parity
    assert not _b128_conflict_free((j * 32 + (2 * 0 + kk) * 4))


def test_fc_dma16_fragments():
    """gemm.hip fc_mfma_dma16_kernel<10> (the product fp32 InnerProduct): same copy and row swizzle as fc_mfma_dma_kernel, fragments
    for v_mfma_f32_16x16x4_f32.  Lane (r = l % 16, g = l / 16) of wave (wm, wn) reads chunk 4 G + g of row r of every 16-row
    sub-tile with one ds_read_b128; MFMA q of K-group G multiplies element q of those chunks, i.e. k = 16 G + 4 g + q in lane group
    g.  Checks: the reads find their chunk in the copied image, every ds_read_b128 service group is conflict-free, one XOR
    switches buffers, the 8 waves x (10 sub-tiles x 2 column halves) x D-register map cover the 320 x 128 outputs once, and a
    stage's two groups x four MFMAs x four lane groups take every k of the stage once -- the same k on both operands."""
    rows, kBM = 448, 320
    where = {}
    for wave in range(8):
        for i in range(7):
            p = wave + 8 * i
            for lane in range(64):
                slot = p * 64 + lane
                r, c = slot >> 3, (slot & 7) ^ ((slot >> 4) & 7)
                assert (r, c) not in where
                where[(r, c)] = slot * 16
    assert len(where) == rows * 8
    r16, g4 = LANES & 15, LANES >> 4
    covered = np.zeros((320, 128), np.int32)
    for wave in range(8):
        wn, wm = wave & 3, wave >> 2
        ks = []
        for G in range(2):
            sl = ((4 * G + g4) ^ ((r16 >> 1) & 7)) * 16
            a_off = (wm * 160 + r16) * 128 + sl
            b_off = (kBM + wn * 32 + r16) * 128 + sl
            for i in range(10):
                got = a_off + i * 2048
                want = np.array([where[(wm * 160 + 16 * i + int(r16[l]), 4 * G + int(g4[l]))] for l in LANES])
                assert np.array_equal(got, want) and _b128_conflict_free(got // 4)
            for c in range(2):
                got = b_off + c * 2048
                want = np.array([where[(kBM + wn * 32 + 16 * c + int(r16[l]), 4 * G + int(g4[l]))] for l in LANES])
                assert np.array_equal(got, want) and _b128_conflict_free(got // 4)
                assert got.max() + 16 <= rows * 128 <= 65536 and np.array_equal(got ^ 65536, got + 65536)
            for q in range(4):                                        # MFMA q: lane group g supplies k = 4 (4 G + g) + q
                ks.extend(sorted(set((4 * (4 * G + g4) + q).tolist())))
        assert sorted(ks) == list(range(32))                          # every k of the stage exactly once
        for i in range(10):
            for c in range(2):
                for e in range(4):                                    # D register e: row 4 g + e, column r
                    m = wm * 160 + i * 16 + 4 * g4 + e
                    n = wn * 32 + c * 16 + r16
                    np.add.at(covered, (m, n), 1)
    assert (covered == 1).all()


def _wino_plan(H, W, Cin, Cout, slots=512):
    """wino_impl's tail plan (conv_wino.hip) for the two-row-group kernel: -> (pix_a, ksplit_a, ksplit_b)."""
    pix = -(-W // 32) * -(-H // 8)
    ncot, blocks = Cout // 32, Cin // 8
    full_pix = (pix * ncot // slots) * slots // ncot
    rest = (pix - full_pix) * ncot
    if rest > 0:
        light = rest <= slots // 4
        sb = min(slots // rest, blocks // (4 if light else 8), 8)
        if sb >= 3 or (sb == 2 and light):
            return full_pix, 1, sb
    return pix, 1, 1


def test_winograd_tail_plan_covers_every_tile_once():
    """The launcher's two grid sections and the kernel's block decode: every (pixel tile, channel tile) appears with each of
    its K ranges exactly once, the ranges partition the blocks, and the XCD renumbering is a bijection inside a section."""
    def xcd(b, total):
        q, r, x, idx = total >> 3, total & 7, b & 7, b >> 3
        return (x * (q + 1) if x < r else r * (q + 1) + (x - r) * q) + idx

    for (H, W, Cin, Cout), want in (((75, 125, 512, 512), (32, 1, 4)), ((38, 63, 512, 512), (0, 1, 3)),
                                    ((150, 250, 256, 256), (152, 1, 1)), ((600, 1000, 64, 64), (2400, 1, 1)),
                                    ((37, 63, 256, 512), (0, 1, 3)), ((13, 33, 128, 256), (0, 1, 4)),
                                    ((38, 63, 128, 64), (0, 1, 4)), ((75, 125, 64, 64), (0, 1, 2)), ((5, 3, 8, 32), (1, 1, 1))):
        pix_a, sa, sb = _wino_plan(H, W, Cin, Cout)
        assert (pix_a, sa, sb) == want
        tiles_x, ncot, nb = -(-W // 32), Cout // 32, Cin // 8
        pix = tiles_x * -(-H // 8)
        n_a = pix_a * ncot * sa
        grid = n_a + (pix - pix_a) * ncot * sb
        seen = {}
        for blk in range(grid):
            b, total, pix0, s = (blk, n_a, 0, sa) if blk < n_a else (blk - n_a, grid - n_a, pix_a, sb)
            logical = xcd(b, total)
            assert 0 <= logical < total
            nz = ncot * s
            bz, p = logical % nz, pix0 + logical // nz
            split, cot = bz // ncot, bz % ncot
            c0, c1 = split * nb // s, (split + 1) * nb // s
            assert p < pix and c1 > c0
            seen.setdefault((p, cot), []).append((c0, c1))
        assert len(seen) == pix * ncot
        for ranges in seen.values():
            ranges.sort()
            assert ranges[0][0] == 0 and ranges[-1][1] == nb
            assert all(a[1] == b[0] for a, b in zip(ranges, ranges[1:]))


def test_winograd_stream_partition_and_fixup_ownership():
    """conv_wino_stream.hip (tuning builds): the layer as units u = tile * (Cin / 8) + block cut into G ranges [start(l), start(l + 1)),
    start(l) = 2 * floor(l * U / 2 / G).  The kernel walks a range segment by segment (a segment ends with its tile or with the
    range); a segment that is not a whole tile leaves a piece in slot 0 (the range's first segment) or slot 1 (a later one).
    wino_stream_fix_kernel's block l owns the tile that begins inside range l and ends outside it and adds range l's piece, then the
    slot-0 pieces of the following ranges up to the tile's end.  Checks, for the trunk's shapes and awkward G: the ranges partition the
    units, every segment has >= 2 blocks (the prefetch runs two ahead), every tile is either whole in one range or cut into pieces
    that exactly one fix block collects, in block order, with the slots the kernel wrote."""
    def start(l, u2, G):
        return 2 * (l * u2 // G)

    for tiles, nch, G in ((4800, 8, 512), (2432, 16, 512), (1216, 32, 512), (640, 64, 512), (160, 64, 512), (160, 64, 1024),
                          (7, 2, 7), (33, 6, 512), (1216, 32, 2560)):
        u2 = tiles * nch // 2
        G = min(G, u2)
        assert start(0, u2, G) == 0 and start(G, u2, G) == tiles * nch
        pieces = {}                                   # tile -> [(range, slot, first block, end block)]
        whole = set()
        for l in range(G):
            u0, u1 = start(l, u2, G), start(l + 1, u2, G)
            assert u1 - u0 >= 2 and u0 % 2 == 0
            u, t = u0, u0 // nch
            while u < u1:
                seg_end = min((t + 1) * nch, u1)
                assert seg_end - u >= 2
                if seg_end - u == nch:
                    assert t not in whole and t not in pieces
                    whole.add(t)
                else:
                    pieces.setdefault(t, []).append((l, 0 if u == u0 else 1, u - t * nch, seg_end - t * nch))
                u, t = seg_end, t + 1
        assert len(whole) + len(pieces) == tiles and not (whole & set(pieces))
        for t, ps in pieces.items():                  # the pieces of a cut tile cover its blocks once, in range order
            assert ps[0][2] == 0 and ps[-1][3] == nch and all(a[3] == b[2] and a[0] + 1 == b[0] for a, b in zip(ps, ps[1:]))
        owned = {}
        for l in range(G):                            # wino_stream_fix_kernel, block l
            s_l, e_l = start(l, u2, G), start(l + 1, u2, G)
            if e_l % nch == 0:
                continue
            t = e_l // nch
            tb, te = t * nch, t * nch + nch
            if tb < s_l:
                continue
            got = [(l, 0 if s_l // nch == t else 1)]
            l2 = l + 1
            while l2 < G and start(l2, u2, G) < te:
                got.append((l2, 0))
                l2 += 1
            assert t not in owned
            owned[t] = got
        assert set(owned) == set(pieces)
        for t, got in owned.items():
            assert got == [(l, slot) for l, slot, _, _ in pieces[t]]


def test_fc_dma_kernel_index_math():
    """fc_mfma_dma_kernel<10, 0, 2> lane by lane: the copy's slot -> (row, k-chunk) map with clamped rows, the swizzled fragment
    addresses of wave (wm, wn), the v_mfma_f32_32x32x2_f32 operand layout, the epilogue's accumulator -> (m, n) map and the
    K-split partial sums, against a float64 product.  Two row blocks (the second ragged), a ragged column tile, two K splits of
    unequal length."""
    rng = np.random.default_rng(3)
    M, N, K, kper = 333, 150, 192, 128           # splits: stages [0, 4) and [4, 6)
    A = rng.normal(size=(M, K)).astype(np.float32)
    Wt = rng.normal(size=(N, K)).astype(np.float32)
    kBM, kBN, kNW, TR = 320, 128, 8, 5
    rows = kBM + kBN
    splits = -(-K // kper)
    part = np.full((splits, M, N), np.nan, np.float64)
    for bmz in range(-(-M // kBM)):
        m0 = bmz * kBM
        mrows = min(M - m0, kBM)
        mtiles = (mrows + 31) >> 5
        for bn in range(-(-N // kBN)):
            n0 = bn * kBN
            for split in range(splits):
                kbeg, kend = split * kper, min(K, split * kper + kper)
                acc = np.zeros((kNW, TR, 64, 16), np.float64)
                for s in range((kend - kbeg) // 32):
                    buf = np.full(rows * 32, np.nan, np.float32)          # one stage buffer, in floats
                    for wave in range(kNW):
                        for i in range(rows // 8 // kNW):
                            for lane in range(64):
                                slot = (wave + kNW * i) * 64 + lane
                                r, c = slot >> 3, (slot & 7) ^ ((slot >> 4) & 7)
                                src = A[m0 + min(r, mrows - 1)] if r < kBM else Wt[min(n0 + r - kBM, N - 1)]
                                k0 = kbeg + s * 32 + c * 4
                                buf[slot * 4: slot * 4 + 4] = src[k0: k0 + 4]
                    assert not np.isnan(buf).any()
                    for wave in range(kNW):
                        wn, wm = wave & 3, wave >> 2
                        for kc in range(4):
                            c = (2 * kc + KK) ^ ((J >> 1) & 7)
                            a_off = (wm * TR * 32 + J) * 32 + c * 4
                            b_off = (kBM + wn * 32 + J) * 32 + c * 4
                            for t in range(TR):
                                for e in range(4):
                                    mfma_32x32x2(buf[a_off + t * 1024 + e], buf[b_off + e], acc[wave, t])
                for wave in range(kNW):
                    wn, wm = wave & 3, wave >> 2
                    for lane in range(64):
                        j, kk = lane & 31, lane >> 5
                        n = n0 + wn * 32 + j
                        if n >= N:
                            continue
                        for t in range(TR):
                            if wm * TR + t < mtiles:
                                for e in range(16):
                                    m = m0 + (wm * TR + t) * 32 + (e & 3) + 8 * (e >> 2) + 4 * kk
                                    if m < M:
                                        assert np.isnan(part[split, m, n])               # every element written once
                                        part[split, m, n] = acc[wave, t, lane, e]
    assert not np.isnan(part).any()
    assert np.abs(part.sum(0) - A.astype(np.float64) @ Wt.astype(np.float64).T).max() < 1e-4


def test_winograd_wave_pair_roles_and_partial_output_transform():
    """conv_wino.hip, flat / rotated builds, one tile and one channel pair in scalar form: wave hf of a pair names its three
    halo rows by role -- hf = 0: (A, B, C) = (d0, d1, d2), hf = 1: (d2, d3, d1) -- builds t0 = A - C, t1 = fma(sgn, B, C) with
    sgn = +1 / -1 (rows 2 hf, 2 hf + 1 of B^T d), runs the column pass, multiplies with rows 2 hf, 2 hf + 1 of U = G g G^T and
    applies the output transform to its own rows (hf = 0: s0 = M0 + M1, s1 = M1; hf = 1: s0 = M2, s1 = -(M2 + M3)); the pair's
    partial 2x2 outputs add up to the 3x3 correlation.  Exact in float64 up to rounding of the G transform."""
    rng = np.random.default_rng(5)
    G = np.array([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]])
    for _ in range(20):
        d = rng.normal(size=(4, 4))
        g = rng.normal(size=(3, 3))
        U = G @ g @ G.T
        y = np.zeros((2, 2))
        for hf in (0, 1):
            rowA, rowB, rowC = (2, 3, 1) if hf else (0, 1, 2)
            sgn = -1.0 if hf else 1.0
            t0 = d[rowA] - d[rowC]
            t1 = sgn * d[rowB] + d[rowC]
            want = (d[2] - d[1], d[1] - d[3]) if hf else (d[0] - d[2], d[1] + d[2])
            assert np.array_equal(t0, want[0]) and np.array_equal(t1, want[1])
            m = []
            for r, t in enumerate((t0, t1)):
                v = np.array([t[0] - t[2], t[1] + t[2], t[2] - t[1], t[1] - t[3]])
                m.append(U[2 * hf + r] * v)                       # positions 8 hf + 4 r + c
            s0 = m[0] if hf else m[0] + m[1]
            s1 = -(m[0] + m[1]) if hf else m[1]
            for dy, s in enumerate((s0, s1)):
                y[dy, 0] += s[0] + s[1] + s[2]
                y[dy, 1] += s[1] - s[2] - s[3]
        direct = np.array([[(d[i:i + 3, j:j + 3] * g).sum() for j in range(2)] for i in range(2)])
        assert np.abs(y - direct).max() < 1e-12
