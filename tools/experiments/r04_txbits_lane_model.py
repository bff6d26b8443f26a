#!/usr/bin/env python3
"""The lane algorithm of k_tx_qpsk_bits (round 4: input as bit-reversed words, start registers of the 64 lanes by a log-step scan over
GF(2) affine maps, coded bits as XORs of shifted words, symbol sums as popcounts) against a plain serial model of
scrambler_bb(0x8A, seed, 7) -> cc_encoder(7, {109, 79}) -> map{0,1,3,2} -> diff_encoder_bb(4), for ragged sizes and random carried state.
Written and run BEFORE the kernel was rewritten; kept as the executable statement of what the kernel computes per lane."""
import numpy as np
rng = np.random.default_rng(1)
def par(x): return bin(x).count("1") & 1
def serial(data, sr, enc, prev):
    bits = np.unpackbits(data)  # MSB first
    out = []
    reg = 0
    for k in range(6): reg |= ((enc >> k) & 1) << (k + 1)   # reg bit k+1 = s(-1-k)
    for b in bits:
        s = sr & 1
        nb = par(sr & 0x8A) ^ int(b)
        sr = (sr >> 1) | (nb << 7)
        reg = ((reg << 1) | s) & 127 if False else None
        out.append(s)
    return out, sr
def model(data, sr, enc, prev):
    bits = np.unpackbits(data)
    s_hist = [(enc >> k) & 1 for k in range(6)]   # s(-1-k)
    sb = []; syms = []
    run = prev
    for i, b in enumerate(bits):
        s = sr & 1
        nb = par(sr & 0x8A) ^ int(b)
        sr = (sr >> 1) | (nb << 7)
        sb.append(s)
        def sbit(j): return sb[j] if j >= 0 else s_hist[-j - 1]
        reg = 0
        for k in range(7): reg |= sbit(i - k) << k
        c0, c1 = par(reg & 109), par(reg & 79)
        m = [0, 1, 3, 2][(c0 << 1) | c1]
        run = (run + m) & 3
        syms.append(run)
    n = len(bits)
    enc2 = 0
    for k in range(6):
        j = n - 1 - k
        enc2 |= (sb[j] if j >= 0 else s_hist[-j - 1]) << k
    return syms, sr, enc2, run

def lfsr_power(L):
    cols = []
    for k in range(8):
        sr = 1 << k
        for _ in range(L):
            nb = par(sr & 0x8A); sr = (sr >> 1) | (nb << 7)
        cols.append(sr)
    return cols
def gf2(cols, v):
    r = 0
    for k in range(8):
        if (v >> k) & 1: r ^= cols[k]
    return r
def brev(x, n): return int(bin(x)[2:].zfill(n)[::-1], 2)

def lanes(data, sr0, enc, prev):
    nbytes = len(data); nbits = 8 * nbytes
    L = ((nbits + 63) // 64 + 31) // 32 * 32
    A = [lfsr_power(L * (1 << d)) for d in range(6)]
    nw = L // 32
    padded = np.concatenate([data, np.zeros(64 * L // 8 - nbytes + 8, np.uint8)])
    lo = [min(nbits, l * L) for l in range(64)]; hi = [min(nbits, (l + 1) * L) for l in range(64)]
    # 1. input words
    W = [[0] * nw for _ in range(64)]
    for l in range(64):
        for w in range(nw):
            byte0 = (l * L + 32 * w) // 8
            d = int(padded[byte0]) | int(padded[byte0 + 1]) << 8 | int(padded[byte0 + 2]) << 16 | int(padded[byte0 + 3]) << 24
            r = brev(d, 32)
            W[l][w] = ((r & 0xff) << 24) | ((r & 0xff00) << 8) | ((r >> 8) & 0xff00) | (r >> 24)   # bswap(brev32)
    # 2. pass 1
    sf = []
    for l in range(64):
        sr = 0
        for i in range(lo[l], hi[l]):
            j = i - l * L; b = (W[l][j >> 5] >> (j & 31)) & 1
            nb = par(sr & 0x8A) ^ b; sr = (sr >> 1) | (nb << 7)
        sf.append(sr)
    # 3. scan
    c = sf[:]
    c[0] ^= gf2(A[0], sr0)
    for d in range(6):
        o = 1 << d
        t = [c[l - o] if l >= o else 0 for l in range(64)]
        c = [c[l] ^ gf2(A[d], t[l]) if l >= o else c[l] for l in range(64)]
    start = [sr0] + c[:63]
    # 4. pass 2
    S = [[0] * nw for _ in range(64)]; srl = []
    for l in range(64):
        sr = start[l]
        for i in range(lo[l], hi[l]):
            j = i - l * L; b = (W[l][j >> 5] >> (j & 31)) & 1
            S[l][j >> 5] |= (sr & 1) << (j & 31)
            nb = par(sr & 0x8A) ^ b; sr = (sr >> 1) | (nb << 7)
        srl.append(sr)
    last_lane = (nbits - 1) // L if nbits else 0
    sr_end = srl[last_lane]
    flat = [S[l][w] for l in range(64) for w in range(nw)]
    # 5/6. encoder words
    local = []; CW = [[None] * nw for _ in range(64)]
    for l in range(64):
        loc = 0
        for w in range(nw):
            gi = l * nw + w
            prev6 = brev(enc & 63, 6) if gi == 0 else (flat[gi - 1] >> 26)
            T = (flat[gi] << 6) | prev6
            c0 = ((T >> 6) ^ (T >> 4) ^ (T >> 3) ^ (T >> 1) ^ T) & 0xffffffff
            c1 = ((T >> 6) ^ (T >> 5) ^ (T >> 4) ^ (T >> 3) ^ T) & 0xffffffff
            i0 = l * L + 32 * w
            nvalid = max(0, min(32, hi[l] - i0))
            vm = (1 << nvalid) - 1
            CW[l][w] = (c0, c1, nvalid)
            loc += 2 * bin(c0 & vm).count("1") + bin((c0 ^ c1) & vm).count("1")
        local.append(loc & 3)
    incl = local[:]
    for d in range(6):
        o = 1 << d
        incl = [(incl[l] + incl[l - o]) & 3 if l >= o else incl[l] for l in range(64)]
    syms = []
    for l in range(64):
        run = (prev + incl[l] - local[l]) & 3
        for w in range(nw):
            c0, c1, nv = CW[l][w]
            for t in range(nv):
                m = 2 * ((c0 >> t) & 1) + (((c0 ^ c1) >> t) & 1)
                run = (run + m) & 3
                syms.append(run)
        if l == last_lane: prev_end = run
    enc2 = 0
    for k in range(6):
        j = nbits - 1 - k
        if j >= 0: b = (flat[j >> 5] >> (j & 31)) & 1
        else: b = (enc >> (-j - 1)) & 1
        enc2 |= b << k
    return syms, sr_end, enc2, prev_end

for nbytes in (1, 3, 7, 8, 9, 64, 255, 256, 257, 512, 1527):
    for trial in range(3):
        data = rng.integers(0, 256, nbytes, dtype=np.uint8)
        sr0, enc, prev = int(rng.integers(0, 256)), int(rng.integers(0, 64)), int(rng.integers(0, 4))
        a = model(data, sr0, enc, prev); b = lanes(data, sr0, enc, prev)
        assert a[0] == b[0], (nbytes, "syms")
        assert a[1:] == b[1:], (nbytes, a[1:], b[1:])
print("lane algorithm == serial model")
