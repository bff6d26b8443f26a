#!/usr/bin/env python3
"""NOT in the kernel yet (round 4 ran out of GPU minutes to verify it there): the scrambler of k_tx_qpsk_bits without a bit loop.

scrambler_bb(0x8A, seed, 7) as the kernel steps it: nb = in ^ parity(sr & 0x8A); out = sr & 1; sr = (sr >> 1) | (nb << 7).  With
y[i] = nb of step i the register is (y[i-1] .. y[i-8]) from bit 7 down to bit 0, so
    y[i] = x[i] ^ y[i-1] ^ y[i-5] ^ y[i-7]        (0x8A = register bits 7, 3, 1),      out[i] = y[i-8].
Over GF(2)[[D]]: y = x / (1 + P), P = D + D^5 + D^7, and 1 / (1 + P) = prod_k (1 + P^(2^k)), P^(2^k) = D^(2^k) + D^(5 2^k) + D^(7 2^k):
for a 64-bit word six factors of three shift-xors each give the ZERO-STATE response of the word; the carried register adds its zero-input
response, linear in its 8 bits (8 precomputed 64-bit words R_k and 8 state words Q_k); the register behind the word is the word's last 8
y bits.  ~ 40 64-bit operations per 64 input bits instead of two passes of 64 x 6.  Checked here against the bit-serial recursion."""
import numpy as np

MASK = (1 << 64) - 1
def par(x): return bin(x).count("1") & 1

def serial(bits, sr):
    out = []
    for b in bits:
        out.append(sr & 1)
        nb = par(sr & 0x8A) ^ int(b)
        sr = (sr >> 1) | (nb << 7)
    return out, sr

def zero_state_y(x):                       # y = x / (1 + D + D^5 + D^7) truncated to 64 bits; bit i = step i
    y = x
    for k in range(6):
        s = 1 << k
        y ^= ((y << s) ^ (y << 5 * s) ^ (y << 7 * s)) & MASK
    return y

def tables():
    """zero-input response of register bit k over 64 steps: R[k] = the y word, i.e. what the feedback produces from that state alone"""
    R = []
    for k in range(8):
        sr = 1 << k
        y = 0
        for i in range(64):
            nb = par(sr & 0x8A)
            y |= nb << i
            sr = (sr >> 1) | (nb << 7)
        R.append(y)
    return R

R = tables()

def word_step(x, sr):
    """one 64-bit word: returns (out word, register behind it)"""
    y = zero_state_y(x)
    for k in range(8):
        if (sr >> k) & 1: y ^= R[k]
    # out[i] = y[i-8]: the first 8 outputs are the register itself (bit 0 first), then y shifted
    out = ((y << 8) & MASK) | sr
    sr_new = y >> 56                       # y[56..63] -> register bits 0..7 = (y[i-8] .. y[i-1]) at i = 64
    return out, sr_new

rng = np.random.default_rng(2)
for trial in range(200):
    nw = int(rng.integers(1, 5))
    bits = rng.integers(0, 2, 64 * nw)
    sr0 = int(rng.integers(0, 256))
    ref_out, ref_sr = serial(bits, sr0)
    sr = sr0; got = []
    for w in range(nw):
        x = 0
        for i in range(64): x |= int(bits[64 * w + i]) << i
        o, sr = word_step(x, sr)
        got += [(o >> i) & 1 for i in range(64)]
    assert got == ref_out and sr == ref_sr, trial
print("word-parallel scrambler == bit-serial recursion (200 random cases, 1..4 words, random registers)")
