#!/usr/bin/env python3
"""Generator of forge_ec_amd/csrc/field_asm.inc: hand-allocated gfx950 assembly for the field
multiplications of the three curves.

    python tools/gen_field_asm.py            # rewrites forge_ec_amd/csrc/field_asm.inc

Why generated assembly: hipcc's lowering of the same arithmetic spends ~21 % of the ladder's VALU
instructions on v_mov (profiles/r02_isa_mix_*.md): 64-bit VGPR operands are even-aligned pairs, so
every column of the product scanning re-assembles its {carry-in, overflow} pair, and the Montgomery
recurrence re-assembles a pair per word.  Here every temporary lives in a FIXED register block
(v[VB..255], declared as clobbers), so halves of pairs are addressed directly:

  * column k of the 512-bit product accumulates in its own pair Q_k = v[VB+2k : VB+2k+1]; the finished
    word T[k] stays where the last v_mad_u64_u32 left it (the pair's low register), the pair's high
    register is copied once into the carry pair C (1 v_mov per column instead of 2-3), and the
    overflow word is counted directly in C's high register;
  * secp256k1: the Montgomery word recurrence (one v_mul_lo_u32 + one v_mad_u64_u32 per word) runs
    interleaved with the following column, its 33rd bit travels in an SGPR pair as a carry, and the
    word m_k is parked in the dead high register of Q_k;
  * the compiler sees one asm statement per field multiplication: 8 outputs, 16 inputs, 2-4 scalars.

The statements are semantically the routines they replace (limbs.hpp mul_wide + the per-curve
reduction); tests/test_gpu_parity.py and the host emulation (which keeps the portable C++ form)
pin them against the oracle.
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "forge_ec_amd", "csrc", "field_asm.inc")


class Block:
    """One asm statement under construction."""

    def __init__(self):
        self.lines = []

    def e(self, s):
        self.lines.append(s)

    def text(self):
        return "\n".join('      "%s\\n\\t"' % l for l in self.lines[:-1]) + '\n      "%s"' % self.lines[-1]


def v(n):
    return "v%d" % n


def vp(n):
    assert n % 2 == 0, n
    return "v[%d:%d]" % (n, n + 1)


# The capture of a v_mad_u64_u32's carry, directly behind it: VOP2 (implicit vcc).  In isolation the VOP3 encoding is
# the cheaper one behind a mad (mad + addc_e32 9.6 cycles per pair per SIMD at three wavefronts, mad + addc_e64 8.7:
# tools/microbench/valu_occ.hip); in the kernels it is the slower one (secp256k1 28.72 -> 29.20 ms, P-256 23.79 -> 24.10,
# Ed25519 table kernel 5.03 -> 5.19, same box, round 3) -- eight bytes to fetch instead of four, 1 100 times per ladder step.
# Likewise the column's closing v_mov moved between the last mad and its capture (free in isolation: mad, mov, addc 9.5
# cycles against 9.6 for mad, addc): 28.84 -> 28.97 ms, nothing for the other kernels.
CAPTURE = "v_addc_co_u32_e32 %s, vcc, 0, %s, vcc"


def interleave(main, side):
    """Spread the instructions of `side` through `main`: one after every main instruction that is
    not a v_mad (so a mad and the v_addc that consumes its carry stay adjacent)."""
    out, si = [], 0
    for i, m in enumerate(main):
        out.append(m)
        if si < len(side) and i >= 1 and not m.startswith("v_mad"):
            out.append(side[si])
            si += 1
    out.extend(side[si:])
    return out


def mul_wide_columns(A, B, VB, side_for_column=None, first_col_src=None):
    """Product scanning of A[0..7] x B[0..7] (operand strings).  Column k accumulates in the pair
    Q_k = v[VB+2k : VB+2k+1]; C = v[VB+30 : VB+31] is the carry pair {next column's carry-in low,
    overflow count}.  Returns the instruction list; afterwards T[k] = v(VB+2k) for k <= 14 and
    T[15] = v(VB+29).  side_for_column(k) -> instructions to interleave into column k."""
    C = VB + 30
    ins = ["v_mov_b32_e32 %s, 0" % v(C + 1)]
    for k in range(15):
        q = VB + 2 * k
        lo = max(0, k - 7)
        prods = [(i, k - i) for i in range(lo, min(k, 7) + 1)]
        main = []
        for n, (i, j) in enumerate(prods):
            if n == 0:
                src2 = "0" if k == 0 else vp(C)
                main.append("v_mad_u64_u32 %s, vcc, %s, %s, %s" % (vp(q), A[i], B[j], src2))
                # carry-in < 2^32 for column 1 and the top column is bounded by the true product
                if k >= 2 and k != 14:
                    main.append("v_addc_co_u32_e64 %s, vcc, 0, 0, vcc" % v(C + 1))
            else:
                main.append("v_mad_u64_u32 %s, vcc, %s, %s, %s" % (vp(q), A[i], B[j], vp(q)))
                main.append(CAPTURE % (v(C + 1), v(C + 1)))
        if k != 14:
            main.append("v_mov_b32_e32 %s, %s" % (v(C), v(q + 1)))
        side = side_for_column(k) if side_for_column else []
        ins.extend(interleave(main, side))
    return ins


# ------------------------------------------------------------------------------------------------
# secp256k1 Mul (secp256k1.rs:442-507) = csub_p((T_hi + M - Q) mod 2^256), see secp256k1.hpp
# operands: %0-%7 r, %8 sc (SGPR pair: the recurrence's 33rd bit), %9 bw (SGPR pair: dummy carry
# sink, then the borrow out of word 1), %10-%17 a, %18-%25 b, %26 N0' (s), %27 977 (s)
# ------------------------------------------------------------------------------------------------
def secp_mul(VB):
    A = ["%%%d" % (10 + i) for i in range(8)]
    B = ["%%%d" % (18 + i) for i in range(8)]
    R = ["%%%d" % i for i in range(8)]
    SC, BW, N0P, C977 = "%8", "%9", "%26", "%27"
    P, D = VB + 32, VB + 34  # P = {e_lo, 0}; D = m*977 + P
    T = [VB + 2 * k for k in range(15)] + [VB + 29]
    M = [VB + 2 * k + 1 for k in range(8)]

    def side(col):
        k = col - 1  # recurrence step k needs the finished T[k]
        if k < 0 or k > 7:
            return []
        if k == 0:
            return ["v_mul_lo_u32 %s, %s, %s" % (v(M[0]), v(T[0]), N0P),
                    "v_mad_u64_u32 %s, %s, %s, %s, 0" % (vp(D), BW, v(M[0]), C977),
                    "v_add_co_u32_e64 %s, %s, %s, %s" % (v(P), SC, v(D + 1), v(M[0]))]
        return ["v_sub_u32_e32 %s, %s, %s" % (v(M[k]), v(T[k]), v(P)),
                "v_mul_lo_u32 %s, %s, %s" % (v(M[k]), v(M[k]), N0P),
                "v_mad_u64_u32 %s, %s, %s, %s, %s" % (vp(D), BW, v(M[k]), C977, vp(P)),
                "v_addc_co_u32_e64 %s, %s, %s, %s, %s" % (v(P), SC, v(D + 1), v(M[k]), SC)]

    b = Block()
    b.e("v_mov_b32_e32 %s, 0" % v(P + 1))
    for s in mul_wide_columns(A, B, VB, side):
        b.e(s)
    # V = T_hi + M - Q (mod 2^256), Q = {P.lo, sc}
    b.e("v_add_co_u32_e32 %s, vcc, %s, %s" % (R[0], v(T[8]), v(M[0])))
    for i in range(1, 8):
        b.e("v_addc_co_u32_e32 %s, vcc, %s, %s, vcc" % (R[i], v(T[8 + i]), v(M[i])))
    b.e("v_cndmask_b32_e64 %s, 0, 1, %s" % (v(P + 1), SC))
    b.e("v_sub_co_u32_e32 %s, vcc, %s, %s" % (R[0], R[0], v(P)))
    b.e("v_subb_co_u32_e32 %s, vcc, %s, %s, vcc" % (R[1], R[1], v(P + 1)))
    b.e("s_mov_b64 %s, vcc" % BW)
    return b, list(range(VB, VB + 36))


# ------------------------------------------------------------------------------------------------
# secp256k1 Mul by a raw small constant (three / eight at secp256k1.rs:1523, 1533): T = a * k has
# nine words, so V = T_hi + M - Q is M + t8 - Q.  The carry of m0 + t8 (t8 < 8) out of word 0 is an
# exception lane (2^-29), the borrow of - Q out of word 1 the same rare continuation as in Mul.
# operands: %0-%7 r, %8 sc, %9 bw (sink, then the borrow mask), %10 exc, %11-%18 a, %19 N0' (s), %20 977 (s)
# ------------------------------------------------------------------------------------------------
def secp_mul_small(VB, K):
    A = ["%%%d" % (11 + i) for i in range(8)]
    R = ["%%%d" % i for i in range(8)]
    SC, BW, EXC, N0P, C977 = "%8", "%9", "%10", "%19", "%20"
    Q = [VB + 2 * i for i in range(8)]      # Q_i = a_i * K + hi(Q_{i-1}): T[i] = lo(Q_i), T[8] = hi(Q_7)
    C = VB + 16                             # {hi(Q_{i-1}), 0}
    P, D = VB + 18, VB + 20
    b = Block()
    b.e("v_mov_b32_e32 %s, 0" % v(C + 1))
    b.e("v_mov_b32_e32 %s, 0" % v(P + 1))
    for i in range(8):
        b.e("v_mad_u64_u32 %s, %s, %s, %d, %s" % (vp(Q[i]), BW, A[i], K, "0" if i == 0 else vp(C)))
        if i < 7:
            b.e("v_mov_b32_e32 %s, %s" % (v(C), v(Q[i] + 1)))
    T = [v(q) for q in Q]
    t8 = v(Q[7] + 1)
    # the Montgomery word recurrence, m_k straight into the output operands
    b.e("v_mul_lo_u32 %s, %s, %s" % (R[0], T[0], N0P))
    b.e("v_mad_u64_u32 %s, %s, %s, %s, 0" % (vp(D), BW, R[0], C977))
    b.e("v_add_co_u32_e64 %s, %s, %s, %s" % (v(P), SC, v(D + 1), R[0]))
    for k in range(1, 8):
        b.e("v_sub_u32_e32 %s, %s, %s" % (R[k], T[k], v(P)))
        b.e("v_mul_lo_u32 %s, %s, %s" % (R[k], R[k], N0P))
        b.e("v_mad_u64_u32 %s, %s, %s, %s, %s" % (vp(D), BW, R[k], C977, vp(P)))
        b.e("v_addc_co_u32_e64 %s, %s, %s, %s, %s" % (v(P), SC, v(D + 1), R[k], SC))
    b.e("v_add_co_u32_e32 %s, vcc, %s, %s" % (R[0], R[0], t8))
    b.e("s_mov_b64 %s, vcc" % EXC)
    b.e("v_cndmask_b32_e64 %s, 0, 1, %s" % (v(P + 1), SC))
    b.e("v_sub_co_u32_e32 %s, vcc, %s, %s" % (R[0], R[0], v(P)))
    b.e("v_subb_co_u32_e32 %s, vcc, %s, %s, vcc" % (R[1], R[1], v(P + 1)))
    b.e("s_mov_b64 %s, vcc" % BW)
    return b, list(range(VB, VB + 22))


# ------------------------------------------------------------------------------------------------
# secp256k1 square() (secp256k1.rs:634-713), common path: see secp256k1.hpp sqr_cxx for the literal
# restatement.  Every data-dependent continuation of the reference (a +1 that ripples past the limb it
# is added to; the fold's general carry rule) fires only when a 64-bit limb is all ones; the lanes
# where one would fire are collected in the exception mask %9 and the caller recomputes those
# wavefronts with sqr_cxx.
# operands: %0-%7 r, %8 tmp (SGPR pair), %9 exc (SGPR pair), %10-%17 a, %18 977 (s)
# ------------------------------------------------------------------------------------------------
def secp_sqr(VB):
    A = ["%%%d" % (10 + i) for i in range(8)]
    R = ["%%%d" % i for i in range(8)]
    TMP, EXC, C977 = "%8", "%9", "%18"
    # register homes: every 64-bit product gets its own aligned pair
    nxt = [VB]

    def pair():
        r = nxt[0]
        nxt[0] += 2
        return r

    Z, Y = pair(), pair()  # Z = {x, 0} zero-extension pair; Y = {acc.hi, ovf}
    b = Block()
    b.e("v_mov_b32_e32 %s, 0" % v(Z + 1))
    b.e("s_mov_b64 %s, 0" % EXC)

    def mul64(x0, x1, y0, y1, pa, pb, pc):
        """p = (x1:x0)*(y1:y0): p0 = lo(pa), p1 = lo(pb), p2 = lo(pc), p3 = hi(pc)."""
        b.e("v_mad_u64_u32 %s, vcc, %s, %s, 0" % (vp(pa), x0, y0))
        b.e("v_mov_b32_e32 %s, %s" % (v(Z), v(pa + 1)))
        b.e("v_mad_u64_u32 %s, vcc, %s, %s, %s" % (vp(pb), x0, y1, vp(Z)))
        b.e("v_mad_u64_u32 %s, vcc, %s, %s, %s" % (vp(pb), x1, y0, vp(pb)))
        b.e("v_addc_co_u32_e64 %s, vcc, 0, 0, vcc" % v(Y + 1))
        b.e("v_mov_b32_e32 %s, %s" % (v(Y), v(pb + 1)))
        b.e("v_mad_u64_u32 %s, vcc, %s, %s, %s" % (vp(pc), x1, y1, vp(Y)))
        return [v(pa), v(pb), v(pc), v(pc + 1)]

    W = [None] * 16
    for i in range(4):  # 643-649: limb squares
        pa, pb, pc = pair(), pair(), pair()
        W[4 * i:4 * i + 4] = mul64(A[2 * i], A[2 * i + 1], A[2 * i], A[2 * i + 1], pa, pb, pc)
    xa, xb, xc = pair(), pair(), pair()  # cross products reuse one set of pairs
    for i in range(4):  # 652-679: doubled cross terms
        for j in range(i + 1, 4):
            p0, p1, p2, p3 = mul64(A[2 * i], A[2 * i + 1], A[2 * j], A[2 * j + 1], xa, xb, xc)
            # u128 wrapping_mul(2): bit 127 is lost.  Top word first so every source is still intact.
            b.e("v_alignbit_b32 %s, %s, %s, 31" % (p3, p3, p2))
            b.e("v_alignbit_b32 %s, %s, %s, 31" % (p2, p2, p1))
            b.e("v_alignbit_b32 %s, %s, %s, 31" % (p1, p1, p0))
            b.e("v_lshlrev_b32_e32 %s, 1, %s" % (p0, p0))
            B = 2 * (i + j)
            b.e("v_add_co_u32_e32 %s, vcc, %s, %s" % (W[B], W[B], p0))
            b.e("v_addc_co_u32_e32 %s, vcc, %s, %s, vcc" % (W[B + 1], W[B + 1], p1))
            b.e("s_mov_b64 %s, vcc" % TMP)
            b.e("v_add_co_u32_e32 %s, vcc, %s, %s" % (W[B + 2], W[B + 2], p2))
            b.e("v_addc_co_u32_e32 %s, vcc, %s, %s, vcc" % (W[B + 3], W[B + 3], p3))
            b.e("s_or_b64 vcc, vcc, %s" % TMP)
            b.e("v_addc_co_u32_e32 %s, vcc, 0, %s, vcc" % (W[B + 4], W[B + 4]))
            b.e("v_addc_co_u32_e32 %s, vcc, 0, %s, vcc" % (W[B + 5], W[B + 5]))
            if B + 6 < 16:
                b.e("s_or_b64 %s, %s, vcc" % (EXC, EXC))
    # 681-707: every high limb folded into limb 0 with the low 64 bits of limb * 0x1000003D1
    F = xa
    T = xb  # scratch word
    for i in range(4, 8):
        h0, h1 = W[2 * i], W[2 * i + 1]
        b.e("v_mad_u64_u32 %s, vcc, %s, %s, 0" % (vp(F), h0, C977))
        b.e("v_mul_lo_u32 %s, %s, %s" % (v(T), h1, C977))
        b.e("v_add3_u32 %s, %s, %s, %s" % (v(T), v(F + 1), v(T), h0))
        dst = R if i == 7 else W  # the last fold leaves limbs 0..1 in the output operands
        b.e("v_add_co_u32_e32 %s, vcc, %s, %s" % (dst[0], W[0], v(F)))
        b.e("v_addc_co_u32_e32 %s, vcc, %s, %s, vcc" % (dst[1], W[1], v(T)))
        b.e("v_addc_co_u32_e32 %s, vcc, 0, %s, vcc" % (dst[2], W[2]))
        b.e("v_addc_co_u32_e32 %s, vcc, 0, %s, vcc" % (dst[3], W[3]))
        b.e("s_or_b64 %s, %s, vcc" % (EXC, EXC))
    # limbs 2..3 (W[4..7]) are final after the cross terms: their last writers target the output
    # operands directly (those four outputs are early-clobber in the C++ wrapper)
    for i in range(4, 8):
        last = max(n for n, l in enumerate(b.lines) if l.split()[1].rstrip(",") == W[i])
        assert not any(W[i] in l for l in b.lines[last + 1:]), W[i]
        parts = b.lines[last].split(" ", 2)
        b.lines[last] = "%s %s, %s" % (parts[0], R[i], parts[2])
    return b, list(range(VB, nxt[0]))


def sqr_wide_columns(A, VB, sink):
    """Exact 512-bit square of A[0..7]: the 28 cross products a_i a_j (i < j) by product scanning
    (column k in the pair X_k = v[VB+2k : VB+2k+1], k = 1..13, carry pair C = v[VB+28 : VB+29]),
    doubled with v_alignbit, plus the eight squares a_i^2 (pairs S_i = v[VB+30+2i : ...]).
    Returns (instructions, T) where T[k] names the register of word k of the square.
    `sink` = an SGPR-pair operand for carries that cannot occur."""
    C = VB + 28
    S = [VB + 30 + 2 * i for i in range(8)]
    ins = ["v_mov_b32_e32 %s, 0" % v(C + 1)]
    for k in range(1, 14):
        q = VB + 2 * k
        lo = max(0, k - 7)
        prods = [(i, k - i) for i in range(lo, (k - 1) // 2 + 1)]
        for n, (i, j) in enumerate(prods):
            if n == 0:
                src2 = "0" if k == 1 else vp(C)
                ins.append("v_mad_u64_u32 %s, vcc, %s, %s, %s" % (vp(q), A[i], A[j], src2))
                if k >= 4:  # columns 1..3: the carry-in is < 2^32, the first product cannot overflow
                    ins.append("v_addc_co_u32_e64 %s, vcc, 0, 0, vcc" % v(C + 1))
            else:
                ins.append("v_mad_u64_u32 %s, vcc, %s, %s, %s" % (vp(q), A[i], A[j], vp(q)))
                ins.append(CAPTURE % (v(C + 1), v(C + 1)))
        if k != 13:
            ins.append("v_mov_b32_e32 %s, %s" % (v(C), v(q + 1)))
        if k % 2 == 1:  # spread the independent squares through the columns
            i = (k - 1) // 2
            ins.append("v_mad_u64_u32 %s, %s, %s, %s, 0" % (vp(S[i]), sink, A[i], A[i]))
    ins.append("v_mad_u64_u32 %s, %s, %s, %s, 0" % (vp(S[7]), sink, A[7], A[7]))
    # X[0] = 0, X[k] = v(VB+2k) for 1 <= k <= 13, X[14] = hi of column 13, X[15] = its overflow count
    X = [None] + [VB + 2 * k for k in range(1, 14)] + [VB + 27, C + 1]
    # Y = 2X in place, top word first
    for k in range(15, 1, -1):
        ins.append("v_alignbit_b32 %s, %s, %s, 31" % (v(X[k]), v(X[k]), v(X[k - 1])))
    ins.append("v_lshlrev_b32_e32 %s, 1, %s" % (v(X[1]), v(X[1])))
    # T = Y + D, D[2i] = lo(S_i), D[2i+1] = hi(S_i); T[0] = D[0] (Y[0] = 0), sums in place in X
    ins.append("v_add_co_u32_e32 %s, vcc, %s, %s" % (v(X[1]), v(X[1]), v(S[0] + 1)))
    for k in range(2, 16):
        ins.append("v_addc_co_u32_e32 %s, vcc, %s, %s, vcc" % (v(X[k]), v(X[k]), v(S[k // 2] + (k & 1))))
    T = [S[0]] + X[1:]
    return ins, T


# ------------------------------------------------------------------------------------------------
# P-256 reduce_wide_p256 (p256.rs:544-704) on T[0..15] (register numbers), result into the operands
# R[0..7].  S = s1 + 2 s2 + 2 s3 + s4 + s5 - s6 - s7 - s8 - s9 as 256-bit chains on the accumulator
# T[0..7] with a signed ninth word; then r = L - carry*p mod 2^256 (677-698).  The closing reduce()
# (701) is left to the caller (one compare of the top word).  `free` = 8 dead registers.
# ------------------------------------------------------------------------------------------------
def p256_reduce(T, R, free):
    c = [v(t) for t in T]
    w = c[:8]
    U = [v(f) for f in free[:6]]   # s2 + s3, words 3..8
    top, sx, v3, c3, v6, v7 = [v(f) for f in free[6:12]]
    ins = []
    # U = s2 + s3 = (c11+c12, c12+c13, c13+c14, c14+c15, c15, carry) at words 3..8
    ins.append("v_add_co_u32_e32 %s, vcc, %s, %s" % (U[0], c[11], c[12]))
    ins.append("v_addc_co_u32_e32 %s, vcc, %s, %s, vcc" % (U[1], c[12], c[13]))
    ins.append("v_addc_co_u32_e32 %s, vcc, %s, %s, vcc" % (U[2], c[13], c[14]))
    ins.append("v_addc_co_u32_e32 %s, vcc, %s, %s, vcc" % (U[3], c[14], c[15]))
    ins.append("v_addc_co_u32_e32 %s, vcc, 0, %s, vcc" % (U[4], c[15]))
    ins.append("v_addc_co_u32_e64 %s, vcc, 0, 0, vcc" % U[5])
    ins.append("v_mov_b32_e32 %s, 0" % top)

    def chain(words, sub):
        """acc +-= words (list of 8 operand strings or None for a zero word); top +-= carry."""
        first = True
        for i in range(8):
            x = words[i]
            if first:
                if x is None:
                    continue  # leading zero words change nothing
                ins.append("%s %s, vcc, %s, %s" % ("v_sub_co_u32_e32" if sub else "v_add_co_u32_e32", w[i], w[i], x))
                first = False
            elif x is None:
                ins.append("%s %s, vcc, 0, %s, vcc" % ("v_subbrev_co_u32_e32" if sub else "v_addc_co_u32_e32", w[i], w[i]))
            else:
                ins.append("%s %s, vcc, %s, %s, vcc" % ("v_subb_co_u32_e32" if sub else "v_addc_co_u32_e32", w[i], w[i], x))
        return None

    for rep in range(2):  # 2 (s2 + s3)
        chain([None, None, None, U[0], U[1], U[2], U[3], U[4]], False)
        ins.append("v_addc_co_u32_e32 %s, vcc, %s, %s, vcc" % (top, top, U[5]))
    Z = None
    chain([c[8], c[9], c[10], Z, Z, Z, c[14], c[15]], False)  # s4
    ins.append("v_addc_co_u32_e32 %s, vcc, 0, %s, vcc" % (top, top))
    chain([c[9], c[10], c[11], c[13], c[14], c[15], c[13], c[8]], False)  # s5
    ins.append("v_addc_co_u32_e32 %s, vcc, 0, %s, vcc" % (top, top))
    for words in ([c[11], c[12], c[13], Z, Z, Z, c[8], c[10]],       # s6
                  [c[12], c[13], c[14], c[15], Z, Z, c[9], c[11]],   # s7
                  [c[13], c[14], c[15], c[8], c[9], c[10], Z, c[12]],  # s8
                  [c[14], c[15], Z, c[9], c[10], c[11], Z, c[13]]):  # s9
        chain(words, True)
        ins.append("v_subbrev_co_u32_e32 %s, vcc, 0, %s, vcc" % (top, top))
    # r = L + t*(1 - 2^96 - 2^192 + 2^224) mod 2^256, t = top (signed): the words of t*K are
    # (t, s, s, s - t, c3, c3, c3 - t, t + c6) with s = t >> 31, c3 = (s - t) >> 31, c6 = (c3 - t) >> 31
    ins.append("v_ashrrev_i32_e32 %s, 31, %s" % (sx, top))
    ins.append("v_sub_u32_e32 %s, %s, %s" % (v3, sx, top))
    ins.append("v_ashrrev_i32_e32 %s, 31, %s" % (c3, v3))
    ins.append("v_sub_u32_e32 %s, %s, %s" % (v6, c3, top))
    ins.append("v_ashrrev_i32_e32 %s, 31, %s" % (v7, v6))
    ins.append("v_add_u32_e32 %s, %s, %s" % (v7, v7, top))
    adds = [top, sx, sx, v3, c3, c3, v6, v7]
    ins.append("v_add_co_u32_e32 %s, vcc, %s, %s" % (R[0], w[0], adds[0]))
    for i in range(1, 8):
        ins.append("v_addc_co_u32_e32 %s, vcc, %s, %s, vcc" % (R[i], w[i], adds[i]))
    return ins


def p256_mul_small(VB, K):
    """FieldElement::from(K) * a (p256.rs:1893-1904, K = 3 or 8): the nine-word product a*K, then
    reduce_wide_p256 with c9..c15 = 0, which collapses to S = T_lo + c8*(2^224 - 2^192 - 2^96 + 1);
    the carry of that sum out of 2^256 (c8 < 8, so about 2^-29 per lane) is the exception mask.
    operands: %0-%7 r, %8 sink, %9 exc, %10-%17 a"""
    A = ["%%%d" % (10 + i) for i in range(8)]
    R = ["%%%d" % i for i in range(8)]
    SINK, EXC = "%8", "%9"
    Q = [VB + 2 * i for i in range(8)]
    C = VB + 16
    t, sx, v3, c3, v6, v7 = [v(VB + 18 + i) for i in range(6)]
    b = Block()
    b.e("v_mov_b32_e32 %s, 0" % v(C + 1))
    for i in range(8):
        b.e("v_mad_u64_u32 %s, %s, %s, %d, %s" % (vp(Q[i]), SINK, A[i], K, "0" if i == 0 else vp(C)))
        if i < 7:
            b.e("v_mov_b32_e32 %s, %s" % (v(C), v(Q[i] + 1)))
    top = v(Q[7] + 1)   # c8 >= 0: the general signed formula of p256_reduce with s = 0
    b.e("v_sub_u32_e32 %s, 0, %s" % (v3, top))                 # word 3 of c8*K: -c8
    b.e("v_ashrrev_i32_e32 %s, 31, %s" % (c3, v3))             # its sign extension (0 or -1)
    b.e("v_sub_u32_e32 %s, %s, %s" % (v6, c3, top))            # word 6: c3 - c8
    b.e("v_ashrrev_i32_e32 %s, 31, %s" % (v7, v6))
    b.e("v_add_u32_e32 %s, %s, %s" % (v7, v7, top))            # word 7: c8 + sign(word 6)
    adds = [top, "0", "0", v3, c3, c3, v6, v7]
    b.e("v_add_co_u32_e32 %s, vcc, %s, %s" % (R[0], v(Q[0]), adds[0]))
    for i in range(1, 8):
        if adds[i] == "0":
            b.e("v_addc_co_u32_e32 %s, vcc, 0, %s, vcc" % (R[i], v(Q[i])))
        else:
            b.e("v_addc_co_u32_e32 %s, vcc, %s, %s, vcc" % (R[i], v(Q[i]), adds[i]))
    b.e("s_mov_b64 %s, vcc" % EXC)
    return b, list(range(VB, VB + 24))


def p256_mul(VB):
    """operands: %0-%7 r, %8-%15 a, %16-%23 b"""
    A = ["%%%d" % (8 + i) for i in range(8)]
    B = ["%%%d" % (16 + i) for i in range(8)]
    R = ["%%%d" % i for i in range(8)]
    b = Block()
    for s in mul_wide_columns(A, B, VB):
        b.e(s)
    T = [VB + 2 * k for k in range(15)] + [VB + 29]
    free = [VB + 2 * k + 1 for k in range(14)]  # the dead high halves of the column pairs
    for s in p256_reduce(T, R, free):
        b.e(s)
    return b, list(range(VB, VB + 32))


def p256_sqr(VB):
    """operands: %0-%7 r, %8 sink (SGPR pair), %9-%16 a"""
    A = ["%%%d" % (9 + i) for i in range(8)]
    R = ["%%%d" % i for i in range(8)]
    b = Block()
    ins, T = sqr_wide_columns(A, VB, "%8")
    for s in ins:
        b.e(s)
    S = [VB + 30 + 2 * i for i in range(8)]
    used = set(T)
    free = [r for r in range(VB, VB + 46) if r not in used and r not in (VB + 28,)]
    assert len(free) >= 12, free
    for s in p256_reduce(T, R, free):
        b.e(s)
    return b, list(range(VB, VB + 46))


# ------------------------------------------------------------------------------------------------
# Ed25519 reduce_wide (ed25519.rs:260-289) + reduce (214-247), common path: low + 38*high word by
# word, the carry word times 19 added to word 0, bit 255 cleared and 19 added for it.  Each of the
# two small additions carries out of word 0 with probability ~2^-22, and the final value reaches
# p only with a top word of 0x7FFFFFFF: those lanes are returned in the exception mask and the
# caller recomputes the wavefront with the compiler-scheduled routine.
# ------------------------------------------------------------------------------------------------
def ed_reduce(T, R, sink, exc, tmp):
    """T: register numbers of the 16 product words; the registers T[0..7]+1 must be free (they
    are zeroed to zero-extend T[k]) -- true for the column pairs of mul_wide_columns."""
    ins = []
    for i in range(8):
        assert T[i] % 2 == 0
        ins.append("v_mov_b32_e32 %s, 0" % v(T[i] + 1))
    for i in range(8):  # D_i = T[8+i]*38 + T[i]  (in place in the pair of T[i])
        ins.append("v_mad_u64_u32 %s, %s, %s, 38, %s" % (vp(T[i]), sink, v(T[8 + i]), vp(T[i])))
    lo = [v(T[i]) for i in range(8)]
    hi = [v(T[i] + 1) for i in range(8)]
    t = v(tmp)
    # words 1..7 first: w_i = lo_i + hi_{i-1} + carry; the carry word (<= 38) ends in t
    ins.append("v_add_co_u32_e32 %s, vcc, %s, %s" % (R[1], lo[1], hi[0]))
    for i in range(2, 8):
        ins.append("v_addc_co_u32_e32 %s, vcc, %s, %s, vcc" % (R[i], lo[i], hi[i - 1]))
    ins.append("v_addc_co_u32_e32 %s, vcc, 0, %s, vcc" % (t, hi[7]))
    ins.append("v_mul_u32_u24_e32 %s, 19, %s" % (t, t))
    ins.append("v_add_co_u32_e32 %s, vcc, %s, %s" % (R[0], lo[0], t))
    ins.append("s_mov_b64 %s, vcc" % exc)
    ins.append("v_lshrrev_b32_e32 %s, 31, %s" % (t, R[7]))
    ins.append("v_and_b32_e32 %s, 0x7fffffff, %s" % (R[7], R[7]))
    ins.append("v_mul_u32_u24_e32 %s, 19, %s" % (t, t))
    ins.append("v_add_co_u32_e32 %s, vcc, %s, %s" % (R[0], R[0], t))
    ins.append("s_or_b64 %s, %s, vcc" % (exc, exc))
    ins.append("v_cmp_eq_u32_e32 vcc, 0x7fffffff, %s" % R[7])
    ins.append("s_or_b64 %s, %s, vcc" % (exc, exc))
    return ins


def ed_mul(VB):
    """operands: %0-%7 r, %8 sink, %9 exc, %10-%17 a, %18-%25 b"""
    A = ["%%%d" % (10 + i) for i in range(8)]
    B = ["%%%d" % (18 + i) for i in range(8)]
    R = ["%%%d" % i for i in range(8)]
    b = Block()
    for s in mul_wide_columns(A, B, VB):
        b.e(s)
    T = [VB + 2 * k for k in range(15)] + [VB + 29]
    for s in ed_reduce(T, R, "%8", "%9", VB + 31):
        b.e(s)
    return b, list(range(VB, VB + 32))


def ed_sqr(VB):
    """operands: %0-%7 r, %8 sink, %9 exc, %10-%17 a"""
    A = ["%%%d" % (10 + i) for i in range(8)]
    R = ["%%%d" % i for i in range(8)]
    b = Block()
    ins, T = sqr_wide_columns(A, VB, "%8")
    for s in ins:
        b.e(s)
    # T[0] = lo(S_0) and T[k] = lo(X_k), k = 1..7: even registers whose odd partners are dead by now
    for s in ed_reduce(T, R, "%8", "%9", VB + 28):
        b.e(s)
    return b, list(range(VB, VB + 46))


def clobbers(regs):
    return ", ".join('"v%d"' % r for r in regs)


HEADER = """// field_asm.inc -- GENERATED by tools/gen_field_asm.py; do not edit.
// Hand-allocated gfx950 assembly for the field multiplications (one asm statement each).  All
// temporaries live in the fixed VGPR block named in each statement's clobber list.
#pragma once
"""


SECP_TOP = 168
P256_TOP = int(os.environ.get("FEC_P256_TOP", "168"))  # the P-256 scheduler runs three wavefronts per SIMD (168 VGPRs)
ED_TOP = int(os.environ.get("FEC_ED_TOP", "168"))    # the Ed25519 fixed-base kernel runs three wavefronts per SIMD


def main():
    parts = [HEADER]
    report = []

    def add(name, blk, regs):
        assert regs[-1] in (255, SECP_TOP - 1, P256_TOP - 1, ED_TOP - 1) and regs[0] % 2 == 0, (name, regs[0], regs[-1])
        parts.append("#define FEC_%s_ASM \\\n" % name + blk.text().replace("\n", " \\\n") + "\n")
        parts.append("#define FEC_%s_CLOBBERS \"vcc\", " % name + clobbers(regs) + "\n")
        parts.append("// FEC_%s_ASM: %d instructions, fixed block v[%d:%d]\n" % (name, len(blk.lines), regs[0], regs[-1]))
        report.append("%s %d" % (name, len(blk.lines)))

    # the secp256k1 blocks end at v167 so that the ladder kernel fits 168 VGPRs (3 waves per SIMD)
    add("SECP_MUL", *secp_mul(SECP_TOP - 36))
    add("SECP_SQR", *secp_sqr(SECP_TOP - 34))
    add("SECP_MUL3", *secp_mul_small(SECP_TOP - 22, 3))
    add("SECP_MUL8", *secp_mul_small(SECP_TOP - 22, 8))
    add("P256_MUL", *p256_mul(P256_TOP - 32))
    add("P256_SQR", *p256_sqr(P256_TOP - 46))
    add("P256_MUL3", *p256_mul_small(P256_TOP - 24, 3))
    add("P256_MUL8", *p256_mul_small(P256_TOP - 24, 8))
    add("ED_MUL", *ed_mul(ED_TOP - 32))
    add("ED_SQR", *ed_sqr(ED_TOP - 46))
    with open(OUT, "w") as f:
        f.write("\n".join(parts))
    print("wrote %s (%s)" % (OUT, ", ".join(report)))


if __name__ == "__main__":
    main()
