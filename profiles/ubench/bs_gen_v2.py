#!/usr/bin/env python3
"""Generator of the bit-sliced ntHash ring filter for gfx950 (hash_bs_k32.inc) + a numpy model of the generated code.

What the generated code computes (reference: `indexlr`'s ntHash as restated in SURVEY.md App. A; the same candidate filter
as k_hash_sparse in sketch.hip): for every base position p of an assembly, whether the 32-mer starting at p MAY have
canonical hash fwd + rev < tau -- decided on the top 31-bit rings F, R of the two strand hashes (srol / sror rotate bits
33..63 of a hash within themselves), as a superset: the sum is formed on the top B_PLANES bits without the carry from below.

Why bit-sliced.  One ring update is `rotate by one, xor a term that depends on the outgoing and the incoming base`.  With
one k-mer per lane the rotations and the compare are v_bfe / v_lshlrev / v_alignbit / v_cmp / v_addc -- instructions that
not only issue at half rate on gfx950 but slow the WHOLE instruction stream of the SIMD down to half rate
(profiles/ubench/README.md).  Here ONE LANE HOLDS 32 STRIPS: register j holds bit j of the ring for 32 k-mers (bit s = slot s),
so a rotation is a renaming of registers (free: the code is unrolled) and the xor of the table term is ONE v_bitop3_b32 per
ring bit for 32 k-mers: state ^= A_j(out) ^ B_j(in), where A_j, B_j are boolean functions of one base's two bits, i.e. one of
seven masks made of the base's two bit planes (5 VALU per base).  The sum test is a ripple adder over the top planes and a
bit-sliced compare.  The stream consists of full-rate ("fast class") instructions only: v_xor/and/or/mov, v_bitop3_b32,
v_add_u32, v_lshrrev_b32 -- no left shift, no compare, no add-with-carry, no popcount, no permute.

Geometry (k = 32).  A chunk = 65 536 consecutive base positions = 64 lanes x 32 strips x 32 positions.  The bases come
TRANSPOSED (mxg bs layout, built once per assembly by k_bs_transpose in sketch_bs.hip):
    T[chunk][t / 2][lane][2 (t & 1) + beta]  (u32)   bit s = bit beta of the base at chunk * 65536 + (32 lane + s) * 32 + t
    Q[chunk][lane][beta]                     (u32)   bit t = bit beta of the base at chunk * 65536 + (32 lane - 1) * 32 + t
(16 bytes per lane and pair of steps: one global_load_dwordx4 of 1 KB per wave; with 4-byte loads of 256 B per wave the
filter stayed at 3.0 TB/s of traffic, a third below what its instruction stream allows)
Slot s of lane L rolls the k-mers of strip 32 L + s - 1 (the strip BEFORE the one whose bits sit at position s): the
incoming base of step t is then W[t] itself and the outgoing base is W[t] shifted up by one slot (v_add_u32 W, W) with
the previous lane's last strip coming in at the bottom (bit t of Q: a running v_lshrrev_b32 and one v_bitop3_b32).
(v_addc_co_u32 with the carry in an SGPR pair would do it in one instruction -- and is slow class: the body ran at 3.7
cycles per instruction with 126 of them among 5 860, at 2.25 without.)  The 32
bases of a strip are at once the warm-up input of its own slot, the outgoing bases of that slot and the incoming bases of
the slot below.  Per chunk: 32 warm-up steps (no outgoing base, no test), 32 productive steps (test, then roll); the 32
result words of a lane (word t: bit s = slot s) are transposed in registers (32 x 32 bits, left shifts as chains of
v_add_u32) into position order and go out as a plain bitmap, one bit per base position:
    OUT[p / 32] bit p % 32 = the 32-mer at position p passed      (word index chunk * 2048 + 32 lane + s - 1 for slot s)
(slots 1..31 of a lane are 124 contiguous bytes on a 128-byte boundary, slot 0 is the word in front of them)
The words of the next chunk are requested into the registers of the current one as soon as a step has read them.

The same instruction list is (1) printed as gfx950 assembly for one inline-asm block with fixed registers (the chunk loop
included) and (2) executed by a numpy model (class VM): tests/test_bs_gen_cpu.py runs the model against the direct ntHash
formula, so renaming, truth tables and address arithmetic are checked without a GPU.
"""
import argparse
import sys

import numpy as np

SEED = [0x3c8bfbb395c60474, 0x3193c18562a02b4c, 0x20323ed082572324, 0x295549f54be24456]  # A C G T
RING = 31
M31 = (1 << 31) - 1
B_PLANES = 14
CHUNK = 65536


def rotl31(x, n):
    n %= 31
    return ((x << n) | (x >> (31 - n))) & M31 if n else x


def top31(x):
    return x >> 33


def plane_funcs(k):
    """Per ring bit j: truth vectors (value for base code 0..3) of the four table terms.
    forward:  F' = rotl31(F) ^ rotl31^k(top SEED[out]) ^ top SEED[in]
    reverse:  R' = rotr31(R ^ top SEED[3-out] ^ rotl31^k(top SEED[3-in]))"""
    fo, fi, ro, ri = [], [], [], []
    for j in range(RING):
        fo.append(tuple((rotl31(top31(SEED[c]), k) >> j) & 1 for c in range(4)))
        fi.append(tuple((top31(SEED[c]) >> j) & 1 for c in range(4)))
        ro.append(tuple((top31(SEED[3 - c]) >> j) & 1 for c in range(4)))
        ri.append(tuple((rotl31(top31(SEED[3 - c]), k) >> j) & 1 for c in range(4)))
    return fo, fi, ro, ri


# a boolean function of a base code c = b0 + 2 b1, normalised to g(0) = 0: which of the seven masks it is
MASK_OF = {(0, 0, 0): None, (1, 0, 0): 'c1', (0, 1, 0): 'c2', (0, 0, 1): 'a', (1, 1, 0): 'x', (1, 0, 1): 'b0',
           (0, 1, 1): 'b1', (1, 1, 1): 'o'}


def norm(tv):
    """truth vector -> (mask name or None, constant)"""
    c = tv[0]
    g = tuple(v ^ c for v in tv[1:])
    return MASK_OF[g], c


def fmt_src(x):
    if isinstance(x, int):
        return str(x) if -16 <= x <= 64 else hex(x & 0xFFFFFFFF)
    return x


def sp(i):
    """SGPR pair i, i+1"""
    return f"s[{i}:{i + 1}]"


def to_asm(ins):
    """one instruction -> assembly text"""
    op = ins[0]
    a = [fmt_src(x) for x in ins[1:]]
    if op in ('xor', 'and', 'or'):
        return f"v_{op}_b32 {a[0]}, {a[1]}, {a[2]}"
    if op == 'mov':
        return f"v_mov_b32 {a[0]}, {a[1]}"
    if op == 'bitop3':
        return f"v_bitop3_b32 {a[0]}, {a[1]}, {a[2]}, {a[3]} bitop3:{hex(ins[5])}"
    if op == 'add':
        return f"v_add_u32 {a[0]}, {a[1]}, {a[2]}"
    if op == 'lshr':
        return f"v_lshrrev_b32 {a[0]}, {a[2]}, {a[1]}"
    if op == 'gload4':  # first dst register, base SGPR pair, byte offset; the lane's offset (lane * 16) is operand %[voff]
        d = int(ins[1][1:])
        return f"global_load_dwordx4 v[{d}:{d + 3}], %[voff], {sp(ins[2])} offset:{ins[3]}"
    if op == 'gload2':  # (lane * 8: %[voff8])
        d = int(ins[1][1:])
        return f"global_load_dwordx2 v[{d}:{d + 1}], %[voff8], {sp(ins[2])} offset:{ins[3]}"
    if op in ('gstore4', 'gstore3', 'gstore1'):  # first data register, base SGPR pair, byte offset; lane * 128: %[voff128]
        d = int(ins[1][1:])
        n = int(op[-1])
        regs = f"v[{d}:{d + n - 1}]" if n > 1 else f"v{d}"
        suffix = {4: 'dwordx4', 3: 'dwordx3', 1: 'dword'}[n]
        return f"global_store_{suffix} %[voff128], {regs}, {sp(ins[2])} offset:{ins[3]}"
    if op == 'waitcnt':
        return f"s_waitcnt {ins[1]}"
    if op == 'comment':
        return f"; {ins[1]}"
    raise ValueError(op)


# ---------------------------------------------------------------------------------------------------------------
# register map (physical VGPRs / SGPRs the generated block owns: the inline-asm clobber list)
#
# VGPR banks.  Measured on MI355X (profiles/ubench): v_bitop3_b32 with three VGPR sources issues every 2.3-2.4 cycles when
# the three registers lie in different banks (register index mod 4) and every 4.5 cycles when two share a bank; VOP2 does
# not care.  The three-source instructions are the rolls (state, mask of the outgoing base, mask of the incoming base) and
# the adder (forward plane, reverse plane, carry):
#     bank 0: forward ring planes FP[0..30], le     bank 1: reverse ring planes RP[0..30], ones
#     bank 2: masks of the outgoing base, carry     bank 3: masks of the incoming base, sum
# W (the chunk's 64 transposed words, 4 consecutive registers per pair of steps) is only read by v_add / v_mov.
# The block stays below 256 registers = two waves per SIMD, which is what the grid is sized for anyway (an odd number of
# waves per SIMD issues slower than an even one).
# ---------------------------------------------------------------------------------------------------------------
B0 = 8                     # first VGPR of the block (the compiler keeps v0..v7)
W0 = B0                    # 64 registers
G0 = W0 + 64               # 31 groups of 4: FP[i], RP[i], bank 2, bank 3
X0 = G0 + 124              # the rest
VEND = X0 + 52


def grp(i, q):
    return f"v{G0 + 4 * i + q}"


def bank(reg):
    return int(reg[1:]) % 4


# SGPRs of the block
S0 = 36
S_C = S0            # chunk index (pair: high word 0)
S_N = S0 + 2        # one past the last chunk
S_STRIDE = S0 + 3
S_TT = S0 + 4
S_T = S0 + 6        # T base (pair), S_P = Q base (pair), S_O = OUT base (pair)
S_P = S0 + 8
S_O = S0 + 10
S_TN = S0 + 12      # 4 pairs: next chunk's T words + k * 4096
S_OC = S0 + 20      # 2 pairs: this chunk's OUT words - 4 bytes, this chunk's OUT words
S_QN = S0 + 24      # pair: the next chunk's Q words
S_TMP = S0 + 26     # pair
S_CM = S0 + 28      # B_PLANES compare masks
S_M16, S_M8, S_M4, S_M2, S_M1 = (S0 + 28 + B_PLANES + i for i in range(5))  # the transpose's select masks
SEND = S_M1 + 1


class Gen:
    def __init__(self, k=32, b_planes=B_PLANES):
        assert k == 32, "strips of 32 k-mers: k = 32 only (other k: k_hash_sparse)"
        self.k = k
        self.b = b_planes
        self.ins = []
        self.fo, self.fi, self.ro, self.ri = plane_funcs(k)
        self.FP = [grp(i, 0) for i in range(31)]
        self.RP = [grp(i, 1) for i in range(31)]
        self.W = {(t, be): f"v{W0 + 2 * t + be}" for t in range(32) for be in (0, 1)}
        self.A = [grp(g, 2) for g in range(7)]   # o0 o1 x o a c1 c2 of the outgoing base
        self.B = [grp(g, 3) for g in range(7)]   # in0 in1 ... of the incoming base
        self.cy, self.s = grp(7, 2), grp(7, 3)
        assert X0 % 4 == 0
        # the steps' results, then their transpose: M[1..] go out as dwordx4 (register tuples must start on an even register)
        self.M = [f"v{X0 + 1 + i}" for i in range(32)]
        self.Qn = [f"v{X0 + 34}", f"v{X0 + 35}"]   # the next chunk's Q words (a dwordx2), the chunk's, their running shifts
        self.Q = [f"v{X0 + 36}", f"v{X0 + 37}"]
        self.Qr = [f"v{X0 + 40}", f"v{X0 + 41}"]   # banks 0, 1 (read with a bank 2 register by one v_bitop3_b32)
        self.le, self.ones = f"v{X0 + 44}", f"v{X0 + 45}"
        self.TT = [f"v{X0 + 48 + i}" for i in range(4)]  # temporaries of the transpose, one per bank
        self.neg = {}

    def e(self, *t):
        self.ins.append(t)

    @staticmethod
    def tt3(fn, na=0, nb=0, nc=0):
        """truth table of fn(a ^ na, b ^ nb, c ^ nc), index a*4 + b*2 + c"""
        tt = 0
        for a in (0, 1):
            for b in (0, 1):
                for c in (0, 1):
                    if fn(a ^ na, b ^ nb, c ^ nc):
                        tt |= 1 << (a * 4 + b * 2 + c)
        return tt

    def masks(self, regs):
        """the five derived masks of a base whose two bit planes are regs[0], regs[1] -> dict name -> register"""
        e = self.e
        b0, b1 = regs[0], regs[1]
        m = {'b0': b0, 'b1': b1, 'x': regs[2], 'o': regs[3], 'a': regs[4], 'c1': regs[5], 'c2': regs[6]}
        e('xor', m['x'], b0, b1)
        e('or', m['o'], b0, b1)
        e('and', m['a'], b0, b1)
        e('xor', m['c1'], b0, m['a'])
        e('xor', m['c2'], b1, m['a'])
        return m

    def plane_update(self, dst, tvA, mA, tvB, mB, first=False):
        """dst ^= A(out) ^ B(in)   (tvA None: no outgoing base; first: dst is zero before).  The constant terms are not
        computed: self.neg[dst] says whether the register holds the complement of the logical plane; whoever reads the
        plane folds the flag into its truth table."""
        e = self.e
        sa, ca = norm(tvA) if tvA is not None else (None, 0)
        sb, cb = norm(tvB)
        c = ca ^ cb
        ra = mA[sa] if sa else None
        rb = mB[sb] if sb else None
        if first:
            assert ra is None
            e('mov', dst, rb if rb else 0)
            self.neg[dst] = c
            return
        self.neg[dst] ^= c
        if ra is None and rb is None:
            pass
        elif ra is None or rb is None:
            e('xor', dst, dst, ra or rb)
        else:
            e('bitop3', dst, dst, ra, rb, 0x96)

    def o_stream(self, t):
        """A[0], A[1] <- the two bit planes of the outgoing base of step t: W[t] moved up by one slot, bit t of Q at the bottom"""
        e = self.e
        tt = self.tt3(lambda a, b2, c: a | (b2 & c))
        for be in (0, 1):
            e('add', self.A[be], self.W[(t, be)], self.W[(t, be)])
            e('bitop3', self.A[be], self.A[be], self.Qr[be], 1, tt)
            if t < 31:
                e('lshr', self.Qr[be], self.Qr[be], 1)

    def chunk(self):
        """the body of the chunk loop: W and Qn hold the chunk's words (their loads may still be in flight)"""
        e = self.e
        b = self.b
        FP, RP, W = self.FP, self.RP, self.W
        A, B = self.A, self.B
        # ---- warm-up: steps n = 0..31: the slot's own strip, i.e. the o-stream (W shifted up by one slot)
        for n in range(32):
            # W[n] has arrived when at most the loads issued after it are outstanding (loads return in order; stores in
            # between only make the wait stricter)
            if n % 2 == 0:
                e('waitcnt', f'vmcnt({15 - n // 2})')
            if n == 0:  # (the Q words were requested before all of W)
                for be in (0, 1):
                    e('mov', self.Q[be], self.Qn[be])
                    e('mov', self.Qr[be], self.Qn[be])
            self.o_stream(n)
            mB = self.masks(A)
            for r in range(31):
                jf = (r + n + 1) % 31
                self.plane_update(FP[r], None, None, self.fi[jf], mB, first=(n == 0))
            for r in range(31):
                jr = (r - n) % 31
                self.plane_update(RP[r], None, None, self.ri[jr], mB, first=(n == 0))
        # ---- productive steps t = 0..31 (n = 32 + t): test the k-mer, then roll
        s, cy, le, ones = self.s, self.cy, self.le, self.ones
        for t in range(32):
            n = 32 + t
            if t == 0:
                for be in (0, 1):
                    e('mov', self.Qr[be], self.Q[be])
                e('gload2', self.Qn[0], S_QN, 0)  # the next chunk's Q words (before its W words: they wait for less)
            if t < 31:
                # masks of the outgoing base (bank 2) and of the incoming base (bank 3), made before the test
                self.o_stream(t)
                e('mov', B[0], W[(t, 0)])
                e('mov', B[1], W[(t, 1)])
            if t % 2 == 1:  # W[t - 1], W[t] are dead: the next chunk's words can come
                e('gload4', W[(t - 1, 0)], S_TN + 2 * ((t // 2) // 4), ((t // 2) % 4) * 1024)
            if t < 31:
                mA = self.masks(A)
                mB = self.masks(B)
            # test: top b planes of F + R.  s (sum plane), cy (carry), le, ones hold true values; the planes' complement
            # flags go into the truth tables.  Banks: f 0, r 1, cy 2, s 3, le 0, ones 1.
            jlo = 31 - b
            e('mov', le, -1)
            for j in range(jlo, 31):
                f, r = FP[(j - n) % 31], RP[(j + n) % 31]
                nf, nr = self.neg[f], self.neg[r]
                if j == jlo:  # (third source: ignored by the truth table)
                    e('bitop3', s, f, r, cy, self.tt3(lambda a, b2, c: a ^ b2, nf, nr, 0))
                    e('bitop3', cy, f, r, cy, self.tt3(lambda a, b2, c: a & b2, nf, nr, 0))
                else:
                    e('bitop3', s, f, r, cy, self.tt3(lambda a, b2, c: a ^ b2 ^ c, nf, nr, 0))
                    if j < 30:
                        e('bitop3', cy, f, r, cy, self.tt3(lambda a, b2, c: (a & b2) | (a & c) | (b2 & c), nf, nr, 0))
                e('bitop3', le, s, le, f"s{S_CM + (j - jlo)}", 0x8E)
                idx = j - jlo  # all-ones over planes jlo+1 .. 30
                if idx == 1:
                    e('mov', ones, s)
                elif idx >= 2:
                    e('and', ones, ones, s)
            e('or', self.M[t], le, ones)
            if t < 31:
                for r in range(31):
                    jf = (r + n + 1) % 31
                    self.plane_update(FP[r], self.fo[jf], mA, self.fi[jf], mB)
                for r in range(31):
                    jr = (r - n) % 31
                    self.plane_update(RP[r], self.ro[jr], mA, self.ri[jr], mB)
        self.transpose_out()
        # M[s] = the 32 positions of strip 32 lane + s - 1: slots 1..31 at bytes 0..123 of the lane's 128, slot 0 in front
        e('gstore1', self.M[0], S_OC, 0)
        for q in range(7):
            e('gstore4', self.M[4 * q + 1], S_OC + 2, 16 * q)
        e('gstore3', self.M[29], S_OC + 2, 112)
        self.check_banks()
        return self.ins

    def transpose_out(self):
        """M[t] bit s -> M[s] bit t, in place.  Stage j pairs rows k, k + j: new_k = (k & m) | ((k+j << j) & ~m),
        new_k+j = ((k >> j) & m) | (k+j & ~m), m = the bits whose index has bit j clear (an SGPR: v_bitop3_b32 with one SGPR source
        stays fast class); the left shift is j times v_add_u32 x, x (v_lshlrev_b32 is slow class)."""
        e = self.e
        sel = self.tt3(lambda a, b2, c: a if c else b2)
        for j, sm in ((16, S_M16), (8, S_M8), (4, S_M4), (2, S_M2), (1, S_M1)):
            for k in range(32):
                if k & j:
                    continue
                a, bq = self.M[k], self.M[k + j]
                t0 = self.TT[(bank(a) + 1) % 4]
                t1 = self.TT[(bank(bq) + 1) % 4]
                if t1 == t0:
                    t1 = self.TT[(bank(bq) + 2) % 4]
                e('add', t0, bq, bq)
                for _ in range(j - 1):
                    e('add', t0, t0, t0)
                e('lshr', t1, a, j)
                e('bitop3', a, a, t0, f"s{sm}", sel)
                e('bitop3', bq, t1, bq, f"s{sm}", sel)

    def check_banks(self):
        """no v_bitop3_b32 may read two VGPRs of one bank"""
        bad = 0
        for ins in self.ins:
            if ins[0] == 'bitop3':
                srcs = set(x for x in ins[2:5] if isinstance(x, str) and x.startswith('v'))
                banks = [bank(x) for x in srcs]
                if len(banks) != len(set(banks)):
                    bad += 1
        assert bad == 0, f"{bad} v_bitop3_b32 with a register bank conflict"

    # ---- the whole block: prologue, chunk loop ---------------------------------------------------------------
    def address_setup(self, lines):
        """SALU: bases of this chunk's OUT / P words and of the NEXT chunk's T / P words (the last chunk of a wave asks for
        its own words again)"""
        L = lines.append
        t0, t1 = S_TMP, S_TMP + 1
        L(f"s_add_u32 s{t0}, s{S_C}, s{S_STRIDE}")
        L(f"s_cmp_lt_u32 s{t0}, s{S_N}")
        L(f"s_cselect_b32 s{t0}, s{t0}, s{S_C}")
        L(f"s_mov_b32 s{t1}, 0")
        for k in range(4):
            d = S_TN + 2 * k
            L(f"s_lshl_b64 {sp(d)}, {sp(t0)}, 14")
            L(f"s_add_u32 s{d}, s{d}, s{S_T}")
            L(f"s_addc_u32 s{d + 1}, s{d + 1}, s{S_T + 1}")
            if k:
                L(f"s_add_u32 s{d}, s{d}, {hex(4096 * k)}")
                L(f"s_addc_u32 s{d + 1}, s{d + 1}, 0")
        L(f"s_lshl_b64 {sp(S_QN)}, {sp(t0)}, 9")
        L(f"s_add_u32 s{S_QN}, s{S_QN}, s{S_P}")
        L(f"s_addc_u32 s{S_QN + 1}, s{S_QN + 1}, s{S_P + 1}")
        d = S_OC + 2  # the chunk's 2048 words; S_OC: the same minus one word (slot 0 of a lane = the word in front of its 31)
        L(f"s_lshl_b64 {sp(d)}, {sp(S_C)}, 13")
        L(f"s_add_u32 s{d}, s{d}, s{S_O}")
        L(f"s_addc_u32 s{d + 1}, s{d + 1}, s{S_O + 1}")
        L(f"s_add_u32 s{S_OC}, s{d}, -4")
        L(f"s_addc_u32 s{S_OC + 1}, s{d + 1}, -1")

    def asm(self):
        """the inline-asm text.  Operands: %[t] %[p] %[o] (SGPR pairs: T, Q, OUT bases), %[c0] first chunk of the wave,
        %[n] one past the last chunk, %[stride] chunks between a wave's chunks, %[tt] threshold, VGPRs %[voff] = lane * 16, %[voff8] = lane * 8, %[voff128] = lane * 128"""
        body = self.chunk()
        L = []
        A = L.append
        A(f"s_mov_b32 s{S_C}, %[c0]")
        A(f"s_mov_b32 s{S_C + 1}, 0")
        A(f"s_mov_b32 s{S_N}, %[n]")
        A(f"s_mov_b32 s{S_STRIDE}, %[stride]")
        A(f"s_mov_b32 s{S_TT}, %[tt]")
        A(f"s_mov_b64 {sp(S_T)}, %[t]")
        A(f"s_mov_b64 {sp(S_P)}, %[p]")
        A(f"s_mov_b64 {sp(S_O)}, %[o]")
        for i in range(self.b):  # compare masks: Cm_i = all ones iff bit i of the threshold is set
            A(f"s_bfe_u32 s{S_TMP}, s{S_TT}, {hex((1 << 16) | i)}")
            A(f"s_sub_u32 s{S_CM + i}, 0, s{S_TMP}")
        for sm, val in ((S_M16, 0x0000FFFF), (S_M8, 0x00FF00FF), (S_M4, 0x0F0F0F0F), (S_M2, 0x33333333), (S_M1, 0x55555555)):
            A(f"s_mov_b32 s{sm}, {hex(val)}")
        A(f"s_cmp_ge_u32 s{S_C}, s{S_N}")
        A("s_cbranch_scc1 L_bs_end_%=")
        # prologue: the first chunk's words and P entries
        A(f"s_lshl_b64 {sp(S_TMP)}, {sp(S_C)}, 14")
        A(f"s_add_u32 s{S_TMP}, s{S_TMP}, s{S_T}")
        A(f"s_addc_u32 s{S_TMP + 1}, s{S_TMP + 1}, s{S_T + 1}")
        for k in range(4):
            d = S_TN + 2 * k
            A(f"s_add_u32 s{d}, s{S_TMP}, {hex(4096 * k)}")
            A(f"s_addc_u32 s{d + 1}, s{S_TMP + 1}, 0")
        A(f"s_lshl_b64 {sp(S_QN)}, {sp(S_C)}, 9")
        A(f"s_add_u32 s{S_QN}, s{S_QN}, s{S_P}")
        A(f"s_addc_u32 s{S_QN + 1}, s{S_QN + 1}, s{S_P + 1}")
        A(to_asm(('gload2', self.Qn[0], S_QN, 0)))
        for t2 in range(16):
            A(to_asm(('gload4', self.W[(2 * t2, 0)], S_TN + 2 * (t2 // 4), (t2 % 4) * 1024)))
        A("L_bs_loop_%=:")
        self.address_setup(L)
        for ins in body:
            if ins[0] != 'comment':
                A(to_asm(ins))
        A(f"s_add_u32 s{S_C}, s{S_C}, s{S_STRIDE}")
        A(f"s_cmp_lt_u32 s{S_C}, s{S_N}")
        A("s_cbranch_scc1 L_bs_loop_%=")
        A("s_waitcnt vmcnt(0)")  # (the requests for a chunk that does not follow)
        A("L_bs_end_%=:")
        return L

    def clobbers(self):
        return [f"v{i}" for i in range(B0, VEND)] + [f"s{i}" for i in range(S0, SEND)] + ["vcc", "scc", "memory"]


# ---------------------------------------------------------------------------------------------------------------
# numpy model of the chunk body (64 lanes)
# ---------------------------------------------------------------------------------------------------------------
class VM:
    def __init__(self, T, P, tt, c, c_next):
        """T: uint32 [n_chunks][16][64][4], P (= Q): uint32 [n_chunks][64][2]; runs chunk c (whose words are preloaded into W, as the
        prologue / the previous iteration does) and collects OUT[32][64] and the words requested for chunk c_next"""
        self.T, self.P = T, P
        self.c, self.cn = c, c_next
        self.vr = {}
        self.sr = {S_CM + i: (0xFFFFFFFF if (tt >> i) & 1 else 0) for i in range(B_PLANES)}
        self.sr.update({S_M16: 0x0000FFFF, S_M8: 0x00FF00FF, S_M4: 0x0F0F0F0F, S_M2: 0x33333333, S_M1: 0x55555555})
        self.out = np.zeros(2048 + 1, dtype=np.uint32)  # word index + 1 (slot 0 of lane 0 lies in front of the chunk)

    def V(self, x):
        if isinstance(x, int):
            return np.full(64, x & 0xFFFFFFFF, dtype=np.uint32)
        if x.startswith('v'):
            if x not in self.vr:  # (read before written: only as an operand that the truth table ignores -- junk on purpose)
                self.vr[x] = np.full(64, 0xDEADBEEF, dtype=np.uint32)
            return self.vr[x]
        if x.startswith('s'):
            return np.full(64, self.sr[int(x[1:])] & 0xFFFFFFFF, dtype=np.uint32)
        raise KeyError(x)

    def run(self, g):
        U = np.uint32
        for (t, be), reg in g.W.items():
            self.vr[reg] = self.T[self.c, t // 2, :, 2 * (t & 1) + be].copy()
        for be in (0, 1):
            self.vr[g.Qn[be]] = self.P[self.c, :, be].copy()
        pend = {}  # loads in flight: they land when the body ends (no instruction of this chunk may see them)
        for ins in g.ins:
            op = ins[0]
            if op == 'xor':
                self.vr[ins[1]] = self.V(ins[2]) ^ self.V(ins[3])
            elif op == 'and':
                self.vr[ins[1]] = self.V(ins[2]) & self.V(ins[3])
            elif op == 'or':
                self.vr[ins[1]] = self.V(ins[2]) | self.V(ins[3])
            elif op == 'mov':
                self.vr[ins[1]] = self.V(ins[2]).copy()
            elif op == 'bitop3':
                a, b, c, tt = self.V(ins[2]), self.V(ins[3]), self.V(ins[4]), ins[5]
                r = np.zeros(64, dtype=U)
                for idx in range(8):
                    if (tt >> idx) & 1:
                        r |= (a if idx & 4 else ~a) & (b if idx & 2 else ~b) & (c if idx & 1 else ~c)
                self.vr[ins[1]] = r
            elif op == 'add':
                self.vr[ins[1]] = (self.V(ins[2]) + self.V(ins[3])).astype(U)
            elif op == 'lshr':
                self.vr[ins[1]] = (self.V(ins[2]) >> U(ins[3])).astype(U)
            elif op == 'gload2':
                d = int(ins[1][1:])
                for j in range(2):
                    pend[f"v{d + j}"] = self.P[self.cn, :, j].copy()
            elif op == 'gload4':
                k = (ins[2] - S_TN) // 2
                t2 = (k * 4096 + ins[3]) // 1024
                d = int(ins[1][1:])
                for j in range(4):
                    pend[f"v{d + j}"] = self.T[self.cn, t2, :, j].copy()
            elif op in ('gstore4', 'gstore3', 'gstore1'):
                n = int(op[-1])
                base = -1 if ins[2] == S_OC else 0  # word offset of the SGPR pair relative to the chunk
                d = int(ins[1][1:])
                lanes = np.arange(64)
                for j in range(n):
                    self.out[1 + base + 32 * lanes + ins[3] // 4 + j] = self.vr[f"v{d + j}"]
            elif op in ('waitcnt', 'comment'):
                pass
            else:
                raise ValueError(op)
        for reg, val in pend.items():
            self.vr[reg] = val
        return self.out


# ---------------------------------------------------------------------------------------------------------------
# layouts and the direct formula (what the generated code must reproduce)
# ---------------------------------------------------------------------------------------------------------------
def transpose_layout(codes, n_chunks):
    """base codes (0..3) of n_chunks * 65536 positions -> T [n_chunks][16][64][4] u32, Q [n_chunks][64][2] u32"""
    codes = np.asarray(codes, dtype=np.uint8)
    assert len(codes) == n_chunks * CHUNK
    c4 = codes.reshape(n_chunks, 64, 32, 32)  # chunk, lane, slot, t
    T = np.zeros((n_chunks, 16, 64, 4), dtype=np.uint32)
    for be in (0, 1):
        bits = ((c4 >> be) & 1).astype(np.uint32)  # chunk, lane, slot, t
        for t in range(32):
            w = np.zeros((n_chunks, 64), dtype=np.uint32)
            for s in range(32):
                w |= bits[:, :, s, t] << np.uint32(s)
            T[:, t // 2, :, 2 * (t & 1) + be] = w
    Q = np.zeros((n_chunks, 64, 2), dtype=np.uint32)
    tsh = np.arange(32, dtype=np.uint32)
    for c in range(n_chunks):
        for lane in range(64):
            p0 = c * CHUNK + (32 * lane - 1) * 32
            if p0 < 0:
                continue
            seg = codes[p0:p0 + 32].astype(np.uint32)
            for be in (0, 1):
                Q[c, lane, be] = int((((seg >> be) & 1) << tsh).sum())
    return T, Q


def reference_bits(codes, k, tt, b_planes=B_PLANES):
    """codes: base codes (0..3) of n + k - 1 bases -> bool[n]: the ring test of the k-mer starting at each position.
    St = top b bits of (F + R) mod 2^31 without any carry from below; accepted iff St in [-2, tt] (mod 2^b)."""
    codes = np.asarray(codes, dtype=np.int64)
    n = len(codes) - k + 1
    tf = np.array([top31(SEED[c]) for c in range(4)], dtype=np.uint64)
    tr = np.array([top31(SEED[3 - c]) for c in range(4)], dtype=np.uint64)
    F = np.zeros(n, dtype=np.uint64)
    R = np.zeros(n, dtype=np.uint64)
    m = np.uint64(M31)

    def rot(x, r):
        r %= 31
        if r == 0:
            return x
        return ((x << np.uint64(r)) | (x >> np.uint64(31 - r))) & m

    for j in range(k):
        c = codes[j:j + n]
        F ^= rot(tf[c], k - 1 - j)
        R ^= rot(tr[c], j)
    low = 31 - b_planes
    St = ((F >> np.uint64(low)) + (R >> np.uint64(low))) & np.uint64((1 << b_planes) - 1)
    return (St <= np.uint64(tt)) | (St >= np.uint64((1 << b_planes) - 2))


def out_position(c, t, lane, s):
    """position of the k-mer behind bit t of slot s of a lane (= bit index in OUT)"""
    return c * CHUNK + (32 * lane + s - 1) * 32 + t


def emit_inc(path, k):
    g = Gen(k)
    lines = g.asm()
    n_valu = sum(1 for i in g.ins if i[0] in ('xor', 'and', 'or', 'mov', 'bitop3', 'add', 'lshr'))
    with open(path, 'w') as fh:
        fh.write(f"// GENERATED by gen/bs_gen.py (k = {k}, {B_PLANES} sum planes): the bit-sliced ring filter, chunk loop included.\n")
        fh.write(f"// {n_valu} VALU per chunk of 65 536 base positions per wave, all of them full-rate (see gen/bs_gen.py).  Do not edit.\n")
        fh.write("// operands: [t] [p] [o] SGPR pairs (T, Q, OUT bases), [c0] [n] [stride] [tt] SGPRs, [voff] VGPR = lane * 16, [voff8] = lane * 8, [voff128] = lane * 128\n")
        fh.write(f"#define HASH_BS_VGPR_END {VEND}\n")
        fh.write(f"#define HASH_BS_VALU_PER_CHUNK {n_valu}\n")
        fh.write(f"#define HASH_BS_PLANES {B_PLANES}\n")
        fh.write("#define HASH_BS_ASM \\\n")
        for ln in lines:
            fh.write(f'    "{ln}\\n" \\\n')
        fh.write("\n")
        fh.write("#define HASH_BS_CLOBBERS " + ", ".join(f'"{c}"' for c in g.clobbers()) + "\n")
    return len(lines), n_valu


if __name__ == '__main__':
    ap = argparse.ArgumentParser()
    ap.add_argument('-k', type=int, default=32)
    ap.add_argument('-o', default='hash_bs_k32.inc')
    a = ap.parse_args()
    n, nv = emit_inc(a.o, a.k)
    print(f"{a.o}: {n} lines, {nv} VALU per chunk", file=sys.stderr)
