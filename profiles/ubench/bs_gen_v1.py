#!/usr/bin/env python3
"""Generator of the bit-sliced ntHash ring kernel body for gfx950 (hash_bs_k<k>.inc) + a numpy model of the same code.

What the generated code computes (reference: `indexlr`'s ntHash as restated in SURVEY.md App. A; same candidate filter as
k_hash_sparse in sketch.hip): for every base position p of a 65 536-base chunk of 2-bit packed bases, whether the k-mer
starting at p MAY have canonical hash fwd+rev < tau -- decided on the top 31-bit rings F, R of the two strand hashes.

Why bit-sliced.  srol/sror rotate the top 31 bits of a hash within themselves and the table terms are XORed in, so a
ring update is `rotate, xor a term that depends on the outgoing and the incoming base`.  With one k-mer per lane the
rotations and the compare cost half-rate VALU instructions (v_bfe, v_lshrrev, v_alignbit, v_cmp, v_addc: 5 of the 10 per
base in k_hash_sparse).  Here ONE LANE HOLDS 32 STRIPS: register j holds bit j of the ring for 32 k-mers (bit s = strip s),
so a rotation is a renaming of registers (free: the code is unrolled) and the xor of the table term is ONE v_bitop3 per
ring bit for 32 k-mers: state ^= A_j(out) ^ B_j(in) where A_j, B_j are boolean functions of one base's two bits, i.e.
one of seven masks built from the two bit-planes of the base (5 VALU per base plane pair).  The sum test is a ripple
adder over the top B_PLANES planes and a bit-sliced compare with the threshold.  Everything is full-rate VALU.

Geometry.  A wave works on a chunk of 2048 strips x 32 k-mers = 65 536 consecutive base positions = 16 KB of packed
bases; lane L owns the 256 bytes [256 L, 256 L + 256) = its 32 strips (8 bytes each).  Strip s of a lane covers k-mers
[32 s, 32 s + 32) of the lane: bases 32 s .. 32 s + 62.  At k = 32 the k bases of a strip's first k-mer are the strip's
own 32 bases, the outgoing base of step t is base t of the strip and the incoming base is base t of the NEXT strip:
one transposed copy W[t] of the lane's bases serves all three roles (in-bits = W[t] shifted by one strip; the next
lane's first strip comes from 8 more bytes).  Per chunk: two 32 x 32 bit transposes in, 32 warm-up steps (no outgoing
base, no test), 32 productive steps (test, then roll), one transpose out: word s of a lane = the 32 test results of
strip s in position order, i.e. a bitmap with one bit per base position.  It is ANDed with the assembly's valid-k-mer
bitmap, counted and stored.

The same instruction list is (1) printed as gfx950 assembly for an inline-asm block with fixed registers and
(2) executed by a numpy model (class VM) -- tests/test_bs_gen_cpu.py runs the model against the direct ntHash formula,
so the register renaming, truth tables and transposes are checked without a GPU.
"""
import argparse
import sys

import numpy as np

SEED = [0x3c8bfbb395c60474, 0x3193c18562a02b4c, 0x20323ed082572324, 0x295549f54be24456]  # A C G T
RING = 31
M31 = (1 << 31) - 1


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


class Prog:
    """instruction list: tuples (op, dst, srcs...) over named registers 'v<n>' / 's<n>' / immediates"""

    def __init__(self):
        self.ins = []

    def emit(self, *t):
        self.ins.append(t)


def fmt_src(x):
    if isinstance(x, int):
        return str(x) if -16 <= x <= 64 else hex(x & 0xFFFFFFFF)
    return x


def to_asm(ins):
    """one instruction -> assembly text"""
    op = ins[0]
    a = [fmt_src(x) for x in ins[1:]]
    if op in ('xor', 'and', 'or', 'xnor'):
        return f"v_{op}_b32 {a[0]}, {a[1]}, {a[2]}"
    if op == 'add':
        return f"v_add_u32 {a[0]}, {a[1]}, {a[2]}"
    if op == 'not':
        return f"v_not_b32 {a[0]}, {a[1]}"
    if op == 'mov':
        return f"v_mov_b32 {a[0]}, {a[1]}"
    if op == 'bitop3':
        return f"v_bitop3_b32 {a[0]}, {a[1]}, {a[2]}, {a[3]} bitop3:{hex(ins[5])}"
    if op == 'lshl':
        return f"v_lshlrev_b32 {a[0]}, {a[2]}, {a[1]}"
    if op == 'lshr':
        return f"v_lshrrev_b32 {a[0]}, {a[2]}, {a[1]}"
    if op == 'alignbit':
        return f"v_alignbit_b32 {a[0]}, {a[1]}, {a[2]}, {a[3]}"
    if op == 'perm':
        return f"v_perm_b32 {a[0]}, {a[1]}, {a[2]}, {a[3]}"
    if op == 'bcnt':
        return f"v_bcnt_u32_b32 {a[0]}, {a[1]}, {a[2]}"
    if op == 's_mov':
        return f"s_mov_b32 {a[0]}, {a[1]}"
    if op == 's_bfe':  # dst, src, offset, width
        return f"s_bfe_u32 {a[0]}, {a[1]}, {hex((ins[4] << 16) | ins[3])}"
    if op == 's_sub':
        return f"s_sub_u32 {a[0]}, {a[1]}, {a[2]}"
    if op == 'gload4':  # dst first reg, addr operand, byte offset
        d = int(ins[1][1:])
        return f"global_load_dwordx4 v[{d}:{d + 3}], {ins[2][0]}, {ins[2][1]} offset:{ins[3]}"
    if op == 'gload2':
        d = int(ins[1][1:])
        return f"global_load_dwordx2 v[{d}:{d + 1}], {ins[2][0]}, {ins[2][1]} offset:{ins[3]}"
    if op == 'gstore2':
        d = int(ins[1][1:])
        return f"global_store_dwordx2 {ins[2][0]}, v[{d}:{d + 1}], {ins[2][1]} offset:{ins[3]}"
    if op == 'gstore4':
        d = int(ins[1][1:])
        return f"global_store_dwordx4 {ins[2][0]}, v[{d}:{d + 3}], {ins[2][1]} offset:{ins[3]}"
    if op == 'waitcnt':
        return f"s_waitcnt {ins[1]}"
    if op == 'nop':
        return f"s_nop {ins[1]}"
    if op == 'comment':
        return f"; {ins[1]}"
    raise ValueError(op)


# ---------------------------------------------------------------------------------------------------------------
# register map (physical VGPRs / SGPRs the generated block owns: the inline-asm clobber list)
#
# VGPR banks.  Measured on MI355X (profiles/ubench/issue_bench_mi355x.txt): v_bitop3_b32 with three VGPR sources issues
# every 2.4 cycles when the three registers lie in different banks (register index mod 4) and every 4.4-4.5 cycles when
# two of them share a bank; VOP2 (v_xor_b32 ...) does not care.  So the map is built around the banks:
#     bank 0: forward ring planes FP[0..30]          bank 1: reverse ring planes RP[0..30]
#     bank 2: word 0 of the lane's 32 strips,        bank 3: word 1 of the strips,
#             then W[t] for t < 16                            then W[t] for t >= 16
# (group i = registers B0 + 4 i .. + 3; dwordx2 loads fill the bank 2 / 3 pair of a group), and 8 more groups hold the
# masks of the outgoing base (the bank of its W registers), the masks of the incoming base (the other of banks 2 / 3),
# the adder's carry (bank 2) and sum (bank 3), and in banks 0 / 1 the step's result, temporaries and transpose masks.
# ---------------------------------------------------------------------------------------------------------------
B0 = 8                     # first VGPR of the block (the compiler keeps v0..v7 for its own values)
NGRP_X = 8                 # extra groups
VEND = B0 + 4 * (32 + NGRP_X)  # one past the last VGPR used: 168 = three waves per SIMD


def grp(i, q):
    return B0 + 4 * i + q


def ex(g, q):
    return B0 + 128 + 4 * g + q


def bank(reg):
    return int(reg[1:]) % 4


S0 = 36                    # first SGPR of the block
S_M4, S_M2, S_M1 = f"s{S0}", f"s{S0 + 1}", f"s{S0 + 2}"
S_P16L, S_P16H, S_P8L, S_P8H = (f"s{S0 + 3 + i}" for i in range(4))
S_TMP = f"s{S0 + 7}"
S_M16, S_M8 = f"s{S0 + 8}", f"s{S0 + 9}"
S_CM = S0 + 10             # B_PLANES compare masks
B_PLANES = 14


def v(i):
    return f"v{i}"


class Gen:
    def __init__(self, k=32, b_planes=B_PLANES, use_perm=False):
        assert k == 32, "strips of 32 k-mers: k = 32 only (other k: k_hash_sparse)"
        self.k = k
        self.b = b_planes
        self.use_perm = use_perm
        self.p = Prog()
        self.fo, self.fi, self.ro, self.ri = plane_funcs(k)
        self.FP = [v(grp(i, 0)) for i in range(31)]
        self.RP = [v(grp(i, 1)) for i in range(31)]
        self.RAW0 = [v(grp(i, 2)) for i in range(32)]
        self.RAW1 = [v(grp(i, 3)) for i in range(32)]
        self.XW = [v(ex(0, 0)), v(ex(0, 1))]     # the 8 bytes behind the lane's region (a dwordx2 pair)
        self.XR = [v(ex(1, 0)), v(ex(1, 1))]     # their running shifts
        self.le, self.ones = v(ex(2, 0)), v(ex(2, 1))
        self.tmp0 = [v(ex(g, 0)) for g in range(3, 8)] + [v(ex(2, 0))]       # bank 0 temporaries of the transposes (le is idle then)
        self.tmask_regs = [v(ex(g, 1)) for g in range(3, 8)]              # bank 1: the transposes' select masks
        self.E2 = [v(ex(g, 2)) for g in range(7)]
        self.E3 = [v(ex(g, 3)) for g in range(7)]
        self.cy, self.s = v(ex(7, 2)), v(ex(7, 3))
        self.send = S_CM + self.b
        self.neg = {}

    # ---- 32 x 32 bit transpose by renaming --------------------------------------------------------------
    def transpose_masks(self):
        """the select masks of the transposes in bank 1 VGPRs (v_bitop3_b32 with an SGPR source issues at half rate)"""
        e = self.p.emit
        self.tmask = {}
        for i, (j, sreg) in enumerate(((16, S_M16), (8, S_M8), (4, S_M4), (2, S_M2), (1, S_M1))):
            if self.use_perm and j >= 8:
                continue
            self.tmask[j] = self.tmask_regs[i]
            e('mov', self.tmask_regs[i], sreg)

    def transpose(self, rows, spare, home, final=None):
        """rows: 32 registers, row i = input word i.  Afterwards out[i] bit s = in[s] bit i; returns the registers holding
        out[0..31].  Rows move by renaming: a pair's new rows are written to free registers (`spare` at first, then the
        registers earlier pairs gave up; registers of `home` are preferred), shifted copies go to bank 0 temporaries and
        the masks sit in bank 1, so no v_bitop3_b32 reads two registers of one bank.  With `final` the last stage writes
        row i to final[i]; without it, rows that ended outside `home` are moved there."""
        e = self.p.emit
        rows = list(rows)
        pool = list(spare)
        home = set(home)
        tq = list(self.tmp0)
        sel = self.tt3(lambda a, b2, c: a if c else b2)

        def take():
            for r in pool:
                if r in home:
                    pool.remove(r)
                    return r
            return pool.pop(0)

        # A select right behind the half-rate shift it depends on stalls the wave (measured: the mixed stream took 1.4 x the
        # sum of its parts), so the shifts run two pairs ahead of the selects: three pairs of temporaries in flight.
        stages = [16, 8, 4, 2, 1]
        for si, j in enumerate(stages):
            last = si == len(stages) - 1
            pairs = [kk for kk in range(32) if not kk & j]
            sh, se = [], []
            for kk in pairs:
                a, bq = rows[kk], rows[kk + j]
                if self.use_perm and j >= 8:
                    sh.append([])
                else:
                    t0, t1 = tq[0], tq[1]
                    tq = tq[2:] + [t0, t1]
                    # (v_lshlrev_b32 issues at half rate, v_add_u32 at full rate)
                    sh.append([('add', t0, bq, bq) if j == 1 else ('lshl', t0, bq, j), ('lshr', t1, a, j)])
                se.append((kk, a, bq, sh[-1]))
            ahead = 2
            for q in range(min(ahead, len(pairs))):
                for ins in sh[q]:
                    e(*ins)
            for q, (kk, a, bq, shq) in enumerate(se):
                if q + ahead < len(pairs):
                    for ins in sh[q + ahead]:
                        e(*ins)
                if last and final:
                    d0, d1 = final[kk], final[kk + j]
                else:
                    d0, d1 = take(), take()
                if self.use_perm and j >= 8:
                    selL, selH = (S_P16L, S_P16H) if j == 16 else (S_P8L, S_P8H)
                    e('perm', d0, bq, a, selL)   # new row kk
                    e('perm', d1, bq, a, selH)   # new row kk + j
                else:
                    m = self.tmask[j]
                    e('bitop3', d0, a, shq[0][1], m, sel)
                    e('bitop3', d1, shq[1][1], bq, m, sel)
                pool += [a, bq]
                rows[kk], rows[kk + j] = d0, d1
        if not final:
            for i, r in enumerate(rows):
                if r not in home:
                    d = take()
                    assert d in home
                    e('mov', d, r)
                    rows[i] = d
        return rows

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

    # ---- the seven masks of a base given its two bit planes ---------------------------------------------
    def masks(self, b0, b1, regs):
        e = self.p.emit
        m = {'b0': b0, 'b1': b1, 'x': regs[0], 'o': regs[1], 'a': regs[2], 'c1': regs[3], 'c2': regs[4]}
        e('xor', m['x'], b0, b1)
        e('or', m['o'], b0, b1)
        e('and', m['a'], b0, b1)
        e('xor', m['c1'], b0, m['a'])
        e('xor', m['c2'], b1, m['a'])
        return m

    def plane_update(self, dst, tvA, mA, tvB, mB, first=False):
        """dst ^= A(out) ^ B(in)   (tvA None: no outgoing base; first: dst is zero before).  The constant terms are not
        computed: self.neg[dst] says whether the register holds the complement of the logical plane (v_xnor_b32 and
        v_not_b32 would cost an issue slot, v_xnor_b32 a half-rate one); whoever reads the plane folds the flag into its
        truth table."""
        e = self.p.emit
        sa, ca = norm(tvA) if tvA is not None else (None, 0)
        sb, cb = norm(tvB)
        c = ca ^ cb
        ra = mA[sa] if sa else None
        rb = mB[sb] if sb else None
        if first:
            assert ra is None
            if rb is None:
                e('mov', dst, 0)
            else:
                e('mov', dst, rb)
            self.neg[dst] = c
            return
        self.neg[dst] ^= c
        if ra is None and rb is None:
            pass
        elif ra is None or rb is None:
            e('xor', dst, dst, ra or rb)
        else:
            e('bitop3', dst, dst, ra, rb, 0x96)

    # ---- one chunk ------------------------------------------------------------------------------------
    def build(self, addr_in=('%5', '%1'), addr_kv=('%6', '%2'), addr_out=('%6', '%3'), s_tt='%4', v_cnt='%0'):
        """addresses: (VGPR byte offset of the lane, SGPR pair holding the chunk's base)"""
        e = self.p.emit
        b = self.b
        # constants
        e('s_mov', S_M16, 0x0000FFFF)
        e('s_mov', S_M8, 0x00FF00FF)
        e('s_mov', S_M4, 0x0F0F0F0F)
        e('s_mov', S_M2, 0x33333333)
        e('s_mov', S_M1, 0x55555555)
        e('s_mov', S_P16L, 0x05040100)
        e('s_mov', S_P16H, 0x07060302)
        e('s_mov', S_P8L, 0x06020400)
        e('s_mov', S_P8H, 0x07030501)
        for i in range(b):  # compare masks: Cm_i = all ones iff bit i of the threshold is set
            e('s_bfe', S_TMP, s_tt, i, 1)
            e('s_sub', f"s{S_CM + i}", 0, S_TMP)
        # loads: the lane's 32 strips (8 bytes each: word 0 -> bank 2, word 1 -> bank 3) + the 8 bytes behind them
        for i in range(32):
            e('gload2', self.RAW0[i], addr_in, 8 * i)
        e('gload2', self.XW[0], addr_in, 256)
        e('waitcnt', 'vmcnt(0)')
        e('comment', 'PHASE transpose_in')
        self.transpose_masks()
        rowsA = self.transpose(self.RAW0, self.E2 + [self.cy], self.RAW0)
        rowsB = self.transpose(self.RAW1, self.E3 + [self.s], self.RAW1)
        # W[t][beta]: bit s = bit beta of base t of strip s
        W = {}
        for t in range(16):
            for be in (0, 1):
                W[(t, be)] = rowsA[2 * t + be]
                W[(t + 16, be)] = rowsB[2 * t + be]
        FP, RP = self.FP, self.RP
        # ---- warm-up: steps n = 0..31, incoming base = base n of the strip itself
        e('comment', 'PHASE warmup')
        for n in range(32):
            mB = self.masks(W[(n, 0)], W[(n, 1)], (self.E2 if n < 16 else self.E3)[2:7])
            for r in range(31):
                jf = (r + n + 1) % 31
                self.plane_update(FP[r], None, None, self.fi[jf], mB, first=(n == 0))
            for r in range(31):
                jr = (r - n) % 31
                self.plane_update(RP[r], None, None, self.ri[jr], mB, first=(n == 0))
        # ---- productive steps t = 0..31 (n = 32 + t): test the k-mer, then roll
        e('comment', 'PHASE productive')
        XR, XW = self.XR, self.XW
        e('mov', XR[0], XW[0])
        e('lshr', XR[1], XW[0], 1)
        M = {}
        s, cy, le, ones = self.s, self.cy, self.le, self.ones
        for t in range(32):
            n = 32 + t
            if t == 16:
                e('mov', XR[0], XW[1])
                e('lshr', XR[1], XW[1], 1)
            mt = W[(t, 0)]  # the step's result replaces the step's first base plane after the roll
            if t < 31:
                # masks of the outgoing base in the bank of its W registers, those of the incoming base in the other one.
                # They are made BEFORE the test: the roll must not start right behind the half-rate v_alignbit_b32.
                EA, EB = (self.E2, self.E3) if t < 16 else (self.E3, self.E2)
                in0, in1 = EB[0], EB[1]
                # in-bits: W shifted by one strip, bit 31 from the next lane's first strip
                e('alignbit', in0, XR[0], W[(t, 0)], 1)
                e('alignbit', in1, XR[1], W[(t, 1)], 1)
                if t % 16 != 15:
                    e('lshr', XR[0], XR[0], 2)
                    e('lshr', XR[1], XR[1], 2)
                mA = self.masks(W[(t, 0)], W[(t, 1)], EA[0:5])
                mB = self.masks(in0, in1, EB[2:7])
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
            if t < 31:
                for r in range(31):
                    jf = (r + n + 1) % 31
                    self.plane_update(FP[r], self.fo[jf], mA, self.fi[jf], mB)
                for r in range(31):
                    jr = (r - n) % 31
                    self.plane_update(RP[r], self.ro[jr], mA, self.ri[jr], mB)
            e('or', mt, le, ones)
            M[t] = mt
        # ---- out: transpose the 32 step masks into position order (the state registers are free now): words 2 i, 2 i + 1
        # of the lane land in the (bank 0, bank 1) pair of group i, the valid-k-mer words in the pairs of groups 16..31
        e('comment', 'PHASE out')
        OUT, KV = [], []
        for i in range(16):
            OUT += [v(grp(i, 0)), v(grp(i, 1))]
            KV += [v(grp(16 + i, 0)), v(grp(16 + i, 1))]
        for i in range(16):
            e('gload2', KV[2 * i], addr_kv, 8 * i)
        self.transpose_masks()
        free_raw = [W[(t, 1)] for t in range(32)]
        self.transpose([M[t] for t in range(32)], free_raw, self.RAW0 + self.RAW1, final=OUT)
        e('waitcnt', 'vmcnt(0)')
        for i in range(32):
            e('and', OUT[i], OUT[i], KV[i])
        acc = self.tmp0[0:4]  # (four chains: v_bcnt_u32_b32 issues at half rate and a dependent one would wait for it)
        for q in range(4):
            e('mov', acc[q], 0)
        for i in range(32):
            e('bcnt', acc[i % 4], OUT[i], acc[i % 4])
        e('add', acc[0], acc[0], acc[1])
        e('add', acc[2], acc[2], acc[3])
        e('add', v_cnt, acc[0], acc[2])
        for i in range(16):
            e('gstore2', OUT[2 * i], addr_out, 8 * i)
        self.check_banks()
        return self.p

    def check_banks(self):
        """no v_bitop3_b32 may read two VGPRs of one bank"""
        bad = 0
        for ins in self.p.ins:
            if ins[0] == 'bitop3':
                srcs = [x for x in ins[2:5] if isinstance(x, str) and x.startswith('v')]
                banks = [bank(x) for x in set(srcs)]
                if len(banks) != len(set(banks)):
                    bad += 1
        assert bad == 0, f"{bad} v_bitop3_b32 with a register bank conflict"

    def clobbers(self):
        return [f"v{i}" for i in range(B0, VEND)] + [f"s{i}" for i in range(S0, self.send)] + ["vcc", "scc", "memory"]


# ---------------------------------------------------------------------------------------------------------------
# numpy model of the instruction list (64 lanes)
# ---------------------------------------------------------------------------------------------------------------
class VM:
    def __init__(self, mem_in, mem_kv, s_tt):
        """mem_in: uint32 array, the wave's packed words (lane L reads bytes [256 L, 256 L + 264));
        mem_kv: uint32 [64 x 32] valid-k-mer words of the chunk"""
        self.vr = {}
        self.sr = {}
        self.mem_in = np.asarray(mem_in, dtype=np.uint32)
        self.mem_kv = np.asarray(mem_kv, dtype=np.uint32).reshape(64, 32)
        self.out = np.zeros((64, 32), dtype=np.uint32)
        self.s_tt = int(s_tt)
        self.lane = np.arange(64)

    def V(self, x):
        if isinstance(x, int):
            return np.full(64, x & 0xFFFFFFFF, dtype=np.uint32)
        if x.startswith('v'):
            return self.vr[x]
        if x.startswith('s'):
            return np.full(64, self.sr[x] & 0xFFFFFFFF, dtype=np.uint32)
        if x == '%0':
            return self.vr.get('%0', np.zeros(64, dtype=np.uint32))
        raise KeyError(x)

    def S(self, x):
        if isinstance(x, int):
            return x & 0xFFFFFFFF
        if x == '%4':
            return self.s_tt
        return self.sr[x]

    def run(self, ins_list):
        U = np.uint32
        for ins in ins_list:
            op = ins[0]
            if op == 'xor':
                self.vr[ins[1]] = self.V(ins[2]) ^ self.V(ins[3])
            elif op == 'xnor':
                self.vr[ins[1]] = ~(self.V(ins[2]) ^ self.V(ins[3]))
            elif op == 'and':
                self.vr[ins[1]] = self.V(ins[2]) & self.V(ins[3])
            elif op == 'add':
                self.vr[ins[1]] = (self.V(ins[2]) + self.V(ins[3])).astype(U)
            elif op == 'or':
                self.vr[ins[1]] = self.V(ins[2]) | self.V(ins[3])
            elif op == 'not':
                self.vr[ins[1]] = ~self.V(ins[2])
            elif op == 'mov':
                self.vr[ins[1]] = self.V(ins[2]).copy()
            elif op == 'bitop3':
                a, b, c, tt = self.V(ins[2]), self.V(ins[3]), self.V(ins[4]), ins[5]
                r = np.zeros(64, dtype=U)
                for idx in range(8):
                    if (tt >> idx) & 1:
                        ta = a if idx & 4 else ~a
                        tb = b if idx & 2 else ~b
                        tc = c if idx & 1 else ~c
                        r |= ta & tb & tc
                self.vr[ins[1]] = r
            elif op == 'lshl':
                self.vr[ins[1]] = (self.V(ins[2]) << U(ins[3])).astype(U)
            elif op == 'lshr':
                self.vr[ins[1]] = (self.V(ins[2]) >> U(ins[3])).astype(U)
            elif op == 'alignbit':  # ({hi, lo} >> n) & 0xffffffff
                hi, lo, n = self.V(ins[2]).astype(np.uint64), self.V(ins[3]).astype(np.uint64), ins[4]
                self.vr[ins[1]] = (((hi << np.uint64(32)) | lo) >> np.uint64(n)).astype(U)
            elif op == 'perm':  # bytes of {S0, S1}: selector 0..3 = S1 bytes, 4..7 = S0 bytes
                s0, s1, sel = self.V(ins[2]), self.V(ins[3]), self.S(ins[4])
                comb = (s0.astype(np.uint64) << np.uint64(32)) | s1.astype(np.uint64)
                r = np.zeros(64, dtype=np.uint64)
                for byte in range(4):
                    sb = (sel >> (8 * byte)) & 0xFF
                    assert sb < 8
                    r |= ((comb >> np.uint64(8 * sb)) & np.uint64(0xFF)) << np.uint64(8 * byte)
                self.vr[ins[1]] = r.astype(U)
            elif op == 'bcnt':
                x = self.V(ins[2])
                pc = np.array([bin(int(t)).count('1') for t in x], dtype=U)
                self.vr[ins[1]] = pc + self.V(ins[3])
            elif op == 's_mov':
                self.sr[ins[1]] = ins[2] & 0xFFFFFFFF
            elif op == 's_bfe':
                self.sr[ins[1]] = (self.S(ins[2]) >> ins[3]) & ((1 << ins[4]) - 1)
            elif op == 's_sub':
                self.sr[ins[1]] = (self.S(ins[2]) - self.S(ins[3])) & 0xFFFFFFFF
            elif op in ('gload4', 'gload2'):
                nw = 4 if op == 'gload4' else 2
                d = int(ins[1][1:])
                if ins[2][1] == '%1':
                    base = self.lane * 64 + ins[3] // 4
                    for q in range(nw):
                        self.vr[f"v{d + q}"] = self.mem_in[base + q].astype(U)
                else:
                    for q in range(nw):
                        self.vr[f"v{d + q}"] = self.mem_kv[:, ins[3] // 4 + q].copy()
            elif op in ('gstore4', 'gstore2'):
                d = int(ins[1][1:])
                for q in range(4 if op == 'gstore4' else 2):
                    self.out[:, ins[3] // 4 + q] = self.vr[f"v{d + q}"]
            elif op in ('waitcnt', 'nop', 'comment'):
                pass
            else:
                raise ValueError(op)
        return self.out, self.vr.get('%0')


# ---------------------------------------------------------------------------------------------------------------
# direct formula (what the generated code must reproduce): per base position, the superset ring test
# ---------------------------------------------------------------------------------------------------------------
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


E64_OK = ('xor', 'and', 'or', 'add', 'lshl', 'lshr', 'mov', 'not', 'xnor')


def ins_size(ins):
    """encoded size in bytes"""
    op = ins[0]
    if op in ('bitop3', 'alignbit', 'perm', 'bcnt') or op.startswith(('gload', 'gstore')):
        return 8
    if op in E64_OK:
        lit = any(isinstance(x, int) and not -16 <= x <= 64 for x in ins[2:])
        return 8 if lit else 4
    if op == 's_mov':
        return 8 if not -16 <= ins[2] <= 64 else 4
    if op == 's_bfe':
        return 8
    if op in ('s_sub', 'waitcnt', 'nop'):
        return 4
    if op == 'comment':
        return 0
    raise ValueError(op)


def asm_lines(ins_list, align8=True):
    """assembly text; with align8 every 8-byte instruction starts on an 8-byte boundary (the block begins with .p2align 3):
    a 4-byte VALU instruction in front of an 8-byte one is encoded as VOP3 (_e64), anything else gets an s_nop behind it.
    (Measured on MI355X: the aligned stream is SLOWER, 595 vs 519 us per 3 Gbp -- the VOP3 encodings of the promoted
    instructions cost more than the straddling; the option remains for the record.)"""
    real = [i for i in ins_list if ins_size(i) > 0]
    lines = ['.p2align 3'] if align8 else []
    off = 0
    n_e64 = n_nop = 0
    for idx, ins in enumerate(real):
        sz = ins_size(ins)
        txt = to_asm(ins)
        if align8 and sz == 4 and off % 8 == 0:
            nxt = ins_size(real[idx + 1]) if idx + 1 < len(real) else 8
            if nxt == 8:
                if ins[0] in E64_OK:
                    head, rest = txt.split(' ', 1)
                    txt = f"{head}_e64 {rest}"
                    sz = 8
                    n_e64 += 1
                else:
                    lines.append(txt)
                    lines.append('s_nop 0')
                    off += 8
                    n_nop += 1
                    continue
        lines.append(txt)
        off += sz
    return lines, n_e64, n_nop


def emit_inc(path, k, use_perm=False, prefix="HASH_BS", align8=False):
    g = Gen(k, use_perm=use_perm)
    prog = g.build()
    lines, n_e64, n_nop = asm_lines(prog.ins, align8)
    n_valu = sum(1 for i in prog.ins if not i[0].startswith(('s_', 'g', 'wait', 'nop', 'comment')))
    with open(path, 'w') as fh:
        fh.write(f"// GENERATED by gen/bs_gen.py (k = {k}, {B_PLANES} sum planes): one chunk of the bit-sliced ring filter.\n")
        fh.write(f"// {len(lines)} instructions, {n_valu} VALU per 65 536 base positions per wave.  Do not edit.\n")
        fh.write("// operands: %0 = lane count (out, early clobber); SGPR pairs %1 = the chunk's packed bases (16 KB + 8 bytes are read),\n")
        fh.write("//           %2 = its valid-k-mer words (8 KB), %3 = its result words (8 KB); %4 = threshold (SGPR);\n")
        fh.write("//           VGPRs %5 = lane * 256, %6 = lane * 128\n")
        fh.write(f"#define {prefix}_VGPR_END {VEND}\n")
        fh.write(f"#define {prefix}_ASM \\\n")
        for ln in lines:
            fh.write(f'    "{ln}\\n" \\\n')
        fh.write("\n")
        fh.write(f"#define {prefix}_CLOBBERS " + ", ".join(f'"{c}"' for c in g.clobbers()) + "\n")
    return len(lines), n_valu


if __name__ == '__main__':
    ap = argparse.ArgumentParser()
    ap.add_argument('-k', type=int, default=32)
    ap.add_argument('-o', default='hash_bs_k32.inc')
    ap.add_argument('--perm', action='store_true', help="byte stages of the transposes with v_perm_b32 (bench variant)")
    ap.add_argument('--prefix', default='HASH_BS')
    ap.add_argument('--align8', action='store_true', help="8-byte encodings on 8-byte boundaries (bench variant: it is slower)")
    a = ap.parse_args()
    n, nv = emit_inc(a.o, a.k, a.perm, a.prefix, a.align8)
    print(f"{a.o}: {n} instructions, {nv} VALU", file=sys.stderr)
