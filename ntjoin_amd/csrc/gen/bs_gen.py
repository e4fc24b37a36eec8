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
# ---------------------------------------------------------------------------------------------------------------
V0 = 16                    # first VGPR of the block (the compiler keeps v0..v15 for its own values)
ST = V0                    # 64 state registers: FP = ST[0..30], RP = ST[31..61], ST[62..63] spare
RAW = ST + 64              # 64 registers: the lane's 256 bytes, then the transposed bases W, then the step masks
XW = RAW + 64              # 2: the 8 bytes after the lane's region (first strip of the next lane)
XR = XW + 2                # 2: running shifts of them (bit 0 = the next lane's bit of this step)
MA = XR + 2                # 5: masks of the outgoing base (x o a c1 c2; b0 b1 are W registers)
MB = MA + 5                # 7: in-bits b0 b1 and their masks
TT = MB + 7                # test temporaries: s0 s1 carry le ones
TR = TT + 5                # 2 rotating temporaries of the transposes
VEND = TR + 2              # one past the last VGPR used
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
    def __init__(self, k=32, b_planes=B_PLANES, use_perm=True):
        assert k == 32, "strips of 32 k-mers: k = 32 only (other k: k_hash_sparse)"
        self.k = k
        self.b = b_planes
        self.use_perm = use_perm
        self.p = Prog()
        self.fo, self.fi, self.ro, self.ri = plane_funcs(k)
        self.FP = [v(ST + i) for i in range(31)]
        self.RP = [v(ST + 31 + i) for i in range(31)]
        self.send = S_CM + self.b
        self.neg = {}

    # ---- 32 x 32 bit transpose by renaming --------------------------------------------------------------
    def transpose_masks(self):
        """the select masks of the transposes' shift stages in VGPRs that are idle while a transpose runs (v_bitop3_b32 with
        an SGPR source issues at half rate: profiles/ubench)"""
        e = self.p.emit
        self.tmask = {}
        for i, (j, sreg) in enumerate(((16, S_M16), (8, S_M8), (4, S_M4), (2, S_M2), (1, S_M1))):
            if self.use_perm and j >= 8:
                continue
            self.tmask[j] = v(MA + i)
            e('mov', v(MA + i), sreg)

    def transpose(self, rows, free, final=None):
        """rows: 32 register names, row i = input word i.  Afterwards out[i] bit s = in[s] bit i; returns the list of
        registers holding out[0..31] (a permutation of rows + free, or `final` when given: the last stage writes there).
        free: 2 scratch registers (which ones are scratch afterwards changes: returned as second value)."""
        e = self.p.emit
        rows = list(rows)
        free = list(free)
        stages = [16, 8, 4, 2, 1]
        for si, j in enumerate(stages):
            last = si == len(stages) - 1
            for kk in range(32):
                if kk & j:
                    continue
                a, bq = rows[kk], rows[kk + j]
                d0 = final[kk] if (last and final) else None
                d1 = final[kk + j] if (last and final) else None
                if self.use_perm and j >= 8:
                    selL, selH = (S_P16L, S_P16H) if j == 16 else (S_P8L, S_P8H)
                    t0 = d0 or free.pop()
                    e('perm', t0, bq, a, selL)   # new row kk
                    t1 = d1 or free.pop()
                    e('perm', t1, bq, a, selH)   # new row kk + j
                    if not d0:
                        free += [a, bq]
                    rows[kk], rows[kk + j] = t0, t1
                else:
                    m = self.tmask[j]
                    t0 = free.pop()
                    t1 = free.pop()
                    if j == 1:
                        e('add', t0, bq, bq)  # (v_lshlrev_b32 issues at half rate, v_add_u32 at full rate)
                    else:
                        e('lshl', t0, bq, j)
                    e('lshr', t1, a, j)
                    n0 = d0 or t0
                    n1 = d1 or t1
                    e('bitop3', n0, a, t0, m, self._sel_tt())
                    e('bitop3', n1, t1, bq, m, self._sel_tt())
                    if d0:
                        free += [t0, t1]
                    else:
                        free += [a, bq]
                    rows[kk], rows[kk + j] = n0, n1
        return rows, free

    @staticmethod
    def _sel_tt():
        # f(a, b, c) = c ? a : b  with index a*4 + b*2 + c
        tt = 0
        for a in (0, 1):
            for b in (0, 1):
                for c in (0, 1):
                    if (a if c else b):
                        tt |= 1 << (a * 4 + b * 2 + c)
        return tt

    # ---- the seven masks of a base given its two bit planes ---------------------------------------------
    def masks(self, b0, b1, base):
        e = self.p.emit
        m = {'b0': b0, 'b1': b1, 'x': v(base), 'o': v(base + 1), 'a': v(base + 2), 'c1': v(base + 3), 'c2': v(base + 4)}
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
        # loads: 16 x 16 bytes + the 8 bytes behind them
        for q in range(16):
            e('gload4', v(RAW + 4 * q), addr_in, 16 * q)
        e('gload2', v(XW), addr_in, 256)
        e('waitcnt', 'vmcnt(0)')
        # transposes: matrix A = word 0 of every strip (even registers), B = word 1 (odd registers)
        free = [v(TR), v(TR + 1)]
        self.transpose_masks()
        rowsA, free = self.transpose([v(RAW + 2 * s) for s in range(32)], free)
        rowsB, free = self.transpose([v(RAW + 2 * s + 1) for s in range(32)], free)
        # W[t][beta]: bit s = bit beta of base t of strip s
        W = {}
        for t in range(16):
            for be in (0, 1):
                W[(t, be)] = rowsA[2 * t + be]
                W[(t + 16, be)] = rowsB[2 * t + be]
        tr_free = free
        FP, RP = self.FP, self.RP
        # ---- warm-up: steps n = 0..31, incoming base = base n of the strip itself
        for n in range(32):
            mB = self.masks(W[(n, 0)], W[(n, 1)], MB + 2)
            for r in range(31):
                jf = (r + n + 1) % 31
                self.plane_update(FP[r], None, None, self.fi[jf], mB, first=(n == 0))
            for r in range(31):
                jr = (r - n) % 31
                self.plane_update(RP[r], None, None, self.ri[jr], mB, first=(n == 0))
        # ---- productive steps t = 0..31 (n = 32 + t): test the k-mer, then roll
        e('mov', v(XR), v(XW))
        e('lshr', v(XR + 1), v(XW), 1)
        M = {}
        sA, sB, cy, le, ones = (v(TT + i) for i in range(5))
        for t in range(32):
            n = 32 + t
            if t == 16:
                e('mov', v(XR), v(XW + 1))
                e('lshr', v(XR + 1), v(XW + 1), 1)
            # test: top b planes of F + R.  s (sum plane), cy (carry), le, ones hold true values; the planes' complement
            # flags go into the truth tables.
            jlo = 31 - b
            e('mov', le, -1)
            for j in range(jlo, 31):
                f, r = FP[(j - n) % 31], RP[(j + n) % 31]
                nf, nr = self.neg[f], self.neg[r]
                s = sA if (j - jlo) % 2 == 0 else sB
                if j == jlo:
                    e('bitop3', s, f, r, r, self.tt3(lambda a, b2, c: a ^ b2, nf, nr, nr))
                    e('bitop3', cy, f, r, r, self.tt3(lambda a, b2, c: a & b2, nf, nr, nr))
                else:
                    e('bitop3', s, f, r, cy, self.tt3(lambda a, b2, c: a ^ b2 ^ c, nf, nr, 0))
                    if j < 30:
                        e('bitop3', cy, f, r, cy, self.tt3(lambda a, b2, c: (a & b2) | (a & c) | (b2 & c), nf, nr, 0))
                e('bitop3', le, s, le, f"s{S_CM + (j - jlo)}", 0x8E)
                # all-ones over planes jlo+1 .. 30, two planes per op
                idx = j - jlo
                if idx == 1:
                    e('mov', ones, s)
                elif idx >= 2 and idx % 2 == 1:
                    prev = sA if s is sB else sB
                    e('bitop3', ones, ones, prev, s, 0x80)
                elif idx >= 2 and j == 30:
                    e('and', ones, ones, s)
            mt = W[(t, 0)]  # the step's result replaces the step's first base plane after the roll
            if t < 31:
                # in-bits: W shifted by one strip, bit 31 from the next lane's first strip
                in0, in1 = v(MB), v(MB + 1)
                e('alignbit', in0, v(XR), W[(t, 0)], 1)
                e('alignbit', in1, v(XR + 1), W[(t, 1)], 1)
                if t % 16 != 15:
                    e('lshr', v(XR), v(XR), 2)
                    e('lshr', v(XR + 1), v(XR + 1), 2)
                mA = self.masks(W[(t, 0)], W[(t, 1)], MA)
                mB = self.masks(in0, in1, MB + 2)
                for r in range(31):
                    jf = (r + n + 1) % 31
                    self.plane_update(FP[r], self.fo[jf], mA, self.fi[jf], mB)
                for r in range(31):
                    jr = (r - n) % 31
                    self.plane_update(RP[r], self.ro[jr], mA, self.ri[jr], mB)
            e('or', mt, le, ones)
            M[t] = mt
        # ---- out: transpose the 32 step masks into position order (the state registers are free now)
        OUT = [v(ST + i) for i in range(32)]
        KV = [v(ST + 32 + i) for i in range(32)]
        for q in range(8):
            e('gload4', KV[4 * q], addr_kv, 16 * q)
        self.transpose_masks()
        rowsM, _ = self.transpose([M[t] for t in range(32)], tr_free, final=OUT)
        e('waitcnt', 'vmcnt(0)')
        e('mov', v_cnt, 0)
        for s in range(32):
            e('and', OUT[s], OUT[s], KV[s])
            e('bcnt', v_cnt, OUT[s], v_cnt)
        for q in range(8):
            e('gstore4', OUT[4 * q], addr_out, 16 * q)
        return self.p

    def clobbers(self):
        return [f"v{i}" for i in range(V0, VEND)] + [f"s{i}" for i in range(S0, self.send)] + ["vcc", "scc", "memory"]


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
            elif op == 'gstore4':
                d = int(ins[1][1:])
                for q in range(4):
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


def emit_inc(path, k, use_perm=True, prefix="HASH_BS"):
    g = Gen(k, use_perm=use_perm)
    prog = g.build()
    lines = [to_asm(i) for i in prog.ins]
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
    ap.add_argument('--no-perm', action='store_true', help="transposes with shifts only (bench variant)")
    ap.add_argument('--prefix', default='HASH_BS')
    a = ap.parse_args()
    n, nv = emit_inc(a.o, a.k, not a.no_perm, a.prefix)
    print(f"{a.o}: {n} instructions, {nv} VALU", file=sys.stderr)
