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

Geometry (k = 32).  A chunk = 65 536 consecutive base positions = 64 lanes x 32 strips x 32 positions = 4096 words of the
2-bit packed assembly (16 bases per u32, base i of a word in bits 2i, 2i+1).  Lane L reads ITS 64 words (strip 32 L + s = words
64 L + 2 s, + 1) with 16 global_load_dwordx4 and turns them into bit planes in registers: two 32 x 32 bit transposes (the strips'
first / second words), after which
    plane (t, beta) = a register whose bit s = bit beta of the base at chunk * 65536 + (32 lane + s) * 32 + t.
A 32 x 32 transpose = 5 stages x 16 register pairs; a pair is swapped with v_lshrrev_b32, one v_bitop3_b32 against an SGPR mask,
two v_xor_b32 and the left shift as a chain of v_add_u32 x, x (v_lshlrev_b32 is slow class): 1 632 instructions per chunk,
+24 % on the filter itself.  (Until late in round 3 a pre-transposed copy of the assembly was kept instead, built by a kernel
of its own: 0.26 B/bp of HBM and one more pass over the assembly per sketch, 0.33 ms per 3 Gbp -- as long as the filter's launch.)
The strip in front of a lane's first one (its bases are the outgoing bases of slot 0) is the previous lane's last strip: every
lane loads those two words itself (the eight bytes in front of its 256: one global_load_dwordx2 more per chunk -- a DPP move
and a scalar load with its wait did the same until they were found to cost more than a thousand cycles per chunk), and their
even / odd bits are separated into four registers of 16 steps each.  The assembly's first and last chunk are read from padded
copies (two zero words in front of chunk 0, zeros behind the last chunk's words), so nothing outside the packed array is read.
The result words of a chunk are stored at the start of the NEXT chunk, after its words have arrived: a wait for "all but n
loads" then never waits for a store.
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
The words of the next chunk are requested into the registers of the current one as soon as the steps have read them (a load
fills the planes of steps k and 16 + k: it goes out after step 16 + k).

The same instruction list is (1) printed as gfx950 assembly for one inline-asm block with fixed registers (the chunk loop
included) and (2) executed by a numpy model (class VM): tests/test_bs_gen_cpu.py runs the model against the direct ntHash
formula, so renaming, truth tables and address arithmetic are checked without a GPU.
"""
import argparse
import os
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
        return f"v_{op}_b32{E64} {a[0]}, {a[1]}, {a[2]}"
    if op == 'mov':
        return f"v_mov_b32{E64} {a[0]}, {a[1]}"
    if op == 'bitop3':
        return f"v_bitop3_b32 {a[0]}, {a[1]}, {a[2]}, {a[3]} bitop3:{hex(ins[5])}"
    if op == 'perm':  # v_perm_b32: byte i of dst = byte sel[i] of {src0 (4..7), src1 (0..3)}; the selector in an SGPR
        return f"v_perm_b32 {a[0]}, {a[1]}, {a[2]}, {a[3]}"
    if op == 'add':
        return f"v_add_u32{E64} {a[0]}, {a[1]}, {a[2]}"
    if op == 'lshr':
        return f"v_lshrrev_b32{E64} {a[0]}, {a[2]}, {a[1]}"
    if op in ('gload4', 'gload2') and LDSLOADS_ABLATION:  # (timing only: the words land in LDS -- and stay there -- instead of the registers)
        k = ins[3] // 16 if op == 'gload4' else 16
        return f"s_add_u32 m0, s{S_LB}, {1024 * k}\nglobal_load_lds_{'dwordx4' if op == 'gload4' else 'dword'} %[voff256], {sp(ins[2])} offset:{ins[3]}"
    if MUBUF and op in ('gload4', 'gload2'):  # (experiment: buffer instructions on a resource based at the words - 8 bytes)
        d = int(ins[1][1:])
        n = 4 if op == 'gload4' else 2
        return f"buffer_load_dwordx{n} v[{d}:{d + n - 1}], %[voff256], s[{S_RL}:{S_RL + 3}], 0 offen offset:{ins[3] + 8 if op == 'gload4' else 0}"
    if MUBUF and op in ('gstore4', 'gstore3', 'gstore1'):  # (resource based at OUT - 4 bytes)
        d = int(ins[1][1:])
        n = int(op[-1])
        regs = f"v[{d}:{d + n - 1}]" if n > 1 else f"v{d}"
        suffix = {4: 'dwordx4', 3: 'dwordx3', 1: 'dword'}[n]
        return f"buffer_store_{suffix} {regs}, %[voff128], s[{S_RS}:{S_RS + 3}], 0 offen offset:{ins[3] + (0 if ins[2] == S_OC else 4)}"
    if op == 'gload4':  # first dst register, base SGPR pair, byte offset; the lane's offset (lane * 256) is operand %[voff256]
        d = int(ins[1][1:])
        if COALESCED_ABLATION:  # (timing only, wrong words: the 64 lanes of a load read 1 KB in a row -- 8 cache lines instead of 64)
            k = ins[3] // 16
            return f"global_load_dwordx4 v[{d}:{d + 3}], %[vco{k // 4}], {sp(ins[2])} offset:{1024 * (k % 4)}"
        return f"global_load_dwordx4 v[{d}:{d + 3}], %[voff256], {sp(ins[2])} offset:{ins[3]}{LOAD_MOD}"
    if op == 'gload2':  # the two words in front of the lane's 64: base pair = the chunk's words - 8 bytes
        d = int(ins[1][1:])
        return f"global_load_dwordx2 v[{d}:{d + 1}], %[voff256], {sp(ins[2])} offset:{ins[3]}{LOAD_MOD}"
    if op in ('gstore4', 'gstore3', 'gstore1'):  # first data register, base SGPR pair, byte offset; lane * 128: %[voff128]
        d = int(ins[1][1:])
        n = int(op[-1])
        if COSTORES_ABLATION and op != 'gstore1':  # (timing only, wrong places: a store's 64 lanes write 1 KB in a row)
            q = ins[3] // 16
            regs_ = f"v[{d}:{d + n - 1}]"
            return f"global_store_{ {4: 'dwordx4', 3: 'dwordx3'}[n]} %[vco{q // 4}], {regs_}, {sp(S_OC + 2)} offset:{1024 * (q % 4)}"
        regs = f"v[{d}:{d + n - 1}]" if n > 1 else f"v{d}"
        suffix = {4: 'dwordx4', 3: 'dwordx3', 1: 'dword'}[n]
        return f"global_store_{suffix} %[voff128], {regs}, {sp(ins[2])} offset:{ins[3]}{STORE_MOD}"
    if op == 'dsw16':    # low half of a register -> LDS; the lane's address is operand %[vlds]
        return f"ds_write_b16 %[vlds], {a[0]} offset:{ins[2]}"
    if op == 'dsw16hi':  # high half
        return f"ds_write_b16_d16_hi %[vlds], {a[0]} offset:{ins[2]}"
    if op == 'dsr32':
        return f"ds_read_b32 {a[0]}, %[vlds] offset:{ins[2]}"
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
E64 = "_e64" if os.environ.get("BS_E64") == "1" else ""   # experiment: the two-source instructions in their 8-byte encoding (same work, larger code)
MUBUF = os.environ.get("BS_MUBUF") == "1"   # experiment: buffer_load / buffer_store instead of global_load / global_store
LOAD_MOD = os.environ.get("BS_LOAD_MOD", "")    # cache-policy bits of the loads / stores (experiments: " nt", " sc1", ...)
STORE_MOD = os.environ.get("BS_STORE_MOD", "")
LDSLOADS_ABLATION = False   # --ablate ldsloads (timing only: LDS-direct loads, nothing reads the LDS)
COALESCED_ABLATION = False  # --ablate coalesced (tools/bs_bench.hip timing only)
COSTORES_ABLATION = False   # --ablate costores (likewise, the result stores)
LDS_SLOTS = 8              # pairs of the LDS stage in flight per wave (slots are reused in order)
LDS_SLOT = 528             # bytes per slot: 64 lanes x 8 bytes + 4 bytes per 16 lanes (lane address = 8 lane + 4 (lane / 16): the
                           # 32 lanes of a read then hit 32 different banks)
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
S_T = S0 + 6        # packed bases (pair), S_P = the copy of the last chunk's words (pair), S_O = OUT base (pair)
S_P = S0 + 8
S_O = S0 + 10
S_TN = S0 + 12      # pair: the next chunk's words
S_CB = S0 + 14      # pair: this chunk's words
S_FIRST = S0 + 16   # 1 in a wave's first chunk (no results of a chunk before to store)
S_CTAIL = S0 + 18   # the chunk whose words are taken from the copy (the assembly's last, ragged chunk)
S_OC = S0 + 20      # 2 pairs: this chunk's OUT words - 4 bytes, this chunk's OUT words
S_TQ = S0 + 24      # pair: the next chunk's words - 8 bytes
S_TMP = S0 + 26     # pair
S_CM = S0 + 28      # B_PLANES compare masks
S_M16, S_M8, S_M4, S_M2, S_M1 = (S0 + 28 + B_PLANES + i for i in range(5))  # the transpose's select masks
S_HD = (S_M1 + 2) & ~1   # pair (even register): the padded copy of chunk 0's words
S_PA = S_HD + 2      # v_perm_b32 selectors of the transposes' stage 16: [y.lo : x.lo] and [y.hi : x.hi] of (src0 = y, src1 = x)
S_PB = S_HD + 3
SEND = S_HD + 4
S_LB = SEND         # (--ablate ldsloads only: the wave's LDS base)
S_RL = 88           # (BS_MUBUF only: buffer resources of the loads and of the stores, four SGPRs each)
S_RS = 92


class Gen:
    def __init__(self, k=32, b_planes=B_PLANES, ablate=(), lds16=(), perm16=()):
        """ablate (timing experiments only, tools/bs_ablate.sh -- the results are wrong on purpose): 'loads' = no vector loads in the
        chunk loop (every chunk works on the first one's words) and no waits for them, 'stores' = no result stores"""
        assert k == 32, "strips of 32 k-mers: k = 32 only (other k: k_hash_sparse)"
        self.ablate = set(ablate)
        self.lds16 = set(lds16) if not isinstance(lds16, bool) else ({'in', 'out'} if lds16 else set())
        self.perm16 = set(perm16)
        self.warm_pairs = 'nowarmpairs' not in self.ablate
        self.lds_slot = 0
        self.k = k
        self.b = b_planes
        self.ins = []
        self.fo, self.fi, self.ro, self.ri = plane_funcs(k)
        self.FP = [grp(i, 0) for i in range(31)]
        self.RP = [grp(i, 1) for i in range(31)]
        # the lane's 64 packed words as loaded: word 2 s + h (h = 0: bases 0..15 of strip s, 1: bases 16..31) in v(W0 + 2 s + h);
        # after the two transposes plane (t, beta) of the first words sits where word 2 (2 t + beta) was, of the second words likewise
        self.RAW = [[f"v{W0 + 2 * s_ + h_}" for s_ in range(32)] for h_ in (0, 1)]
        self.W = {(t, be): f"v{W0 + 4 * (t % 16) + 2 * be + t // 16}" for t in range(32) for be in (0, 1)}
        self.A = [grp(g, 2) for g in range(7)]   # o0 o1 x o a c1 c2 of the outgoing base
        self.B = [grp(g, 3) for g in range(7)]   # in0 in1 ... of the incoming base
        self.cy, self.s = grp(7, 2), grp(7, 3)
        assert X0 % 4 == 0
        # the steps' results, then their transpose: M[1..] go out as dwordx4 (register tuples must start on an even register)
        self.M = [f"v{X0 + 1 + i}" for i in range(32)]
        # the strip in front of the lane's first one, bit planes: QL[beta] bit t = bases 0..15, QH[beta] = bases 16..31; Qr: running
        self.QL = [f"v{X0 + 34}", f"v{X0 + 35}"]
        self.QH = [f"v{X0 + 36}", f"v{X0 + 37}"]
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
            if ra is not None and rb is not None:
                e('xor', dst, ra, rb)
            else:
                e('mov', dst, ra or rb or 0)
            self.neg[dst] = c
            return
        self.neg[dst] ^= c
        if ra is None and rb is None:
            pass
        elif ra is None or rb is None:
            e('xor', dst, dst, ra or rb)
        else:
            e('bitop3', dst, dst, ra, rb, 0x96)

    def o_stream(self, t, dst=None):
        """dst[0], dst[1] (default: A) <- the two bit planes of the outgoing base of step t: W[t] moved up by one slot, bit t of
        Q at the bottom"""
        e = self.e
        tt = self.tt3(lambda a, b2, c: a | (b2 & c))
        dst = dst or self.A
        for be in (0, 1):
            if t % 16 == 0:
                e('mov', self.Qr[be], (self.QL if t == 0 else self.QH)[be])
            e('add', dst[be], self.W[(t, be)], self.W[(t, be)])
            e('bitop3', dst[be], dst[be], self.Qr[be], 1, tt)
            if t % 16 != 15:
                e('lshr', self.Qr[be], self.Qr[be], 1)

    def chunk(self):
        """the body of the chunk loop: W and Qn hold the chunk's words (their loads may still be in flight)"""
        e = self.e
        b = self.b
        FP, RP, W = self.FP, self.RP, self.W
        A, B = self.A, self.B
        # ---- the chunk's 64 words per lane -> bit planes; then (the loads are all back) the results of the chunk before go out
        self.planes_in()
        e('prev_stores')
        # ---- warm-up: steps n = 0..31: the slot's own strip, i.e. the o-stream (W shifted up by one slot)
        # Two steps per pass over the ring: a roll has room for two masks (the productive steps' outgoing and incoming base),
        # the warm-up has no outgoing base, so steps n and n + 1 share one instruction per plane (masks of n in the A set,
        # of n + 1 in the B set, which nothing else uses before the first productive step).  The first pair writes the ring
        # (dst = m1 ^ m2), the others add to it.
        n = 0
        while n < 32:
            if not self.warm_pairs:
                self.o_stream(n)
                m1 = self.masks(A)
                for r in range(31):
                    self.plane_update(FP[r], None, None, self.fi[(r + n + 1) % 31], m1, first=(n == 0))
                for r in range(31):
                    self.plane_update(RP[r], None, None, self.ri[(r - n) % 31], m1, first=(n == 0))
                n += 1
                continue
            self.o_stream(n)
            m1 = self.masks(A)
            self.o_stream(n + 1, B)
            m2 = self.masks(B)
            for r in range(31):
                self.plane_update(FP[r], self.fi[(r + n + 1) % 31], m1, self.fi[(r + n + 2) % 31], m2, first=(n == 0))
            for r in range(31):
                self.plane_update(RP[r], self.ri[(r - n) % 31], m1, self.ri[(r - n - 1) % 31], m2, first=(n == 0))
            n += 2
        # ---- productive steps t = 0..31 (n = 32 + t): test the k-mer, then roll
        s, cy, le, ones = self.s, self.cy, self.le, self.ones
        for t in range(32):
            n = 32 + t
            if t < 31:
                # masks of the outgoing base (bank 2) and of the incoming base (bank 3), made before the test
                self.o_stream(t)
                e('mov', B[0], W[(t, 0)])
                e('mov', B[1], W[(t, 1)])
            if t >= 16:  # the planes of steps t - 16 and t are dead: the next chunk's words 4 (t - 16) .. + 3 can come
                e('gload4', f"v{W0 + 4 * (t - 16)}", S_TN, 16 * (t - 16))
            if t == 31:  # (the running Q registers are dead since step 30: the words in front of the lane's 64 land there)
                e('gload2', self.Qr[0], S_TQ, 0)
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
        # M[s] = the 32 positions of strip 32 lane + s - 1: slots 1..31 at bytes 0..123 of the lane's 128, slot 0 in front.
        # They stay in their registers (nothing touches M before the next chunk's first productive step) and are stored by
        # the next chunk, or behind the loop.
        self.store_ins = [('gstore1', self.M[0], S_OC, 0)] + [('gstore4', self.M[4 * q + 1], S_OC + 2, 16 * q) for q in range(7)] + \
                         [('gstore3', self.M[29], S_OC + 2, 112)]
        self.check_banks()
        return self.ins

    def temp_pool(self):
        """registers that are dead while the bit planes are made and while the results are transposed back: the ring planes
        (the warm-up makes them anew), the masks, the adder's registers and the transposes' own four -- eight per bank"""
        return self.TT + self.FP[:7] + self.RP[:7] + self.A + self.B

    def pick(self, pool, used, avoid_banks):
        for t_ in pool:
            if t_ not in used and bank(t_) not in avoid_banks:
                used.add(t_)
                return t_
        raise RuntimeError("no temporary register of a suitable bank left")

    def swap_pairs(self, pairs, j, sm, width=8):
        """one stage of a 32 x 32 bit transpose in place on the register pairs (x, y) = rows (k, k + j): they exchange the bits
        whose index has bit j set / clear:  t = ((x >> j) ^ y) & m;  y ^= t;  x ^= t << j   (m in an SGPR; the left shift = j
        times v_add_u32 t, t).  `width` pairs are interleaved, so that an instruction never follows the one it depends on: a
        chain of dependent v_add_u32 issued back to back runs at half the rate."""
        e = self.e
        tt = self.tt3(lambda a, b2, c: (a ^ b2) & c)
        pool = self.temp_pool()
        for i in range(0, len(pairs), width):
            grp_ = pairs[i:i + width]
            used = set()
            tmp = [self.pick(pool, used, {bank(y)}) for (x, y) in grp_]
            for (x, y), t_ in zip(grp_, tmp):
                e('lshr', t_, x, j)
            for (x, y), t_ in zip(grp_, tmp):
                e('bitop3', t_, t_, y, f"s{sm}", tt)
            for (x, y), t_ in zip(grp_, tmp):
                e('xor', y, y, t_)
            for _ in range(j):
                for t_ in tmp:
                    e('add', t_, t_, t_)
            for (x, y), t_ in zip(grp_, tmp):
                e('xor', x, x, t_)

    def swap16_lds(self, pairs):
        """the stage j = 16 of a transpose (rows k, k + 16 exchange x's high with y's low half) through LDS: no VALU at all.
        A pair has 8 bytes per lane, [x.lo][y.lo][x.hi][y.hi], written half by half (ds_write_b16 / ds_write_b16_d16_hi) and read
        back as the two new words.  In registers the stage is lshr, bitop3, xor, SIXTEEN v_add_u32 (the left shift; v_lshlrev_b32
        is slow class) and xor per pair: 20 of the 8 520 instructions of a chunk x 48 pairs.  The LDS operations of one wave are
        carried out in order, so the slots are reused without waiting; the caller waits (lgkmcnt) before it uses the registers.
        Measured on MI355X (round 6, profiles/r06/filter_lds16_ab.txt): 7 576 VALU + 288 LDS operations per chunk run as long as the
        8 520 VALU of the register version (470-500 us per 3 Gbp launch either way), so the shipped kernel keeps the registers."""
        e = self.e
        for (x, y) in pairs:
            o = LDS_SLOT * (self.lds_slot % LDS_SLOTS)
            self.lds_slot += 1
            e('dsw16', x, o)
            e('dsw16', y, o + 2)
            e('dsw16hi', x, o + 4)
            e('dsw16hi', y, o + 6)
            e('dsr32', x, o)
            e('dsr32', y, o + 4)

    def swap16_perm(self, pairs, width=8):
        """the stage j = 16 of a transpose as byte permutes: x' = [y.lo : x.lo], y' = [y.hi : x.hi] -- two v_perm_b32 and a move per
        pair instead of lshr, bitop3, xor, SIXTEEN v_add_u32 and xor.  v_perm_b32 is not of the fast class (profiles/ubench), but
        three instructions stand for twenty."""
        e = self.e
        pool = self.temp_pool()
        for i in range(0, len(pairs), width):
            grp_ = pairs[i:i + width]
            used = set()
            tmp = [self.pick(pool, used, set()) for _ in grp_]
            for (x, y), t_ in zip(grp_, tmp):
                e('perm', t_, y, x, f"s{S_PB}")
            for (x, y), t_ in zip(grp_, tmp):
                e('perm', x, y, x, f"s{S_PA}")
            for (x, y), t_ in zip(grp_, tmp):
                e('mov', y, t_)

    @staticmethod
    def stage_pairs(regs, j):
        return [(regs[k], regs[k + j]) for k in range(32) if not k & j]

    def even_bits(self, dst, src, shift):
        """dst = the bits 2 i + shift of src, packed into bits 0..15 (right shifts only)"""
        e = self.e
        tt = self.tt3(lambda a, b2, c: (a | b2) & c)
        t_ = self.TT[(bank(dst) + 1) % 4]
        if shift:
            e('lshr', dst, src, shift)
            e('and', dst, dst, f"s{S_M1}")
        else:
            e('and', dst, src, f"s{S_M1}")
        for j, sm in ((1, S_M2), (2, S_M4), (4, S_M8), (8, S_M16)):
            e('lshr', t_, dst, j)
            e('bitop3', dst, t_, dst, f"s{sm}", tt)

    def planes_in(self):
        """the lane's 64 packed words (RAW) and the two in front of them (in the Qr registers) -> the strip in front (QL, QH)
        and the 64 bit planes (W), in place"""
        e = self.e
        pa, pb = self.Qr[0], self.Qr[1]
        # the first stage pairs strips s and s + 16, i.e. the words of loads s / 2 and s / 2 + 8: it starts as soon as nine of the
        # sixteen loads are back and follows the others in (loads return in order; behind load j come 15 - j loads and the
        # one of the two words in front), instead of waiting for the last one first.  No store is outstanding here.
        p16 = [self.stage_pairs(self.RAW[h_], 16) for h_ in (0, 1)]
        for k in range(0, 16, 2):
            e('waitcnt', f'vmcnt({8 - k // 2 if k < 14 else 0})')
            if 'in' in self.lds16:
                self.swap16_lds(p16[0][k:k + 2] + p16[1][k:k + 2])
            elif 'in' in self.perm16:
                self.swap16_perm(p16[0][k:k + 2] + p16[1][k:k + 2])
            else:
                self.swap_pairs(p16[0][k:k + 2] + p16[1][k:k + 2], 16, S_M16)
        for be in (0, 1):
            self.even_bits(self.QL[be], pa, be)
            self.even_bits(self.QH[be], pb, be)
        if 'in' in self.lds16:
            e('waitcnt', 'lgkmcnt(0)')
        for j, sm in ((8, S_M8), (4, S_M4), (2, S_M2), (1, S_M1)):
            self.swap_pairs(self.stage_pairs(self.RAW[0], j) + self.stage_pairs(self.RAW[1], j), j, sm)

    def transpose_out(self, width=8):
        """M[t] bit s -> M[s] bit t, in place.  Stage j pairs rows k, k + j: new_k = (k & m) | ((k+j << j) & ~m),
        new_k+j = ((k >> j) & m) | (k+j & ~m), m = the bits whose index has bit j clear (an SGPR: v_bitop3_b32 with one SGPR source
        stays fast class); the left shift is j times v_add_u32 x, x (v_lshlrev_b32 is slow class).  `width` pairs interleaved."""
        e = self.e
        sel = self.tt3(lambda a, b2, c: a if c else b2)
        pool = self.temp_pool()
        for j, sm in ((16, S_M16), (8, S_M8), (4, S_M4), (2, S_M2), (1, S_M1)):
            pairs = self.stage_pairs(self.M, j)
            if j == 16 and 'out' in self.lds16:  # (new_a = [b.lo : a.lo], new_b = [b.hi : a.hi]: the same exchange)
                self.swap16_lds(pairs)
                e('waitcnt', 'lgkmcnt(0)')
                continue
            if j == 16 and 'out' in self.perm16:
                self.swap16_perm(pairs)
                continue
            for i in range(0, len(pairs), width):
                grp_ = pairs[i:i + width]
                used = set()
                t0 = [self.pick(pool, used, {bank(a)}) for (a, bq) in grp_]
                t1 = [self.pick(pool, used, {bank(bq)}) for (a, bq) in grp_]
                for (a, bq), t_ in zip(grp_, t0):
                    e('add', t_, bq, bq)
                for _ in range(j - 1):
                    for t_ in t0:
                        e('add', t_, t_, t_)
                for (a, bq), t_ in zip(grp_, t1):
                    e('lshr', t_, a, j)
                for (a, bq), t_ in zip(grp_, t0):
                    e('bitop3', a, a, t_, f"s{sm}", sel)
                for (a, bq), t_ in zip(grp_, t1):
                    e('bitop3', bq, t_, bq, f"s{sm}", sel)

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
    def next_pointers(self, lines, which):
        """SALU: S_TN = the words of chunk s`which` (chunk 0 and the assembly's last chunk: their padded copies), S_TQ = S_TN - 8"""
        L = lines.append
        L(f"s_mov_b32 s{which + 1}, 0")
        L(f"s_lshl_b64 {sp(S_TN)}, {sp(which)}, 14")
        L(f"s_add_u32 s{S_TN}, s{S_TN}, s{S_T}")
        L(f"s_addc_u32 s{S_TN + 1}, s{S_TN + 1}, s{S_T + 1}")
        L(f"s_cmp_eq_u32 s{which}, s{S_CTAIL}")
        L(f"s_cselect_b64 {sp(S_TN)}, {sp(S_P)}, {sp(S_TN)}")
        L(f"s_cmp_eq_u32 s{which}, 0")
        L(f"s_cselect_b64 {sp(S_TN)}, {sp(S_HD)}, {sp(S_TN)}")
        L(f"s_add_u32 s{S_TQ}, s{S_TN}, -8")
        L(f"s_addc_u32 s{S_TQ + 1}, s{S_TN + 1}, -1")
        if MUBUF:
            L(f"s_mov_b32 s{S_RL}, s{S_TQ}")
            L(f"s_and_b32 s{S_RL + 1}, s{S_TQ + 1}, 0xffff")

    def address_setup(self, lines):
        """SALU at the top of the loop: the NEXT chunk's words (the last chunk of a wave asks for its own words again)"""
        L = lines.append
        t0 = S_TMP
        L(f"s_add_u32 s{t0}, s{S_C}, s{S_STRIDE}")
        L(f"s_cmp_lt_u32 s{t0}, s{S_N}")
        L(f"s_cselect_b32 s{t0}, s{t0}, s{S_C}")
        self.next_pointers(lines, t0)

    def out_pointers(self, lines):
        """SALU: this chunk's 2048 OUT words (S_OC + 2) and the same minus one word (slot 0 of a lane = the word in front of its 31)"""
        L = lines.append
        d = S_OC + 2
        L(f"s_lshl_b64 {sp(d)}, {sp(S_C)}, 13")
        L(f"s_add_u32 s{d}, s{d}, s{S_O}")
        L(f"s_addc_u32 s{d + 1}, s{d + 1}, s{S_O + 1}")
        L(f"s_add_u32 s{S_OC}, s{d}, -4")
        L(f"s_addc_u32 s{S_OC + 1}, s{d + 1}, -1")
        if MUBUF:
            L(f"s_mov_b32 s{S_RS}, s{S_OC}")
            L(f"s_and_b32 s{S_RS + 1}, s{S_OC + 1}, 0xffff")

    def asm(self):
        """the inline-asm text.  Operands: %[t] %[p] %[hd] %[o] (SGPR pairs: packed bases, padded copies of the last and of the first
        chunk's words, OUT), %[c0] first chunk of the wave, %[n] one past the last chunk, %[stride] chunks between a wave's chunks,
        %[tt] threshold, %[ctail] the last chunk, VGPRs %[voff256] = lane * 256, %[voff128] = lane * 128"""
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
        A(f"s_mov_b64 {sp(S_HD)}, %[hd]")
        A(f"s_mov_b64 {sp(S_O)}, %[o]")
        A(f"s_mov_b32 s{S_CTAIL}, %[ctail]")
        A(f"s_mov_b32 s{S_FIRST}, 1")
        if MUBUF:
            for r_ in (S_RL, S_RS):
                A(f"s_mov_b32 s{r_ + 2}, -1")
                A(f"s_mov_b32 s{r_ + 3}, 0x00020000")
        for i in range(self.b):  # compare masks: Cm_i = all ones iff bit i of the threshold is set
            A(f"s_bfe_u32 s{S_TMP}, s{S_TT}, {hex((1 << 16) | i)}")
            A(f"s_sub_u32 s{S_CM + i}, 0, s{S_TMP}")
        for sm, val in ((S_M16, 0x0000FFFF), (S_M8, 0x00FF00FF), (S_M4, 0x0F0F0F0F), (S_M2, 0x33333333), (S_M1, 0x55555555),
                        (S_PA, 0x05040100), (S_PB, 0x07060302)):
            A(f"s_mov_b32 s{sm}, {hex(val)}")
        A(f"s_cmp_ge_u32 s{S_C}, s{S_N}")
        A("s_cbranch_scc1 L_bs_end_%=")
        # prologue: the first chunk's words
        if LDSLOADS_ABLATION:
            A(f"v_readfirstlane_b32 s{S_LB}, %[vlds]")
        A(f"s_mov_b32 s{S_TMP}, s{S_C}")
        self.next_pointers(L, S_TMP)
        for k in range(16):
            for piece in to_asm(('gload4', f"v{W0 + 4 * k}", S_TN, 16 * k)).split("\n"):
                A(piece)
        for piece in to_asm(('gload2', self.Qr[0], S_TQ, 0)).split("\n"):
            A(piece)
        A("L_bs_loop_%=:")
        self.address_setup(L)
        for ins in body:
            if ins[0] == 'prev_stores':
                A(f"s_cmp_eq_u32 s{S_FIRST}, 1")
                A("s_cbranch_scc1 L_bs_nostore_%=")
                for st in self.store_ins:
                    if 'stores' not in self.ablate:
                        A(to_asm(st))
                A("L_bs_nostore_%=:")
                A(f"s_mov_b32 s{S_FIRST}, 0")
                self.out_pointers(L)
            elif ins[0] in ('gload4', 'gload2', 'waitcnt') and 'loads' in self.ablate:
                continue
            elif ins[0] == 'waitcnt' and 'vmcnt' in ins[1] and 'waits' in self.ablate:  # (the loads stay, nothing waits for them)
                continue
            elif ins[0] != 'comment':
                for piece in to_asm(ins).split("\n"):
                    A(piece)
        A(f"s_add_u32 s{S_C}, s{S_C}, s{S_STRIDE}")
        A(f"s_cmp_lt_u32 s{S_C}, s{S_N}")
        A("s_cbranch_scc1 L_bs_loop_%=")
        for st in self.store_ins:  # the last chunk's results (kept in every ablation: something must depend on the work)
            A(to_asm(st))
        A("s_waitcnt vmcnt(0)")  # (and the requests for a chunk that does not follow)
        A("L_bs_end_%=:")
        return L

    def clobbers(self):
        return [f"v{i}" for i in range(B0, VEND)] + [f"s{i}" for i in range(S0, SEND)] + ["vcc", "scc", "memory"] + (["m0", f"s{S_LB}"] if LDSLOADS_ABLATION else []) + ([f"s{i}" for i in range(S_RL, S_RS + 4)] if MUBUF else [])


# ---------------------------------------------------------------------------------------------------------------
# numpy model of the chunk body (64 lanes)
# ---------------------------------------------------------------------------------------------------------------
class VM:
    def __init__(self, packed, tt, c, c_next):
        """packed: uint32 [n_chunks][4096], the chunks' packed words; runs chunk c (whose words are preloaded into the W
        registers, as the prologue / the previous iteration does) and collects the chunk's OUT words and the words requested
        for chunk c_next"""
        self.packed = packed
        self.c, self.cn = c, c_next
        self.vr = {}
        self.sr = {S_CM + i: (0xFFFFFFFF if (tt >> i) & 1 else 0) for i in range(B_PLANES)}
        self.sr.update({S_M16: 0x0000FFFF, S_M8: 0x00FF00FF, S_M4: 0x0F0F0F0F, S_M2: 0x33333333, S_M1: 0x55555555,
                        S_PA: 0x05040100, S_PB: 0x07060302})
        self.out = np.zeros(2048 + 1, dtype=np.uint32)  # word index + 1 (slot 0 of lane 0 lies in front of the chunk)
        self.lds = {}  # byte offset of a half word (lane address left out: every lane has its own bytes) -> uint16[64]

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

    def lane_words(self, chunk, k):
        """what global_load_dwordx4 number k brings: words 64 lane + 4 k .. + 3 of the chunk"""
        lanes = np.arange(64)
        return [self.packed[chunk, 64 * lanes + 4 * k + j].copy() for j in range(4)]

    def front_words(self, chunk):
        """what the global_load_dwordx2 brings: the two words in front of the lane's 64 (in front of chunk 0: zeros, the padded copy)"""
        flat = np.concatenate([np.zeros(2, dtype=np.uint32), self.packed.reshape(-1)])
        lanes = np.arange(64)
        at = chunk * 4096 + 64 * lanes  # (+ 2 for the pad, - 2 for "in front")
        return [flat[at].copy(), flat[at + 1].copy()]

    def run(self, g):
        U = np.uint32
        for k in range(16):
            for j, wds in enumerate(self.lane_words(self.c, k)):
                self.vr[f"v{W0 + 4 * k + j}"] = wds
        for j, wds in enumerate(self.front_words(self.c)):
            self.vr[f"v{int(g.Qr[0][1:]) + j}"] = wds
        pend = {}  # loads in flight: they land when the body ends (no instruction of this chunk may see them)
        for ins in list(g.ins) + list(getattr(g, 'store_ins', [])):  # (the chunk's results go out at the start of the next chunk / behind the loop)
            op = ins[0]
            if op == 'prev_stores':
                continue
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
            elif op == 'perm':
                both = (self.V(ins[2]).astype(np.uint64) << np.uint64(32)) | self.V(ins[3]).astype(np.uint64)
                sel = int(self.V(ins[4])[0]) if not isinstance(self.V(ins[4]), int) else self.V(ins[4])
                r = np.zeros(64, dtype=U)
                for i in range(4):
                    q = (sel >> (8 * i)) & 0xFF
                    assert q < 8
                    r |= ((both >> np.uint64(8 * q)) & np.uint64(0xFF)).astype(U) << U(8 * i)
                self.vr[ins[1]] = r
            elif op == 'add':
                self.vr[ins[1]] = (self.V(ins[2]) + self.V(ins[3])).astype(U)
            elif op == 'lshr':
                self.vr[ins[1]] = (self.V(ins[2]) >> U(ins[3])).astype(U)
            elif op == 'gload2':
                d = int(ins[1][1:])
                for j, wds in enumerate(self.front_words(self.cn)):
                    pend[f"v{d + j}"] = wds
            elif op == 'gload4':
                k = ins[3] // 16
                d = int(ins[1][1:])
                assert d == W0 + 4 * k
                for j, wds in enumerate(self.lane_words(self.cn, k)):
                    pend[f"v{d + j}"] = wds
            elif op in ('gstore4', 'gstore3', 'gstore1'):
                n = int(op[-1])
                base = -1 if ins[2] == S_OC else 0  # word offset of the SGPR pair relative to the chunk
                d = int(ins[1][1:])
                lanes = np.arange(64)
                for j in range(n):
                    self.out[1 + base + 32 * lanes + ins[3] // 4 + j] = self.vr[f"v{d + j}"]
            elif op == 'dsw16':
                self.lds[ins[2]] = (self.V(ins[1]) & U(0xFFFF)).astype(np.uint16)
            elif op == 'dsw16hi':
                self.lds[ins[2]] = (self.V(ins[1]) >> U(16)).astype(np.uint16)
            elif op == 'dsr32':
                assert ins[2] % 4 == 0
                self.vr[ins[1]] = self.lds[ins[2]].astype(U) | (self.lds[ins[2] + 2].astype(U) << U(16))
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
def pack_chunks(codes, n_chunks):
    """base codes (0..3) of n_chunks * 65536 positions -> uint32 [n_chunks][4096]: 16 bases per word, base i in bits 2i, 2i+1"""
    codes = np.asarray(codes, dtype=np.uint32)
    assert len(codes) == n_chunks * CHUNK
    sh = (2 * np.arange(16, dtype=np.uint32))[None, :]
    return (codes.reshape(-1, 16) << sh).sum(axis=1, dtype=np.uint32).reshape(n_chunks, 4096)


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


def emit_inc(path, k, ablate=(), lds16=(), perm16=()):
    g = Gen(k, ablate=ablate, lds16=lds16, perm16=perm16)
    lines = g.asm()
    n_valu = sum(1 for i in g.ins if i[0] in ('xor', 'and', 'or', 'mov', 'bitop3', 'add', 'lshr', 'perm'))
    with open(path, 'w') as fh:
        fh.write(f"// GENERATED by gen/bs_gen.py (k = {k}, {B_PLANES} sum planes): the bit-sliced ring filter, chunk loop included.\n")
        fh.write(f"// {n_valu} VALU per chunk of 65 536 base positions per wave, all of them full-rate (see gen/bs_gen.py).  Do not edit.\n")
        fh.write("// operands: [t] [p] [hd] [o] SGPR pairs (packed bases, padded copies of the last / first chunk, OUT), [c0] [n] [stride] [tt] [ctail] SGPRs, [voff256] VGPR = lane * 256, [voff128] = lane * 128, [vlds] = the lane's LDS address (HASH_BS_LDS_PER_WAVE bytes per wave: 8 lane + 4 (lane / 16))\n")
        fh.write(f"#define HASH_BS_VGPR_END {VEND}\n")
        fh.write(f"#define HASH_BS_VALU_PER_CHUNK {n_valu}\n")
        fh.write(f"#define HASH_BS_PLANES {B_PLANES}\n")
        fh.write(f"#define HASH_BS_LDS_PER_WAVE {18 * 1024 if LDSLOADS_ABLATION else LDS_SLOTS * LDS_SLOT if g.lds16 else 0}\n")
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
    ap.add_argument('--ablate', default='', help="comma-separated: loads, stores, coalesced (timing experiments: tools/bs_ablate.sh; coalesced = every load's 64 lanes read 1 KB in a row, wrong words)")
    ap.add_argument('--perm16', default='', help="which transposes' stage 16 is two v_perm_b32 and a move per pair instead of twenty fast-class instructions: in, out, in,out")
    ap.add_argument('--lds16', default='', help="which transposes' stage 16 goes through LDS instead of registers: in, out, in,out (round 6: 944 VALU instructions fewer per chunk for 288 LDS operations, and no faster -- profiles/r06/filter_lds16_ab.txt)")
    a = ap.parse_args()
    COALESCED_ABLATION = 'coalesced' in a.ablate.split(',')
    LDSLOADS_ABLATION = 'ldsloads' in a.ablate.split(',')
    COSTORES_ABLATION = 'costores' in a.ablate.split(',')
    n, nv = emit_inc(a.o, a.k, tuple(x for x in a.ablate.split(',') if x), lds16=tuple(x for x in a.lds16.split(',') if x),
                     perm16=tuple(x for x in a.perm16.split(',') if x))
    print(f"{a.o}: {n} lines, {nv} VALU per chunk", file=sys.stderr)
