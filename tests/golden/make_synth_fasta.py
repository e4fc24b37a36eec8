#!/usr/bin/env python3
"""
make_synth_fasta.py -- seeded generator of the small synthetic FASTA fixtures in tests/golden/fasta/
(synth.*.fa, more_seqs.pieces.fa).  They add what the reference's own fixtures lack on the hot path:
duplicated minimizers (repeats, homopolymers, microsatellites), N runs inside windows, single Ns,
soft-masked (lower-case) stretches, records shorter than k+w-1, wrapped lines, header comments.
Run from anywhere; output is deterministic.
"""
import os
import random

HERE = os.path.dirname(os.path.abspath(__file__))
FASTA = os.path.join(HERE, "fasta")
COMP = str.maketrans("ACGTNacgtn", "TGCANtgcan")


def rc(s):
    return s[::-1].translate(COMP)


def rand_seq(rng, n):
    return "".join(rng.choice("ACGT") for _ in range(n))


def mutate(rng, s, rate):
    out = list(s)
    for i, c in enumerate(out):
        if c in "ACGT" and rng.random() < rate:
            out[i] = rng.choice([b for b in "ACGT" if b != c])
    return "".join(out)


def write_fasta(path, records, width):
    with open(path, "w", encoding="ascii") as fh:
        for header, seq in records:
            fh.write(">" + header + "\n")
            if width:
                for i in range(0, len(seq), width):
                    fh.write(seq[i:i + width] + "\n")
            else:
                fh.write(seq + "\n")


def main():
    os.makedirs(FASTA, exist_ok=True)
    rng = random.Random(20240928)
    # --- reference with repeats / N / soft-masking ---
    a = list(rand_seq(rng, 40000))
    rep = a[5000:7500]
    a[25000:27500] = rep                        # exact 2.5-kbp duplicate -> duplicated minimizers
    a[12000:12300] = "N" * 300                  # N run
    for p in (3000, 3001, 18000, 33333):        # isolated Ns
        a[p] = "N"
    a[20000:21000] = [c.lower() for c in a[20000:21000]]  # soft-masked stretch
    chr_a = "".join(a)
    b = list(rand_seq(rng, 15000))
    b[2000:3200] = "A" * 1200                   # homopolymer longer than w+k for w<=1000
    b[6000:7500] = list("AC" * 750)             # microsatellite
    b[9000:9040] = "N" * 40
    chr_b = "".join(b)
    chr_c = rand_seq(rng, 900)                  # shorter than k+w-1 at w=1000
    ref = [("chrA first record", chr_a), ("chrB", chr_b), ("chrC\tshort one", chr_c), ("chrD_tiny", "ACGTNNACGT")]
    write_fasta(os.path.join(FASTA, "synth.ref.fa"), ref, 60)
    # --- a second, diverged reference ---
    ref2 = [(h.split()[0] + "_v2", mutate(rng, s, 0.01)) for h, s in ref[:3]]
    write_fasta(os.path.join(FASTA, "synth.ref2.fa"), ref2, 0)
    # --- target: pieces of chrA / chrB, some reverse-complemented, lightly mutated, shuffled ---
    pieces = []
    for name, s in (("A", chr_a), ("B", chr_b)):
        p, i = 0, 0
        while p < len(s):
            ln = rng.randint(2000, 8000)
            seg = s[p:p + ln]
            p += ln + rng.randint(20, 200)
            if len(seg) < 500:
                continue
            seg = mutate(rng, seg, 0.005)
            strand = "f"
            if rng.random() < 0.5:
                seg, strand = rc(seg), "r"
            pieces.append((f"ctg{name}{i}_{strand} len={len(seg)}", seg))
            i += 1
    rng.shuffle(pieces)
    pieces.append(("ctg_short", rand_seq(rng, 40)))
    write_fasta(os.path.join(FASTA, "synth.tgt.fa"), pieces, 70)
    # --- pieces of the reference's own 20-record fixture ---
    src = os.path.join(FASTA, "scaf.more_seqs.fa")
    if os.path.exists(src):
        recs, cur = [], None
        with open(src, encoding="ascii") as fh:
            for line in fh:
                line = line.rstrip("\n")
                if line.startswith(">"):
                    cur = [line[1:].split()[0], []]
                    recs.append(cur)
                elif cur is not None:
                    cur[1].append(line)
        out = []
        for rid, chunks in recs:
            s = "".join(chunks)
            cut = len(s) // 2 + rng.randint(-200, 200)
            left, right = s[:cut], s[cut + 50:]
            if rng.random() < 0.5:
                right = rc(right)
            out.append((rid + "_L", left))
            out.append((rid + "_R", right))
        rng.shuffle(out)
        write_fasta(os.path.join(FASTA, "more_seqs.pieces.fa"), out, 80)


if __name__ == "__main__":
    main()
