"""Seeded random FASTA/FASTQ text for parity tests (test inputs only)."""
import random


def random_fastq(rng, nrec, min_len=0, max_len=300, final_newline=True, name_space=True, trailing_blank=0,
                 qual_lo=33, qual_hi=126, alphabet="ACGTN"):
    out = []
    for i in range(nrec):
        L = rng.randint(min_len, max_len)
        seq = "".join(rng.choice(alphabet) for _ in range(L))
        if L and rng.random() < 0.05:
            seq = seq[:L // 2] + "-." [rng.randrange(2)] + seq[L // 2 + 1:]
        qual = "".join(chr(rng.randint(qual_lo, qual_hi)) for _ in range(L))
        # quality lines that start with '@' or '+' are legal and must not confuse the boundary repair
        if L and rng.random() < 0.3:
            qual = rng.choice("@+") + qual[1:]
        name = f"r{i}" + (f" desc {rng.randint(0, 99)}" if name_space and rng.random() < 0.5 else "")
        plus = "+" + (name if rng.random() < 0.1 else "")
        out.append(f"@{name}\n{seq}\n{plus}\n{qual}\n")
    s = "".join(out)
    if not final_newline and s.endswith("\n"):
        s = s[:-1]
    return (s + "\n" * trailing_blank).encode()


def random_fasta(rng, nrec, min_len=0, max_len=500, width=60, final_newline=True, trailing_blank=0,
                 alphabet="ACGTN", gt_in_header=False):
    out = []
    for i in range(nrec):
        L = rng.randint(min_len, max_len)
        seq = "".join(rng.choice(alphabet) for _ in range(L))
        if L and rng.random() < 0.1:
            k = rng.randrange(L)
            seq = seq[:k] + rng.choice("-. ") + seq[k + 1:]
        w = width if width > 0 else max(L, 1)
        lines = [seq[j:j + w] for j in range(0, L, w)]
        hdr = f">s{i}" + (" a>b" if gt_in_header and rng.random() < 0.3 else "")
        out.append(hdr + "\n" + "".join(l + "\n" for l in lines))
    s = "".join(out)
    if not final_newline and s.endswith("\n"):
        s = s[:-1]
    return (s + "\n" * trailing_blank).encode()
