"""The 24 NCBI genetic codes, pinned from two sides.

* The PRODUCT holds them as gc.prt lines (ncbieaa / sncbieaa, base order TCAG):
  bigseqkit_amd/csrc/genetic_codes.inc.
* The ORACLE derives the same lines from a different notation -- the standard code as a codon list
  (alphabetical over ACGT) plus, per table, the differences NCBI documents and the list of initiation
  codons: oracle/genetic_codes_diff.inc.
This file compares the two (24 x 2 x 64 letters) and, independently of both, holds the oracle's
translation of single codons to hand-written known answers taken from NCBI's "The Genetic Codes" page
("Differences from the Standard Code"), and to the reference's own ambiguous-codon examples
(/root/reference/bigseqkit-cli/translate.go:42-52).  [source: hand]
"""
import os
import re

import pytest

import oracle

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TABLE_IDS = [1, 2, 3, 4, 5, 6, 9, 10, 11, 12, 13, 14, 16, 21, 22, 23, 24, 25, 26, 27, 28, 29, 30, 31]  # cli/translate.go:55-78


def product_tables():
    text = open(os.path.join(ROOT, "bigseqkit_amd", "csrc", "genetic_codes.inc")).read()
    rows = re.findall(r'\{(\d+),\s*"([A-Z*]{64})",\s*"([-M*]{64})"\}', text)
    return {int(i): (aa, st) for i, aa, st in rows}


def test_product_and_oracle_tables_agree_letter_for_letter():
    prod = product_tables()
    assert sorted(prod) == TABLE_IDS
    for tid in TABLE_IDS:
        got = oracle.genetic_code(tid)
        assert got is not None, tid
        assert got[0] == prod[tid][0], "ncbieaa of table %d" % tid
        assert got[1] == prod[tid][1], "sncbieaa of table %d" % tid
    assert oracle.genetic_code(7) is None and oracle.genetic_code(32) is None


# codon -> amino acid where a table differs from the standard code (hand-typed from NCBI's page, RNA letters as DNA)
DIFFS = {
    2: {"AGA": "*", "AGG": "*", "ATA": "M", "TGA": "W"},
    3: {"ATA": "M", "CTT": "T", "CTC": "T", "CTA": "T", "CTG": "T", "TGA": "W"},
    4: {"TGA": "W"},
    5: {"AGA": "S", "AGG": "S", "ATA": "M", "TGA": "W"},
    6: {"TAA": "Q", "TAG": "Q"},
    9: {"AAA": "N", "AGA": "S", "AGG": "S", "TGA": "W"},
    10: {"TGA": "C"},
    11: {},
    12: {"CTG": "S"},
    13: {"AGA": "G", "AGG": "G", "ATA": "M", "TGA": "W"},
    14: {"AAA": "N", "AGA": "S", "AGG": "S", "TAA": "Y", "TGA": "W"},
    16: {"TAG": "L"},
    21: {"TGA": "W", "ATA": "M", "AGA": "S", "AGG": "S", "AAA": "N"},
    22: {"TCA": "*", "TAG": "L"},
    23: {"TTA": "*"},
    24: {"AGA": "S", "AGG": "K", "TGA": "W"},
    25: {"TGA": "G"},
    26: {"CTG": "A"},
    27: {"TAG": "Q", "TAA": "Q", "TGA": "W"},
    28: {"TAA": "Q", "TAG": "Q", "TGA": "W"},
    29: {"TAA": "Y", "TAG": "Y"},
    30: {"TAA": "E", "TAG": "E"},
    31: {"TGA": "W", "TAG": "E", "TAA": "E"},
}
# the standard code by amino acid (textbook table), a third notation
STANDARD = {
    "F": "TTT TTC", "L": "TTA TTG CTT CTC CTA CTG", "I": "ATT ATC ATA", "M": "ATG", "V": "GTT GTC GTA GTG",
    "S": "TCT TCC TCA TCG AGT AGC", "P": "CCT CCC CCA CCG", "T": "ACT ACC ACA ACG", "A": "GCT GCC GCA GCG",
    "Y": "TAT TAC", "*": "TAA TAG TGA", "H": "CAT CAC", "Q": "CAA CAG", "N": "AAT AAC", "K": "AAA AAG",
    "D": "GAT GAC", "E": "GAA GAG", "C": "TGT TGC", "W": "TGG", "R": "CGT CGC CGA CGG AGA AGG", "G": "GGT GGC GGA GGG",
}


def test_standard_code_all_64_codons():
    seen = 0
    for aa, codons in STANDARD.items():
        for c in codons.split():
            assert oracle.translate_seq(c, 1) == aa, c
            assert oracle.translate_seq(c.lower().replace("t", "u"), 1) == aa, c  # case-insensitive, U == T
            seen += 1
    assert seen == 64


@pytest.mark.parametrize("tid", sorted(DIFFS))
def test_every_codon_of_every_table_against_the_documented_differences(tid):
    std = {c: aa for aa, cs in STANDARD.items() for c in cs.split()}
    for codon, aa in std.items():
        want = DIFFS[tid].get(codon, aa)
        assert oracle.translate_seq(codon, tid) == want, (tid, codon)


# initiation codons per table (NCBI's page), used by -M / --init-codon-as-M
STARTS = {
    1: "TTG CTG ATG", 2: "ATT ATC ATA ATG GTG", 3: "ATA ATG GTG", 4: "TTA TTG CTG ATT ATC ATA ATG GTG",
    5: "TTG ATT ATC ATA ATG GTG", 6: "ATG", 9: "ATG GTG", 10: "ATG", 11: "TTG CTG ATT ATC ATA ATG GTG", 12: "CTG ATG",
    13: "TTG ATA ATG GTG", 14: "ATG", 16: "ATG", 21: "ATG GTG", 22: "ATG", 23: "ATT ATG GTG", 24: "TTG CTG ATG GTG",
    25: "TTG ATG GTG", 26: "CTG ATG", 27: "ATG", 28: "ATG", 29: "ATG", 30: "ATG", 31: "ATG",
}


@pytest.mark.parametrize("tid", sorted(STARTS))
def test_initiation_codons_become_M_only_with_init_m(tid):
    std = {c: aa for aa, cs in STANDARD.items() for c in cs.split()}
    starts = set(STARTS[tid].split())
    for codon, aa in std.items():
        plain = DIFFS.get(tid, {}).get(codon, aa)
        want = "M" if codon in starts else plain
        assert oracle.translate_seq(codon + "GGG", tid, init_m=True) == want + "G", (tid, codon)
        assert oracle.translate_seq("GGG" + codon, tid, init_m=True) == "G" + plain, (tid, codon)  # only the first codon


def test_reference_ambiguous_codon_examples():
    # the ten examples in the reference's help text, bigseqkit-cli/translate.go:42-52 ("for standard table")
    for codon, aa in {"ACN": "T", "CCN": "P", "CGN": "R", "CTN": "L", "GCN": "A", "GGN": "G", "GTN": "V", "TCN": "S",
                      "MGR": "R", "YTR": "L"}.items():
        assert oracle.translate_seq(codon, 1) == aa, codon
    # expansions that disagree -> X with -x (an error without it)
    assert oracle.translate_seq("RAY", 1, allow_unknown=True) == "X"
    assert oracle.translate_seq("NNN", 1, allow_unknown=True) == "X"
    with pytest.raises(oracle.OracleError):
        oracle.translate_seq("A-G", 1)
