"""Pins the CPU oracle against the known-answer material the reference holds
 and against hand-derived cases.  No GPU."""
import random

import pytest

import oracle
import seqgen

FQ = (b"@r1 d\nACGT\n+\nIIII\n"
      b"@r2\nAC-N\n+\n5#5I\n")


def test_wrap_byte_slice_in_tree_algorithm():
    # bigseqkit-lib/helper.go:81-117
    assert oracle.wrap(b"ACGTACGTAC", 4) == b"ACGT\nACGT\nAC"
    assert oracle.wrap(b"ACGTACGTAC", 5) == b"ACGTA\nCGTAC"
    assert oracle.wrap(b"ACGTACGTAC", 10) == b"ACGTACGTAC"
    assert oracle.wrap(b"ACGTACGTAC", 0) == b"ACGTACGTAC"
    assert oracle.wrap(b"ACGTACGTAC", 60) == b"ACGTACGTAC"
    assert oracle.wrap(b"", 4) == b""


def test_parse_head_id_and_desc_in_tree_algorithm():
    # bigseqkit-lib/helper.go:329-369, including the skip-two-per-iteration loop
    assert oracle.parse_head("id1 desc here") == ("id1", "desc here")
    assert oracle.parse_head("id1  two") == ("id1", "wo")
    assert oracle.parse_head("id\tx") == ("id", "x")
    assert oracle.parse_head("nospace") == ("nospace", "")
    assert oracle.parse_head(" lead") == (" lead", "")
    assert oracle.parse_head("id ") == ("id", "")
    # --id-ncbi regexp (bigseqkit/helper.go:97-100)
    ncbi = r"\|([^\|]+)\| "
    assert oracle.parse_head("gi|110645304|ref|NC_002516.2| Pseudomonas", ncbi) == ("NC_002516.2", "")


def test_stats_map_hand_case():
    m = oracle.stats_map(FQ, True, '{"All": true}')
    assert m == {4: 2, -1: 7, -2: 5, -3: 1, -4: ord("D")}
    m = oracle.stats_map(FQ, True, "{}")
    assert m == {4: 2, -4: ord("D")}


def test_stats_string_tabular_hand_case():
    s = oracle.stats_string(FQ, True, '{"All": true, "Tabular": true}')
    head, row, _ = s.split("\n")
    assert head == "file\tformat\ttype\tnum_seqs\tsum_len\tmin_len\tavg_len\tmax_len\tQ1\tQ2\tQ3\tsum_gap\tN50\tQ20(%)\tQ30(%)"
    assert row == "input0\tN/A\tDNA\t2\t8\t4\t4.0\t4\t4.0\t4.0\t4.0\t1\t4\t87.50\t62.50"


def test_stats_quartiles_and_n50():
    fa = b"".join(b">s%d\n%s\n" % (i, b"A" * i) for i in range(1, 6))
    s = oracle.stats_string(fa, False, '{"All": true, "Tabular": true}')
    row = s.split("\n")[1].split("\t")
    assert row[3:8] == ["5", "15", "1", "3.0", "5"]
    assert row[8:11] == ["2.0", "3.0", "4.0"]  # Q1 Q2 Q3
    assert row[12] == "4"  # N50


def test_expected_c2_row_from_baseline_md():
    # expected row for the 100 GB FASTQ-150 file
    m = {150: 315457413, -4: ord("D")}
    s = oracle.stats_string_from_map(m, b"@S0000000000\nACGT\n+\nIIII", '{"Tabular": true}')
    assert s.split("\n")[1] == "input0\tN/A\tDNA\t315457413\t47318611950\t150\t150.0\t150"


def test_pretty_table_and_humanize():
    m = {150: 1234567, -4: ord("D")}
    s = oracle.stats_string_from_map(m, b"@a\nA\n+\nI", "{}")
    lines = s.split("\n")
    assert lines[0].split() == ["file", "format", "type", "num_seqs", "sum_len", "min_len", "avg_len", "max_len"]
    assert lines[1].split() == ["input0", "N/A", "DNA", "1,234,567", "185,185,050", "150", "150", "150"]
    assert len(lines[0]) == len(lines[1])


def test_type_column():
    assert oracle.stats_map(b">p\nMKVLAAGIVGLLLAQ\n", False)[-4] == ord("F")
    s = oracle.stats_string(b">p\nMKVLAAGIVGLLLAQ\n", False, '{"Tabular": true}')
    assert s.split("\n")[1].split("\t")[2] == "Protein"
    assert oracle.stats_map(b">r\nACGUACGU\n", False)[-4] == ord("R")
    assert oracle.stats_map(b"", False) == {-4: ord("U")}
    assert oracle.stats_map(b">x\nACGT\n", False, '{"Config": {"SeqType": "rna"}}')[-4] == ord("R")


def test_split_edge_cases():
    assert oracle.count_records(b"", True) == 0
    assert oracle.count_records(b"\n\n", True) == 0
    # no final newline, '@' / '+' leading quality lines
    fq = b"@a\nAC\n+\n@+\n@b\nGT\n+a\n+@"
    assert oracle.record_spans(fq, True) == [(0, 10), (11, 11)]
    assert oracle.is_strict_4line_fastq(fq)
    # empty sequence, trailing blank lines
    fq = b"@a\n\n+\n\n@b\nA\n+\nI\n\n\n"
    assert oracle.count_records(fq, True) == 2
    assert oracle.stats_map(fq, True) == {0: 1, 1: 1, -4: ord("F")}
    assert oracle.is_strict_4line_fastq(fq)
    # multi-line FASTQ is parsed by the oracle but is not strict
    ml = b"@a\nACGT\nAC\n+\nIIII\nII\n"
    assert oracle.count_records(ml, True) == 1
    assert oracle.stats_map(ml, True)[6] == 1
    assert not oracle.is_strict_4line_fastq(ml)
    # '>' inside a header does not start a record (PARITY.md SPLIT)
    fa = b">s1 a>b\nAC\nGT\n>s2\n\n>s3"
    assert oracle.record_spans(fa, False) == [(0, 13), (14, 4), (19, 3)]
    assert oracle.stats_map(fa, False) == {4: 1, 0: 2, -4: ord("D")}


def test_unmatched_quality_length_is_an_error():
    with pytest.raises(oracle.OracleError, match="unmatched length"):
        oracle.stats_map(b"@a\nACGT\n+\nIII\n@b\nA\n+\nI\n", True)


def test_option_errors():
    with pytest.raises(oracle.OracleError, match="should not be empty"):
        oracle.stats_map(FQ, True, '{"GapLetters": ""}')
    with pytest.raises(oracle.OracleError, match="invalid sequence type"):
        oracle.stats_map(FQ, True, '{"Config": {"SeqType": "dnaa"}}')
    with pytest.raises(oracle.OracleError, match="unsupported quality encoding"):
        oracle.stats_map(FQ, True, '{"FqEncoding": "phred"}')


def test_partitions_reduce_sums():
    rng = random.Random(7)
    fq = seqgen.random_fastq(rng, 200, 0, 80)
    a = oracle.stats_map(fq, True, '{"All": true}', nparts=1)
    b = oracle.stats_map(fq, True, '{"All": true}', nparts=7)
    assert a == b


def test_solexa_offset():
    fq = b"@a\nACGT\n+\nTT^^\n"  # 'T'=84 -> 20 at offset 64, '^'=94 -> 30
    assert oracle.stats_map(fq, True, '{"All": true, "FqEncoding": "solexa"}') == {4: 1, -1: 4, -2: 2, -3: 0, -4: ord("D")}
    assert oracle.stats_map(fq, True, '{"All": true}')[-2] == 4


def test_xxh64_golden_vectors():
    # tests/golden/xxh64_vectors.json: generated with python-xxhash (make_xxh64_vectors.py);
    # xxh64("") = ef46db3751d8e999, xxh64("ACGT") = f40a8ecfa26af897
    import json
    import os
    g = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "xxh64_vectors.json")))
    assert len(g["vectors"]) > 80
    for v in g["vectors"]:
        assert "%016x" % oracle.xxh64(bytes.fromhex(v["hex"])) == v["xxh64"]
    assert "%016x" % oracle.xxh64(b"") == "ef46db3751d8e999"
    assert "%016x" % oracle.xxh64(b"ACGT") == "f40a8ecfa26af897"


def test_region_table_kat():
    # bigseqkit-cli/helper.go:348-361
    s = b"ACGTNacgtn"
    for r, w in [("1:1", "A"), ("2:4", "CGT"), ("-4:-2", "cgt"), ("-4:-1", "cgtn"), ("-1:-1", "n"),
                 ("2:-2", "CGTNacgt"), ("1:-1", "ACGTNacgtn"), ("1:12", "ACGTNacgtn"), ("-12:-1", "ACGTNacgtn")]:
        a, b = map(int, r.split(":"))
        x, y = oracle.sub_location(10, a, b)
        assert s[x:y].decode() == w, r
        assert oracle.subseq(b">s\n" + s + b"\n", False, '{"Region": "%s"}' % r) == b">s\n" + w.encode() + b"\n"


def test_ambiguous_codon_kat_and_frames():
    # bigseqkit-cli/translate.go:42-52
    for c, a in [("ACN", "T"), ("CCN", "P"), ("CGN", "R"), ("CTN", "L"), ("GCN", "A"), ("GGN", "G"), ("GTN", "V"),
                 ("TCN", "S"), ("MGR", "R"), ("YTR", "L")]:
        assert oracle.translate_seq(c) == a
    assert oracle.translate_seq("ATGGCCTAAGG") == "MA*"
    assert oracle.translate_seq("ATGGCCTAAGG", frame=2) == "WPK"
    assert oracle.translate_seq("ATGGCCTAAGG", frame=-1) == "P*A"
    assert oracle.translate_seq("AUGGCCUAA") == "MA*"          # RNA: U == T
    assert oracle.translate_seq("atggcctaa") == "MA*"
    assert oracle.translate_seq("ATGNNNTAA", clean=True) == "MXX"
    assert oracle.translate_seq("ATGTAATAA", trim=True) == "M"
    assert oracle.translate_seq("TTGGCC", init_m=True) == "MA"   # TTG is a start codon of table 1
    assert oracle.translate_seq("TGA", table=2) == "W" and oracle.translate_seq("TGA", table=1) == "*"
    with pytest.raises(oracle.OracleError):
        oracle.translate_seq("AT-GCC")
    assert oracle.translate_seq("AT-GCC", allow_unknown=True) == "XA"
    fa = b">s1 d\nATGGCCTAAGGA\n"
    out = oracle.translate(fa, False, '{"Frame": ["6"], "AppendFrame": true}')
    assert out == (b">s1_frame=1 d\nMA*G\n>s1_frame=2 d\nWPK\n>s1_frame=3 d\nGLR\n"
                   b">s1_frame=-1 d\nSLGH\n>s1_frame=-2 d\nP*A\n>s1_frame=-3 d\nLRP\n")


def test_rmdup_hand_cases():
    fa = b">a\nACGT\n>b\nacgt\n>c\nACGT\n>d\nTTTT\n"
    assert oracle.rmdup(fa, False, '{"BySeq": true}') == b">a\nACGT\n>b\nacgt\n>d\nTTTT\n"
    assert oracle.rmdup(fa, False, '{"BySeq": true, "IgnoreCase": true}') == b">a\nACGT\n>d\nTTTT\n"
    assert oracle.rmdup(b">a x\nA\n>a y\nC\n>b\nG\n", False, "{}") == b">a x\nA\n>b\nG\n"
    assert oracle.rmdup(b">a x\nA\n>a y\nC\n>a x\nG\n", False, '{"ByName": true}') == b">a x\nA\n>a y\nC\n"


def test_codon_lookup_table_equals_the_definition_on_every_triple():
    """translate_seq looks amino acids up in a 4 096-entry table per genetic code (round 6: the CPU baseline of `translate`);
    the table is codon_aa -- the cited definition, bio's ambiguous-codon rule -- evaluated on every triple: held to it here
    on all 33^3 triples over both cases of the 15 IUPAC letters, U / u and an invalid byte, for every NCBI table"""
    checked = 0
    for table in range(1, 34):
        bad = oracle._lib.orc_codon_table_mismatches(table)
        if bad < 0:
            continue
        assert bad == 0, table
        checked += 1
    assert checked >= 24


def test_rmdup_on_several_threads_equals_the_sequential_restatement():
    """rmdup_call_mt (bench.py's all-cores CPU baseline of `rmdup`: per-thread parsing, key groups settled by the thread that
    owns key % threads) against rmdup_call on random inputs, every subject kind, FASTA and FASTQ"""
    import random
    import seqgen
    rng = random.Random(606)
    for it in range(30):
        fastq = rng.random() < 0.5
        gen = (lambda r, k: seqgen.random_fastq(r, k, 0, 40)) if fastq else (lambda r, k: seqgen.random_fasta(r, k, 0, 90))
        seed, k = rng.randrange(1 << 30), rng.randint(0, 200)
        first, again, other = gen(random.Random(seed), k), gen(random.Random(seed), k), gen(random.Random(seed + 1), rng.randint(0, 100))
        assert first == again
        data = first + other + (again.lower() if (rng.random() < 0.3 and not fastq) else again)   # every record of `first` once more
        for opts in ({"BySeq": True}, {"BySeq": True, "IgnoreCase": True}, {}, {"ByName": True}, {"BySeq": True, "OnlyPositiveStrand": True}):
            import json as _json
            want = oracle.rmdup(data, fastq, _json.dumps(opts))
            for threads in (1, 3, 8):
                assert oracle.rmdup_mt(data, fastq, _json.dumps(opts), threads) == want, (it, opts, threads)
