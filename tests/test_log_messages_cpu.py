"""The reference's option advice (log.Warn / log.Info of Before(): /root/reference/bigseqkit-lib/seq.go:52-69,
grep.go:57-98 + 140-207, locate.go:50-70 + 96-98 + 143-145, subseq.go:98-100 + 127-133 + 157-159): same texts, same
conditions, and --quiet (Config.Quiet) silences exactly the messages the reference guards with it.  Options-only contexts
(device -1): no GPU needed."""
import json

import bigseqkit_amd as bsk
from bigseqkit_amd._lib import lib


def log_of(name, opts):
    with bsk.Operator(name, json.dumps(opts), -1) as op:
        return lib.bsk_log_text(op.ctx).decode()


def test_seq_messages():
    assert log_of("SeqTransform", {}) == ""
    assert log_of("SeqTransform", {"MinLen": 5}) == "[WARN] you may switch on flag -g/--remove-gaps to remove spaces\n"
    assert log_of("SeqTransform", {"MinLen": 5, "RemoveGaps": True}) == ""
    assert log_of("SeqTransform", {"Complement": True}) == \
        "[WARN] flag -t (--seq-type) (DNA/RNA) is recommended for computing complement sequences\n"
    assert log_of("SeqTransform", {"Complement": True, "Config": {"SeqType": "dna"}}) == \
        "[INFO] when flag -t (--seq-type) given, flag -v (--validate-seq) is automatically switched on\n"
    # the info line is guarded by Quiet (seq.go:67), the warnings are not (seq.go:52-64)
    assert log_of("SeqTransform", {"Config": {"SeqType": "dna", "Quiet": True}}) == ""
    assert log_of("SeqTransform", {"MaxLen": 9, "Config": {"Quiet": True}}) == \
        "[WARN] you may switch on flag -g/--remove-gaps to remove spaces\n"


def test_grep_messages(tmp_path):
    assert log_of("Grep", {"Pattern": ["abc"]}) == ""
    assert log_of("Grep", {"Pattern": ["A{2"]}).startswith("[WARN] possible unquoted comma detected")
    assert log_of("Grep", {"Pattern": ["3}"]}).startswith("[WARN] possible unquoted comma detected")
    assert log_of("Grep", {"Pattern": ["A{2,3}"]}) == ""
    assert log_of("Grep", {"Pattern": ["ACN"], "Degenerate": True}) == \
        "[INFO] when flag -d (--degenerate) given, flag -s (--by-seq) is automatically on\n"
    assert log_of("Grep", {"Pattern": ["ACGTACGT"], "MaxMismatch": 5}) == \
        "[INFO] when value of flag -m (--max-mismatch) > 0, flag -s (--by-seq) is automatically on\n" \
        "[WARN] large value flag -m/--max-mismatch will slow down the search\n"
    assert log_of("Grep", {"Pattern": ["ACG"], "MaxMismatch": 1, "BySeq": True}) == ""
    assert log_of("Grep", {"Pattern": ["ACG"], "Region": "1:5"}) == \
        "[INFO] when flag -R (--region) given, flag -s (--by-seq) is automatically on\n"
    # (these three are NOT guarded by Quiet in the reference: grep.go:70, 80, 96)
    assert log_of("Grep", {"Pattern": ["ACG"], "Region": "1:5", "Config": {"Quiet": True}}) != ""
    assert log_of("Grep", {"Pattern": [">id"]}) == '[WARN] symbol ">" detected, it should not be a part of the sequence ID/name: >id\n'
    assert log_of("Grep", {"Pattern": ["@id"]}) == '[WARN] symbol "@" detected, it should not be a part of the sequence ID/name. @id\n'
    assert log_of("Grep", {"Pattern": ["id 1"]}) == "[WARN] space found in pattern, you may need use -n/--by-name: id 1\n"
    assert log_of("Grep", {"Pattern": ["id 1"], "ByName": True}) == ""
    assert log_of("Grep", {"Pattern": ["id 1"], "Config": {"IDRegexp": "^(\\S+)"}}) == ""   # not the default ID expression
    assert log_of("Grep", {"Pattern": [">id", "a b"], "Config": {"Quiet": True}}) == ""      # guarded (grep.go:199)
    f = tmp_path / "pats.txt"
    f.write_text("id1\n\nid2\n")
    assert log_of("Grep", {"PatternFile": str(f)}) == "[INFO] 2 patterns loaded from file\n"
    assert log_of("Grep", {"PatternFile": str(f), "Config": {"Quiet": True}}) == ""
    e = tmp_path / "none.txt"
    e.write_text("\n\n")
    assert log_of("Grep", {"PatternFile": str(e)}) == "[WARN] 0 patterns loaded from file\n"


def test_locate_messages(tmp_path):
    assert log_of("Locate", {"Pattern": ["ACGT"]}) == ""
    assert log_of("Locate", {"Pattern": ["AC GT"]}) == "[WARN] space found in sequence: 'AC GT'\n"
    assert log_of("Locate", {"Pattern": ["AC GT"], "Config": {"Quiet": True}}) == ""
    assert log_of("Locate", {"Pattern": ["ACGT"], "MaxMismatch": 1, "NonGreedy": True}) == \
        "[INFO] flag -G (--non-greedy) ignored when giving flag -m (--max-mismatch)\n"
    assert log_of("Locate", {"Pattern": ["ACGT"], "MaxMismatch": 1, "NonGreedy": True, "Config": {"Quiet": True}}) == ""
    assert log_of("Locate", {"Pattern": ["AC}"], "UseRegexp": True}).startswith("[WARN] possible unquoted comma detected")
    # a pattern file: bytes.Contains(seq, "\t ") -- tab THEN blank, as written (locate.go:96)
    f = tmp_path / "m.fa"
    f.write_text(">m1\nAC GT\n>m2\nAC\t GT\n")
    assert log_of("Locate", {"PatternFile": str(f), "UseRegexp": True}) == "[WARN] space found in sequence: m2\n"


def test_subseq_feature_messages(tmp_path):
    g = tmp_path / "a.gtf"
    g.write_text("# comment\nchr1\tsrc\tgene\t2\t5\t.\t+\t.\tgene_id \"g1\";\nchr1\tsrc\texon\t3\t4\t.\t-\t.\tgene_id \"g2\";\nshort\tline\n")
    assert log_of("SubseqTransform", {"Gtf": str(g)}) == "[INFO] read GTF file ...\n[INFO] 2 GTF features loaded\n"
    assert log_of("SubseqTransform", {"Gtf": str(g), "Feature": ["exon"]}) == "[INFO] read GTF file ...\n[INFO] 1 GTF features loaded\n"
    assert log_of("SubseqTransform", {"Gtf": str(g), "Config": {"Quiet": True}}) == ""
    b = tmp_path / "a.bed"
    b.write_text("track x\nchr1\t1\t5\nchr2\t0\t3\tname\t0\t-\n")
    assert log_of("SubseqTransform", {"Bed": str(b)}) == "[INFO] read BED file ...\n[INFO] 2 BED features loaded\n"
    assert log_of("SubseqTransform", {"Region": "1:5"}) == ""
