#!/usr/bin/env bash
# non-temporal tile loads, the other streaming passes: FASTA stats (same file as k_stats: on with it), k_index (FASTQ via
# grep with the filter off, FASTA via translate_index=full), k_subseq_stream, the FASTA light pass
cd $GRAFT_REPO_ROOT
export BSK_BENCH_PROFILE=1
echo "== FASTA stats, stream_stats.hip as committed (nt on)"; python scripts/bench_stats_fasta.py | tail -1
bash scripts/variant.sh "-DBSK_LOAD_NT=0" >/dev/null; echo "== FASTA stats, nt off"; python scripts/bench_stats_fasta.py | tail -1
bash scripts/variant.sh "" >/dev/null
for f in "" "-DBSK_LOAD_NT=1"; do
  BSK_FILTER=off bash scripts/variant_src.sh stream_index.hip "$f" grep
  BSK_TRANSLATE_INDEX=full bash scripts/variant_src.sh stream_index.hip "$f" translate
done
bash scripts/variant_src.sh stream_index.hip "" grep >/dev/null
for f in "" "-DBSK_LOAD_NT=1"; do bash scripts/variant_src.sh stream_subseq.hip "$f" subseq; done
bash scripts/variant_src.sh stream_subseq.hip "" grep >/dev/null
