#!/usr/bin/env bash
# index sink (k_index on FASTQ): deferral mode / window; measured through seq -r -p (25 GB), records (range), grepid
cd $GRAFT_REPO_ROOT
export BSK_BENCH_PROFILE=1
bash scripts/variant_src.sh stream_index.hip "" seqrc,grepid
bash scripts/variant_src.sh stream_index.hip "-DBSK_INDEX_WINDOW=384" seqrc,grepid
bash scripts/variant_src.sh stream_index.hip "-DBSK_INDEX_WINDOW=256 -DBSK_INDEX_TE=0" seqrc,grepid
