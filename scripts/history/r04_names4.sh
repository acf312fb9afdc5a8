#!/usr/bin/env bash
# k_names: the stores of a sink call are waited for with the next tile's data (loads and stores share vmcnt on gfx9, a mixed
# queue is waited down to zero) -- does the pass do better without the register prefetch of the next tile?  LDS-assembled
# output on / off
cd $GRAFT_REPO_ROOT
export BSK_BENCH_PROFILE=1
for f in "" "-DBSK_PREFETCH=0" "-DBSK_NAMES_LDS_OUT=0" "-DBSK_NAMES_LDS_OUT=0 -DBSK_PREFETCH=0"; do bash scripts/variant_src.sh stream_names.hip "$f" seq; done
bash scripts/variant_src.sh stream_names.hip "" grep > /dev/null
for f in "-DBSK_PREFETCH=0"; do BSK_FILTER=off bash scripts/variant_src.sh stream_index.hip "$f" grep; bash scripts/variant_src.sh stream_rmdup.hip "$f" rmdup; bash scripts/variant_src.sh stream_subseq.hip "$f" subseq; done
