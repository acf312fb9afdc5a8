#!/usr/bin/env bash
# names sink: deferral mode / window (after the probe loads went away k_names rose 20.4 -> 22.1 ms)
cd $GRAFT_REPO_ROOT
export BSK_BENCH_PROFILE=1
bash scripts/variant_src.sh stream_names.hip "" seq
bash scripts/variant_src.sh stream_names.hip "-DBSK_NAMES_WINDOW=256 -DBSK_NAMES_TE=0" seq
bash scripts/variant_src.sh stream_names.hip "-DBSK_NAMES_WINDOW=256 -DBSK_NAMES_TE=1" seq
bash scripts/variant_src.sh stream_names.hip "-DBSK_NAMES_WINDOW=384 -DBSK_NAMES_TE=1" seq
