#!/usr/bin/env bash
# k_names moves 125 - 145 GB for a 100 GB file (the header sectors are fetched a second time): does a sink that runs sooner
# find them in L2?  smaller windows, both schedules
cd $GRAFT_REPO_ROOT
export BSK_BENCH_PROFILE=1
for f in "" "-DBSK_NAMES_WINDOW=128 -DBSK_NAMES_TE=0" "-DBSK_NAMES_WINDOW=64 -DBSK_NAMES_TE=0" "-DBSK_NAMES_WINDOW=128 -DBSK_NAMES_TE=1" "-DBSK_NAMES_WINDOW=192 -DBSK_NAMES_TE=1" "-DBSK_NAMES_WINDOW=256 -DBSK_NAMES_TE=0 -DBSK_NAMES_WAVES=5"; do
  bash scripts/variant_src.sh stream_names.hip "$f" seq
done
bash scripts/variant_src.sh stream_names.hip "" grep > /dev/null
