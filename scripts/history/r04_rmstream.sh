#!/usr/bin/env bash
# what the hashing costs inside k_rmdup_stream: no hash at all / k1 only / k1 + k2, at 4 and 5 waves per SIMD
cd $GRAFT_REPO_ROOT
export BSK_BENCH_PROFILE=1
for f in "" "-DBSK_RMSTREAM_K2=0" "-DBSK_RMSTREAM_K2=0 -DBSK_RMSTREAM_WAVES=5" "-DBSK_RMSTREAM_NOHASH=1" "-DBSK_RMSTREAM_NOHASH=1 -DBSK_RMSTREAM_WAVES=5"; do
  bash scripts/variant_src.sh stream_rmdup.hip "$f" rmdup
done
