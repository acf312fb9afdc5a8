#!/usr/bin/env bash
# round 6, A/B 4g: instructions per tile of k_names by variant (deterministic, where the times spread by a millisecond)
cd "$(dirname "$0")/.."
{
for v in "-DBSK_NAMES_TE=0 -DBSK_NAMES_WINDOW=256 -DBSK_NAMES_DIAG=2 -DBSK_NAMES_HEAD16=0" "-DBSK_NAMES_DIAG=2 -DBSK_NAMES_HEAD16=0" "-DBSK_NAMES_TE=0 -DBSK_NAMES_WINDOW=256" "-DBSK_NAMES_TE=0 -DBSK_NAMES_WINDOW=256 -DBSK_NAMES_HEAD16=0"; do
  BSK_OUT=slices bash scripts/variant_valu.sh stream_names.hip "$v" seq k_names 100e9
done
} > gpurun_out/r06_ab4g.log 2>&1
grep "^==" gpurun_out/r06_ab4g.log
