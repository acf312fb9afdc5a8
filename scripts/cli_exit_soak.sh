#!/usr/bin/env bash
# 600 short runs of the command line: does any of them die on the way out?  (round 3: one CLI run in ~200 ended with SIGSEGV
# after its output was complete -- the teardown of the HIP runtime at exit(); the CLI now leaves through _exit)
cd "$(dirname "$0")/.."
python - <<PY
import random
rng=random.Random(5)
open("/tmp/exit_soak.fq","w").write("".join("@r%d\n%s\n+\n%s\n"%(i,"".join(rng.choice("ACGT") for _ in range(80)),"I"*80) for i in range(2000)))
PY
bad=0
for i in $(seq 1 600); do
  case $((i % 4)) in
    0) ./bigseqkit_amd/bin/bigseqkit stats -T /tmp/exit_soak.fq > /dev/null 2>/tmp/exit_soak.err ;;
    1) ./bigseqkit_amd/bin/bigseqkit grep -s -p ACGTAC /tmp/exit_soak.fq -o - > /dev/null 2>/tmp/exit_soak.err ;;
    2) ./bigseqkit_amd/bin/bigseqkit seq -n /tmp/exit_soak.fq -o - > /dev/null 2>/tmp/exit_soak.err ;;
    3) ./bigseqkit_amd/bin/bigseqkit rmdup -s /tmp/exit_soak.fq -o - > /dev/null 2>/tmp/exit_soak.err ;;
  esac
  rc=$?
  if [ $rc -ne 0 ]; then bad=$((bad+1)); echo "run $i rc=$rc"; head -5 /tmp/exit_soak.err; fi
done
echo "runs 600 failed $bad"
