#!/usr/bin/env bash
# round 5: the native driver after bsk_shard_load + no communicator library for a lone worker: tests, then the wall clock
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_devices_native_gpu.py tests/test_run_multi_gpu.py -q -x -m gpu 2>&1 | tail -5
python - <<PY
import ctypes as C, torch
from bigseqkit_amd import _lib
from bigseqkit_amd._lib import lib, check
n = 8_000_000_000 // 317 * 317
t = torch.empty(n, dtype=torch.uint8, device="cuda")
check(lib.bsk_synth_device(_lib.SYNTH_FASTQ150, 42, _lib.SYNTH_FLAG_MOTIF, 0, C.c_void_p(t.data_ptr()), n, 0, None))
torch.cuda.synchronize()
h = t.cpu().numpy()
with open("/dev/shm/in.fastq", "wb", buffering=0) as f:
    at = 0
    while at < n: at += f.write(memoryview(h)[at:at + (256 << 20)])
PY
CLI=bigseqkit_amd/bin/bigseqkit
for k in 1 2; do ( time BSK_CLI_TIMING=1 $CLI stats -T /dev/shm/in.fastq --devices 0 ) 2>&1 | grep -v amdgpu.ids; done
( time BSK_CLI_TIMING=1 $CLI grep -s -p ACGTTGCAAGCT /dev/shm/in.fastq -o /dev/shm/out.fastq --merge --devices 0 ) 2>&1 | grep -v amdgpu.ids
( time BSK_CLI_TIMING=1 $CLI seq -n /dev/shm/in.fastq -o /dev/shm/out.fastq --merge --devices 0 ) 2>&1 | grep -v amdgpu.ids
for T in 2 4 16 32; do echo "readers via piece env: BSK_SHARD_PIECE_BYTES=$((T << 20))"; ( time BSK_SHARD_PIECE_BYTES=$((T << 20)) $CLI stats -T /dev/shm/in.fastq --devices 0 ) 2>&1 | grep real; done
rm -f /dev/shm/in.fastq /dev/shm/out.fastq
