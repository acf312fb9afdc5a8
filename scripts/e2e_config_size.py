#!/usr/bin/env python3
"""FILE -> result on the whole C2 file in /dev/shm (bench.py's end_to_end_config_size leg on its own).  Usage: e2e_config_size.py [GB]"""
import argparse, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
import bigseqkit_amd as bsk
from bigseqkit_amd import _lib
from bigseqkit_amd._lib import lib, check
gb = float(sys.argv[1]) if len(sys.argv) > 1 else 100.0
args = argparse.Namespace(ops_calls=1, cpu_seconds=1.0, no_cpu_baseline=True, ops_scale=1.0)
H = bench._Helpers(args, torch, bsk, _lib, lib, check, torch.device("cuda", 0), 0)
print(json.dumps(bench.file_to_result_config_size(H, gb), indent=1))
