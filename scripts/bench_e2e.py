#!/usr/bin/env python3
"""The end-to-end object of bench.py alone (host bytes -> result and FILE -> result), at its full sample size."""
import argparse, json, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import bench
import bigseqkit_amd as bsk
from bigseqkit_amd import _lib
from bigseqkit_amd._lib import lib, check
ap = argparse.ArgumentParser()
ap.add_argument("--gb", type=float, default=8.0)
a = ap.parse_args()
args = argparse.Namespace(ops_calls=5, ops_scale=1.0)
torch.cuda.set_device(0)
H = bench._Helpers(args, torch, bsk, _lib, lib, check, torch.device("cuda", 0), 0)
print(json.dumps(bench.run_end_to_end(H, a.gb), indent=1))
