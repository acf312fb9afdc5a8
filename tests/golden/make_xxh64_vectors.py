#!/usr/bin/env python3
"""Generates tests/golden/xxh64_vectors.json with the python `xxhash` package (3.8.1 here).
XXH64 with seed 0 is what github.com/cespare/xxhash/v2 Sum64 computes
(/root/reference/bigseqkit-lib/rmdup.go:69-83).  Run in the build container; the
vectors (data, not code) travel to the GPU box."""
import json
import random

import xxhash

rng = random.Random(64)
vec = []
fixed = [b"", b"a", b"ACGT", b"ACGT" * 9, b"N" * 31, b"N" * 32, b"N" * 33]
for n in list(range(0, 70)) + [96, 127, 128, 150, 151, 255, 256, 1000, 5001]:
    fixed.append(bytes(rng.choice(b"ACGTacgtN") for _ in range(n)))
for b in fixed:
    vec.append({"hex": b.hex(), "xxh64": "%016x" % xxhash.xxh64(b, seed=0).intdigest()})
json.dump({"generator": "python-xxhash " + xxhash.VERSION, "seed": 0, "vectors": vec},
          open(__file__.replace("make_xxh64_vectors.py", "xxh64_vectors.json"), "w"), indent=0)
print(len(vec), "vectors")
