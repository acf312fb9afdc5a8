import sys, json, random
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import torch
import bigseqkit_amd as bsk
class O:
    def __init__(s, d): s.d = d; s._v = d
    def to_json(s): return json.dumps(s.d)
rng = random.Random(1)
bad = []
for width in (60, 0):
    for L in [0, 1, 139, 190, 193, 194, 195, 200, 400, 767]:
        s = "".join(rng.choice("ACGT") for _ in range(L))
        w = width or max(1, L)
        data = (">r\n" + "".join(s[j:j + w] + "\n" for j in range(0, L, w))).encode()
        t = torch.frombuffer(bytearray(data), dtype=torch.uint8).cuda()
        for fr in (["1"], ["-1"]):
            try:
                bsk.Translate(bsk.SeqFrame(bsk.FORMAT_FASTA, [t]), O({"Frame": fr}))
            except Exception as e:
                bad.append((width, L, fr[0], str(e)[:60]))
print(bad[:80], len(bad))
