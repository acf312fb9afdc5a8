"""CPU stand-in for the four device phases of the multi-GPU rmdup (TEST infrastructure only: lets the world_size-2
gloo test exercise bigseqkit_amd.dist.rmdup_distributed -- routing, split sizes, reply mapping -- without a GPU).
Keys come from the oracle's XXH64; strict 4-line FASTQ, subject = sequence (-s) or ID (default)."""
import torch

import oracle

M64 = (1 << 64) - 1
SEED2 = 0x9E3779B97F4A7C15


def _s64(u):
    return u - (1 << 64) if u >= (1 << 63) else u


class OracleRmDupBackend:
    def __init__(self, opts):
        self.by_seq = bool(opts.get("BySeq"))
        self.fold = bool(opts.get("IgnoreCase"))

    def keys(self, shard, fmt):
        import xxhash
        data = bytes(shard.numpy().tobytes())
        lines = data.split(b"\n")
        if lines and lines[-1] == b"":
            lines.pop()
        assert len(lines) % 4 == 0
        self.records, self.k1, self.k2, self.subjects = [], [], [], []
        for r in range(0, len(lines), 4):
            self.records.append(b"\n".join(lines[r:r + 4]) + b"\n")
            subject = lines[r + 1] if self.by_seq else lines[r][1:].split(b" ")[0]
            if self.fold:
                subject = subject.lower()
            self.subjects.append(subject)
            self.k1.append(oracle.xxh64(subject))
            self.k2.append(xxhash.xxh64(subject, seed=SEED2).intdigest())
        return len(self.records)

    def pack(self, base, world):
        buckets = [[] for _ in range(world)]
        for i, k in enumerate(self.k1):
            buckets[k % world].append((_s64(k), _s64(self.k2[i]), base + i))
        rows = [t for b in buckets for t in b]
        send = torch.tensor(rows, dtype=torch.int64).reshape(len(rows), 3)
        return send, [len(b) for b in buckets]

    def resolve(self, tuples):
        first = {}
        rows = tuples.tolist()
        for k, k2, g in rows:
            cur = first.get(k)
            if cur is None or g < cur[0]:
                first[k] = (g, k2)
            assert first[k][1] == k2 or cur is None or cur[1] == k2
        return torch.tensor([1 if first[k][0] == g else 0 for k, k2, g in rows], dtype=torch.uint8)

    def resolve_ex(self, tuples):
        """resolve + the global index of every tuple's survivor (what bsk_rmdup_dist_resolve_ex answers)"""
        keep = self.resolve(tuples)
        first = {}
        rows = tuples.tolist()
        for k, k2, g in rows:
            first[k] = min(first.get(k, g), g)
        return keep, torch.tensor([first[k] for k, k2, g in rows], dtype=torch.int64)

    def emit(self, send, reply, base, to_host=True, surv_reply=None):
        keep = {}
        self.local_pairs = 0
        for p, ((k, k2, g), r) in enumerate(zip(send.tolist(), reply.tolist())):
            keep[g - base] = r
            if surv_reply is not None:
                s = int(surv_reply[p])
                assert (s == g) == bool(r)
                if not r and base <= s < base + len(self.records):   # the survivor lives in this shard: the bytes must agree
                    assert self.subjects[s - base] == self.subjects[g - base]
                    self.local_pairs += 1
        text = b"".join(rec for i, rec in enumerate(self.records) if keep[i])
        if to_host:
            return text
        return ResidentText(text, sum(1 for i in range(len(self.records)) if keep[i]))


class ResidentText:
    """what HipRmDupBackend.emit(to_host=False) returns (dist.DeviceText), on the host: the text stays with the backend"""

    def __init__(self, text, records):
        self.text, self.len, self.records = text, len(text), records

    def __bytes__(self):
        return self.text
