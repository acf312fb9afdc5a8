"""The headline kernels must not spill: `k_stats<FASTQ, default, DPP>` runs at 7 waves per SIMD with every value in registers.
A harmless-looking extra branch in its range loop once cost two spilled VGPRs and 1 ms of 17 at 100 GB without any test
noticing; this compiles the file for gfx950 (no GPU needed) and reads the compiler's resource report."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not installed")
def test_k_stats_keeps_its_registers(tmp_path):
    src = os.path.join(ROOT, "bigseqkit_amd", "csrc", "stream_stats.hip")
    r = subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Rpass-analysis=kernel-resource-usage",
                        "-c", src, "-o", str(tmp_path / "s.o")], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    blocks = re.split(r"remark: Function Name: ", r.stderr)
    seen = set()
    # k_stats<FASTQ = true, ALL = false, DPP = true>: the variant that runs, 7 waves per SIMD;
    # k_stats<true, true, true>: `stats -a` by line roles, VALU-bound at 5 waves (6 waves spilled 178 registers: 44 ms for 25)
    want = {"7k_statsILb1ELb0ELb1E": 7, "7k_statsILb1ELb1ELb1E": 5}
    for b in blocks:
        name = b.split(" ", 1)[0]
        key = next((k for k in want if k in name), None)
        if key is None:
            continue
        seen.add(key)
        scratch = int(re.search(r"ScratchSize \[bytes/lane\]: (\d+)", b).group(1))
        spill = int(re.search(r"VGPRs Spill: (\d+)", b).group(1))
        occ = int(re.search(r"Occupancy \[waves/SIMD\]: (\d+)", b).group(1))
        assert (scratch, spill) == (0, 0), (name, scratch, spill)
        assert occ >= want[key], (name, occ)
    assert seen == set(want)


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not installed")
def test_translate_wide_leaves_its_pending_window_alone(tmp_path):
    """k_translate_wide<64> requests its next window with inline-asm loads and waits for it with a counted s_waitcnt of its
    own (ops_translate.hip, window_issue / window_arrive): the compiler does not know that the destination registers are
    pending, so nothing may read or write them between the request and the wait -- a register copy there (one per
    s_waitcnt variant, before the waits were folded into one asm statement) reads text that has not arrived.  The
    compiler's output is checked, instruction by instruction."""
    src = os.path.join(ROOT, "bigseqkit_amd", "csrc", "ops_translate.hip")
    out = tmp_path / "tr.s"
    # (the variant is off by default -- it measured slower, see DESIGN.md -- and kept behind -DBSK_TRW_ASM=1)
    r = subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "--cuda-device-only", "-DBSK_TRW_ASM=1", "-S", src, "-o", str(out)],
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    text = out.read_text().split("\n")
    begin = next(i for i, l in enumerate(text) if re.match(r"^_ZN3bsk\S*k_translate_wideILi64E\S*:", l))
    end = next(i for i in range(begin, len(text)) if ".amdhsa_next_free_vgpr" in text[i])
    lines = text[begin:end]
    vg = int(re.search(r"(\d+)", text[end]).group(1))
    assert vg <= 128, vg   # 4 waves per SIMD

    def regs(l):
        s = set()
        for m in re.finditer(r"v\[(\d+):(\d+)\]", l):
            s.update(range(int(m.group(1)), int(m.group(2)) + 1))
        for m in re.finditer(r"\bv(\d+)\b", l):
            s.add(int(m.group(1)))
        return s

    sites = [i for i, l in enumerate(lines) if "global_load_dwordx4" in l and "ASMSTART" in lines[i - 1]]
    waits = [i for i, l in enumerate(lines) if re.search(r"s_cmp_ge_u32 s\d+, 6", l) and "ASMSTART" in lines[i - 1]]
    assert len(sites) == 2 and len(waits) == 1, (sites, waits)   # before the loop and inside it; one wait at the loop top
    wait = waits[0]
    hdr = None
    for i in range(wait, 0, -1):  # the loop the wait belongs to
        m = re.search(r"Header=(BB\d+_\d+)", lines[i]) or re.match(r"^\.L(BB\d+_\d+):.*Loop Header", lines[i])
        if m:
            hdr = m.group(1)
            break
    assert hdr
    in_loop = [i for i, l in enumerate(lines) if ("Header=%s " % hdr) in l or l.startswith(".L%s:" % hdr)]
    first, last = min(in_loop), max(in_loop)
    while last + 1 < len(lines) and not lines[last + 1].startswith(".LBB"):  # to the end of the loop's last block
        last += 1
    assert first < wait < sites[1] <= last and sites[0] < first

    def pending(site):
        p, j = set(), site
        while "ASMEND" not in lines[j]:
            m = re.search(r"global_load_dword(?:x4)?\s+(v\[\d+:\d+\]|v\d+)", lines[j])
            if m:
                p |= regs(m.group(1))
            j += 1
        return p, j + 1

    for site in sites:
        p, after = pending(site)
        assert len(p) == 13, p
        span = list(range(after, wait)) if site < first else list(range(after, last + 1)) + list(range(first, wait))
        for k in span:
            code = lines[k].split(";")[0].strip()
            if code and not code.startswith("."):
                assert not (regs(code) & p), (k, code, sorted(p))
