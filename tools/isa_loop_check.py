#!/usr/bin/env python
"""Diagnostic: compile one HIP source to gfx950 assembly and report, for every kernel whose mangled name contains a given
substring, the registers / scratch it uses and what its hottest loop (the longest backward-branch span) contains: MFMAs,
LDS-DMA loads, scratch reloads (each one drains the LDS-DMA pipeline through a compiler-inserted s_waitcnt vmcnt(0)) and the
s_waitcnt vmcnt values.  Usage: python tools/isa_loop_check.py prompt_tts_amd/csrc/gemm.hip wgrad8p_group [extra hipcc flags]"""
import re
import subprocess
import sys

src, pat = sys.argv[1], sys.argv[2]
asm = "/tmp/isa_loop_check.s"
subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=fast", "-S",
                "--cuda-device-only", src, "-o", asm] + sys.argv[3:], check=True, stderr=subprocess.DEVNULL)
lines = open(asm).read().split("\n")
i = 0
while i < len(lines):
    m = re.match(r"(_Z\w+):", lines[i])
    if not (m and pat in m.group(1)):
        i += 1
        continue
    name = m.group(1)
    j = i
    while not lines[j].startswith(".Lfunc_end"):
        j += 1
    body = lines[i:j]
    lab = {mm.group(1): k for k, l in enumerate(body) if (mm := re.match(r"(\.LBB\d+_\d+):", l))}
    loops = []
    for k, l in enumerate(body):
        mm = re.search(r"s_c?branch\w*\s+(\.LBB\d+_\d+)", l)
        if mm and lab.get(mm.group(1), 1 << 30) < k:
            loops.append((k - lab[mm.group(1)], lab[mm.group(1)], k))
    meta = {}
    for l in lines[j:j + 400]:
        for key in (".vgpr_count", ".vgpr_spill_count", ".sgpr_spill_count", ".private_segment_fixed_size"):
            if key + ":" in l and key not in meta:
                meta[key] = l.split(":")[1].strip()
    print(f"{name[:110]}\n   {meta}")
    if loops:
        span, lo, hi = max(loops)
        reg = body[lo:hi + 1]
        cnt = lambda s: sum(1 for l in reg if s in l)
        vm = sorted(set(re.findall(r"vmcnt\((\d+)\)", "\n".join(reg))), key=int)
        print(f"   hottest loop: {span} lines, mfma {cnt('v_mfma')}, global_load_lds {cnt('global_load_lds')}, ds_read {cnt('ds_read')}, "
              f"scratch_load {cnt('scratch_load')}, scratch_store {cnt('scratch_store')}, s_barrier {cnt('s_barrier')}, vmcnt values {vm}")
    i = j
