#!/usr/bin/env python3
"""Scan gfx950 ISA (hipcc -S --cuda-device-only output) for VGPR spill stores issued while EXEC is narrowed.

Why: hipcc (ROCm 7.2, clang 22) can place the spill of a VGPR that is live across a divergent region INSIDE that region
-- e.g. behind `if (lane == 0) {...}` just before a call -- so the store runs with EXEC = {lane 0}; the reload after the
reconvergence point runs with full EXEC and lanes 1..63 read bytes nobody wrote.  This cost round 1 a wrong Wald
covariance on Count outcomes in glmm_general_kernel ("clustered GLM with extra groupings").

Heuristic, linear in program order: a region opens at `s_and_saveexec_b64 s[a:b]`, `s_andn2_saveexec_b64`, or
`s_mov_b64 s[a:b], exec` followed by a narrowing `s_mov_b64 exec, ...`; it closes at `s_or_b64 exec, exec, s[a:b]`.
A `Folded Spill` store inside an open region is reported when the spilled register's last definition precedes the
region's opening (the value was produced under the wider mask).  Reports are leads to read, not proofs.

    python tools/scan_exec_spills.py file.s [file.s ...]
"""
import re
import sys

SAVE = re.compile(r"\ts_(?:and|andn2)_saveexec_b64 (s\[\d+:\d+\])")  # or/xor_saveexec ..., -1 WIDEN the mask (CSR saves)
COPY = re.compile(r"\ts_mov_b64 (s\[\d+:\d+\]), exec\b")
NARROW = re.compile(r"\ts_(?:mov|and|andn2)_b64 exec, ")
RESTORE = re.compile(r"\ts_or_b64 exec, exec, (s\[\d+:\d+\])")
RESTORE_MOV = re.compile(r"\ts_mov_b64 exec, (s\[\d+:\d+\])")
SDEST = re.compile(r"\ts_\w+ (s\[\d+:\d+\]), ")
SPILL = re.compile(r"\tscratch_store_dword(?:x(\d))? off, (v\[(\d+):(\d+)\]|v(\d+)), .*Folded Spill")
DEF = re.compile(r"\t(?:v_|ds_read|ds_bpermute|ds_permute|ds_swizzle|scratch_load|global_load|buffer_load|flat_load|global_atomic|ds_add_rtn)\S* (v\[(\d+):(\d+)\]|v(\d+))(?=[,\s])")
FUNC = re.compile(r"^(_Z\w+):")


def regs(m, base):
    if m.group(base + 1) is not None:
        return range(int(m.group(base + 1)), int(m.group(base + 2)) + 1)
    return range(int(m.group(base + 3)), int(m.group(base + 3)) + 1)


def scan(path):
    hits = []
    func, opened, pending_copy, last_def = None, {}, None, {}
    for no, line in enumerate(open(path, errors="replace"), 1):
        f = FUNC.match(line)
        if f:
            func, opened, pending_copy, last_def = f.group(1), {}, None, {}
            continue
        m = SAVE.match(line)
        if m:
            opened[m.group(1)] = no
            continue
        m = COPY.match(line)
        if m:
            pending_copy = (m.group(1), no)
            continue
        if NARROW.match(line) and pending_copy and not (RESTORE_MOV.match(line) and RESTORE_MOV.match(line).group(1) in opened):
            opened[pending_copy[0]] = pending_copy[1]
            pending_copy = None
            continue
        m = RESTORE.match(line) or RESTORE_MOV.match(line)
        if m:
            opened.pop(m.group(1), None)
            continue
        m = SDEST.match(line)
        if m and m.group(1) in opened:  # the pair that held the saved mask is reused: that region is over
            opened.pop(m.group(1), None)
        m = SPILL.match(line)
        if m and opened:
            for start in sorted(opened.values(), reverse=True):
                if any(last_def.get(r, 0) < start for r in regs(m, 2)):
                    hits.append((func, no, line.strip(), start))
                    break
            continue
        m = DEF.match(line)
        if m:
            for r in regs(m, 1):
                last_def[r] = no
    return hits


if __name__ == "__main__":
    total = 0
    for p in sys.argv[1:]:
        for func, no, text, start in scan(p):
            total += 1
            print(f"{p}:{no}: in {func[:90]}: {text}   [EXEC narrowed since line {start}]")
    print(f"{total} suspicious spill store(s)")
