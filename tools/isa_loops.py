#!/usr/bin/env python3
"""Loops of one kernel in a gfx950 code object: instruction mix per loop body (backward branches of the disassembly).
    tools/isa_loops.py <object-or-library> <kernel-name-substring>
Counts per loop: all instructions, vector ALU (of which fp64), scalar, LDS, global loads, scratch (spill) accesses."""
import re
import subprocess
import sys
import tempfile
import os

LLVM = "/opt/rocm/lib/llvm/bin"
obj, pat = sys.argv[1], sys.argv[2]
tmp = tempfile.mkdtemp()
subprocess.check_call([f"{LLVM}/llvm-objcopy", "--dump-section", f".hip_fatbin={tmp}/fb.bin", obj])
subprocess.check_call([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", f"--input={tmp}/fb.bin",
                       "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={tmp}/dev.co"])
asm = subprocess.check_output([f"{LLVM}/llvm-objdump", "-d", f"{tmp}/dev.co"], text=True).split("\n")
ins, on = [], False
for l in asm:
    m = re.match(r"^([0-9a-f]+) <(.*)>:", l)
    if m:
        on = pat in m.group(2)
        if on:
            print("kernel", m.group(2))
        continue
    if on:
        m = re.match(r"^\s+(\S+)\s*(.*?)\s*//\s*([0-9A-Fa-f]+):", l)
        if m:
            ins.append((int(m.group(3), 16), m.group(1), m.group(2)))
addr_index = {a: i for i, (a, _, _) in enumerate(ins)}
loops = []
for i, (a, op, args) in enumerate(ins):
    if op.startswith("s_cbranch") or op == "s_branch":
        m = re.search(r"<.*\+0x([0-9a-f]+)>", l) if False else None
    # objdump prints the target as a comment-less operand: compute it from the encoding instead
for i, (a, op, args) in enumerate(ins):
    if op.startswith("s_cbranch") or op == "s_branch":
        try:
            off = int(args.split()[-1])
        except ValueError:
            continue
        if off >= 32768:
            off -= 65536
        tgt = a + 4 + 4 * off
        if tgt <= a and tgt in addr_index:
            loops.append((addr_index[tgt], i))
for lo, hi in sorted(loops):
    body = ins[lo:hi + 1]
    n = len(body)
    valu = sum(1 for _, op, _ in body if op.startswith("v_"))
    f64 = sum(1 for _, op, _ in body if op.startswith("v_") and "f64" in op)
    sal = sum(1 for _, op, _ in body if op.startswith("s_") and not op.startswith("s_waitcnt") and not op.startswith("s_nop"))
    lds = sum(1 for _, op, _ in body if op.startswith("ds_"))
    gl = sum(1 for _, op, _ in body if op.startswith("global_load") or op.startswith("buffer_load"))
    gs = sum(1 for _, op, _ in body if op.startswith("global_store") or op.startswith("global_atomic"))
    sc = sum(1 for _, op, _ in body if op.startswith("scratch_"))
    dv = sum(1 for _, op, _ in body if op.startswith("v_div_") or op.startswith("v_rcp") or op.startswith("v_rsq") or op.startswith("v_sqrt"))
    if n >= 24:
        print(f"loop @{ins[lo][0]:#x}..{ins[hi][0]:#x}: {n:5d} instr  valu {valu:5d} (f64 {f64:4d}, div/rcp {dv:3d})  salu {sal:4d}  lds {lds:3d}  "
              f"gload {gl:3d}  gstore/atomic {gs:3d}  scratch {sc:3d}")
