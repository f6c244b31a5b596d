#!/usr/bin/env python3
"""Builds conv_split.o WITH the SLP vectoriser but with the device assembly patched before it is assembled (the bisect of
the epilogue defect, see tools/slp_hazard_repro.hip): replays hipcc's own pipeline (hipcc -### -save-temps) and edits the
gfx950 .s between the code generator and the assembler.
    slp_patch_build.py <mode> <out.o>      mode: nop_after_pk  -> `s_nop N` after every v_pk_* instruction (N = 4)
                                                 nop_before_mov_hi -> only between a v_pk_* with op_sel and the next VALU write
                                                                      to either half of its destination
                                                 nop_before_opsel -> `s_nop 7` before AND after every op_sel-swapped v_pk_add_f32
                                                 scalarize_opsel -> every `v_pk_add_f32 vD, vD, vS op_sel:[0,1] op_sel_hi:[1,0]` becomes
                                                                    two v_add_f32 (same arithmetic, no packed op_sel swap)"""
import os, re, shlex, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CS = os.path.join(ROOT, "doubly-contrastive-semseg_amd", "dcs_amd", "csrc")
mode, out = sys.argv[1], os.path.abspath(sys.argv[2])
work = tempfile.mkdtemp(prefix="slp_patch_")
cmd = ["/opt/rocm/bin/hipcc", "-###", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-I" + os.path.join(ROOT, "include"),
       "-I" + CS, "-Wno-unused-function", "-save-temps", "-c", os.path.join(CS, "conv_split.hip"), "-o", out]
txt = subprocess.run(cmd, cwd=work, stderr=subprocess.PIPE, stdout=subprocess.PIPE).stderr.decode()
steps = [shlex.split(l) for l in txt.splitlines() if l.startswith(' "')]
dev_s = "conv_split-hip-amdgcn-amd-amdhsa-gfx950.s"


def patch(path):
    lines = open(path).read().split("\n")
    outl, n = [], 0
    pk = re.compile(r"^\s+v_pk_\w+\s+v\[(\d+):(\d+)\]")
    swapped = re.compile(r"^\s+v_pk_add_f32 v\[(\d+):(\d+)\], v\[(\d+):(\d+)\], v\[(\d+):(\d+)\] op_sel:\[0,1\] op_sel_hi:\[1,0\]\s*$")
    for i, l in enumerate(lines):
        if mode == "nop_before_opsel" and swapped.match(l):
            outl.append("\ts_nop 7"); outl.append(l); outl.append("\ts_nop 7"); n += 1
            continue
        if mode == "scalarize_opsel":
            # lo = src0.lo + src1.HI, hi = src0.hi + src1.LO  ->  two scalar adds (destination pair == src0 pair, src1 disjoint)
            w = swapped.match(l)
            if w:
                d0, d1, a0, a1, b0, b1 = (int(v) for v in w.groups())
                if (d0, d1) == (a0, a1) and not {b0, b1} & {d0, d1}:
                    outl.append(f"\tv_add_f32_e32 v{d0}, v{a0}, v{b1}")
                    outl.append(f"\tv_add_f32_e32 v{d1}, v{a1}, v{b0}")
                    n += 1
                    continue
            outl.append(l)
            continue
        outl.append(l)
        m = pk.match(l)
        if not m:
            continue
        if mode == "nop_after_pk":
            outl.append("\ts_nop 4"); n += 1
        elif mode == "nop_before_mov_hi" and "op_sel" in l:
            lo, hi = int(m.group(1)), int(m.group(2))
            for j in range(i + 1, min(i + 6, len(lines))):
                w = re.match(r"^\s+v_\w+\s+v(\d+),", lines[j])
                if w and int(w.group(1)) in (lo, hi):
                    outl.append("\ts_nop 4"); n += 1
                    break
    open(path, "w").write("\n".join(outl))
    print(f"{mode}: {n} sites patched", flush=True)


for st in steps:
    if "-cc1as" in st and "amdgcn-amd-amdhsa" in st:
        patch(os.path.join(work, dev_s))
    subprocess.run(st, cwd=work, check=True)
print("built", out)
