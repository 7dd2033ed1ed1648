#!/usr/bin/env python3
"""Compile-only evidence for block_sync() (varscot_amd/csrc/vsc_device.h): builds vsc_sort.hip for gfx950 twice -
as committed, and with -DVSC_PLAIN_SYNCTHREADS (block_sync() = plain __syncthreads()) - and lists every s_barrier
of bin_hist_kernel with the instructions in front of it.  No GPU needed.

    python tools/isa_barrier_excerpt.py > profiles/r03_isa_barrier_bin_hist.txt
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "varscot_amd", "csrc")


def build(flag, out):
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-I" + os.path.join(ROOT, "include"),
           "-I" + CSRC] + flag + ["-x", "hip", os.path.join(CSRC, "vsc_sort.hip"), "-S", "--cuda-device-only", "-o", out]
    subprocess.run(cmd, check=True, stderr=subprocess.DEVNULL)
    return open(out).read().splitlines()


def instructions_before(lines, i, n):
    out = []
    j = i - 1
    while j >= 0 and len(out) < n:
        t = lines[j].strip()
        if t and not t.startswith((";", ".")) and not t.endswith(":"):
            out.append(t.split(";")[0].strip())
        elif t.endswith(":") or re.match(r"^\.LBB\d+_\d+:", t):
            out.append("<" + t.split(":")[0] + ":>")
        j -= 1
    return out[::-1]


def main():
    print("hipcc:", subprocess.run(["/opt/rocm/bin/hipcc", "--version"], capture_output=True, text=True).stdout.splitlines()[0])
    with tempfile.TemporaryDirectory() as tmp:
        for name, flag in (("plain __syncthreads()  (-DVSC_PLAIN_SYNCTHREADS)", ["-DVSC_PLAIN_SYNCTHREADS"]),
                           ("block_sync() as committed (s_waitcnt lgkmcnt(0) + __syncthreads())", [])):
            lines = build(flag, os.path.join(tmp, "sort.s"))
            print("\n==== %s" % name)
            cur, unfenced = None, 0
            for i, l in enumerate(lines):
                m = re.match(r"^(_ZN3vsc\S+):", l)
                if m:
                    cur = m.group(1)
                if "s_barrier" in l and cur and "bin_hist_kernelILb0E" in cur:
                    before = instructions_before(lines, i, 5)
                    fenced = any("lgkmcnt(0)" in x for x in before[-2:])
                    unfenced += not fenced
                    print("  line %5d  %s  <- %s" % (i + 1, "fenced  " if fenced else "UNFENCED", " | ".join(before)))
            print("  s_barrier without s_waitcnt lgkmcnt(0) directly in front: %d" % unfenced)
            if flag:
                # the unfenced barrier's loop: where the back-edge comes from
                for i, l in enumerate(lines):
                    if "s_barrier" in l and "bin_hist_kernelILb0E" in (cur_of(lines, i) or ""):
                        before = instructions_before(lines, i, 3)
                        if not any("lgkmcnt(0)" in x for x in before):
                            print("\n  excerpt around that barrier (loop header -> barrier; the back-edge arrives from the tile's")
                            print("  ds_add_u32 counting code, whose LDS atomics are still in flight):")
                            for x in lines[max(0, i - 14):i + 4]:
                                print("    " + x.rstrip())
                            break


def cur_of(lines, i):
    for j in range(i, -1, -1):
        m = re.match(r"^(_ZN3vsc\S+):", lines[j])
        if m:
            return m.group(1)
    return None


if __name__ == "__main__":
    sys.exit(main())
