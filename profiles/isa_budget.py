#!/usr/bin/env python3
"""Instruction census of one kernel of a gfx950 code object, basic block by basic block.

  isa_budget.py <object.o | code object> <kernel name substring> [--blocks] [--dump]

Extracts the device code of a hipcc object (llvm-objdump --offloading), disassembles it
(llvm-objdump -d --mcpu=gfx950; both work in a GPU-less container), cuts the kernel into basic
blocks at branch targets and branches, classifies every instruction -- FP64 (v_fma/mul/add_f64 and
packed forms), other VALU, SALU, SMEM, LDS, VMEM (buffer/global/scratch/flat), branch, waitcnt /
nop, lane (v_readlane / v_writelane / v_readfirstlane) -- and prints the totals per block with the
block's successors, so that the steady-state path of a loop can be read off and priced."""
import collections
import os
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"


def device_code(path):
    """the gfx950 code object inside a hipcc object (llvm-objdump --offloading writes the bundles
    next to its input: work on a copy), or `path` itself if it already is one"""
    import glob
    import shutil
    d = tempfile.mkdtemp()
    tmp = os.path.join(d, os.path.basename(path))
    shutil.copy(path, tmp)
    subprocess.run([LLVM + "/llvm-objdump", "--offloading", tmp], capture_output=True, text=True)
    got = [f for f in glob.glob(tmp + ".*") if "gfx950" in f]
    return got[0] if got else path


def classify(m):
    if m.startswith(("v_readlane", "v_writelane", "v_readfirstlane")):
        return "lane"
    if re.match(r"v_(pk_)?(fma|mul|add|fmac|mfma)_f64|v_mfma_f64", m):
        return "fp64"
    if m.startswith("v_"):
        return "valu"
    if m.startswith(("s_load", "s_buffer_load", "s_scratch", "s_store", "s_dcache")):
        return "smem"
    if m.startswith(("s_cbranch", "s_branch", "s_setpc", "s_swappc", "s_endpgm", "s_getpc")):
        return "branch"
    if m.startswith(("s_waitcnt", "s_nop", "s_sleep", "s_barrier")):
        return "wait"
    if m.startswith("s_"):
        return "salu"
    if m.startswith("ds_"):
        return "lds"
    if m.startswith(("buffer_", "global_", "scratch_", "flat_")):
        return "vmem"
    return "other"


KINDS = ["fp64", "valu", "lane", "salu", "smem", "lds", "vmem", "branch", "wait", "other"]


def main():
    if len(sys.argv) < 3:
        raise SystemExit(__doc__)
    co = device_code(sys.argv[1])
    want = sys.argv[2]
    asm = subprocess.run([LLVM + "/llvm-objdump", "-d", "--mcpu=gfx950", co], capture_output=True, text=True).stdout
    lines = asm.splitlines()
    start = None
    name = None
    for i, ln in enumerate(lines):
        mt = re.match(r"^[0-9a-f]+ <(.+)>:$", ln)
        if mt and not mt.group(1).startswith("L"):          # (objdump may print local labels too)
            if start is not None:
                end = i
                break
            if want in mt.group(1):
                start, name = i + 1, mt.group(1)
    else:
        end = len(lines)
    if start is None:
        raise SystemExit("no kernel matches " + want)
    ins = []      # (addr, mnemonic, text)
    for ln in lines[start:end]:
        mt = re.match(r"^\s+(\S+)\s+(.*?)\s*//\s*([0-9A-Fa-f]+):", ln)
        if not mt:
            continue
        ins.append((int(mt.group(3), 16), mt.group(1), (mt.group(1) + " " + mt.group(2)).strip()))
    addr_index = {a: k for k, (a, _, _) in enumerate(ins)}
    # branch targets: objdump prints them as "<label+0x..>"-free absolute? -- compute from the simm16
    targets = set()
    succ = {}
    for k, (a, m, text) in enumerate(ins):
        if m.startswith(("s_cbranch", "s_branch")):
            mt = re.search(r"\s(-?\d+)\s*$", text) or re.search(r"(0x[0-9a-f]+|\d+)\s*$", text)
            if mt:
                imm = int(mt.group(1), 0)
                if imm >= 0x8000:
                    imm -= 0x10000
                t = a + 4 + 4 * imm
                if t in addr_index:
                    targets.add(t)
                    succ[k] = addr_index[t]
    # basic blocks
    leaders = {0} | {addr_index[t] for t in targets} | {k + 1 for k in succ if k + 1 < len(ins)}
    leaders = sorted(leaders)
    blocks = []
    for bi, lo in enumerate(leaders):
        hi = leaders[bi + 1] if bi + 1 < len(leaders) else len(ins)
        cnt = collections.Counter(classify(m) for _, m, _ in ins[lo:hi])
        last = hi - 1
        out = []
        if last in succ:
            out.append(succ[last])
        if not ins[last][1].startswith(("s_branch", "s_endpgm", "s_setpc")):
            out.append(hi if hi < len(ins) else None)
        blocks.append((lo, hi, cnt, out))
    leader_to_block = {lo: bi for bi, (lo, _, _, _) in enumerate(blocks)}
    total = collections.Counter(classify(m) for _, m, _ in ins)
    print("kernel", name)
    print("instructions %d:" % len(ins), "  ".join("%s %d" % (k, total[k]) for k in KINDS if total[k]))
    mn = collections.Counter(m for _, m, _ in ins)
    print("most frequent:", ", ".join("%s %d" % kv for kv in mn.most_common(24)))
    if "--blocks" in sys.argv:
        print("\nblock  first-addr  n   " + " ".join("%5s" % k for k in KINDS) + "  -> successors (back edges marked <)")
        for bi, (lo, hi, cnt, out) in enumerate(blocks):
            s = []
            for o in out:
                if o is None:
                    continue
                tb = leader_to_block.get(o)
                s.append("%s%s" % ("<" if tb is not None and tb <= bi else "", tb))
            print("%5d  %#10x %4d  " % (bi, ins[lo][0], hi - lo) + " ".join("%5d" % cnt[k] for k in KINDS) + "  -> " + ",".join(s))
    if "--dump" in sys.argv:
        for bi, (lo, hi, cnt, out) in enumerate(blocks):
            print("\n; ---- block %d" % bi)
            for a, m, text in ins[lo:hi]:
                print("%#8x  %-5s %s" % (a, classify(m), text))


if __name__ == "__main__":
    main()
