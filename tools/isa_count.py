#!/usr/bin/env python3
"""Static instruction counts of the blind-rotation kernels' step loops, from the compiler's assembly (no GPU needed).

    python3 tools/isa_count.py [pattern ...] > profiles/r04/isa_step_loop.txt

Compiles csrc/fbs_blind_rotate.hip, fbs_blind_rotate_cu.hip, fbs_blind_rotate_k2.hip and fbs_blind_rotate_glwe.hip with the Makefile's flags to gfx950 assembly (`hipcc -S`), finds
every kernel whose demangled name matches one of the patterns (default: the instantiations the profile sets of
tools/profile_round.sh time) and prints, for the basic blocks the compiler marks as inside a loop: vector (VALU) instructions, the
FP64 share, the opcode histogram, and the kernel's register / scratch / LDS figures.  The PMC counter SQ_INSTS_VALU per wave and
step (profiles/r03/summary_*.json) agrees with the VALU total to within the blocks a step may skip."""
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "tfhe_fbs_map_amd", "csrc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "--cuda-device-only", "-S"]
DEFAULT = [r"k_blind_rotate<10, 6, 3, 4, true>", r"k_blind_rotate<10, 6, 3, 1, true>", r"k_blind_rotate<10, 6, 6, 1, true>",
           r"k_blind_rotate<11, 7, 4, 1, true>", r"k_blind_rotate<11, 7, 5, 1, true>", r"k_blind_rotate<12, 8, 5, 1, true>",
           r"k_blind_rotate_pairs<11, 7, 4>", r"k_blind_rotate_cu<10, 3, 2, false>", r"k_blind_rotate_cu<10, 3, 2, true>",
           r"k_blind_rotate_cu_pairs<11, 1>", r"k_blind_rotate_cu_pairs<11, 2>", r"k_blind_rotate_pairs_k2<10, 4>", r"k_blind_rotate_cu_k2",
           r"k_blind_rotate_glwe<9, 4, 2, 3>", r"k_blind_rotate_glwe<9, 4, 2, 1>"]


def demangle(names):
    out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout
    return dict(zip(names, out.splitlines()))


def kernels(asm):
    lines = asm.split("\n")
    starts = [(i, m.group(1)) for i, l in enumerate(lines) for m in [re.match(r"^(_Z\w+):", l)] if m]
    for i, sym in starts:
        ops, in_loop, info = collections.Counter(), False, {}
        for l in lines[i + 1:]:
            s = l.strip()
            if l.startswith(".Lfunc_end"):
                break
            m = re.match(r"^\.LBB\d+_\d+:(.*)", s)
            if m:
                in_loop = "Loop" in m.group(1)
                continue
            if in_loop and s and s[0] not in ";.":
                ops[s.split()[0]] += 1
        for l in lines[i:]:
            m = re.match(r"\s*;\s*(NumVgprs|ScratchSize|LDSByteSize|Occupancy):\s*(\d+)", l)
            if m and m.group(1) not in info:
                info[m.group(1)] = int(m.group(2))
            if len(info) == 4:
                break
        yield sym, ops, info


def main():
    pats = sys.argv[1:] or DEFAULT
    print("# python3 tools/isa_count.py: vector instructions in the loop blocks of each kernel, from hipcc -S (a nested loop's body is")
    print("# counted once: the headline kernel's level loop, 849 of the 2664, runs twice per step -> 3 513 per wave and step)")
    with tempfile.TemporaryDirectory() as tmp:
        for src in ("fbs_blind_rotate.hip", "fbs_blind_rotate_cu.hip", "fbs_blind_rotate_k2.hip", "fbs_blind_rotate_glwe.hip"):
            out = os.path.join(tmp, src + ".s")
            subprocess.run(["/opt/rocm/bin/hipcc"] + FLAGS + ["-o", out, os.path.join(CSRC, src)], check=True, stderr=subprocess.DEVNULL)
            found = list(kernels(open(out).read()))
            names = demangle([k[0] for k in found])
            for sym, ops, info in found:
                name = names[sym].replace("void fbs::", "").replace("(fbs::BrArgs)", "")
                if not any(p in name for p in pats):
                    continue
                valu = sum(n for o, n in ops.items() if o.startswith("v_"))
                f64 = sum(n for o, n in ops.items() if o.startswith("v_") and "f64" in o)
                print("%s\n  step loop: %d VALU (%d FP64), %d instructions in all; registers %s, scratch %s B, LDS %s B" % (
                    name, valu, f64, sum(ops.values()), info.get("NumVgprs"), info.get("ScratchSize"), info.get("LDSByteSize")))
                print("  " + " ".join("%s:%d" % kv for kv in ops.most_common()))


if __name__ == "__main__":
    main()
