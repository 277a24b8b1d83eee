"""Per-ply instruction table of the rollout kernel, from the ISA hipcc emits (not an estimate).

usage: python tools/isa_ply_table.py [asm file]      (default: compiles csrc/mnk_rollout.hip with -S into /tmp)
Finds k_rollout_random<3, 9, 5, true, 0, true> (the headline kernel: 9x9x5, records, 32-bit-offset stores), takes its
innermost loops that hold four plies (16 global stores), and counts the instructions of one trip by class / 4.
The kernel has two such loops since round 3: the FAST one (waves of consistent games) comes first in the text."""
import collections
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KERNEL = "_Z16k_rollout_randomILi3ELi9ELi5ELb1ELi0ELb1E"

CLASSES = [
    ("bit logic (v_bitop3 / and / or / xor / or3 / and_or / not)", r"v_(bitop3|and_b32|or_b32|xor_b32|or3_b32|and_or_b32|not_b32|xad_u32)"),
    ("shifts / funnel shifts (v_alignbit / lshl / lshr / ashr / lshl_or / lshl_add)", r"v_(alignbit|lshlrev|lshrrev|ashrrev|lshl_or|lshl_add)"),
    ("bit count / field extract (v_bcnt / v_bfe)", r"v_(bcnt|bfe)"),
    ("selects and compares (v_cndmask / v_cmp / v_max / v_min)", r"v_(cndmask|cmp|max_u32|min_u32)"),
    ("integer arithmetic (v_add / sub / mad / mul / add3)", r"v_(add|sub|mad|mul|add3)"),
    ("moves (v_mov)", r"v_mov"),
    ("other VALU", r"v_"),
    ("global stores", r"global_store"),
    ("global loads", r"global_load"),
    ("scalar ALU / moves", r"s_(?!cbranch|branch|nop|waitcnt|barrier)"),
    ("branches", r"s_(cbranch|branch)"),
    ("s_nop / s_waitcnt", r"s_(nop|waitcnt)"),
]


def main():
    if len(sys.argv) > 1:
        asm = sys.argv[1]
    else:
        asm = "/tmp/mnk_rollout_isa.s"
        subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off",
                        "--cuda-device-only", "-S", "mnk_rollout.hip", "-o", asm],
                       cwd=os.path.join(ROOT, "rl-selfplay-mnk_amd", "csrc"), check=True, stderr=subprocess.DEVNULL)
    lines, on = [], False
    for line in open(asm):
        if line.startswith(KERNEL):
            on = True
        if on:
            lines.append(line.rstrip("\n"))
            if line.startswith(".Lfunc_end"):
                break
    # basic blocks that belong to one "Inner Loop Header": from the header label to the backward branch to it
    loops = []
    for i, line in enumerate(lines):
        m = re.match(r"^(\.LBB\d+_\d+):.*Inner Loop Header", line)
        if not m:
            continue
        label = m.group(1)
        for j in range(i + 1, len(lines)):
            if re.search(r"s_(c?branch\w*)\s+" + re.escape(label) + r"\b", lines[j]):
                body = [x.strip() for x in lines[i + 1:j + 1] if x.startswith("\t") and not x.strip().startswith((";", "."))]
                loops.append((label, body))
                break
    four = [(lab, b) for lab, b in loops if sum(x.startswith("global_store") for x in b) == 16]
    print("# Per-ply instruction table of the headline rollout kernel, from the ISA\n")
    print("`python tools/isa_ply_table.py`: hipcc -O3 -S of `csrc/mnk_rollout.hip`, kernel `k_rollout_random<3, 9, 5, true, 0, true>` "
          "(9x9x5, records on, 32-bit-offset record stores); the loops that hold four plies per trip (16 global stores), "
          "instructions of one trip / 4.  A wave that is alone on its SIMD pays one ~4-cycle issue slot per instruction of "
          "any kind (DESIGN.md section 5), so the column sums are what the headline time is made of.\n")
    names = ["FAST loop (waves of consistent games: what the bench runs)", "general loop (a wave that holds a poked state)"]
    table = collections.OrderedDict((c, []) for c, _ in CLASSES)
    totals = []
    for lab, body in four:
        counts = collections.Counter()
        for ins in body:
            for cname, pat in CLASSES:
                if re.match(pat, ins):
                    counts[cname] += 1
                    break
        for cname in table:
            table[cname].append(counts[cname] / 4)
        totals.append(len(body) / 4)
    print("| instructions per ply | " + " | ".join(names[:len(four)]) + " |")
    print("|---|" + "---|" * len(four))
    for cname, vals in table.items():
        print(f"| {cname} | " + " | ".join(f"{v:.2f}" for v in vals) + " |")
    valu = [sum(v[i] for c, v in table.items() if c.split()[0] not in ("global", "scalar", "branches", "s_nop")) for i in range(len(four))]
    print("| **VALU total** | " + " | ".join(f"**{v:.2f}**" for v in valu) + " |")
    print("| **all instructions** | " + " | ".join(f"**{v:.2f}**" for v in totals) + " |")
    print("\nPer four plies the trip also holds one Philox4x32-10 block: 20 `v_mad_u64_u32` (inline assembly) + 20 `v_bitop3_b32` "
          "(three-input xor), counted above under integer arithmetic / bit logic; the r-th-set-bit select is 23 instructions "
          "of inline assembly per ply (5 per halving level).")


if __name__ == "__main__":
    main()
