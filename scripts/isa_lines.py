#!/usr/bin/env python3
"""Per-source-line instruction counts of one kernel from `hipcc -S -gline-tables-only` output.

  hipcc --offload-arch=gfx950 -O3 -std=c++17 -gline-tables-only -S --cuda-device-only csrc/dpe_bcs.hip -o bcs.s
  scripts/isa_lines.py bcs.s <mangled-name-prefix> [--blocks] [--from LABEL --to LABEL] [--only BLOCK,BLOCK+1,...]

Prints, for the blocks between two labels (default: the whole function), vector / LDS / scalar instruction counts per source
line (file:line of the innermost inlined frame as the .loc directives give it) -- which lines of a pass loop the instructions
of the shipped build belong to."""
import collections
import re
import sys


def main():
    path, prefix = sys.argv[1], sys.argv[2]
    args = sys.argv[3:]
    lo = args[args.index("--from") + 1] if "--from" in args else None
    hi = args[args.index("--to") + 1] if "--to" in args else None
    only = set(args[args.index("--only") + 1].split(",")) if "--only" in args else None   # sub-block names as --blocks prints them
    files, body, inside = {}, [], False
    for ln in open(path, errors="replace"):
        m = re.match(r'\s*\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?', ln)
        if m:
            files[int(m.group(1))] = (m.group(3) or m.group(2)).split("/")[-1]
        if ln.startswith(prefix) and ln.rstrip().split(":")[0].startswith(prefix) and ":" in ln:
            inside = True
            continue
        if inside:
            if ln.startswith(".Lfunc_end"):
                break
            body.append(ln.rstrip("\n"))
    cur, label, active = ("?", 0), "entry", lo is None
    per_line = collections.defaultdict(lambda: [0, 0, 0, 0])
    per_block = collections.OrderedDict()
    sub = {}
    for ln in body:
        s = ln.strip()
        m = re.match(r"\.loc\s+(\d+)\s+(\d+)", s)
        if m:
            cur = (files.get(int(m.group(1)), m.group(1)), int(m.group(2)))
            continue
        m = re.match(r"(\.LBB\d+_\d+):", s)
        if m:
            label = m.group(1)
            if lo and label == lo:
                active = True
            if hi and label == hi:
                active = False
            continue
        if not s or s.startswith((";", ".", "//")):
            continue
        op = s.split()[0]
        kind = 0 if op.startswith("v_") else 1 if op.startswith("ds_") else 2 if op.startswith("s_") else 3
        b = per_block.setdefault(label, [0, 0, 0, 0, set(), []])
        owner = label
        b[kind] += 1
        b[4].add(cur[1] if cur[0].endswith("chip2.h") else -1)
        if op.startswith("s_cbranch") or op == "s_branch":   # what follows a branch is a block of its own: LABEL+n
            b[5].append(s.split()[-1])
            base = label.split("+")[0]
            sub[base] = sub.get(base, 0) + 1
            label = "%s+%d" % (base, sub[base])
        if active and (only is None or owner in only):
            per_line[cur][kind] += 1
    if "--blocks" in args:
        for k, b in per_block.items():
            ls = sorted(x for x in b[4] if x > 0)
            print("%-12s v %4d ds %3d s %4d other %3d  lines %s..%s  -> %s" % (k, b[0], b[1], b[2], b[3], ls[0] if ls else "-", ls[-1] if ls else "-", ",".join(b[5])))
        return
    tot = [0, 0, 0, 0]
    for k in sorted(per_line):
        v = per_line[k]
        print("%-22s %5d  v %4d  ds %3d  s %4d  other %3d" % (k[0], k[1], *v))
        tot = [a + b for a, b in zip(tot, v)]
    print("total v %d ds %d s %d other %d" % tuple(tot))


if __name__ == "__main__":
    main()
