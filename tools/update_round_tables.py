"""Replaces the generated results table of a round in DESIGN.md (section 5) and BASELINE.md (the round's section) with the current output of
tools/round_table.py -- the block that starts at the header row `| config (`profiles/<tag>_bench_*.json`) ...` and runs to the last
consecutive table row.   python tools/update_round_tables.py r05"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r05"
table = [l for l in subprocess.run([sys.executable, os.path.join(ROOT, "tools", "round_table.py"), tag], stdout=subprocess.PIPE, text=True,
                                   check=True).stdout.splitlines() if l.startswith("|")]
assert table and table[0].startswith(f"| config (`profiles/{tag}_bench_"), table[:1]
for name in ("DESIGN.md", "BASELINE.md"):
    path = os.path.join(ROOT, name)
    lines = open(path).read().split("\n")
    starts = [i for i, l in enumerate(lines) if l.startswith(f"| config (`profiles/{tag}_bench_")]
    assert len(starts) == 1, (name, starts)
    i = j = starts[0]
    while j < len(lines) and lines[j].startswith("|"):
        j += 1
    lines[i:j] = table
    open(path, "w").write("\n".join(lines))
    print(f"{name}: {j - i} rows -> {len(table)}")
