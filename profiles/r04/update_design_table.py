#!/usr/bin/env python3
"""Replace the generated table of DESIGN.md section 5 (between the r04-table markers) with the current measurement set."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
out = subprocess.run([sys.executable, os.path.join(ROOT, "profiles", "r04", "make_design_table.py")] + sys.argv[1:], capture_output=True, text=True).stdout
table = "\n".join(l for l in out.split("\n") if l.startswith("|"))
p = os.path.join(ROOT, "DESIGN.md")
s = open(p).read()
a = s.index("<!-- r04-table-begin")
a = s.index("\n", a) + 1
b = s.index("<!-- r04-table-end -->")
open(p, "w").write(s[:a] + table + "\n" + s[b:])
print(out[out.index("`bench_driver"):] if "`bench_driver" in out else "")
