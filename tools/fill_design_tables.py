#!/usr/bin/env python3
"""Puts the current round's bench tables (tools/bench_table.py <dir>, reflowed to <= 150 columns) between the ROUND_TABLES markers
of DESIGN.md:  python tools/fill_design_tables.py profiles/r5"""
import os
import re
import subprocess
import sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
d = sys.argv[1]
tab = subprocess.run([sys.executable, os.path.join(root, "tools", "bench_table.py"), d], capture_output=True, text=True, check=True).stdout
tmp = os.path.join("/tmp", "xwb_tables.md")
open(tmp, "w").write(tab)
lst = subprocess.run([sys.executable, os.path.join(root, "tools", "reflow_md.py"), tmp], capture_output=True, text=True, check=True).stdout
p = os.path.join(root, "DESIGN.md")
s = open(p).read()
block = "<!-- ROUND_TABLES_BEGIN (tools/fill_design_tables.py %s) -->\n%s\n<!-- ROUND_TABLES_END -->" % (d, lst.strip())
if "ROUND5_TABLES" in s:
    s = s.replace("ROUND5_TABLES", block)
else:
    s = re.sub(r"<!-- ROUND_TABLES_BEGIN.*?<!-- ROUND_TABLES_END -->", lambda m: block, s, flags=re.S)
open(p, "w").write(s)
