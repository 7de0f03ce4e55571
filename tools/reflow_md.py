#!/usr/bin/env python3
"""Reflows a markdown file to <= 160 columns: paragraphs and list items are wrapped; a table with a cell over 300 characters
becomes a bullet list (one block per row, one sub-bullet per column) -- such tables cannot be diffed or read in a terminal.
Usage: tools/reflow_md.py in.md [first_line last_line] > out.md"""
import re
import sys
import textwrap

W = 150


def wrap(text, first, rest):
    return textwrap.fill(text, width=W, initial_indent=first, subsequent_indent=rest, break_long_words=False, break_on_hyphens=False)


def cells(row):
    row = row.strip()
    if row.startswith("|"):
        row = row[1:]
    if row.endswith("|"):
        row = row[:-1]
    out, cur, code = [], "", False
    i = 0
    while i < len(row):
        c = row[i]
        if c == "`":
            code = not code
        if c == "\\" and i + 1 < len(row) and row[i + 1] == "|":
            cur += "|"; i += 2; continue
        if c == "|" and not code:
            out.append(cur.strip()); cur = ""
        else:
            cur += c
        i += 1
    out.append(cur.strip())
    return out


def table(rows):
    hdr = cells(rows[0])
    body = [cells(r) for r in rows[2:]]
    if all(len(c) <= 300 for r in body for c in r) and all(len(r) <= W for r in rows):
        return rows
    if all(len(c) <= 300 for r in body for c in r):
        pass                                              # short cells, long rows: still a list (a 160-column terminal wraps the row)
    out = []
    for r in body:
        out.append(wrap(("**%s**" % r[0]) if r and r[0] else "(row)", "- ", "  "))
        for h, c in zip(hdr[1:], r[1:]):
            if c:
                out.append(wrap("%s: %s" % (h, c) if h else c, "  - ", "    "))
    return out


def main():
    lines = open(sys.argv[1]).read().split("\n")
    if len(sys.argv) > 3:
        lines = lines[int(sys.argv[2]) - 1:int(sys.argv[3])]
    out, i = [], 0
    while i < len(lines):
        ln = lines[i]
        if ln.startswith("```"):
            out.append(ln); i += 1
            while i < len(lines) and not lines[i].startswith("```"):
                out.append(lines[i]); i += 1
            if i < len(lines):
                out.append(lines[i]); i += 1
            continue
        if ln.lstrip().startswith("|") and i + 1 < len(lines) and re.match(r"^\s*\|?[\s:|-]+\|[\s:|-]*$", lines[i + 1]):
            j = i
            while j < len(lines) and lines[j].lstrip().startswith("|"):
                j += 1
            out.extend(table(lines[i:j])); i = j
            continue
        if not ln.strip() or ln.startswith("#"):
            out.append(ln if len(ln) <= W or not ln.startswith("#") else wrap(ln, "", "  ")); i += 1
            continue
        # a paragraph or a list item: gather its continuation lines
        m = re.match(r"^(\s*)([-*+]|\d+\.)\s+", ln)
        indent = (m.group(1) + " " * (len(m.group(2)) + 1)) if m else re.match(r"^\s*", ln).group(0)
        first = ln[:len(m.group(0))] if m else indent
        text = ln[len(first):] if m else ln.strip()
        i += 1
        while i < len(lines) and lines[i].strip() and not lines[i].startswith("#") and not lines[i].lstrip().startswith("|") \
                and not re.match(r"^\s*([-*+]|\d+\.)\s+", lines[i]) and not lines[i].startswith("```"):
            text += " " + lines[i].strip(); i += 1
        out.append(wrap(text, first, indent))
    print("\n".join(out))


if __name__ == "__main__":
    main()
