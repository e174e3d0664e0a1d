#!/usr/bin/env python3
"""Re-wrap the prose of a markdown file at <= N columns (default 160) without touching tables, code fences, headings or list structure.
A list item keeps its marker and hanging indent; table rows cannot be wrapped in markdown and stay as they are.
usage: python tools/wrap_md.py FILE [N]   (rewrites FILE in place)"""
import re
import sys
import textwrap

path = sys.argv[1]
width = int(sys.argv[2]) if len(sys.argv) > 2 else 160
out, para, in_code = [], [], False


def flush():
    global para
    if not para:
        return
    first = para[0]
    m = re.match(r"^(\s*)([*+-] |\d+[.)] |> )?", first)
    lead, marker = m.group(1), m.group(2) or ""
    text = " ".join([first[len(lead) + len(marker):].strip()] + [ln.strip() for ln in para[1:]])
    out.extend(textwrap.wrap(text, width=width, initial_indent=lead + marker, subsequent_indent=lead + " " * len(marker),
                             break_long_words=False, break_on_hyphens=False) or [""])
    para = []


for line in open(path).read().split("\n"):
    s = line.strip()
    if s.startswith("```"):
        flush(); in_code = not in_code; out.append(line); continue
    if in_code or s.startswith("|") or s.startswith("#") or s == "" or re.match(r"^[-=*_]{3,}$", s):
        flush(); out.append(line); continue
    if re.match(r"^\s*([*+-] |\d+[.)] )", line):     # a new list item starts a new paragraph
        flush()
    elif para and len(line) - len(line.lstrip()) < len(para[0]) - len(para[0].lstrip()):
        flush()                                      # dedent: the list item ended
    para.append(line)
flush()
open(path, "w").write("\n".join(out))
