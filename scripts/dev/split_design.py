"""One-off (round 5): cut the 137 KB DESIGN.md into docs/*.md and re-wrap every paragraph / list item to <= 120
characters (tables and code blocks are left as they are).  usage: python scripts/dev/split_design.py DESIGN.md docs/"""
import re
import sys
import textwrap

src, out = sys.argv[1], sys.argv[2]
lines = open(src, encoding="utf-8").read().split("\n")
heads = [(i, l) for i, l in enumerate(lines) if re.match(r"^#{1,3} ", l)]


def section(start_pat, end_pat=None):
    a = next(i for i, l in heads if re.match(start_pat, l))
    b = next((i for i, l in heads if i > a and end_pat and re.match(end_pat, l)), len(lines))
    return lines[a:b]


def wrap(block):
    res, para = [], []

    def flush():
        if not para:
            return
        first = para[0]
        m = re.match(r"^(\s*)([*\-] |\d+\. )?", first)
        indent = m.group(1) + (" " * len(m.group(2)) if m.group(2) else "")
        text = " ".join(p.strip() for p in para)
        lead = m.group(1) + (m.group(2) or "")
        body = text[len((m.group(2) or "")):] if m.group(2) else text
        res.extend(textwrap.wrap(body, 120, initial_indent=lead, subsequent_indent=indent, break_long_words=False, break_on_hyphens=False))
        para.clear()

    in_code = False
    for l in block:
        if l.strip().startswith("```"):
            flush(); in_code = not in_code; res.append(l); continue
        if in_code or l.startswith("|") or re.match(r"^#{1,6} ", l) or re.match(r"^\s{4,}\S", l) and not para:
            flush(); res.append(l); continue
        if not l.strip():
            flush(); res.append(""); continue
        if re.match(r"^\s*([*\-] |\d+\. )", l):
            flush()
        para.append(l)
    flush()
    return res


parts = {
    "oracle.md": section(r"^## 2\. ", r"^## 3\. "),
    "knn.md": section(r"^### 4\.1 ", r"^### 4\.2 "),
    "geometry.md": section(r"^### 4\.2 ", r"^### 4\.5 ") + section(r"^### 4\.6 ", r"^## 5\. "),
    "sift.md": section(r"^### 4\.5 ", r"^### 4\.6 "),
    "measurement.md": section(r"^## 5\. ", r"^## 6\. "),
    "multigpu.md": section(r"^## 6\. ", r"^## 7\. "),
    "history.md": section(r"^## 8\. "),
    "_overview_src.md": section(r"^# DESIGN", r"^## 2\. ") + section(r"^## 3\. ", r"^## 4\. ") + section(r"^## 7\. ", r"^## 8\. "),
}
for name, block in parts.items():
    w = wrap(block)
    open(f"{out}/{name}", "w", encoding="utf-8").write("\n".join(w).rstrip("\n") + "\n")
    longest = max((len(x) for x in w if not x.startswith("|")), default=0)
    print(f"{name}: {len(block)} -> {len(w)} lines, longest non-table line {longest}")
