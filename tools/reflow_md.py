"""Re-wrap the prose of a markdown file at 118 columns (tables, headings, code fences and blank lines untouched; list items keep
their hanging indent).  Usage: reflow_md.py FILE"""
import re, sys, textwrap
path = sys.argv[1]
lines = open(path).read().split("\n")
out, para, fence = [], [], False
def flush():
    global para
    if not para:
        return
    first = para[0]
    m = re.match(r"^(\s*)([*\-]|\d+\.)\s+", first)
    if m:
        lead = first[:m.end()]
        hang = " " * len(lead)
        text = " ".join([first[m.end():].strip()] + [l.strip() for l in para[1:]])
        out.extend(textwrap.wrap(text, 118, initial_indent=lead, subsequent_indent=hang, break_long_words=False, break_on_hyphens=False))
    else:
        ind = re.match(r"^\s*", first).group(0)
        text = " ".join(l.strip() for l in para)
        out.extend(textwrap.wrap(text, 118, initial_indent=ind, subsequent_indent=ind, break_long_words=False, break_on_hyphens=False))
    para = []
for l in lines:
    if l.startswith("```"):
        flush(); fence = not fence; out.append(l); continue
    if fence or l.startswith("|") or l.startswith("#") or not l.strip():
        flush(); out.append(l); continue
    if re.match(r"^\s*([*\-]|\d+\.)\s+", l):
        flush()
    para.append(l)
flush()
open(path, "w").write("\n".join(out))
