"""Prints the entry-point index of INTEGRATION.md (appendix): every function include/dsf_hip.h declares, the Python wrapper that
binds it, and the reference interface its header comment cites.   python tools/abi_index.py > /tmp/index.md"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
hdr = open(os.path.join(ROOT, "include", "dsf_hip.h")).read()
decl = re.compile(r"^(?:int|int64_t|const char\*) (dsf_[a-z0-9_]+)\(", re.M)
# section comments: /* ---- ... ---- */ blocks; the citation = first "/root/reference/<path>:<lines>" or "<file>.py:<lines>" in the block
blocks = [(m.start(), m.group(0)) for m in re.finditer(r"/\* -{10,}.*?-{10,} \*/", hdr, re.S)]


def section(pos):
    txt = ""
    for s, b in blocks:
        if s < pos:
            txt = b
    lines = [l.strip(" */-") for l in txt.splitlines()]
    title = next((l for l in lines if l), "")
    rest = " ".join(l for l in lines[lines.index(title) + 1:] if l) if title else ""
    what = rest[:220] + ("..." if len(rest) > 220 else "")
    return title, what.replace('|', '/')


def wrapper(sym):
    out = subprocess.run(["grep", "-rl", "--include=*.py", r"\b%s\b" % sym, os.path.join(ROOT, "dsf_amd")], capture_output=True, text=True).stdout
    files = sorted(os.path.relpath(f, ROOT) for f in out.split() if not f.endswith("_lib.py"))
    return ", ".join("`%s`" % f for f in files) or "(tools / bench only)"


print("| entry point | bound in | what `include/dsf_hip.h` says it is / replaces |")
print("|---|---|---|")
def preceding_comment(pos):
    """text of the comment that ends right before the declaration at ``pos`` (per-function citations)"""
    end = hdr.rfind("*/", 0, pos)
    if end < 0 or hdr[end + 2:pos].strip():
        return ""
    start = hdr.rfind("/*", 0, end)
    txt = " ".join(l.strip(" */") for l in hdr[start + 2:end].splitlines())
    return " ".join(txt.split())


for m in decl.finditer(hdr):
    title, cite = section(m.start())
    if not cite:
        c = preceding_comment(m.start())
        cite = c[:220] + ("..." if len(c) > 220 else "")
        cite = cite.replace("|", "/")
    if not cite and m.group(1).endswith("_backward"):
        cite = "(backward of the entry above)"
    print("| `%s` | %s | %s |" % (m.group(1), wrapper(m.group(1)), (title + " " + cite).strip()))
