#!/usr/bin/env python3
"""difflib similarity of the product's option-path functions against the reference functions that define the same options
(build container only).  Normalisation: comments / blank lines / leading whitespace dropped, string literals kept.
    python tools/similarity.py"""
import ast, difflib, os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"


def funcs(path):
    src = open(path).read()
    out = {}
    for node in ast.walk(ast.parse(src)):
        if isinstance(node, (ast.FunctionDef,)):
            seg = ast.get_source_segment(src, node)
            out[node.name] = seg
    return out


def norm(text):
    lines = []
    for ln in text.split("\n"):
        ln = re.sub(r"#.*$", "", ln).strip()
        if ln and not ln.startswith(('"""', "'''")):
            lines.append(re.sub(r"\s+", " ", ln))
    return "\n".join(lines)


ref = {}
for f in ("mac_cell.py", "ops.py"):
    for k, v in funcs(os.path.join(REF, f)).items():
        ref[f + ":" + k] = norm(v)
worst = 0.0
for f in ("plan.py", "generic.py"):
    for k, v in funcs(os.path.join(ROOT, "mac-network_amd", f)).items():
        a = norm(v)
        if len(a) < 200:
            continue
        best = max(((difflib.SequenceMatcher(None, a, b, autojunk=False).ratio(), rk) for rk, b in ref.items() if len(b) > 200), default=(0, ""))
        worst = max(worst, best[0])
        print("%-10s %-28s %.2f  %s" % (f, k, best[0], best[1]))
print("max %.2f" % worst)
