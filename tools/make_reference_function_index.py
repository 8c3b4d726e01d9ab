"""Writes tests/golden/reference_functions.json: the NAMES of the functions the reference defines (Chicdiff/R/chicdiff.R) and
of their formal arguments — interface facts the R host in r/R/ is linted against (tests/test_r_shim.py).  Run where
/root/reference exists; the fixture travels, the reference does not."""
import json, os, re, sys

SRC = sys.argv[1] if len(sys.argv) > 1 else "/root/reference/Chicdiff/R/chicdiff.R"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "reference_functions.json")


def formals(src, start):
    """names of the formals of the function( whose opening parenthesis is at src[start]"""
    depth, i, cur, names, in_str = 0, start, "", [], None
    while i < len(src):
        c = src[i]
        if in_str:
            if c == "\\":
                i += 1
            elif c == in_str:
                in_str = None
        elif c in "\"'":
            in_str = c
        elif c == "#":
            while i < len(src) and src[i] != "\n":
                i += 1
        elif c in "([{":
            depth += 1
            if depth > 1:
                cur += c
        elif c in ")]}":
            depth -= 1
            if depth == 0:
                names.append(cur)
                break
            cur += c
        elif c == "," and depth == 1:
            names.append(cur)
            cur = ""
        elif depth >= 1:
            cur += c
        i += 1
    out = []
    for a in names:
        a = a.strip()
        if a:
            out.append(re.split(r"\s*=", a, maxsplit=1)[0].strip())
    return out


src = open(SRC).read()
index = {}
for m in re.finditer(r"^([A-Za-z.][A-Za-z0-9._]*)\s*(?:<-|=)\s*function\s*\(", src, re.M):
    index[m.group(1)] = formals(src, m.end() - 1)
json.dump(index, open(OUT, "w"), indent=1, sort_keys=True)
print(len(index), "functions ->", os.path.normpath(OUT))
