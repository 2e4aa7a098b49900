#!/usr/bin/env python3
"""oracle/AUDIT.md cites line ranges of oracle/cpu_ref.c as `function` :first-last.  This checks that every such range lies
inside the function it names (so that an edit of cpu_ref.c that moves the code is noticed).  Exit status 1 on a stale range."""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def function_spans(path):
    """name -> (first line, last line) of every function defined at column 0."""
    lines = open(path).read().split("\n")
    spans, name, start = {}, None, 0
    for i, line in enumerate(lines, 1):
        m = re.match(r"^(?:static\s+)?(?:inline\s+)?[A-Za-z_][\w\s\*]*?\b(\w+)\s*\([^;]*$", line)
        if m and not line.startswith((" ", "\t", "#", "/", "}")) and name is None:
            name, start = m.group(1), i
            if line.rstrip().endswith("}"):  # one-line function
                spans[name] = (start, i)
                name = None
        elif name is not None and line.startswith("}"):
            spans[name] = (start, i)
            name = None
    return spans


def main():
    spans = function_spans(os.path.join(ROOT, "oracle", "cpu_ref.c"))
    text = open(os.path.join(ROOT, "oracle", "AUDIT.md")).read()
    bad, checked = [], 0
    for m in re.finditer(r"`(\w+)`((?:\s*:\d+(?:-\d+)?,?)+)", text):
        fn = m.group(1)
        if fn not in spans:
            continue
        lo, hi = spans[fn]
        for r in re.findall(r":(\d+)(?:-(\d+))?", m.group(2)):
            a, b = int(r[0]), int(r[1] or r[0])
            checked += 1
            if not (lo <= a <= b <= hi):
                bad.append(f"{fn} is at {lo}-{hi}, the audit cites {a}-{b}")
    for b in bad:
        print(b)
    print(f"{checked} cited ranges checked, {len(bad)} stale")
    return 1 if bad or checked < 20 else 0


if __name__ == "__main__":
    sys.exit(main())
