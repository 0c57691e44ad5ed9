"""Checks every measured figure DESIGN.md (or another document) quotes against the tracked file it cites.

A quoted figure carries a tag ON ITS OWN LINE OF TEXT (an HTML comment: invisible when the markdown is rendered):

    ... `k_entity_stream` **8.56 ms** ...  <!--track csv profiles/r5_wikimel_b4096_kernel_stats.csv k_entity_stream AverageNs 8.56-->
    ... headline 25.6 M pairs/s ...        <!--track json profiles/r5_wikimel_b4096_bench_all_legs.json value 25.6e6-->

  csv  FILE SUBSTRING FIELD VALUE   rocprofv3 `*_kernel_stats.csv`: the first row whose Name contains SUBSTRING; FIELD is AverageNs /
                                    MinNs / MaxNs / TotalDurationNs (compared in MILLISECONDS with VALUE) or Calls / Percentage (as is)
  json FILE DOTTED.PATH VALUE       a bench record (the first line of FILE when it holds several); list indices as numbers

The check fails when the file's value and VALUE disagree by more than 3 % (`--tol`), when the file or the row is missing, or when
VALUE's digits do not occur in the visible text of the tagged line (a tag must sit next to the number it vouches for).

    python tools/check_design_numbers.py [DESIGN.md ...] [--tol 0.03]         # exit code 1 on any disagreement
"""
from __future__ import annotations

import csv
import json
import os
import re
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TAG = re.compile(r"<!--track\s+(csv|json)\s+(\S+)\s+(.*?)\s+(-?[0-9.]+(?:e-?[0-9]+)?)\s*-->")


def _csv_value(path, substring, field):
    with open(path, newline="") as f:
        for row in csv.DictReader(f):
            if substring in row["Name"]:
                v = float(row[field])
                return v / 1e6 if field.endswith("Ns") else v
    raise KeyError(f"no kernel whose name contains {substring!r}")


def _json_value(path, dotted):
    with open(path) as f:
        text = f.read()
    try:
        node = json.loads(text)                       # one document (indented or not)
    except ValueError:
        node = json.loads(text.splitlines()[0])       # several records, one per line: the first
    for key in dotted.split("."):
        node = node[int(key)] if isinstance(node, list) else node[key]
    return float(node)


def _digits_in_text(value_text: str, visible: str) -> bool:
    """'8.56' vouches for '8.56 ms'; '25.6e6' for '25.6 M'; '0.716' for '0.72' is NOT accepted (quote what the file says)."""
    mantissa = value_text.lower().split("e")[0].rstrip("0").rstrip(".") if "." in value_text.lower().split("e")[0] else value_text.lower().split("e")[0]
    squeezed = visible.replace(" ", "").replace(" ", "").replace(",", "")
    return mantissa in squeezed


def check(doc: str, tol: float = 0.03):
    problems, n = [], 0
    with open(doc) as f:
        lines = f.read().splitlines()
    for ln, line in enumerate(lines, 1):
        for kind, rel, selector, quoted in TAG.findall(line):
            n += 1
            where = f"{os.path.relpath(doc, REPO)}:{ln}"
            path = os.path.join(REPO, rel)
            try:
                if kind == "csv":
                    substring, field = selector.rsplit(None, 1)
                    actual = _csv_value(path, substring, field)
                else:
                    actual = _json_value(path, selector)
            except (OSError, KeyError, ValueError, IndexError, TypeError) as e:
                problems.append(f"{where}: {rel} [{selector}]: {type(e).__name__}: {e}")
                continue
            q = float(quoted)
            if abs(actual - q) > tol * max(abs(actual), 1e-30):
                problems.append(f"{where}: quotes {quoted} but {rel} [{selector}] says {actual:.6g} ({(q / actual - 1) * 100:+.1f} %)")
            visible = TAG.sub("", line)
            if not _digits_in_text(quoted, visible):
                problems.append(f"{where}: the tag vouches for {quoted}, which the line's visible text does not show")
    return n, problems


def main(argv):
    tol = 0.03
    docs = []
    it = iter(argv)
    for a in it:
        if a == "--tol":
            tol = float(next(it))
        else:
            docs.append(a)
    docs = docs or [os.path.join(REPO, "DESIGN.md")]
    total, bad = 0, []
    for d in docs:
        n, problems = check(d, tol)
        total += n
        bad += problems
    for p in bad:
        print(p)
    print(f"{total} tracked figures checked, {len(bad)} problem(s)")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
