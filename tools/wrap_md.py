#!/usr/bin/env python3
"""Re-flow a Markdown file to a column limit (default 120): paragraphs and list items are re-wrapped (list continuation lines get
a hanging indent), headings, tables, code fences and blank lines are kept as they are; a table row or code line over the limit is
reported, not altered.  `python tools/wrap_md.py DESIGN.md [--width 120] [--check]`"""
import argparse
import re
import sys
import textwrap

ITEM = re.compile(r"^(\s*)([-*]|\d+\.)\s+")


def reflow(lines, width):
    out, para, fence = [], [], False

    def flush():
        if not para:
            return
        first = para[0]
        m = ITEM.match(first)
        if m:
            lead = m.group(0)
            body = first[len(lead):].strip() + " " + " ".join(p.strip() for p in para[1:])
            out.extend(textwrap.wrap(body.strip(), width=width, initial_indent=lead, subsequent_indent=" " * len(lead),
                                     break_long_words=False, break_on_hyphens=False))
        else:
            indent = re.match(r"^\s*", first).group(0)
            body = " ".join(p.strip() for p in para)
            out.extend(textwrap.wrap(body, width=width, initial_indent=indent, subsequent_indent=indent, break_long_words=False,
                                     break_on_hyphens=False))
        para.clear()

    for ln in lines:
        raw = ln.rstrip("\n")
        if raw.lstrip().startswith("```"):
            flush()
            fence = not fence
            out.append(raw)
            continue
        if fence or raw.startswith("#") or raw.lstrip().startswith("|") or raw.strip() in ("", "---"):
            flush()
            out.append(raw)
            continue
        if ITEM.match(raw) and para:
            flush()
        para.append(raw)
    flush()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("path")
    ap.add_argument("--width", type=int, default=120)
    ap.add_argument("--check", action="store_true", help="only report lines over the limit (exit 1 if any)")
    args = ap.parse_args()
    lines = open(args.path).read().split("\n")
    if not args.check:
        lines = reflow(lines, args.width)
        open(args.path, "w").write("\n".join(lines).rstrip("\n") + "\n")
    over = [(i + 1, len(ln)) for i, ln in enumerate(lines) if len(ln) > args.width]
    for i, n in over:
        print(f"{args.path}:{i}: {n} columns", file=sys.stderr)
    sys.exit(1 if over and args.check else 0)


if __name__ == "__main__":
    main()
