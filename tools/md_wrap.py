"""Reflow the prose of a Markdown file to at most WIDTH characters per line: paragraphs and list items are re-wrapped (continuation lines of
an item are indented to its text), code fences, tables, headings, HTML blocks and link-reference lines are left alone.
usage: python tools/md_wrap.py [--width 150] [--check] file.md ...   (--check: exit 1 if a non-table line exceeds 200 characters)"""
from __future__ import annotations

import re
import sys
import textwrap

ITEM = re.compile(r"^(\s*)([-*+]|\d+[.)])\s+")


def wrap(text: str, width: int) -> str:
    out: list[str] = []
    lines = text.split("\n")
    i, fence = 0, False
    while i < len(lines):
        line = lines[i]
        if line.lstrip().startswith("```"):
            fence = not fence
            out.append(line)
            i += 1
            continue
        if fence or not line.strip() or line.lstrip().startswith(("|", "#", "<", ">")) or re.match(r"^\s*\[[^\]]+\]:", line) or set(line.strip()) <= set("-=*_"):
            out.append(line)
            i += 1
            continue
        m = ITEM.match(line)
        if m:
            indent, first = m.group(1) + " " * (len(m.group(0)) - len(m.group(1))), m.group(0)
            body = [line[len(m.group(0)):]]
        else:
            indent = first = re.match(r"^\s*", line).group(0)
            body = [line.strip()]
        i += 1
        # continuation lines: same block until a blank line, a new item, a table / heading / fence
        while i < len(lines):
            nxt = lines[i]
            if not nxt.strip() or ITEM.match(nxt) or nxt.lstrip().startswith(("|", "#", "```", "<", ">")):
                break
            if m is None and len(re.match(r"^\s*", nxt).group(0)) != len(indent):
                break
            body.append(nxt.strip())
            i += 1
        para = " ".join(b for b in body if b)
        wrapped = textwrap.wrap(para, width=width, initial_indent=first, subsequent_indent=indent, break_long_words=False, break_on_hyphens=False)
        out.extend(wrapped or [first.rstrip()])
    return "\n".join(out)


def main() -> None:
    args = sys.argv[1:]
    width, check = 150, False
    files = []
    while args:
        a = args.pop(0)
        if a == "--width":
            width = int(args.pop(0))
        elif a == "--check":
            check = True
        else:
            files.append(a)
    bad = 0
    for f in files:
        text = open(f).read()
        if check:
            fence = False
            for n, line in enumerate(text.split("\n"), 1):
                if line.lstrip().startswith("```"):
                    fence = not fence
                if not fence and len(line) > 200 and not line.lstrip().startswith("|"):
                    print(f"{f}:{n}: {len(line)} characters")
                    bad += 1
        else:
            open(f, "w").write(wrap(text, width))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
