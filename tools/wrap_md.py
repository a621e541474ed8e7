"""Hard-wrap a Markdown file at 160 columns (VERDICT r5 housekeeping): paragraphs and list items are RE-FLOWED at word boundaries (the lines of a paragraph / list
item are joined first; continuation lines of a list item are indented to its text), code fences, tables, headings and blank lines are left alone.
usage: python tools/wrap_md.py FILE [WIDTH]"""
import re
import sys
import textwrap

MARK = re.compile(r"^(\s*)((?:[-*+]|\d+\.)\s+)")


def wrap_file(path, width=160):
    lines = open(path).read().split("\n")
    out, block, fence = [], None, False      # block = [indent, marker, text]

    def flush():
        nonlocal block
        if block is not None:
            indent, marker, text = block
            out.extend(textwrap.wrap(text, width=width, initial_indent=indent + marker, subsequent_indent=indent + " " * len(marker),
                                     break_long_words=False, break_on_hyphens=False) or [indent + marker.rstrip()])
            block = None

    for line in lines:
        s = line.lstrip()
        if s.startswith("```"):
            flush(); fence = not fence; out.append(line); continue
        if fence or not s or s.startswith("|") or line.startswith("#") or s.startswith("<") or re.match(r"^\s*[-=]{3,}\s*$", line):
            flush(); out.append(line); continue
        m = MARK.match(line)
        if block is not None and not m and s.startswith("**") and block[1] == "":      # a bold lead-in opens a new paragraph
            flush(); out.append("")
        if m:
            flush(); block = [m.group(1), m.group(2), line[m.end():].strip()]
        elif block is not None:
            block[2] += " " + s.rstrip()
        else:
            block = [line[:len(line) - len(s)], "", s.rstrip()]
    flush()
    open(path, "w").write("\n".join(out))


if __name__ == "__main__":
    wrap_file(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 160)
