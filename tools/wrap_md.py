"""Hard-wrap a Markdown file at 160 columns (VERDICT r5 housekeeping): paragraphs and list items are re-flowed at word boundaries (continuation lines of a list
item are indented to its text), code fences and table rows are left alone.  usage: python tools/wrap_md.py FILE [WIDTH]"""
import re
import sys
import textwrap


def wrap_file(path, width=160):
    out, fence = [], False
    for line in open(path).read().split("\n"):
        if line.lstrip().startswith("```"):
            fence = not fence
            out.append(line)
            continue
        if fence or len(line) <= width or line.lstrip().startswith("|") or line.startswith("#"):
            out.append(line)
            continue
        m = re.match(r"^(\s*)((?:[-*+]|\d+\.)\s+)?", line)
        indent, marker = m.group(1), m.group(2) or ""
        body = line[len(indent) + len(marker):]
        sub = indent + " " * len(marker)
        out.extend(textwrap.wrap(body, width=width, initial_indent=indent + marker, subsequent_indent=sub, break_long_words=False, break_on_hyphens=False))
    open(path, "w").write("\n".join(out))


if __name__ == "__main__":
    wrap_file(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 160)
