"""Re-wrap the prose of markdown files to a column limit (default 118) without touching tables, code fences, headings or link reference lines;
list items keep their bullet and get a hanging indent.  Used once per round on the documents whose paragraphs were written as single lines:
    python tools/wrap_md.py FILE [FILE ...] [--width N]"""
import re
import sys
import textwrap


def wrap_text(text, width=118):
    out, para, fence = [], [], False

    def flush():
        if not para:
            return
        first = para[0]
        m = re.match(r"^(\s*)([-*+]|\d+[.)])\s+", first)
        body = " ".join(l.strip() for l in para)
        if m:
            indent = m.group(1)
            bullet = m.group(2)
            body = body[len(bullet):].lstrip() if body.startswith(bullet) else body
            out.extend(textwrap.wrap(body, width=width, initial_indent=f"{indent}{bullet} ", subsequent_indent=indent + " " * (len(bullet) + 1),
                                     break_long_words=False, break_on_hyphens=False))
        else:
            indent = re.match(r"^(\s*)", first).group(1)
            if len(indent) >= 4:   # indented block: leave as written
                out.extend(para)
            else:
                out.extend(textwrap.wrap(body, width=width, initial_indent=indent, subsequent_indent=indent, break_long_words=False, break_on_hyphens=False))
        para.clear()

    for line in text.split("\n"):
        if line.strip().startswith("```"):
            flush(); fence = not fence; out.append(line); continue
        if fence:
            out.append(line); continue
        if not line.strip():
            flush(); out.append(""); continue
        if line.lstrip().startswith(("|", "#", ">")) or re.match(r"^\s*\[[^\]]+\]:", line):
            flush(); out.append(line); continue
        if re.match(r"^\s*([-*+]|\d+[.)])\s+", line):
            flush()
        # documents written one paragraph per line: a very long line is a paragraph of its own, and a line that opens with bold text starts one
        if len(line) > 300 or line.startswith("**") or (para and len(para[-1]) > 300):
            had = bool(para)
            flush()
            if had and out and out[-1] != "":
                out.append("")
        para.append(line)
    flush()
    # a wrapped continuation line must not open with a character that would make it a heading / table row / quote: pull the previous line's last word down
    for i in range(1, len(out)):
        if out[i][:1] in "#|>" and out[i - 1].strip() and not out[i - 1].lstrip().startswith(("#", "|", ">")) and " " in out[i - 1].strip():
            head, last = out[i - 1].rsplit(" ", 1)
            if not text_has_line(text, out[i]):
                out[i - 1] = head; out[i] = last + " " + out[i]
    return "\n".join(out)


def text_has_line(text, line):
    return ("\n" + line + "\n") in ("\n" + text + "\n")


if __name__ == "__main__":
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    width = 118
    if "--width" in sys.argv:
        width = int(sys.argv[sys.argv.index("--width") + 1]); args = [a for a in args if a != str(width)]
    for f in args:
        s = open(f).read()
        open(f, "w").write(wrap_text(s, width))
        print(f, "max line", max(len(l) for l in open(f).read().split("\n")))
