"""Re-wraps a markdown file at WIDTH columns (paragraphs and list items; tables, headings, code fences and html are left alone)."""
import re, sys, textwrap
WIDTH = 128
src = open(sys.argv[1]).read().split("\n")
out, i, in_code = [], 0, False
bullet = re.compile(r"^(\s*)([*\-]|\d+\.)\s+")
def flush(block, first, rest):
    text = " ".join(s.strip() for s in block)
    out.extend(textwrap.fill(text, WIDTH, initial_indent=first, subsequent_indent=rest, break_long_words=False, break_on_hyphens=False).split("\n"))
while i < len(src):
    ln = src[i]
    if ln.strip().startswith("```"):
        in_code = not in_code
        out.append(ln); i += 1; continue
    if in_code or not ln.strip() or ln.lstrip().startswith(("|", "#", "<", "---")):
        out.append(ln); i += 1; continue
    m = bullet.match(ln)
    if m:
        first = m.group(0)
        rest = " " * len(first)
        block = [ln[len(first):]]
        i += 1
        while i < len(src) and src[i].strip() and not bullet.match(src[i]) and not src[i].lstrip().startswith(("|", "#", "```")) and (len(src[i]) - len(src[i].lstrip()) >= len(m.group(1)) + 1 or not src[i].startswith(" ")) :
            # continuation: either indented under the bullet or a lazy continuation line
            if not src[i].startswith(" ") and len(m.group(1)) == 0 and False:
                break
            block.append(src[i]); i += 1
        flush(block, first, rest)
        continue
    ind = ln[:len(ln) - len(ln.lstrip())]
    block = [ln]
    i += 1
    while i < len(src) and src[i].strip() and not bullet.match(src[i]) and not src[i].lstrip().startswith(("|", "#", "```")):
        block.append(src[i]); i += 1
    flush(block, ind, ind)
open(sys.argv[2] if len(sys.argv) > 2 else sys.argv[1], "w").write("\n".join(out))
