#!/usr/bin/env python3
"""Wrap source lines longer than LIMIT columns at statement boundaries (after `;` / `{` outside parentheses), then at `, ` / ` && ` / ` || `,
and long `//` comments at word boundaries.  Whitespace-only: `--check` proves it by comparing the token streams (comments and white space
removed, string and character literals kept verbatim) of the old and the new text.  Preprocessor lines, macro continuations, block comments
and raw strings are left alone.
usage: tools/wrap_long_lines.py [--limit 150] [--check] files..."""
import re
import sys

LIMIT = 150


def scan(line):
    """-> (code_end, marks): code_end = index where a trailing // comment starts (or len); marks = [(index_after, kind, depth)] split candidates.
    A `;` ends a statement when the innermost open bracket (of this line) is a brace or there is none — `for (;;)` headers stay whole, lambda
    bodies inside a call's parentheses split."""
    i, n, stack, marks = 0, len(line), [], []
    while i < n:
        c = line[i]
        if c == '"' or c == "'":
            q = c
            i += 1
            while i < n and line[i] != q:
                i += 2 if line[i] == "\\" else 1
            i += 1
            continue
        if c == "/" and i + 1 < n and line[i + 1] == "/":
            return i, marks
        if c == "/" and i + 1 < n and line[i + 1] == "*":
            j = line.find("*/", i + 2)
            if j < 0:
                return None, None   # block comment runs on: leave the line alone
            i = j + 2
            continue
        in_stmt_ctx = not stack or stack[-1] == "{"
        if c in "([":
            stack.append(c)
        elif c in ")]":
            if stack and stack[-1] != "{":
                stack.pop()
        elif c == "{":
            # a block (or lambda body) opens when something other than an initializer context precedes it
            prev = line[:i].rstrip()[-1:] if line[:i].rstrip() else ""
            block = prev in (")", "", "e", "o", "y", "{", ";", "}") or line[:i].rstrip().endswith(("else", "do", "try", "const", "mutable", "noexcept"))
            stack.append("{" if block else "[")
            if block and line[i + 1:i + 2] == " " and line[i + 2:i + 3] != "}":
                marks.append((i + 1, "stmt", len(stack)))
        elif c == "}":
            if stack:
                stack.pop()
        elif c == ";" and in_stmt_ctx:
            marks.append((i + 1, "stmt", len(stack)))
        elif c == "," and i + 1 < n and line[i + 1] == " ":
            marks.append((i + 1, "expr", len(stack)))
        elif line.startswith(" && ", i) or line.startswith(" || ", i):
            marks.append((i, "expr", len(stack)))
        elif line.startswith(" ? ", i) or line.startswith(" : ", i):
            marks.append((i, "expr", len(stack) + 1))
        i += 1
    return n, marks


def wrap_comment(indent, text, limit):
    # text starts with "//"
    m = re.match(r"(//+!?\s*)", text)
    lead = m.group(1) if m else "// "
    body = text[len(lead):]
    if "|" in body[:3] or body.startswith("  "):   # tables, diagrams: leave
        return [indent + text]
    out, cur = [], ""
    for w in body.split(" "):
        if cur and len(indent) + len(lead) + len(cur) + 1 + len(w) > limit:
            out.append(indent + lead + cur)
            cur = w
        else:
            cur = w if not cur else cur + " " + w
    out.append(indent + lead + cur)
    cont = "//" + " " * (len(lead) - 2) if len(lead) >= 2 else lead
    return [out[0]] + [indent + cont + o[len(indent) + len(lead):] for o in out[1:]]


def wrap_code(indent, code, limit, extra=""):
    """code without indent, no trailing comment"""
    if len(indent) + len(code) <= limit:
        return [indent + code]
    end, marks = scan(code)
    if end is None or end != len(code):
        return [indent + code]
    stmt = [m[0] for m in marks if m[1] == "stmt" and m[0] < len(code) and code[m[0]:].strip()]
    pieces, last = [], 0
    for s in stmt:
        pieces.append(code[last:s])
        last = s
    pieces.append(code[last:])
    pieces = [p for p in pieces if p != ""]
    out, cur, level, cur_level = [], "", 0, 0
    cur_is_cont = False
    for p in pieces:
        ps = p.strip()
        # a continuation line that opens with an unbraced if / for / while / else clause holds nothing else: a statement behind it on the same
        # line reads as guarded (and trips -Wmisleading-indentation)
        lone = bool(out or cur_is_cont) and re.match(r"(if|for|while|else)\b", cur) and not cur.endswith("{") if cur else False
        if not cur:
            cur, cur_level = ps, level
        elif not lone and len(indent) + 2 * max(cur_level, 0) + len(cur) + 1 + len(ps) <= limit:
            cur += " " + ps
        else:
            out.append((cur_level, cur))
            cur, cur_level = ps, level
        e, _ = scan(ps)
        body = ps if e is None else ps[:e]
        level += body.count("{") - body.count("}")
        if ps.startswith("}"):
            pass
    out.append((cur_level, cur))
    lines = []
    for k, (lv, text) in enumerate(out):
        ind = indent + ("" if k == 0 else "  " * max(lv, 0) + ("  " if lv <= 0 else ""))
        if text.startswith("}") and k > 0:
            ind = indent + "  " * max(lv - 1, 0) + ("  " if lv - 1 <= 0 else "")
        if len(ind) + len(text) > limit:
            lines += wrap_expr(ind, text, limit)
        else:
            lines.append(ind + text)
    return lines


def wrap_expr(indent, text, limit):
    end, marks = scan(text)
    if end is None or end != len(text):
        return [indent + text]
    cands = sorted(set((m[0], m[2]) for m in marks if m[1] == "expr" and 0 < m[0] < len(text)))
    if not cands:
        return [indent + text]
    out, start = [], 0
    cont = indent + "    "
    while True:
        ind = indent if not out else cont
        room = limit - len(ind)
        if len(text) - start <= room:
            out.append(ind + text[start:].strip())
            break
        fit = [(p, d) for p, d in cands if start + room // 3 < p <= start + room]
        if fit:
            dmin = min(d for _, d in fit)
            best = max(p for p, d in fit if d == dmin)   # the outermost bracket level, as far right as fits
        else:
            nxt = [p for p, _ in cands if p > start]
            if not nxt:
                out.append(ind + text[start:].strip())
                break
            best = nxt[0]
        out.append(ind + text[start:best].strip())
        start = best
    return out


def process(text, limit):
    out = []
    lines = text.split("\n")
    in_block = False
    in_macro = False
    in_raw = False
    for ln in lines:
        raw_toggle = ln.count('R"') % 2 == 1 and not in_raw
        skip = in_block or in_macro or in_raw or ln.lstrip().startswith("#") or len(ln) <= limit or "\t" in ln
        # state for the following lines
        if in_raw:
            if ')"' in ln or ")'''" in ln:
                in_raw = False
        elif 'R"' in ln and ln.count('R"') > ln.count(')"'):
            in_raw = True
        if not in_raw:
            s = re.sub(r'"(\\.|[^"\\])*"', '""', ln)
            s2 = re.sub(r"//.*", "", s)
            opens, closes = s2.count("/*"), s2.count("*/")
            if in_block:
                if closes > opens - 0 and "*/" in s2:
                    in_block = False
                    if s2.rfind("/*") > s2.rfind("*/"):
                        in_block = True
            elif opens > closes:
                in_block = True
        in_macro_next = ln.endswith("\\")
        if skip or in_macro_next:
            out.append(ln)
            in_macro = in_macro_next
            continue
        in_macro = False
        indent = ln[:len(ln) - len(ln.lstrip())]
        body = ln.strip()
        end, marks = scan(body)
        if end is None:
            out.append(ln)
            continue
        code, comment = body[:end].rstrip(), body[end:]
        if not code:
            out += wrap_comment(indent, comment, limit)
            continue
        res = []
        if comment:
            if len(indent) + len(code) + 3 + len(comment) <= limit:
                out.append(ln)
                continue
            res += wrap_comment(indent, comment, limit)   # the trailing remark goes in front of the statement it belongs to
        res += wrap_code(indent, code, limit)
        out += res
    return "\n".join(out)


def tokens(text):
    """token stream without comments / white space; literals verbatim"""
    toks, i, n = [], 0, len(text)
    cur = []
    def flush():
        if cur:
            toks.append("".join(cur)); cur.clear()
    while i < n:
        c = text[i]
        if text.startswith('R"', i) and (i == 0 or not (text[i - 1].isalnum() or text[i - 1] == "_")):
            j = text.index("(", i)
            delim = text[i + 2:j]
            k = text.index(")" + delim + '"', j)
            flush(); toks.append(text[i:k + len(delim) + 2]); i = k + len(delim) + 2
            continue
        if c == '"' or c == "'":
            j = i + 1
            while text[j] != c:
                j += 2 if text[j] == "\\" else 1
            flush(); toks.append(text[i:j + 1]); i = j + 1
            continue
        if text.startswith("//", i):
            j = text.find("\n", i)
            j = n if j < 0 else j
            # a comment ending in a backslash continues: not expected here
            flush(); i = j
            continue
        if text.startswith("/*", i):
            j = text.index("*/", i)
            flush(); i = j + 2
            continue
        if c == "\n":
            flush()
            # keep line structure of preprocessor lines: a '#' directive ends at its newline
            toks.append("\n") if toks and "\x00pp" in toks[-8:] else None
            i += 1
            continue
        if c.isspace():
            flush(); i += 1
            continue
        if c.isalnum() or c == "_":
            cur.append(c)
        else:
            flush(); toks.append(c)
        i += 1
    flush()
    return toks


def main():
    global LIMIT
    args = sys.argv[1:]
    check = False
    files = []
    k = 0
    while k < len(args):
        if args[k] == "--limit":
            LIMIT = int(args[k + 1]); k += 2
        elif args[k] == "--check":
            check = True; k += 1
        else:
            files.append(args[k]); k += 1
    bad = 0
    for f in files:
        old = open(f).read()
        new = process(old, LIMIT)
        if tokens(old) != tokens(new):
            print("TOKEN MISMATCH", f); bad += 1
            continue
        # preprocessor lines untouched, in order
        if [l for l in old.split("\n") if l.lstrip().startswith("#")] != [l for l in new.split("\n") if l.lstrip().startswith("#")]:
            print("PREPROCESSOR MISMATCH", f); bad += 1
            continue
        longs_old = sum(1 for l in old.split("\n") if len(l) > LIMIT)
        longs_new = sum(1 for l in new.split("\n") if len(l) > LIMIT)
        print("%-50s lines %5d -> %5d   over %d columns: %4d -> %4d" % (f, old.count("\n"), new.count("\n"), LIMIT, longs_old, longs_new))
        if not check and new != old:
            open(f, "w").write(new)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
