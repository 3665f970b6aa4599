#!/usr/bin/env python3
"""Mini unifdef: resolve the preprocessor conditionals that depend ONLY on the given macros and drop the dead branches.
    tools/unifdef.py FILE -DNAME=VALUE ... -UNAME ...  > out
Conditionals that mention any other macro are left alone.  `#ifndef X / #define X v / #endif` default blocks of a known X vanish
(known defined), so the macro's C-level uses must be cleaned by hand afterwards (the tool lists them on stderr)."""
import re
import sys

path = sys.argv[1]
defs, undefs = {}, set()
for a in sys.argv[2:]:
    if a.startswith("-D"):
        k, _, v = a[2:].partition("=")
        defs[k] = int(v or "1")
    elif a.startswith("-U"):
        undefs.add(a[2:])
known = set(defs) | undefs


def evaluate(expr):
    """Returns True / False, or None when the expression mentions an unknown identifier."""
    e = re.sub(r"//.*$", "", expr)
    e = re.sub(r"/\*.*?\*/", "", e).strip()
    e = re.sub(r"defined\s*\(\s*(\w+)\s*\)", lambda m: ("1" if m.group(1) in defs else "0") if m.group(1) in known else "UNKNOWN_" + m.group(1), e)
    e = re.sub(r"defined\s+(\w+)", lambda m: ("1" if m.group(1) in defs else "0") if m.group(1) in known else "UNKNOWN_" + m.group(1), e)

    def ident(m):
        n = m.group(0)
        if n in defs:
            return str(defs[n])
        if n in undefs:
            return "0"
        return "UNKNOWN_" + n
    e = re.sub(r"\b[A-Za-z_]\w*\b", ident, e)
    if "UNKNOWN_" in e:
        return None
    e = e.replace("&&", " and ").replace("||", " or ").replace("!", " not ")
    e = e.replace(" not =", "!=")
    return bool(eval(e))


out = []
stack = []          # entries: dict(known=bool, taken=bool (some branch already taken), active=bool (current branch emitted))
lines = open(path).read().split("\n")


def emitting():
    return all(s["active"] for s in stack)


for ln in lines:
    s = ln.strip()
    m = re.match(r"#\s*(ifdef|ifndef|if|elif|else|endif)\b(.*)", s)
    if not m:
        if emitting():
            out.append(ln)
        continue
    kw, rest = m.group(1), m.group(2)
    if kw in ("if", "ifdef", "ifndef"):
        if kw == "ifdef":
            name = rest.split()[0]
            val = (name in defs) if name in known else None
        elif kw == "ifndef":
            name = rest.split()[0]
            val = (name not in defs) if name in known else None
        else:
            val = evaluate(rest)
        parent = emitting()
        if val is None:
            stack.append(dict(known=False, taken=True, active=True, parent=parent))
            if parent:
                out.append(ln)
        else:
            stack.append(dict(known=True, taken=val, active=val, parent=parent))
    elif kw == "elif":
        top = stack[-1]
        if not top["known"]:
            if all(s_["active"] for s_ in stack[:-1]):
                out.append(ln)
        else:
            if top["taken"]:
                top["active"] = False
            else:
                val = evaluate(rest)
                if val is None:
                    raise SystemExit("%s: #elif on unknown macros inside a known conditional: %s" % (path, ln))
                top["active"] = val
                top["taken"] = val
    elif kw == "else":
        top = stack[-1]
        if not top["known"]:
            if all(s_["active"] for s_ in stack[:-1]):
                out.append(ln)
        else:
            top["active"] = not top["taken"]
            top["taken"] = True
    else:
        top = stack.pop()
        if not top["known"] and emitting():
            out.append(ln)
assert not stack, "unbalanced conditionals"
text = "\n".join(out)
for k in sorted(known):
    uses = [i + 1 for i, l in enumerate(out) if re.search(r"\b%s\b" % k, l)]
    if uses:
        sys.stderr.write("%s still mentioned on output lines %s\n" % (k, uses))
sys.stdout.write(text)
