#!/usr/bin/env python3
"""One-off source cleaner (round 5): resolve the preprocessor conditionals of MEASURED-AND-CLOSED experiment switches in a source file at their shipped values and
delete the dead branches -- a minimal `unifdef`.  Conditionals that mention any other macro are left alone.

    python tools/strip_experiments.py etude_amd/csrc/dec_kernels.hip NAME=VALUE ... NAME=undef ...

The result must preprocess to the same token stream as before (check: hipcc -E of both, line markers and blank lines stripped)."""
import re
import sys


def main():
    path = sys.argv[1]
    known = {}
    for kv in sys.argv[2:]:
        k, v = kv.split("=")
        known[k] = None if v == "undef" else int(v)
    lines = open(path).read().split("\n")

    def evaluate(expr):
        """-> True / False, or None if the expression mentions a macro that is not in `known`"""
        e = re.sub(r"/\*.*?\*/", "", expr).strip()
        names = set(re.findall(r"[A-Za-z_]\w*", e)) - {"defined"}
        if not names or not names <= set(known):
            return None
        e = re.sub(r"defined\s*\(\s*(\w+)\s*\)", lambda m: "1" if known[m.group(1)] is not None else "0", e)
        e = re.sub(r"defined\s+(\w+)", lambda m: "1" if known[m.group(1)] is not None else "0", e)
        e = re.sub(r"[A-Za-z_]\w*", lambda m: str(known[m.group(0)] if known[m.group(0)] is not None else 0), e)
        e = e.replace("&&", " and ").replace("||", " or ").replace("!", " not ").replace(" not =", "!=")
        return bool(eval(e))                     # noqa: S307  (integers and boolean operators only, by construction)

    out = []
    # stack entries: dict(resolved: bool, taken: bool (a branch was already emitted), emit: bool (current branch is live), parent_emit)
    stack = []
    emit = True
    for ln in lines:
        m = re.match(r"\s*#\s*(if|ifdef|ifndef|elif|else|endif)\b(.*)", ln)
        if not m:
            if emit:
                out.append(ln)
            continue
        d, rest = m.group(1), m.group(2)
        if d in ("if", "ifdef", "ifndef"):
            if d == "if":
                v = evaluate(rest)
            else:
                name = re.match(r"\s*(\w+)", rest).group(1)
                v = None if name not in known else ((known[name] is not None) == (d == "ifdef"))
            if not emit:
                stack.append(dict(resolved=True, taken=True, parent=emit, live=False)); continue      # inside a dead branch: swallow everything
            if v is None:
                stack.append(dict(resolved=False, parent=emit)); out.append(ln)
            else:
                stack.append(dict(resolved=True, taken=v, parent=emit, live=v)); emit = v
        elif d == "elif":
            top = stack[-1]
            if not top["resolved"]:
                out.append(ln)
            elif not top["parent"]:
                pass
            else:
                v = evaluate(rest)
                assert v is not None, "mixed known / unknown #elif chain: " + ln
                emit = (not top["taken"]) and v
                top["taken"] = top["taken"] or v
        elif d == "else":
            top = stack[-1]
            if not top["resolved"]:
                out.append(ln)
            elif top["parent"]:
                emit = not top["taken"]
        else:
            top = stack.pop()
            if not top["resolved"]:
                out.append(ln)
            emit = top["parent"]
    assert not stack
    open(path, "w").write("\n".join(out))


if __name__ == "__main__":
    main()
