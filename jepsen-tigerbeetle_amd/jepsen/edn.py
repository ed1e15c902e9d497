"""EDN reader / writer for Jepsen histories (store/<test>/<time>/history.edn: one op map per
line) -- SURVEY.md section 8f row 1.  The reference ignores its own store/ directory
(/root/reference/.gitignore:7) and `serve` only browses it (core.clj:289); with this the
checker can consume a stored history, and anyone with a JVM can produce the stock-Knossos
verdict for the same file to close the parity gap.

Mapping: keywords -> str without the colon (so ops look like knossos/op.py's: {"type": "ok",
"f": "read", ...}), maps -> dict, vectors / lists -> list, sets -> frozenset, nil -> None,
tagged values keep their payload (#inst "..." -> str, #jepsen.history.Op{...} -> dict).
`independent` tuples are written by Jepsen as [k v] vectors: pass tuple_keys=True to
read_history to turn every 2-vector :value back into an independent.Tuple.
"""
from __future__ import annotations

_WS = " \t\r\n,"
_DELIM = _WS + "()[]{}\";"


class EDNError(ValueError):
    pass


class _P:
    def __init__(self, s):
        self.s, self.i, self.n = s, 0, len(s)

    def ws(self):
        s, n = self.s, self.n
        while self.i < n:
            c = s[self.i]
            if c in _WS:
                self.i += 1
            elif c == ";":
                while self.i < n and s[self.i] != "\n":
                    self.i += 1
            else:
                break

    def value(self):
        self.ws()
        if self.i >= self.n:
            raise EDNError("unexpected end of input")
        s, c = self.s, self.s[self.i]
        if c == "{":
            self.i += 1
            return self.map()
        if c == "[":
            self.i += 1
            return self.seq("]")
        if c == "(":
            self.i += 1
            return self.seq(")")
        if c == '"':
            return self.string()
        if c == "#":
            if s.startswith("#{", self.i):
                self.i += 2
                return frozenset(_hashable(x) for x in self.seq("}"))
            if s.startswith("#_", self.i):       # discard
                self.i += 2
                self.value()
                return self.value()
            self.i += 1
            self.token()                          # tag: keep the payload
            return self.value()
        if c == "\\":
            self.i += 1
            t = self.token()
            return {"newline": "\n", "space": " ", "tab": "\t", "return": "\r"}.get(t, t[:1])
        t = self.token()
        if t == "nil":
            return None
        if t == "true":
            return True
        if t == "false":
            return False
        if t[0] == ":":
            return t[1:]
        try:
            if t[-1] in "NM":
                t = t[:-1]
            return int(t)
        except ValueError:
            pass
        try:
            return float(t)
        except ValueError:
            return t                              # symbol

    def token(self):
        s, j = self.s, self.i
        while j < self.n and s[j] not in _DELIM:
            j += 1
        if j == self.i:
            raise EDNError(f"unexpected {s[self.i]!r} at {self.i}")
        t = s[self.i:j]
        self.i = j
        return t

    def string(self):
        s = self.s
        self.i += 1
        out = []
        while True:
            if self.i >= self.n:
                raise EDNError("unterminated string")
            c = s[self.i]
            self.i += 1
            if c == '"':
                return "".join(out)
            if c == "\\":
                e = s[self.i]
                self.i += 1
                out.append({"n": "\n", "t": "\t", "r": "\r", '"': '"', "\\": "\\"}.get(e, e))
            else:
                out.append(c)

    def seq(self, close):
        out = []
        while True:
            self.ws()
            if self.i >= self.n:
                raise EDNError(f"missing {close}")
            if self.s[self.i] == close:
                self.i += 1
                return out
            out.append(self.value())

    def map(self):
        items = self.seq("}")
        if len(items) % 2:
            raise EDNError("map with an odd number of forms")
        return {_hashable(items[i]): items[i + 1] for i in range(0, len(items), 2)}


def _hashable(x):
    if isinstance(x, list):
        return tuple(_hashable(y) for y in x)
    if isinstance(x, dict):
        return tuple(sorted((k, _hashable(v)) for k, v in x.items()))
    return x


def loads(s: str):
    p = _P(s)
    v = p.value()
    p.ws()
    if p.i != p.n:
        raise EDNError(f"trailing data at {p.i}")
    return v


def loads_all(s: str):
    p, out = _P(s), []
    while True:
        p.ws()
        if p.i >= p.n:
            return out
        out.append(p.value())


def read_history(path_or_lines, tuple_keys=False):
    """history.edn -> list of op dicts (one top-level form per op; a single top-level vector of
    ops is accepted too)."""
    if isinstance(path_or_lines, str):
        with open(path_or_lines) as fh:
            text = fh.read()
    else:
        text = "\n".join(path_or_lines)
    forms = loads_all(text)
    if len(forms) == 1 and isinstance(forms[0], list):
        forms = forms[0]
    ops = []
    for f in forms:
        if not isinstance(f, dict):
            raise EDNError(f"not an op map: {f!r}")
        client = isinstance(f.get("process"), int) and not isinstance(f.get("process"), bool)
        if tuple_keys and client and isinstance(f.get("value"), list) and len(f["value"]) == 2:
            from . import independent
            f = dict(f, value=independent.Tuple(f["value"][0], f["value"][1]))
        ops.append(f)
    return ops


_KEYWORD_VALUED = {"type", "f"}


def dumps(v, _key=None) -> str:
    if v is None:
        return "nil"
    if v is True:
        return "true"
    if v is False:
        return "false"
    if isinstance(v, str):
        if _key in _KEYWORD_VALUED or _key == "process":
            return ":" + v
        return '"' + v.replace("\\", "\\\\").replace('"', '\\"') + '"'
    if isinstance(v, (int, float)):
        return repr(v)
    if isinstance(v, dict):
        return "{" + ", ".join(f":{k} {dumps(x, k)}" if isinstance(k, str) else f"{dumps(k)} {dumps(x)}"
                               for k, x in v.items()) + "}"
    if isinstance(v, (set, frozenset)):
        return "#{" + " ".join(dumps(x) for x in sorted(v, key=repr)) + "}"
    if isinstance(v, (list, tuple)):
        return "[" + " ".join(dumps(x) for x in v) + "]"
    raise EDNError(f"cannot write {type(v).__name__}")


def write_history(path, history):
    with open(path, "w") as fh:
        for op in history:
            fh.write(dumps({k: v for k, v in op.items() if not k.endswith("?") or k == "final?"}) + "\n")
