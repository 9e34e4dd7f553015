"""A small parser for the subset of Go composite-literal syntax the reference's table-driven tests use.

It exists only to TRANSCRIBE the reference's test tables into data (tests/golden/*.json, written by
tests/golden/transcribe.py, which runs in the build container where /root/reference is mounted).
No reference code is executed or copied: the parser reads literals — struct/slice/map literals, calls
such as resource.MustParse("2Gi"), identifiers — and returns plain Python values:

    T{a: 1, b: "x"}      -> {"a": 1, "b": "x"}          (keyed literal -> dict; keys that are
    []T{x, y}            -> [x, y]                        identifiers/selectors stay strings)
    f(a, b)              -> Call("f", [a, b])
    x.M(a).N()           -> Call(".N", [Call(".M", [x, a])])   (method chains on call results)
    pkg.Name             -> Ident("pkg.Name")
    &x, *x               -> x
    20*1024*1024         -> 20971520
    func(...) T { ... }  -> Func(source text)
"""
from __future__ import annotations

import re
from dataclasses import dataclass
from typing import Any, List


@dataclass(frozen=True)
class Ident:
    name: str

    def __repr__(self):
        return f"Ident({self.name})"


@dataclass
class Call:
    fn: str
    args: List[Any]


@dataclass
class Func:
    src: str


_TOKEN = re.compile(r"""
    (?P<ws>\s+|//[^\n]*|/\*.*?\*/)
  | (?P<str>"(?:\\.|[^"\\])*"|`[^`]*`)
  | (?P<num>\d+(?:\.\d+)?)
  | (?P<id>[A-Za-z_]\w*)
  | (?P<op><<|>>|[{}()\[\],:.*&+\-/])
  | (?P<other>.)
""", re.S | re.X)


def tokenize(src: str):
    pos, out = 0, []
    while pos < len(src):
        m = _TOKEN.match(src, pos)
        if not m:
            raise SyntaxError(f"cannot tokenize at {pos}: {src[pos:pos+40]!r}")
        pos = m.end()
        if m.lastgroup == "ws":
            continue
        out.append((m.lastgroup, m.group(), m.start()))
    return out


class Parser:
    def __init__(self, src: str):
        self.src = src
        self.t = tokenize(src)
        self.i = 0

    def peek(self, k=0):
        return self.t[self.i + k][1] if self.i + k < len(self.t) else None

    def kind(self, k=0):
        return self.t[self.i + k][0] if self.i + k < len(self.t) else None

    def take(self, want=None):
        tok = self.t[self.i]
        if want is not None and tok[1] != want:
            raise SyntaxError(f"expected {want!r}, got {tok[1]!r} at {tok[2]}: {self.src[tok[2]:tok[2]+60]!r}")
        self.i += 1
        return tok[1]

    # -- types are skipped, never interpreted
    def skip_balanced(self, open_, close):
        depth = 0
        while True:
            t = self.take()
            if t == open_:
                depth += 1
            elif t == close:
                depth -= 1
                if depth == 0:
                    return

    def skip_type(self):
        """Consumes a type expression that precedes a '{' literal body."""
        while True:
            t = self.peek()
            if t == "[":
                self.skip_balanced("[", "]")
            elif t in ("*",):
                self.take()
            elif t == "map":
                self.take()
                self.skip_balanced("[", "]")
            elif t == "struct":
                self.take()
                self.skip_balanced("{", "}")
                return
            elif self.kind() == "id":
                if self.take() == "interface" and self.peek() == "{" and self.peek(1) == "}":  # interface{} as an element type
                    self.take()
                    self.take()
                    return
                while self.peek() == ".":
                    self.take()
                    self.take()
                if self.peek() == "[":  # a generic instantiation, e.g. map[string]sets.Set[string]{...}
                    self.skip_balanced("[", "]")
                return
            else:
                return

    def parse_expr(self):
        left = self.parse_unary()
        while self.peek() in ("*", "+", "-", "/", "<<", ">>") and self.kind(1) in ("num", "id", "op"):
            # arithmetic only between numbers (e.g. 20*1024*1024); a following '*' that starts a
            # pointer deref inside a literal never occurs after a complete operand in these tables
            op = self.take()
            right = self.parse_unary()
            if isinstance(left, (int, float)) and isinstance(right, (int, float)):
                both_int = isinstance(left, int) and isinstance(right, int)
                left = {"*": lambda: left * right, "+": lambda: left + right, "-": lambda: left - right,
                        "/": lambda: left // right if both_int else left / right,
                        "<<": lambda: left << right, ">>": lambda: left >> right}[op]()
            else:
                left = Call("op" + op, [left, right])
        return left

    def parse_unary(self):
        if self.peek() in ("*", "&"):
            self.take()
            return self.parse_unary()
        if self.peek() == "-":
            self.take()
            v = self.parse_unary()
            return -v
        return self.parse_primary()

    def parse_body(self):
        """'{' elements '}' -> dict (if keyed) or list."""
        self.take("{")
        items, keyed = [], False
        while self.peek() != "}":
            v = self.parse_expr()
            if self.peek() == ":":
                self.take()
                val = self.parse_expr()
                items.append((v, val))
                keyed = True
            else:
                items.append((None, v))
            if self.peek() == ",":
                self.take()
        self.take("}")
        if keyed:
            out = {}
            for k, v in items:
                if isinstance(k, Call) and len(k.args) == 1:  # type conversion used as a map key, e.g. v1.ResourceName(x)
                    k = k.args[0]
                key = k.name if isinstance(k, Ident) else k
                out[key] = v
            return out
        return [v for _, v in items]

    def parse_primary(self):
        k, t = self.kind(), self.peek()
        if k == "str":
            self.take()
            return t[1:-1]
        if k == "num":
            self.take()
            return float(t) if "." in t else int(t)
        if t == "{":
            return self.parse_body()
        if t == "(":
            self.take()
            v = self.parse_expr()
            self.take(")")
            return v
        if t == "func":
            start = self.t[self.i][2]
            self.take()
            self.skip_balanced("(", ")")
            while self.peek() != "{":
                self.take()
            self.skip_balanced("{", "}")
            end = self.t[self.i][2] if self.i < len(self.t) else len(self.src)
            return Func(self.src[start:end])
        if t in ("[", "map", "struct"):
            self.skip_type()
            return self.parse_body()
        if k == "id":
            name = self.take()
            while self.peek() == "." and self.kind(1) == "id":
                self.take()
                name += "." + self.take()
            val: Any = Ident(name)
            while True:
                if self.peek() == "(":
                    self.take()
                    args = []
                    while self.peek() != ")":
                        args.append(self.parse_expr())
                        if self.peek() == ",":
                            self.take()
                        elif self.peek() == ".":  # variadic tail "x..."
                            while self.peek() == ".":
                                self.take()
                    self.take(")")
                    val = Call(name, args)
                elif self.peek() == "{" and isinstance(val, Ident):
                    val = self.parse_body()
                elif self.peek() == "[":
                    self.take()
                    idx = self.parse_expr()
                    self.take("]")
                    val = Call("index", [val, idx])
                elif self.peek() == "." and self.kind(1) == "id":  # selector or method call on a call/index result
                    self.take()
                    member = self.take()
                    if self.peek() == "(":  # builder chains: recv.Method(args) -> Call(".Method", [recv, args...])
                        self.take()
                        margs = []
                        while self.peek() != ")":
                            margs.append(self.parse_expr())
                            if self.peek() == ",":
                                self.take()
                            elif self.peek() == ".":
                                while self.peek() == ".":
                                    self.take()
                        self.take(")")
                        val = Call("." + member, [val] + margs)
                    else:
                        val = Call("select", [val, member])
                else:
                    return val
        raise SyntaxError(f"unexpected token {t!r} at {self.t[self.i][2]}: {self.src[self.t[self.i][2]:self.t[self.i][2]+60]!r}")


def parse_literal_after(src: str, marker: str, occurrence: int = 0):
    """Finds `marker` (e.g. 'tests := ') in src and parses the composite literal that follows it."""
    pos = -1
    for _ in range(occurrence + 1):
        pos = src.index(marker, pos + 1)
    p = Parser(src[pos + len(marker):])
    return p.parse_expr()


def line_of(src: str, needle: str, start: int = 0) -> int:
    pos = src.index(needle, start)
    return src.count("\n", 0, pos) + 1
