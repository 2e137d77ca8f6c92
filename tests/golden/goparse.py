"""A small parser/evaluator for the subset of Go composite-literal syntax used by the table-driven tests
of the reference (internal/scheduler/**/*_test.go).  Used ONLY by extract_reference_goldens.py, which
runs in the build container (where /root/reference exists) to turn the reference's own test tables
into the JSON fixtures committed under tests/golden/.

It parses expressions (identifiers, selectors, calls, composite literals with elided inner types,
basic literals, unary/binary arithmetic) into an AST and evaluates them against an environment of
Python callables that re-implement the reference's test fixtures (see gofixtures.py).
Immediately-invoked func literals (`func() T { ... }()`) are supported for the statement forms the tables use: `var x T`,
`x := e`, `x = e` / `x.F = e`, three-clause `for` loops with `i++`, expression statements and `return e`.
Anything outside the subset raises Unsupported so the case is skipped and reported rather than silently mis-transcribed.
"""
from __future__ import annotations

import re
from collections import ChainMap
from typing import Any, List, Optional, Tuple


class Unsupported(Exception):
    pass


TOKEN_RE = re.compile(
    r"""
    (?P<ws>[ \t\r\n]+) |
    (?P<comment>//[^\n]*|/\*.*?\*/) |
    (?P<float>\d+\.\d*(?:[eE][+-]?\d+)?|\d+[eE][+-]?\d+) |
    (?P<int>0x[0-9a-fA-F]+|\d+) |
    (?P<str>"(?:\\.|[^"\\])*"|`[^`]*`) |
    (?P<ident>[A-Za-z_][A-Za-z_0-9]*) |
    (?P<op>:=|==|!=|<=|>=|&&|\|\||[{}()\[\],:.*&\-+/<>=!;])
    """,
    re.X | re.S,
)


def tokenize(src: str, base_line: int = 1):
    toks = []
    pos, line = 0, base_line
    while pos < len(src):
        m = TOKEN_RE.match(src, pos)
        if not m:
            raise Unsupported(f"cannot tokenize at line {line}: {src[pos:pos+30]!r}")
        kind = m.lastgroup
        text = m.group()
        if kind not in ("ws", "comment"):
            toks.append((kind, text, line))
        line += text.count("\n")
        pos = m.end()
    toks.append(("eof", "", line))
    return toks


# ---- AST nodes are tuples: ("lit", v) ("name", dotted) ("call", fn, args) ("comp", type, elems) ("unary", op, x) ("bin", op, a, b)
#      type: ("slice", T) ("map", K, V) ("ptr", T) ("named", dotted) ; elems: list of (key_ast|None, value_ast)


class Parser:
    def __init__(self, toks):
        self.t = toks
        self.i = 0

    def peek(self, k=0):
        return self.t[self.i + k]

    def next(self):
        tok = self.t[self.i]
        self.i += 1
        return tok

    def accept(self, text):
        if self.peek()[1] == text and self.peek()[0] in ("op", "ident"):
            self.i += 1
            return True
        return False

    def expect(self, text):
        tok = self.next()
        if tok[1] != text:
            raise Unsupported(f"line {tok[2]}: expected {text!r}, got {tok[1]!r}")
        return tok

    # ---- types
    def looks_like_type(self) -> bool:
        k, t, _ = self.peek()
        return (k == "op" and t in ("[", "*")) or (k == "ident" and t in ("map", "struct"))

    def parse_type(self):
        k, t, ln = self.peek()
        if t == "[":
            self.next()
            if self.peek()[1] != "]":
                self.next()  # array length / ...
            self.expect("]")
            return ("slice", self.parse_type())
        if t == "map":
            self.next()
            self.expect("[")
            kt = self.parse_type()
            self.expect("]")
            return ("map", kt, self.parse_type())
        if t == "*":
            self.next()
            return ("ptr", self.parse_type())
        if t == "struct":
            raise Unsupported(f"line {ln}: anonymous struct type")
        if t == "func":
            raise Unsupported(f"line {ln}: func type")
        if k == "ident":
            name = self.next()[1]
            while self.peek()[1] == "." and self.peek(1)[0] == "ident":
                self.next()
                name += "." + self.next()[1]
            return ("named", name)
        raise Unsupported(f"line {ln}: bad type at {t!r}")

    # ---- expressions
    def parse_expr(self, prec=0):
        left = self.parse_unary()
        PREC = {"||": 1, "&&": 2, "==": 3, "!=": 3, "<": 3, ">": 3, "<=": 3, ">=": 3, "+": 4, "-": 4, "*": 5, "/": 5}
        while True:
            k, t, _ = self.peek()
            if t == "+" and self.peek(1)[1] == "+":  # i++ (statement), not a binary plus
                return left
            if k == "op" and t in PREC and PREC[t] > prec:
                self.next()
                right = self.parse_expr(PREC[t])
                left = ("bin", t, left, right)
            else:
                return left

    def parse_unary(self):
        k, t, ln = self.peek()
        if k == "op" and t in ("-", "+", "&", "!"):
            self.next()
            return ("unary", t, self.parse_unary())
        return self.parse_postfix(self.parse_primary())

    def parse_primary(self):
        k, t, ln = self.peek()
        if k == "int":
            self.next()
            return ("lit", int(t, 0))
        if k == "float":
            self.next()
            return ("lit", float(t))
        if k == "str":
            self.next()
            if t[0] == "`":
                return ("lit", t[1:-1])
            return ("lit", bytes(t[1:-1], "utf-8").decode("unicode_escape"))
        if t == "(":
            self.next()
            e = self.parse_expr()
            self.expect(")")
            return e
        if t == "func":
            return self.parse_funclit()
        if t == "{":  # elided-type composite literal
            return self.parse_composite(None)
        if self.looks_like_type():
            ty = self.parse_type()
            if self.peek()[1] == "{":
                return self.parse_composite(ty)
            if self.peek()[1] == "(":  # conversion T(x)
                self.next()
                e = self.parse_expr()
                self.expect(")")
                return ("conv", ty, e)
            raise Unsupported(f"line {ln}: type without literal")
        if k == "ident" and t == "make" and self.peek(1)[1] == "(":   # make([]T, n) / make(map[K]V): an empty container
            self.next(); self.next()
            ty = self.parse_type()
            while self.peek()[1] != ")":
                self.next()
            self.expect(")")
            return ("comp", ty, [])
        if k == "ident":
            self.next()
            if t in ("true", "false"):
                return ("lit", t == "true")
            if t == "nil":
                return ("lit", None)
            return ("name", t)
        raise Unsupported(f"line {ln}: unexpected token {t!r}")

    # ---- func literals: func() T { stmts }
    def parse_funclit(self):
        ln = self.peek()[2]
        self.expect("func")
        self.expect("(")
        if self.peek()[1] != ")":
            raise Unsupported(f"line {ln}: func literal with parameters")
        self.expect(")")
        if self.peek()[1] != "{":
            self.parse_type()  # result type
        return ("funclit", self.parse_block())

    def parse_block(self):
        self.expect("{")
        stmts = []
        while self.peek()[1] != "}":
            stmts.append(self.parse_stmt())
            self.accept(";")
        self.expect("}")
        return stmts

    def parse_simple_stmt(self):
        e = self.parse_expr()
        if self.accept(":="):
            return ("assign", e, self.parse_expr(), True)
        if self.accept("="):
            return ("assign", e, self.parse_expr(), False)
        if self.peek()[1] == "+" and self.peek(1)[1] == "+":
            self.next(); self.next()
            return ("inc", e)
        return ("expr", e)

    def parse_stmt(self):
        k, t, ln = self.peek()
        if t == "var":
            self.next()
            name = self.next()[1]
            ty = self.parse_type()
            init = None
            if self.accept("="):
                init = self.parse_expr()
            return ("var", name, ty, init)
        if t == "return":
            self.next()
            if self.peek()[1] == "}":
                return ("return", None)
            return ("return", self.parse_expr())
        if t == "for":
            self.next()
            init = self.parse_simple_stmt()
            self.expect(";")
            cond = self.parse_expr()
            self.expect(";")
            post = self.parse_simple_stmt()
            return ("for", init, cond, post, self.parse_block())
        if t in ("if", "switch", "go", "defer", "range"):
            raise Unsupported(f"line {ln}: statement {t}")
        return self.parse_simple_stmt()

    def parse_postfix(self, e):
        while True:
            k, t, ln = self.peek()
            if t == "." and self.peek(1)[0] == "ident":
                self.next()
                name = self.next()[1]
                if e[0] == "name":
                    e = ("name", e[1] + "." + name)
                else:
                    e = ("sel", e, name)
            elif t == "(":
                self.next()
                args = []
                while self.peek()[1] != ")":
                    args.append(self.parse_expr())
                    if self.peek()[1] == ".":  # variadic spread "..."
                        self.next(); self.next(); self.next()
                        args[-1] = ("spread", args[-1])
                    if not self.accept(","):
                        break
                self.expect(")")
                e = ("call", e, args)
            elif t == "[":
                self.next()
                idx = self.parse_expr()
                self.expect("]")
                e = ("index", e, idx)
            elif t == "{" and e[0] == "name" and self._composite_ok(e[1]):
                e = self.parse_composite(("named", e[1]))
            else:
                return e

    def _composite_ok(self, name: str) -> bool:
        # A named type followed by '{' is a composite literal when the name looks like a type (capitalised last segment)
        last = name.split(".")[-1]
        return last[:1].isupper() or last in ("queueInfo",)

    def parse_composite(self, ty):
        self.expect("{")
        elems = []
        while self.peek()[1] != "}":
            v = self.parse_expr()
            if self.accept(":"):
                key = v
                v = self.parse_expr()
                elems.append((key, v))
            else:
                elems.append((None, v))
            if not self.accept(","):
                break
        self.expect("}")
        return ("comp", ty, elems)


def parse_expr_at(src: str, base_line: int = 1):
    p = Parser(tokenize(src, base_line))
    e = p.parse_expr()
    return e, p


class Struct(dict):
    """Evaluated struct literal; .type holds the Go type name."""

    def __init__(self, type_name, fields):
        super().__init__(fields)
        self.type = type_name


class Evaluator:
    def __init__(self, env):
        self.env = env

    def ev(self, node, ty=None) -> Any:
        kind = node[0]
        if kind == "lit":
            return node[1]
        if kind == "name":
            name = node[1]
            if name in self.env:
                return self.env[name]
            parts = name.split(".")
            for cut in range(len(parts) - 1, 0, -1):  # a local variable followed by field selectors
                head = ".".join(parts[:cut])
                if head in self.env:
                    v = self.env[head]
                    for f in parts[cut:]:
                        if isinstance(v, dict) and f in v:
                            v = v[f]
                        else:
                            raise Unsupported(f"selector .{f} on {head}")
                    return v
            raise Unsupported(f"unknown name {name}")
        if kind == "funclit":
            body = node[1]
            return lambda: Evaluator(self._child_env()).run_block(body)[1]
        if kind == "unary":
            v = self.ev(node[2])
            if node[1] == "-":
                return -v
            if node[1] == "!":
                return not v
            return v
        if kind == "bin":
            a, b = self.ev(node[2]), self.ev(node[3])
            op = node[1]
            if op == "+":
                return a + b
            if op == "-":
                return a - b
            if op == "*":
                return a * b
            if op == "/":
                if isinstance(a, int) and isinstance(b, int):
                    return a // b
                return a / b
            if op == "<":
                return a < b
            if op == "<=":
                return a <= b
            if op == ">":
                return a > b
            if op == ">=":
                return a >= b
            if op == "==":
                return a == b
            if op == "!=":
                return a != b
            raise Unsupported(f"binary op {op}")
        if kind == "conv":
            return self.ev(node[2])
        if kind == "spread":
            return self.ev(node[1])
        if kind == "call":
            fn = self.ev(node[1])
            if not callable(fn):
                raise Unsupported(f"not callable: {node[1]}")
            args = []
            for a in node[2]:
                if a[0] == "spread":
                    args.extend(self.ev(a[1]))
                else:
                    args.append(self.ev(a))
            return fn(*args)
        if kind == "index":
            base = self.ev(node[1])
            return base[self.ev(node[2])]
        if kind == "sel":
            base = self.ev(node[1])
            if isinstance(base, dict) and node[2] in base:
                return base[node[2]]
            raise Unsupported(f"selector .{node[2]}")
        if kind == "comp":
            t = node[1] if node[1] is not None else ty
            return self.comp(t, node[2])
        raise Unsupported(f"node {kind}")

    # ---- statements (func literal bodies)
    class _Return(Exception):
        pass

    def _child_env(self):  # a flat chain: assignments must reach the scope that declared the variable
        return self.env.new_child() if isinstance(self.env, ChainMap) else ChainMap({}, self.env)

    def _scope_of(self, name):
        if isinstance(self.env, ChainMap):
            for m in self.env.maps:
                if name in m:
                    return m
        return None

    def assign(self, target, value, define):
        if target[0] == "name":
            parts = target[1].split(".")
            if len(parts) == 1:
                m = None if define else self._scope_of(parts[0])
                if m is None:
                    m = self.env.maps[0] if isinstance(self.env, ChainMap) else self.env
                m[parts[0]] = value
                return
            obj = self.ev(("name", ".".join(parts[:-1])))
            if not isinstance(obj, dict):
                raise Unsupported(f"assignment to field of non-struct {target[1]}")
            obj[parts[-1]] = value
            return
        if target[0] == "sel":
            obj = self.ev(target[1])
            if not isinstance(obj, dict):
                raise Unsupported("assignment to field of non-struct")
            obj[target[2]] = value
            return
        if target[0] == "index":
            self.ev(target[1])[self.ev(target[2])] = value
            return
        raise Unsupported(f"assignment target {target[0]}")

    def run_block(self, stmts):
        """returns (returned?, value)"""
        for st in stmts:
            k = st[0]
            if k == "var":
                ty = st[2]
                zero = [] if ty[0] == "slice" else ({} if ty[0] == "map" else None)
                self.assign(("name", st[1]), self.ev(st[3], ty) if st[3] is not None else zero, True)
            elif k == "assign":
                self.assign(st[1], self.ev(st[2]), st[3])
            elif k == "inc":
                self.assign(st[1], self.ev(st[1]) + 1, False)
            elif k == "expr":
                self.ev(st[1])
            elif k == "return":
                return True, (self.ev(st[1]) if st[1] is not None else None)
            elif k == "for":
                _, init, cond, post, body = st
                loop = Evaluator(self._child_env())
                loop.run_block([init])
                guard = 0
                while loop.ev(cond):
                    r = Evaluator(loop._child_env()).run_block(body)
                    if r[0]:
                        return r
                    loop.run_block([post])
                    guard += 1
                    if guard > 100000:
                        raise Unsupported("loop does not terminate")
            else:
                raise Unsupported(f"statement {k}")
        return False, None

    def comp(self, ty, elems):
        while ty is not None and ty[0] == "ptr":
            ty = ty[1]
        if ty is None:
            raise Unsupported("composite literal without type")
        if ty[0] == "slice":
            return [self.ev(v, ty[1]) for _, v in elems]
        if ty[0] == "map":
            out = {}
            for k, v in elems:
                out[self.ev(k, ty[1])] = self.ev(v, ty[2])
            return out
        if ty[0] == "named" and ty[1] in self.MAP_TYPES:   # a named map type written as a keyed literal (v1.ResourceList{"cpu": ...})
            return {self.ev(k): self.ev(v) for k, v in elems}
        if ty[0] == "named":
            fields = {}
            for k, v in elems:
                if k is None or k[0] != "name":
                    raise Unsupported(f"positional struct literal of {ty[1]}")
                fields[k[1]] = ("lazy", v)
            # field types are unknown: evaluate values without an elided-type hint, falling back to generic containers
            out = {}
            for name, (_, v) in fields.items():
                out[name] = self.ev_field(ty[1], name, v)
            return Struct(ty[1], out)
        raise Unsupported(f"composite of {ty}")

    # elided-type hints for struct fields we care about
    FIELD_TYPES = {}
    MAP_TYPES = {"v1.ResourceList"}

    def ev_field(self, struct_name, field, v):
        hint = self.FIELD_TYPES.get((struct_name.split(".")[-1], field))
        return self.ev(v, hint)


def find_matching(src: str, open_pos: int, open_ch="{", close_ch="}") -> int:
    """index of the bracket matching src[open_pos], skipping strings and comments."""
    depth = 0
    i = open_pos
    n = len(src)
    while i < n:
        c = src[i]
        if c == '"':
            i += 1
            while src[i] != '"':
                i += 2 if src[i] == "\\" else 1
        elif c == "`":
            i = src.index("`", i + 1)
        elif src.startswith("//", i):
            i = src.index("\n", i)
        elif src.startswith("/*", i):
            i = src.index("*/", i) + 1
        elif c == open_ch:
            depth += 1
        elif c == close_ch:
            depth -= 1
            if depth == 0:
                return i
        i += 1
    raise Unsupported("unbalanced brackets")


def split_table_cases(src: str, table_body_start: int, table_body_end: int, base_line_of) -> List[Tuple[str, str, int]]:
    """Split the body of `map[string]struct{...}{ "name": {...}, ... }` into (name, literal_src, line)."""
    cases = []
    i = table_body_start
    while i < table_body_end:
        m = re.compile(r'\s*(?://[^\n]*\n\s*)*"((?:\\.|[^"\\])*)"\s*:\s*\{', re.S).match(src, i)
        if not m:
            # skip whitespace / trailing commas
            if src[i] in " \t\r\n,":
                i += 1
                continue
            if src.startswith("//", i):
                i = src.index("\n", i)
                continue
            break
        open_pos = m.end() - 1
        close = find_matching(src, open_pos)
        cases.append((m.group(1), src[open_pos:close + 1], base_line_of(open_pos)))
        i = close + 1
    return cases
