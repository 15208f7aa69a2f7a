"""ORACLE (test infrastructure only -- never imported by the product path).

Parser for the Rego subset exercised by the reference's in-tree ConstraintTemplates.

The reference parses Rego with OPA v1.17.1 `ast.ParseModule` (third-party, go.mod:19, source absent from
/root/reference); this is a restatement from the published Rego grammar (OPA docs "policy-reference: grammar"),
covering both the v0 (`violation[x] { .. }`) and v1 (`violation contains x if { .. }`) surface syntax that the
reference's templates use (e.g. test/bats/tests/templates/k8srequiredlabels_template_regov1.yaml).

AST (plain tuples):
  terms    ('scalar', v) ('var', name) ('ref', head, [operands]) ('call', [path], [args]) ('array', [t])
           ('object', [(k, v)]) ('set', [t]) ('arrcomp', t, body) ('setcomp', t, body) ('objcomp', k, v, body)
           ('binop', op, l, r)
  literals ('expr', t) ('assign', l, r) ('unify', l, r) ('not', lit) ('some', [names])
           ('somein', k|None, v, coll) ('every', k|None, v, coll, body)
"""
from __future__ import annotations

import re

KEYWORDS = {"package", "import", "default", "not", "some", "every", "in", "if", "contains", "else", "with", "as",
            "true", "false", "null"}


class RegoSyntaxError(Exception):
    pass


_TOKEN_RE = re.compile(r"""
    (?P<ws>[ \t\r]+)
  | (?P<comment>\#[^\n]*)
  | (?P<nl>\n)
  | (?P<num>(?:\d+\.\d+(?:[eE][+-]?\d+)?|\d+[eE][+-]?\d+|\d+|\.\d+))
  | (?P<ident>[A-Za-z_][A-Za-z_0-9]*)
  | (?P<str>"(?:[^"\\\n]|\\.)*")
  | (?P<raw>`[^`]*`)
  | (?P<op>:=|==|!=|<=|>=|[{}\[\]().,;:|=<>+\-*/%&])
""", re.X)

_UNESC = {'"': '"', "\\": "\\", "/": "/", "b": "\b", "f": "\f", "n": "\n", "r": "\r", "t": "\t"}


def _unescape(s):
    out = []
    i = 0
    while i < len(s):
        c = s[i]
        if c == "\\":
            i += 1
            e = s[i]
            if e == "u":
                out.append(chr(int(s[i + 1:i + 5], 16)))
                i += 4
            elif e in _UNESC:
                out.append(_UNESC[e])
            else:
                raise RegoSyntaxError("bad escape \\%s" % e)
        else:
            out.append(c)
        i += 1
    return "".join(out)


def tokenize(src):
    toks = []
    pos = 0
    line = 1
    while pos < len(src):
        m = _TOKEN_RE.match(src, pos)
        if not m:
            raise RegoSyntaxError("line %d: unexpected character %r" % (line, src[pos]))
        pos = m.end()
        k = m.lastgroup
        if k in ("ws", "comment"):
            continue
        if k == "nl":
            toks.append(("nl", "\n", line))
            line += 1
            continue
        t = m.group(k)
        if k == "num":
            toks.append(("num", float(t) if any(c in t for c in ".eE") else int(t), line))
        elif k == "ident":
            toks.append(("kw" if t in KEYWORDS else "ident", t, line))
        elif k == "str":
            toks.append(("str", _unescape(t[1:-1]), line))
        elif k == "raw":
            toks.append(("str", t[1:-1], line))
            line += t.count("\n")
        else:
            toks.append(("op", t, line))
    toks.append(("eof", None, line))
    return toks


class Parser:
    def __init__(self, src):
        self.toks = tokenize(src)
        self.i = 0
        self.wild = 0

    # -- token helpers
    def peek(self, skip_nl=False):
        i = self.i
        if skip_nl:
            while self.toks[i][0] == "nl":
                i += 1
        return self.toks[i]

    def next(self, skip_nl=False):
        if skip_nl:
            self.skip_nl()
        t = self.toks[self.i]
        self.i += 1
        return t

    def skip_nl(self):
        while self.toks[self.i][0] == "nl":
            self.i += 1

    def at(self, kind, val=None, skip_nl=False):
        t = self.peek(skip_nl)
        return t[0] == kind and (val is None or t[1] == val)

    def at_op(self, val, skip_nl=False):
        return self.at("op", val, skip_nl)

    def accept(self, kind, val=None, skip_nl=False):
        if self.at(kind, val, skip_nl):
            return self.next(skip_nl)
        return None

    def expect(self, kind, val=None, skip_nl=False):
        t = self.next(skip_nl)
        if t[0] != kind or (val is not None and t[1] != val):
            raise RegoSyntaxError("line %d: expected %s %r, got %r" % (t[2], kind, val, t[1]))
        return t

    def err(self, msg):
        t = self.peek()
        raise RegoSyntaxError("line %d: %s (at %r)" % (t[2], msg, t[1]))

    # -- module
    def parse_module(self):
        self.skip_nl()
        self.expect("kw", "package")
        pkg = self.parse_dotted()
        imports = []
        rules = []
        while True:
            self.skip_nl()
            if self.at("eof"):
                break
            if self.accept("kw", "import"):
                path = self.parse_dotted()
                alias = None
                if self.accept("kw", "as"):
                    alias = self.expect("ident")[1]
                imports.append((path, alias))
                continue
            rules.append(self.parse_rule())
        return {"package": pkg, "imports": imports, "rules": rules}

    def parse_dotted(self):
        t = self.next()
        if t[0] not in ("ident", "kw"):
            raise RegoSyntaxError("line %d: expected identifier" % t[2])
        parts = [t[1]]
        while True:
            if self.accept("op", "."):
                t = self.next()
                parts.append(t[1])
            elif self.at_op("["):
                self.next()
                parts.append(self.expect("str")[1])
                self.expect("op", "]")
            else:
                break
        return parts

    # -- rules
    def parse_rule(self):
        line = self.peek()[2]
        is_default = bool(self.accept("kw", "default"))
        name = self.expect("ident")[1]
        rule = {"name": name, "kind": "complete", "args": None, "key": None, "value": None, "body": None,
                "default": is_default, "elses": [], "line": line}
        if self.at_op("("):
            self.next()
            args = []
            self.skip_nl()
            while not self.at_op(")", True):
                args.append(self.parse_term())
                if not self.accept("op", ",", True):
                    break
            self.expect("op", ")", True)
            rule["kind"] = "func"
            rule["args"] = args
        elif self.at_op("["):
            self.next()
            rule["key"] = self.parse_term()
            self.expect("op", "]", True)
            rule["kind"] = "set"
        elif self.accept("kw", "contains"):
            self.skip_nl()
            rule["key"] = self.parse_term()
            rule["kind"] = "set"
        if self.at_op("=") or self.at_op(":="):
            self.next()
            rule["value"] = self.parse_term()
            if rule["kind"] == "set":
                rule["kind"] = "object"
        has_if = bool(self.accept("kw", "if", skip_nl=self.at("kw", "if", True)))
        if has_if:
            self.skip_nl()
        if self.at_op("{"):
            rule["body"] = self.parse_braced_body()
        elif has_if:
            rule["body"] = [self.parse_literal()]
        else:
            rule["body"] = []
        while self.at("kw", "else", True):
            self.next(True)
            val = None
            if self.at_op("=") or self.at_op(":="):
                self.next()
                val = self.parse_term()
            self.accept("kw", "if")
            if self.at_op("{"):
                body = self.parse_braced_body()
            elif self.at("nl") or self.at("eof"):
                body = []
            else:
                body = [self.parse_literal()]
            rule["elses"].append((val, body))
        if is_default and rule["value"] is None:
            self.err("default rule needs a value")
        return rule

    def parse_braced_body(self):
        self.expect("op", "{")
        body = self.parse_body_until("}")
        self.expect("op", "}", True)
        return body

    def parse_body_until(self, closer):
        lits = []
        while True:
            self.skip_nl()
            while self.accept("op", ";"):
                self.skip_nl()
            if self.at_op(closer):
                break
            lits.append(self.parse_literal())
            if not (self.at("nl") or self.at_op(";") or self.at_op(closer)):
                self.err("expected end of literal")
        return lits

    # -- literals
    def parse_literal(self):
        if self.accept("kw", "not"):
            return ("not", self.parse_literal())
        if self.at("kw", "some"):
            self.next()
            first = self.parse_term(no_in=True)
            if self.at_op(","):
                self.next()
                second = self.parse_term(no_in=True)
                if self.accept("kw", "in"):
                    return ("somein", first, second, self.parse_term())
                names = [first, second]
                while self.accept("op", ","):
                    names.append(self.parse_term(no_in=True))
                return ("some", [n[1] for n in names])
            if self.accept("kw", "in"):
                return ("somein", None, first, self.parse_term())
            return ("some", [first[1]])
        if self.at("kw", "every"):
            self.next()
            first = self.parse_term(no_in=True)
            key = None
            if self.accept("op", ","):
                key = first
                first = self.parse_term(no_in=True)
            self.expect("kw", "in")
            coll = self.parse_term()
            body = self.parse_braced_body()
            return ("every", key, first, coll, body)
        lhs = self.parse_term()
        if self.at_op(":="):
            self.next()
            lit = ("assign", lhs, self.parse_term(after_op=True))
        elif self.at_op("="):
            self.next()
            lit = ("unify", lhs, self.parse_term(after_op=True))
        else:
            lit = ("expr", lhs)
        if self.at("kw", "with"):
            self.err("`with` is not supported by the oracle's Rego subset")
        return lit

    # -- terms (precedence climbing)
    def parse_term(self, no_bitor=False, no_in=False, after_op=False):
        if after_op:
            self.skip_nl()
        return self.parse_relation(no_bitor, no_in)

    def parse_relation(self, no_bitor, no_in):
        l = self.parse_bitor(no_bitor)
        while True:
            t = self.peek()
            if t[0] == "op" and t[1] in ("==", "!=", "<", "<=", ">", ">="):
                self.next()
                self.skip_nl()
                r = self.parse_bitor(no_bitor)
                l = ("binop", t[1], l, r)
            elif t[0] == "kw" and t[1] == "in" and not no_in:
                self.next()
                r = self.parse_bitor(no_bitor)
                l = ("binop", "in", l, r)
            else:
                return l

    def parse_bitor(self, no_bitor):
        l = self.parse_bitand()
        while not no_bitor and self.at_op("|"):
            self.next()
            self.skip_nl()
            l = ("binop", "|", l, self.parse_bitand())
        return l

    def parse_bitand(self):
        l = self.parse_arith()
        while self.at_op("&"):
            self.next()
            self.skip_nl()
            l = ("binop", "&", l, self.parse_arith())
        return l

    def parse_arith(self):
        l = self.parse_factor()
        while self.at_op("+") or self.at_op("-"):
            op = self.next()[1]
            self.skip_nl()
            l = ("binop", op, l, self.parse_factor())
        return l

    def parse_factor(self):
        l = self.parse_unary()
        while self.at_op("*") or self.at_op("/") or self.at_op("%"):
            op = self.next()[1]
            self.skip_nl()
            l = ("binop", op, l, self.parse_unary())
        return l

    def parse_unary(self):
        if self.at_op("-"):
            self.next()
            t = self.parse_unary()
            if t[0] == "scalar" and isinstance(t[1], (int, float)) and not isinstance(t[1], bool):
                return ("scalar", -t[1])
            return ("binop", "-", ("scalar", 0), t)
        return self.parse_postfix(self.parse_primary())

    def parse_postfix(self, head):
        ops = []
        while True:
            if self.at_op("."):
                self.next()
                t = self.next()
                if t[0] not in ("ident", "kw"):
                    raise RegoSyntaxError("line %d: expected field name" % t[2])
                ops.append(("scalar", t[1]))
            elif self.at_op("["):
                self.next()
                self.skip_nl()
                ops.append(self.parse_term())
                self.expect("op", "]", True)
            elif self.at_op("("):
                # call: head+ops must be a dotted name
                if head[0] != "var" or any(o[0] != "scalar" or not isinstance(o[1], str) for o in ops):
                    self.err("call on non-name")
                path = [head[1]] + [o[1] for o in ops]
                self.next()
                args = []
                self.skip_nl()
                while not self.at_op(")", True):
                    args.append(self.parse_term())
                    if not self.accept("op", ",", True):
                        break
                self.expect("op", ")", True)
                head = ("call", path, args)
                ops = []
            else:
                break
        if ops:
            return ("ref", head, ops)
        return head

    def parse_primary(self):
        t = self.next()
        k, v = t[0], t[1]
        if k == "num" or k == "str":
            return ("scalar", v)
        if k == "kw":
            if v == "true":
                return ("scalar", True)
            if v == "false":
                return ("scalar", False)
            if v == "null":
                return ("scalar", None)
            if v == "contains":     # OPA's parser: `contains` anywhere BUT in rule heads gets no special treatment (builtin contains(s, sub))
                return ("var", v)
            raise RegoSyntaxError("line %d: unexpected keyword %r" % (t[2], v))
        if k == "ident":
            if v == "_":
                self.wild += 1
                return ("var", "$w%d" % self.wild)
            if v == "set" and self.at_op("("):
                save = self.i
                self.next()
                if self.accept("op", ")"):
                    return ("set", [])
                self.i = save
            return ("var", v)
        if k == "op":
            if v == "(":
                self.skip_nl()
                e = self.parse_term()
                self.expect("op", ")", True)
                return e
            if v == "[":
                return self.parse_array_or_comp()
            if v == "{":
                return self.parse_brace_term()
        raise RegoSyntaxError("line %d: unexpected token %r" % (t[2], v))

    def parse_array_or_comp(self):
        self.skip_nl()
        if self.accept("op", "]"):
            return ("array", [])
        first = self.parse_term(no_bitor=True)
        if self.at_op("|", True):
            self.next(True)
            body = self.parse_body_until("]")
            self.expect("op", "]", True)
            return ("arrcomp", first, body)
        first = self._continue_bitor(first)
        elems = [first]
        while self.accept("op", ",", True):
            self.skip_nl()
            if self.at_op("]"):
                break
            elems.append(self.parse_term())
        self.expect("op", "]", True)
        return ("array", elems)

    def _continue_bitor(self, first):
        # only reached when the element was not followed by a comprehension bar
        return first

    def parse_brace_term(self):
        self.skip_nl()
        if self.accept("op", "}"):
            return ("object", [])
        first = self.parse_term(no_bitor=True)
        if self.at_op(":", True):
            self.next(True)
            self.skip_nl()
            val = self.parse_term(no_bitor=True)
            if self.at_op("|", True):
                self.next(True)
                body = self.parse_body_until("}")
                self.expect("op", "}", True)
                return ("objcomp", first, val, body)
            pairs = [(first, val)]
            while self.accept("op", ",", True):
                self.skip_nl()
                if self.at_op("}"):
                    break
                k = self.parse_term()
                self.expect("op", ":", True)
                self.skip_nl()
                pairs.append((k, self.parse_term()))
            self.expect("op", "}", True)
            return ("object", pairs)
        if self.at_op("|", True):
            self.next(True)
            body = self.parse_body_until("}")
            self.expect("op", "}", True)
            return ("setcomp", first, body)
        elems = [first]
        while self.accept("op", ",", True):
            self.skip_nl()
            if self.at_op("}"):
                break
            elems.append(self.parse_term())
        self.expect("op", "}", True)
        return ("set", elems)


def parse_module(src):
    return Parser(src).parse_module()
