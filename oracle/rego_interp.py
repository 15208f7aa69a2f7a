"""ORACLE (test infrastructure only -- never imported by the product path).

Top-down evaluator for the Rego subset of oracle/rego_parser.py.

Restates what the reference's Rego driver does per (constraint, review):
  frameworks/constraint pkg/client/drivers/rego Driver.Query  (third-party, go.mod:18, source absent) evaluates the
  template's `violation` partial set with  input = {"review": <gkReview JSON>, "parameters": spec.parameters}
  using OPA topdown (go.mod:19).  Call sites in the reference: pkg/webhook/policy.go:826,
  pkg/audit/manager.go:621,719, pkg/gator/test/test.go:118.

Semantics restated from the OPA language reference: bodies are conjunctions evaluated by backtracking search,
undefined propagates (a literal with an undefined operand fails), `not` is negation-as-failure, partial-set rules
are unions over their bodies, functions may have several bodies (disjunction), builtin errors make the call
undefined.  Pinned against the reference's golden vectors in tests/test_oracle_*.py.
"""
from __future__ import annotations

from .rego_builtins import BUILTINS, BuiltinError, arith
from .rego_parser import parse_module
from .values import RObj, RSet, compare, equal, hk

_NONE = object()


class Unbound(Exception):
    """A variable was used in a position that cannot bind it before it was bound."""


class RegoEvalError(Exception):
    pass


class Interp:
    """One compiled template: main module + libs. `data` is the base document (e.g. {"inventory": ...})."""

    def __init__(self, sources, data=None):
        self.modules = [parse_module(s) if isinstance(s, str) else s for s in sources]
        self.rules = {}   # (pkg tuple, name) -> [rule]
        self.pkgs = set()
        for m in self.modules:
            pkg = tuple(m["package"])
            self.pkgs.add(pkg)
            for r in m["rules"]:
                r["pkg"] = pkg
                r["imports"] = {(alias or path[-1]): path for path, alias in m["imports"]}
                self.rules.setdefault((pkg, r["name"]), []).append(r)
        self.main_pkg = tuple(self.modules[0]["package"])
        self.data = data if data is not None else RObj()
        self._check_compile()

    # ------------------------------------------------------------ static checks (compile errors)
    def _check_compile(self):
        """OPA rejects references to undeclared names at compile time (pkg/gator/fixtures/fixtures.go:142-160
        TemplateCompileError: a body consisting of the bare, never-defined var `f`)."""
        for (pkg, _), rules in self.rules.items():
            for r in rules:
                for body in [r["body"]] + [b for _, b in r["elses"]]:
                    for lit in body or []:
                        if lit[0] == "expr" and lit[1][0] == "var":
                            name = lit[1][1]
                            if name.startswith("$w"):
                                continue
                            if name in ("input", "data") or (pkg, name) in self.rules:
                                continue
                            if not self._binds(r, name):
                                raise RegoEvalError("rego_unsafe_var_error: var %s is unsafe" % name)

    def _binds(self, rule, name):
        """True if `name` has a binding occurrence: function argument, unification side, ref operand, some-in."""
        def pattern_vars(t):
            if t[0] == "var":
                return t[1] == name
            if t[0] == "array":
                return any(pattern_vars(x) for x in t[1])
            if t[0] == "object":
                return any(pattern_vars(v) for _, v in t[1])
            return False

        def operand_vars(t):
            if isinstance(t, tuple):
                if t and t[0] == "ref":
                    if any(o[0] == "var" and o[1] == name for o in t[2]):
                        return True
                return any(operand_vars(x) for x in t)
            if isinstance(t, list):
                return any(operand_vars(x) for x in t)
            return False

        if rule["args"] and any(pattern_vars(a) for a in rule["args"]):
            return True
        for lit in rule["body"] or []:
            if lit[0] in ("assign", "unify") and (pattern_vars(lit[1]) or pattern_vars(lit[2])):
                return True
            if lit[0] == "somein" and (pattern_vars(lit[2]) or (lit[1] is not None and pattern_vars(lit[1]))):
                return True
            if operand_vars(lit):
                return True
        return False

    # ------------------------------------------------------------ public
    def violations(self, input_val):
        """The `violation` partial set of the main package, as a list of Rego values."""
        q = _Query(self, input_val)
        rules = self.rules.get((self.main_pkg, "violation"))
        if not rules:
            return []
        s = q.rule_value(self.main_pkg, "violation")
        if s is _NONE:
            return []
        return list(s.elems())


class _Query:
    def __init__(self, interp, input_val):
        self.ip = interp
        self.input = input_val
        self.cache = {}
        self.depth = 0

    # ------------------------------------------------------------ rules
    def rule_value(self, pkg, name):
        key = (pkg, name)
        if key in self.cache:
            v = self.cache[key]
            if v is _Query:
                raise RegoEvalError("recursive rule %s" % name)
            return v
        self.cache[key] = _Query
        rules = self.ip.rules[key]
        kind = rules[0]["kind"]
        if kind == "func":
            raise RegoEvalError("function %s referenced without call" % name)
        if kind == "set":
            out = {}
            for r in rules:
                for env in self.eval_body(r["body"], {}, r):
                    for v, _ in self.eval_term(r["key"], env, r):
                        out[hk(v)] = v
            res = RSet()
            res.d = out
        elif kind == "object":
            pairs = []
            for r in rules:
                for env in self.eval_body(r["body"], {}, r):
                    for k, e2 in self.eval_term(r["key"], env, r):
                        for v, _ in self.eval_term(r["value"], e2, r):
                            pairs.append((k, v))
            res = RObj(pairs)
        else:
            res = _NONE
            default = _NONE
            for r in rules:
                if r["default"]:
                    for v, _ in self.eval_term(r["value"], {}, r):
                        default = v
                    continue
                v = self._complete_def(r, {})
                if v is not _NONE:
                    if res is not _NONE and not equal(res, v):
                        raise RegoEvalError("complete rule %s produced conflicting values" % name)
                    res = v
            if res is _NONE:
                res = default
        self.cache[key] = res
        return res

    def _complete_def(self, r, env):
        """Value of one complete-rule / function definition (with its else chain) under env, or _NONE."""
        chain = [(r["value"], r["body"])] + list(r["elses"])
        for val_t, body in chain:
            res = _NONE
            for e in self.eval_body(body, env, r):
                if val_t is None:
                    v = True
                else:
                    v = _NONE
                    for v, _ in self.eval_term(val_t, e, r):
                        break
                    if v is _NONE:
                        continue
                if res is not _NONE and not equal(res, v):
                    raise RegoEvalError("rule %s produced conflicting values" % r["name"])
                res = v
            if res is not _NONE:
                return res
        return _NONE

    def call_function(self, pkg, name, args):
        rules = self.ip.rules[(pkg, name)]
        self.depth += 1
        if self.depth > 200:
            raise RegoEvalError("recursion too deep in %s" % name)
        try:
            res = _NONE
            default = _NONE
            for r in rules:
                if r["kind"] != "func" or len(r["args"]) != len(args):
                    continue
                if r["default"]:      # `default f(_) := v` (OPA >= 0.51): the value when no other definition is defined
                    for v, _ in self.eval_term(r["value"], {}, r):
                        default = v
                    continue
                envs = [{}]
                for p, a in zip(r["args"], args):
                    envs = [e2 for e in envs for e2 in self.unify_value(p, a, e, r)]
                for e in envs:
                    v = self._complete_def(r, e)
                    if v is not _NONE:
                        if res is not _NONE and not equal(res, v):
                            raise RegoEvalError("function %s produced conflicting outputs" % name)
                        res = v
            return default if res is _NONE else res
        finally:
            self.depth -= 1

    # ------------------------------------------------------------ bodies
    def eval_body(self, lits, env, rule):
        if not lits:
            yield env
            return
        for k in range(len(lits)):
            lit = lits[k]
            try:
                it = self.eval_literal(lit, env, rule)
                first = next(it, _NONE)
            except Unbound:
                continue
            rest = lits[:k] + lits[k + 1:]
            if first is _NONE:
                return
            yield from self.eval_body(rest, first, rule)
            for e in it:
                yield from self.eval_body(rest, e, rule)
            return
        raise Unbound("no evaluable literal in body")

    def eval_literal(self, lit, env, rule):
        k = lit[0]
        if k == "expr":
            for v, e in self.eval_term(lit[1], env, rule):
                if v is not False:
                    yield e
        elif k == "assign" or k == "unify":
            yield from self.unify_terms(lit[1], lit[2], env, rule)
        elif k == "not":
            inner = lit[1]
            for _ in self.eval_literal(inner, env, rule):
                return
            yield env
        elif k == "some":
            e = dict(env)
            for n in lit[1]:
                e.pop(n, None)
            yield e
        elif k == "somein":
            _, kt, vt, ct = lit
            for coll, e in self.eval_term(ct, env, rule):
                for key, val in _iter_kv(coll):
                    for e2 in self.unify_value(vt, val, e, rule):
                        if kt is None:
                            yield e2
                        else:
                            yield from self.unify_value(kt, key, e2, rule)
        elif k == "every":
            _, kt, vt, ct, body = lit
            for coll, e in self.eval_term(ct, env, rule):
                ok = True
                for key, val in _iter_kv(coll):
                    sat = False
                    for e2 in self.unify_value(vt, val, e, rule):
                        envs = [e2] if kt is None else list(self.unify_value(kt, key, e2, rule))
                        for e3 in envs:
                            for _ in self.eval_body(body, e3, rule):
                                sat = True
                                break
                            if sat:
                                break
                        if sat:
                            break
                    if not sat:
                        ok = False
                        break
                if ok:
                    yield e
        else:
            raise RegoEvalError("unknown literal %r" % (k,))

    # ------------------------------------------------------------ unification
    def is_unbound_var(self, t, env, rule):
        return t[0] == "var" and t[1] not in env and not self._is_global(t[1], rule)

    def _is_global(self, name, rule):
        return name in ("input", "data") or (rule["pkg"], name) in self.ip.rules or name in rule["imports"]

    def unify_terms(self, a, b, env, rule):
        if self.is_unbound_var(a, env, rule):
            for v, e in self.eval_term(b, env, rule):
                e2 = dict(e)
                e2[a[1]] = v
                yield e2
        elif self.is_unbound_var(b, env, rule):
            for v, e in self.eval_term(a, env, rule):
                e2 = dict(e)
                e2[b[1]] = v
                yield e2
        elif a[0] in ("array", "object") and self._has_unbound(a, env, rule):
            for v, e in self.eval_term(b, env, rule):
                yield from self.unify_value(a, v, e, rule)
        elif b[0] in ("array", "object") and self._has_unbound(b, env, rule):
            for v, e in self.eval_term(a, env, rule):
                yield from self.unify_value(b, v, e, rule)
        else:
            for va, e in self.eval_term(a, env, rule):
                for vb, e2 in self.eval_term(b, e, rule):
                    if equal(va, vb):
                        yield e2

    def _has_unbound(self, t, env, rule):
        if t[0] == "var":
            return self.is_unbound_var(t, env, rule)
        if t[0] == "array":
            return any(self._has_unbound(x, env, rule) for x in t[1])
        if t[0] == "object":
            return any(self._has_unbound(v, env, rule) for _, v in t[1])
        return False

    def unify_value(self, pat, val, env, rule):
        """Unify a pattern term against a concrete value."""
        k = pat[0]
        if k == "var" and self.is_unbound_var(pat, env, rule):
            e = dict(env)
            e[pat[1]] = val
            yield e
        elif k == "array" and self._has_unbound(pat, env, rule):
            if isinstance(val, tuple) and len(val) == len(pat[1]):
                envs = [env]
                for p, v in zip(pat[1], val):
                    envs = [e2 for e in envs for e2 in self.unify_value(p, v, e, rule)]
                yield from envs
        elif k == "object" and self._has_unbound(pat, env, rule):
            if isinstance(val, RObj) and len(val) == len(pat[1]):
                envs = [env]
                for kt, vt in pat[1]:
                    nxt = []
                    for e in envs:
                        for kv, e1 in self.eval_term(kt, e, rule):
                            if val.has(kv):
                                nxt.extend(self.unify_value(vt, val.get(kv), e1, rule))
                    envs = nxt
                yield from envs
        else:
            for v, e in self.eval_term(pat, env, rule):
                if equal(v, val):
                    yield e

    # ------------------------------------------------------------ terms
    def eval_term(self, t, env, rule):
        k = t[0]
        if k == "scalar":
            yield t[1], env
        elif k == "var":
            name = t[1]
            if name in env:
                yield env[name], env
            elif name == "input":
                yield self.input, env
            elif name == "data":
                yield from self._eval_data_ref([], env, rule)
            elif (rule["pkg"], name) in self.ip.rules:
                v = self.rule_value(rule["pkg"], name)
                if v is not _NONE:
                    yield v, env
            else:
                raise Unbound(name)
        elif k == "ref":
            head, ops = t[1], t[2]
            if head[0] == "var" and head[1] == "data" and "data" not in env:
                yield from self._eval_data_ref(ops, env, rule)
            elif head[0] == "var" and head[1] in rule["imports"] and head[1] not in env:
                path = rule["imports"][head[1]]
                full = [("scalar", p) for p in path[1:]] + list(ops)
                if path[0] == "data":
                    yield from self._eval_data_ref(full, env, rule)
                else:
                    yield from self._walk(self.input, full, 0, env, rule)
            else:
                for hv, e in self.eval_term(head, env, rule):
                    yield from self._walk(hv, ops, 0, e, rule)
        elif k == "call":
            yield from self._eval_call(t, env, rule)
        elif k == "binop":
            op = t[1]
            for a, e in self.eval_term(t[2], env, rule):
                for b, e2 in self.eval_term(t[3], e, rule):
                    if op in ("==", "!=", "<", "<=", ">", ">="):
                        c = compare(a, b)
                        r = {"==": c == 0, "!=": c != 0, "<": c < 0, "<=": c <= 0, ">": c > 0, ">=": c >= 0}[op]
                        yield r, e2
                    elif op == "in":
                        yield any(equal(a, v) for _, v in _iter_kv(b)), e2
                    else:
                        try:
                            yield arith(op, a, b), e2
                        except BuiltinError:
                            pass
        elif k == "array":
            yield from self._eval_seq(t[1], 0, (), env, rule, tuple)
        elif k == "set":
            yield from self._eval_seq(t[1], 0, (), env, rule, RSet)
        elif k == "object":
            flat = [x for kv in t[1] for x in kv]
            for vals, e in self._eval_seq(flat, 0, (), env, rule, tuple):
                yield RObj(zip(vals[0::2], vals[1::2])), e
        elif k == "arrcomp":
            out = []
            for e in self.eval_body(t[2], env, rule):
                for v, _ in self.eval_term(t[1], e, rule):
                    out.append(v)
            yield tuple(out), env
        elif k == "setcomp":
            out = []
            for e in self.eval_body(t[2], env, rule):
                for v, _ in self.eval_term(t[1], e, rule):
                    out.append(v)
            yield RSet(out), env
        elif k == "objcomp":
            out = []
            for e in self.eval_body(t[3], env, rule):
                for kv, e2 in self.eval_term(t[1], e, rule):
                    for vv, _ in self.eval_term(t[2], e2, rule):
                        out.append((kv, vv))
            yield RObj(out), env
        else:
            raise RegoEvalError("unknown term %r" % (k,))

    def _eval_seq(self, terms, i, acc, env, rule, ctor):
        if i == len(terms):
            yield ctor(acc), env
            return
        for v, e in self.eval_term(terms[i], env, rule):
            yield from self._eval_seq(terms, i + 1, acc + (v,), e, rule, ctor)

    def _walk(self, cur, ops, i, env, rule):
        if i == len(ops):
            yield cur, env
            return
        op = ops[i]
        if op[0] == "var" and self.is_unbound_var(op, env, rule):
            for key, val in _iter_kv(cur):
                e = dict(env)
                e[op[1]] = key
                yield from self._walk(val, ops, i + 1, e, rule)
            return
        if op[0] == "scalar":
            nxt = _index(cur, op[1])
            if nxt is not _NONE:
                yield from self._walk(nxt, ops, i + 1, env, rule)
            return
        if op[0] in ("array", "object") and self._has_unbound(op, env, rule):
            # pattern operand, e.g. general_violation[{"msg": msg, "field": "containers"}]
            for key, val in _iter_kv(cur):
                for e in self.unify_value(op, key, env, rule):
                    yield from self._walk(val, ops, i + 1, e, rule)
            return
        for kv, e in self.eval_term(op, env, rule):
            nxt = _index(cur, kv)
            if nxt is not _NONE:
                yield from self._walk(nxt, ops, i + 1, e, rule)

    def _eval_data_ref(self, ops, env, rule):
        # virtual documents: longest package prefix made of constant string operands
        consts = []
        for o in ops:
            if o[0] == "scalar" and isinstance(o[1], str):
                consts.append(o[1])
            else:
                break
        for n in range(len(consts) - 1, -1, -1):
            pkg = tuple(consts[:n])
            if (pkg, consts[n]) in self.ip.rules:
                v = self.rule_value(pkg, consts[n])
                if v is not _NONE:
                    yield from self._walk(v, ops, n + 1, env, rule)
                return
        yield from self._walk(self.ip.data, ops, 0, env, rule)

    def _eval_call(self, t, env, rule):
        path, argts = t[1], t[2]
        name = ".".join(path)
        target = None
        if len(path) == 1 and (rule["pkg"], path[0]) in self.ip.rules:
            target = (rule["pkg"], path[0])
        elif path[0] == "data" and (tuple(path[1:-1]), path[-1]) in self.ip.rules:
            target = (tuple(path[1:-1]), path[-1])
        elif path[0] in rule["imports"]:
            full = rule["imports"][path[0]] + path[1:]
            if full[0] == "data" and (tuple(full[1:-1]), full[-1]) in self.ip.rules:
                target = (tuple(full[1:-1]), full[-1])
        if target is None and name not in BUILTINS:
            raise RegoEvalError("rego_type_error: undefined function %s" % name)
        for args, e in self._eval_seq(argts, 0, (), env, rule, tuple):
            if target is not None:
                v = self.call_function(target[0], target[1], args)
                if v is not _NONE:
                    yield v, e
            else:
                try:
                    v = BUILTINS[name](*args)
                except BuiltinError:
                    continue
                except TypeError as ex:
                    raise RegoEvalError("rego_type_error: %s: %s" % (name, ex))
                yield v, e


def _index(cur, key):
    if isinstance(cur, RObj):
        e = cur.d.get(hk(key))
        return _NONE if e is None else e[1]
    if isinstance(cur, tuple):
        if isinstance(key, bool) or not isinstance(key, (int, float)):
            return _NONE
        if isinstance(key, float):
            if not key.is_integer():
                return _NONE
            key = int(key)
        if 0 <= key < len(cur):
            return cur[key]
        return _NONE
    if isinstance(cur, RSet):
        return key if cur.has(key) else _NONE
    return _NONE


def _iter_kv(coll):
    if isinstance(coll, tuple):
        return list(enumerate(coll))
    if isinstance(coll, RObj):
        return list(coll.items())
    if isinstance(coll, RSet):
        return [(v, v) for v in coll.elems()]
    return []
