"""ORACLE (test infrastructure only -- never imported by the product path).

Rego value model for the CPU restatement of the reference's Rego driver.

The reference evaluates ConstraintTemplate Rego with OPA v1.17.1 (go.mod:19), whose source is NOT in
/root/reference.  This file restates OPA's published value semantics:

  * types and their total order  null < boolean < number < string < array < object < set
    (OPA docs "policy-reference: comparison"; used by `sort`, set printing and `<`),
  * value equality (true != 1, 1 == 1.0),
  * `String()` rendering of composite terms used by sprintf("%v") -- pinned by the reference at
    website/docs/constrainttemplates.md:118 (`you must provide labels: {"gatekeeper"}`) and
    test/gator/test/test.bats:241,
  * number text: a float64 of a review object prints as encoding/json wrote it (json_float_text), a float64
    `sprintf` argument as fmt's %v (go_float_v) -- the Go standard library's two formatters, restated from their
    documented behaviour; the reference holds no vector with a non-integral number in a message: "parity unpinned"
    beyond the well-known outputs tests/test_pe_builtins.py lists.

Python representation
  null -> None, boolean -> bool, number -> int | float, string -> str,
  array -> tuple, object -> RObj, set -> RSet.
"""
from __future__ import annotations

import math
import unicodedata


class RObj:
    """Immutable Rego object. Keys may be any Rego value."""

    __slots__ = ("d", "_h")

    def __init__(self, pairs=()):
        d = {}
        for k, v in pairs:
            d[hk(k)] = (k, v)
        self.d = d
        self._h = None

    @staticmethod
    def from_str_dict(m):
        o = RObj.__new__(RObj)
        o.d = {k: (k, v) for k, v in m.items()}
        o._h = None
        return o

    def get(self, k, default=None):
        e = self.d.get(hk(k))
        return default if e is None else e[1]

    def has(self, k):
        return hk(k) in self.d

    def items(self):
        return self.d.values()

    def keys(self):
        return [k for k, _ in self.d.values()]

    def __len__(self):
        return len(self.d)

    def __repr__(self):
        return "RObj(%s)" % to_string(self)


class RSet:
    """Immutable Rego set."""

    __slots__ = ("d", "_h")

    def __init__(self, elems=()):
        d = {}
        for v in elems:
            d[hk(v)] = v
        self.d = d
        self._h = None

    def has(self, v):
        return hk(v) in self.d

    def elems(self):
        return self.d.values()

    def __len__(self):
        return len(self.d)

    def union(self, o):
        r = RSet()
        r.d = dict(self.d)
        r.d.update(o.d)
        return r

    def intersect(self, o):
        r = RSet()
        r.d = {k: v for k, v in self.d.items() if k in o.d}
        return r

    def diff(self, o):
        r = RSet()
        r.d = {k: v for k, v in self.d.items() if k not in o.d}
        return r

    def __repr__(self):
        return "RSet(%s)" % to_string(self)


def hk(v):
    """Hashable canonical key: equal Rego values <=> equal keys."""
    if isinstance(v, str):
        return v
    if v is None:
        return (0,)
    if isinstance(v, bool):
        return (1, v)
    if isinstance(v, (int, float)):
        if isinstance(v, float) and v.is_integer():
            v = int(v)
        return (2, v)
    if isinstance(v, tuple):
        return (4, tuple(hk(x) for x in v))
    if isinstance(v, RObj):
        if v._h is None:
            v._h = (5, frozenset((k, hk(e[1])) for k, e in v.d.items()))
        return v._h
    if isinstance(v, RSet):
        if v._h is None:
            v._h = (6, frozenset(v.d.keys()))
        return v._h
    raise TypeError("not a rego value: %r" % (v,))


def type_rank(v):
    if v is None:
        return 0
    if isinstance(v, bool):
        return 1
    if isinstance(v, (int, float)):
        return 2
    if isinstance(v, str):
        return 3
    if isinstance(v, tuple):
        return 4
    if isinstance(v, RObj):
        return 5
    if isinstance(v, RSet):
        return 6
    raise TypeError("not a rego value: %r" % (v,))


def type_name(v):
    return ("null", "boolean", "number", "string", "array", "object", "set")[type_rank(v)]


def _cmp_seq(a, b):
    for x, y in zip(a, b):
        c = compare(x, y)
        if c:
            return c
    return (len(a) > len(b)) - (len(a) < len(b))


def sorted_values(vals):
    import functools
    return sorted(vals, key=functools.cmp_to_key(compare))


def compare(a, b):
    """Total order over Rego values (OPA ast.Compare)."""
    ra, rb = type_rank(a), type_rank(b)
    if ra != rb:
        return (ra > rb) - (ra < rb)
    if ra == 0:
        return 0
    if ra in (1, 2, 3):
        return (a > b) - (a < b)
    if ra == 4:
        return _cmp_seq(a, b)
    if ra == 5:
        # objects compare by sorted keys, then values (OPA: key-by-key over sorted keys)
        ka = sorted_values(a.keys())
        kb = sorted_values(b.keys())
        for x, y in zip(ka, kb):
            c = compare(x, y)
            if c:
                return c
            c = compare(a.get(x), b.get(y))
            if c:
                return c
        return (len(ka) > len(kb)) - (len(ka) < len(kb))
    return _cmp_seq(sorted_values(a.elems()), sorted_values(b.elems()))


def equal(a, b):
    return hk(a) == hk(b)


def from_json(j):
    """Python JSON (dict/list/...) -> Rego value."""
    if isinstance(j, dict):
        return RObj.from_str_dict({k: from_json(v) for k, v in j.items()})
    if isinstance(j, (list, tuple)):
        return tuple(from_json(x) for x in j)
    return j


def to_json(v):
    """Rego value -> Python JSON. Sets become sorted arrays (OPA ast.JSON)."""
    if isinstance(v, RObj):
        out = {}
        for k, e in v.items():
            out[k if isinstance(k, str) else to_string(k)] = to_json(e)
        return out
    if isinstance(v, tuple):
        return [to_json(x) for x in v]
    if isinstance(v, RSet):
        return [to_json(x) for x in sorted_values(v.elems())]
    if isinstance(v, float) and math.isfinite(v) and v.is_integer() and abs(v) < 1e21:
        # ast.JSON hands a Number over as json.Number = its TEXT, and the text of a review object's number is what encoding/json
        # wrote for the float64 / int64 the object was decoded into (pkg/target/target.go:140-179 marshals obj.Object; apimachinery's
        # decoder turns `3.0` and `1e2` into float64 3 and 100, the encoder writes `3` and `100`): an integral float is an integer
        # in `details`, exactly as num_to_string prints it in `msg`.  (Negative zero: Go writes `-0`, a JSON reader makes 0 of it.)
        return int(v)
    return v


_ESC = {'"': '\\"', "\\": "\\\\", "\n": "\\n", "\t": "\\t", "\r": "\\r", "\a": "\\a", "\b": "\\b", "\f": "\\f",
        "\v": "\\v"}


def _go_is_print(ch):
    """strconv.IsPrint / unicode.IsPrint: letters, marks, numbers, punctuation, symbols and the ASCII space"""
    return ch == " " or unicodedata.category(ch)[0] in "LMNPS"


def quote(s):
    """Go strconv.Quote (what ast.String.String() and fmt's %q use): printable runes as they are, the C escapes, \\xNN for the other
    ASCII controls, \\uNNNN / \\UNNNNNNNN for every other rune IsPrint refuses (NBSP, format characters, unassigned code points ...)"""
    out = ['"']
    for ch in s:
        cp = ord(ch)
        if ch in _ESC:
            out.append(_ESC[ch])
        elif cp < 0x20 or cp == 0x7F:
            out.append("\\x%02x" % cp)
        elif cp < 0x80 or _go_is_print(ch):
            out.append(ch)
        elif 0xD800 <= cp <= 0xDFFF:
            out.append("\\ufffd")      # (a lone surrogate is no valid rune)
        else:
            out.append("\\u%04x" % cp if cp < 0x10000 else "\\U%08x" % cp)
    out.append('"')
    return "".join(out)


def num_to_string(n):
    """ast.Number.String(): the number's JSON text.  Review objects reach OPA through encoding/json (util.RoundTrip with
    UseNumber), so a float64 carries the text encoding/json's floatEncoder wrote for it."""
    if isinstance(n, int):
        return str(n)
    if math.isfinite(n) and n.is_integer() and abs(n) < 1e21:
        return str(int(n))
    return json_float_text(n)


def _shortest_digits(f):
    """(sign, digits, x): shortest round-trip decimal digits of f != 0 and the decimal exponent of the first one"""
    r = repr(abs(f))
    mant, _, exp = r.partition("e")
    ip, _, fp = mant.partition(".")
    e10 = int(exp) if exp else 0
    if ip.strip("0"):
        x = len(ip.lstrip("0")) - 1 + e10
    else:
        x = -(len(fp) - len(fp.lstrip("0")) + 1) + e10
    digits = (ip + fp).strip("0") or "0"
    return ("-" if f < 0 else ""), digits, x


def _fmt_e(sign, digits, x, min_exp_digits):
    m = digits[0] + ("." + digits[1:] if len(digits) > 1 else "")
    return "%s%se%s%0*d" % (sign, m, "+" if x >= 0 else "-", min_exp_digits, abs(x))


def _fmt_f(sign, digits, x):
    if x >= 0:
        if len(digits) <= x + 1:
            return sign + digits + "0" * (x + 1 - len(digits))
        return sign + digits[: x + 1] + "." + digits[x + 1:]
    return sign + "0." + "0" * (-x - 1) + digits


def json_float_text(f):
    """encoding/json floatEncoder (Go standard library, restated): strconv 'f' with the shortest digits, 'e' when
    abs < 1e-6 or abs >= 1e21, and a one-digit negative exponent written without its leading zero (e-09 -> e-9)."""
    if f != f or f in (float("inf"), float("-inf")):
        return go_float_v(f)      # (not representable in JSON; never reaches here from a document)
    if f == 0:
        return "-0" if math.copysign(1.0, f) < 0 else "0"
    sign, digits, x = _shortest_digits(f)
    if abs(f) < 1e-6 or abs(f) >= 1e21:
        s = _fmt_e(sign, digits, x, 2)
        if len(s) >= 4 and s[-4] == "e" and s[-3] == "-" and s[-2] == "0":
            s = s[:-2] + s[-1]
        return s
    return _fmt_f(sign, digits, x)


def go_float_v(f):
    """Go fmt %v of a float64 = strconv.FormatFloat(f, 'g', -1, 64) (Go standard library, restated): shortest round-trip
    digits; the %e form when the decimal exponent is < -4 or >= 6 (strconv/ftoa.go: with the shortest precision the %g
    decision uses precision 6), exponent of at least two digits -- fmt.Println(6e11) prints 6e+11, 1234567.5 prints
    1.2345675e+06, 123456.5 prints 123456.5."""
    if f != f:
        return "NaN"
    if f in (float("inf"), float("-inf")):
        return "+Inf" if f > 0 else "-Inf"
    if f == 0:
        return "-0" if math.copysign(1.0, f) < 0 else "0"
    sign, digits, x = _shortest_digits(f)
    if x < -4 or x >= 6:
        return _fmt_e(sign, digits, x, 2)
    return _fmt_f(sign, digits, x)


def to_string(v):
    """OPA ast term String()."""
    if v is None:
        return "null"
    if isinstance(v, bool):
        return "true" if v else "false"
    if isinstance(v, (int, float)):
        return num_to_string(v)
    if isinstance(v, str):
        return quote(v)
    if isinstance(v, tuple):
        return "[" + ", ".join(to_string(x) for x in v) + "]"
    if isinstance(v, RObj):
        ks = sorted_values(v.keys())
        return "{" + ", ".join("%s: %s" % (to_string(k), to_string(v.get(k))) for k in ks) + "}"
    if isinstance(v, RSet):
        if len(v) == 0:
            return "set()"
        return "{" + ", ".join(to_string(x) for x in sorted_values(v.elems())) + "}"
    raise TypeError(v)
