"""ORACLE (test infrastructure only -- never imported by the product path).

Builtins used by the reference's in-tree templates (SURVEY.md Appendix B lists the call counts), restated from the
OPA v1.17.1 builtin reference (third-party; go.mod:19).  A builtin that raises BuiltinError makes the calling
expression undefined (OPA's default non-strict mode).
"""
from __future__ import annotations

import json
import math
import re

from .values import (RObj, RSet, compare, equal, from_json, go_float_v, num_to_string, quote, sorted_values,
                     to_json, to_string, type_name)


class BuiltinError(Exception):
    pass


# Go's unicode.IsSpace (strings.TrimSpace): the White_Space code points -- not Python's str.isspace set (which adds U+001C..U+001F)
_GO_SPACES = "\t\n\v\f\r \x85\xa0\u1680" + "".join(chr(c) for c in range(0x2000, 0x200B)) + "\u2028\u2029\u202f\u205f\u3000"


def _go_case(s, fn):
    """Go's strings.ToLower / ToUpper (OPA's lower / upper): rune by rune with the SIMPLE Unicode case mappings -- a code point
    whose full mapping has several code points keeps itself (U+00DF), U+0130 lowers to 'i'; no context rules (final sigma)."""
    out = []
    for c in s:
        if c == "\u0130" and fn is str.lower:
            out.append("i")
            continue
        m = fn(c)
        out.append(m if len(m) == 1 else c)
    return "".join(out)


def _need(v, *types):
    for t in types:
        if t == "number":
            if isinstance(v, (int, float)) and not isinstance(v, bool):
                return v
        elif t == "string":
            if isinstance(v, str):
                return v
        elif t == "array":
            if isinstance(v, tuple):
                return v
        elif t == "object":
            if isinstance(v, RObj):
                return v
        elif t == "set":
            if isinstance(v, RSet):
                return v
        elif t == "boolean":
            if isinstance(v, bool):
                return v
    raise BuiltinError("operand must be %s, got %s" % ("/".join(types), type_name(v)))


def _isnum(v):
    return isinstance(v, (int, float)) and not isinstance(v, bool)


def _norm(n):
    if isinstance(n, float) and n.is_integer() and abs(n) < 2 ** 63:
        return int(n)
    return n


# ---------------------------------------------------------------- Go regexp (RE2) -> python re
# OPA's re_match / regex.match call Go's regexp (regexp/syntax with Perl flags; third-party, absent from /root/reference).
# The oracle evaluates them with Python's `re` after a TRANSLATION that restates Go's syntax rules -- "parity unpinned"
# beyond the patterns the reference's fixtures hold (agilebank allowedRegex, tests/test_oracle_rego.py):
#   * what regexp.Compile rejects raises BuiltinError (re_match is then undefined): unknown alphanumeric escapes, \C,
#     backreferences, lookarounds, stacked repetition (a**), repeat counts > 1000 or {n,m} with n > m, bad classes
#   * Go's ASCII definitions are spelled out: \d [0-9], \w [0-9A-Za-z_], \s [\t\n\f\r ] (no \v), \b ASCII word boundary;
#     `$` without (?m) is end of TEXT (Python's `$` also matches before a final newline); a `{` that does not open a valid
#     repetition is a literal; (?flags) apply to the rest of the enclosing group; (?U) only swaps greediness
#   * matching itself is by code point with Unicode simple case folding under (?i), as in Go
#   * OracleRegexUnsupported: valid Go this translator does not cover (\pL, [[:^alpha:]] / \D \W \S inside a class)
_POSIX = {"alpha": "a-zA-Z", "digit": "0-9", "alnum": "a-zA-Z0-9", "upper": "A-Z", "lower": "a-z", "space": " \t\n\r\f\v",
          "blank": " \t", "cntrl": "\x00-\x1f\x7f", "graph": "!-~", "print": " -~", "ascii": "\x00-\x7f",
          "punct": "!-/:-@\\[-`{-~", "xdigit": "0-9A-Fa-f", "word": "0-9A-Za-z_"}
_PERL = {"d": "0-9", "w": "0-9A-Za-z_", "s": "\t\n\f\r "}
_WORD = "[0-9A-Za-z_]"
_re_cache = {}


class OracleRegexUnsupported(Exception):
    """valid Go syntax outside the oracle's translator (tests skip such patterns)"""


def _go_to_python(pat):
    def bad(why):
        return BuiltinError("error parsing regexp: %s: `%s`" % (why, pat))

    n = len(pat)
    out = []
    i = 0
    # group stack: (multiline flag on entry, specs of the (?flags) scopes opened inside the group: closed at its end and
    # around every `|` of the group, because in Go the flags stay in force across the alternatives that follow)
    stack = [[False, []]]
    multiline = False
    can_repeat = False      # the previous item is an operand
    repeated = False        # ... and already carries a repetition operator

    def esc_rune(k):
        """rune denoted by the escape whose letter is pat[k]; -> (code point, next index)"""
        e = pat[k]
        simple = {"a": 7, "f": 12, "n": 10, "r": 13, "t": 9, "v": 11}
        if e in simple:
            return simple[e], k + 1
        if e == "x":
            if k + 1 >= n:
                raise bad("invalid escape sequence")
            if pat[k + 1] == "{":
                end = pat.find("}", k + 2)
                digits = pat[k + 2:end] if end > 0 else ""
                if not digits or any(c not in "0123456789abcdefABCDEF" for c in digits) or int(digits, 16) > 0x10FFFF:
                    raise bad("invalid escape sequence")
                return int(digits, 16), end + 1
            digits = pat[k + 1:k + 3]
            if len(digits) != 2 or any(c not in "0123456789abcdefABCDEF" for c in digits):
                raise bad("invalid escape sequence")
            return int(digits, 16), k + 3
        if e in "01234567":
            if e != "0" and not (k + 1 < n and pat[k + 1] in "01234567"):
                raise bad("invalid escape sequence")      # a lone digit would be a backreference
            m = k + 1
            while m < n and m < k + 3 and pat[m] in "01234567":
                m += 1
            return int(pat[k:m], 8), m
        if ord(e) < 0x80 and not e.isalnum():
            return ord(e), k + 1
        raise bad("invalid escape sequence")

    def lit(cp):
        return re.escape(chr(cp))

    def parse_class(k):
        """pat[k] is the char after '['; -> (python class text, next index)"""
        body = []
        neg = False
        if k < n and pat[k] == "^":
            neg = True
            k += 1
        first = True
        while True:
            if k >= n:
                raise bad("missing closing ]")
            c = pat[k]
            if c == "]" and not first:
                k += 1
                break
            first = False
            if c == "[" and k + 1 < n and pat[k + 1] == ":":
                end = pat.find(":]", k + 2)
                if end < 0:
                    raise bad("invalid character class range")
                name = pat[k + 2:end]
                if name.startswith("^"):
                    if name[1:] in _POSIX:
                        raise OracleRegexUnsupported(pat)
                    raise bad("invalid character class range")
                if name not in _POSIX:
                    raise bad("invalid character class range")
                body.append(_POSIX[name])
                k = end + 2
                continue
            if c == "\\":
                if k + 1 >= n:
                    raise bad("trailing backslash")
                e = pat[k + 1]
                if e in "dws":
                    body.append(_PERL[e])
                    k += 2
                    continue
                if e in "DWSpP":
                    raise OracleRegexUnsupported(pat)
                lo, k = esc_rune(k + 1)
            else:
                lo, k = ord(c), k + 1
            if k + 1 < n and pat[k] == "-" and pat[k + 1] != "]":
                h = pat[k + 1]
                if h == "\\":
                    if k + 2 >= n:
                        raise bad("trailing backslash")
                    if pat[k + 2] in "dwsDWSpP":
                        raise bad("invalid character class range")
                    hi, k = esc_rune(k + 2)
                else:
                    hi, k = ord(h), k + 2
                if hi < lo:
                    raise bad("invalid character class range")
                body.append(lit(lo) + "-" + lit(hi))
            else:
                body.append(lit(lo))
        return "[" + ("^" if neg else "") + "".join(body) + "]", k

    while i < n:
        c = pat[i]
        if c in "*+?":
            if not can_repeat:
                raise bad("missing argument to repetition operator")
            if repeated:
                raise bad("invalid nested repetition operator")
            out.append(c)
            i += 1
            if i < n and pat[i] == "?":
                out.append("?")
                i += 1
            repeated = True
            continue
        if c == "{":
            m = re.match(r"\{(\d+)(,(\d*))?\}", pat[i:])
            if m and can_repeat:
                lo = int(m.group(1))
                hi = lo if m.group(2) is None else (int(m.group(3)) if m.group(3) else -1)
                if lo > 1000 or hi > 1000 or (hi >= 0 and hi < lo):
                    raise bad("invalid repeat count")
                if repeated:
                    raise bad("invalid nested repetition operator")
                out.append(m.group(0))
                i += len(m.group(0))
                if i < n and pat[i] == "?":
                    out.append("?")
                    i += 1
                repeated = True
                continue
            if m and not can_repeat:
                raise bad("missing argument to repetition operator")
            out.append("\\{")
            i += 1
            can_repeat, repeated = True, False
            continue
        repeated = False
        if c == "(":
            if pat.startswith("(?", i):
                if pat.startswith("(?P<", i) or pat.startswith("(?<", i):
                    start = i + (4 if pat.startswith("(?P<", i) else 3)
                    end = pat.find(">", start)
                    name = pat[start:end] if end > 0 else ""
                    if not name or not re.fullmatch(r"[A-Za-z0-9_]+", name) or pat.startswith("(?<=", i) or pat.startswith("(?<!", i):
                        raise bad("invalid named capture")
                    out.append("(?P<%s>" % name)
                    stack.append([multiline, []])
                    i = end + 1
                    can_repeat = False
                    continue
                m = re.match(r"\(\?([imsU]*)(-[imsU]+)?([:)])", pat[i:])
                if not m or (not m.group(1) and not m.group(2) and m.group(3) == ")"):
                    raise bad("invalid or unsupported Perl syntax")
                on = m.group(1).replace("U", "")
                off = (m.group(2) or "")[1:].replace("U", "")
                new_ml = (multiline or "m" in on) and "m" not in off
                spec = on + ("-" + off if off else "")
                if m.group(3) == ":":
                    stack.append([multiline, []])
                    out.append("(?%s:" % spec if spec else "(?:")
                else:                      # (?flags): in force until the enclosing group ends
                    if spec:
                        out.append("(?%s:" % spec)
                        stack[-1][1].append(spec)
                multiline = new_ml
                i += len(m.group(0))
                can_repeat = False
                if m.group(3) == ")" and (pat[i:i + 1] in ("*", "+", "?") or re.match(r"\\{\\d+(,\\d*)?\\}", pat[i:])):
                    # Go binds a repetition after a flag group to whatever precedes the group on its parse stack
                    raise OracleRegexUnsupported(pat)
                continue
            out.append("(")
            stack.append([multiline, []])
            i += 1
            can_repeat = False
            continue
        if c == ")":
            if len(stack) == 1:
                raise bad("unexpected )")
            saved_ml, owed = stack.pop()
            out.append(")" * len(owed) + ")")
            multiline = saved_ml
            i += 1
            can_repeat = True
            continue
        if c == "|":
            out.append(")" * len(stack[-1][1]) + "|" + "".join("(?%s:" % sp for sp in stack[-1][1]))
            i += 1
            can_repeat = False
            continue
        can_repeat = True
        if c == "[":
            text, i = parse_class(i + 1)
            out.append(text)
        elif c == "^":
            out.append("^")
            i += 1
        elif c == "$":
            out.append("$" if multiline else "\\Z")
            i += 1
        elif c == ".":
            out.append(".")
            i += 1
        elif c == "\\":
            if i + 1 >= n:
                raise bad("trailing backslash")
            e = pat[i + 1]
            if e in "dws":
                out.append("[" + _PERL[e] + "]")
                i += 2
            elif e in "DWS":
                out.append("[^" + _PERL[e.lower()] + "]")
                i += 2
            elif e == "A":
                out.append("\\A")
                i += 2
            elif e == "z":
                out.append("\\Z")
                i += 2
            elif e == "b":
                out.append("(?:(?<=%s)(?!%s)|(?<!%s)(?=%s))" % (_WORD, _WORD, _WORD, _WORD))
                i += 2
            elif e == "B":
                out.append("(?:(?<=%s)(?=%s)|(?<!%s)(?!%s))" % (_WORD, _WORD, _WORD, _WORD))
                i += 2
            elif e in "pP":
                raise OracleRegexUnsupported(pat)
            elif e == "Q":
                end = pat.find("\\E", i + 2)
                text = pat[i + 2:] if end < 0 else pat[i + 2:end]
                i = n if end < 0 else end + 2
                if text:
                    out.append(re.escape(text[:-1]) + "(?:" + re.escape(text[-1]) + ")")
                else:
                    can_repeat = False
            else:
                cp, i = esc_rune(i + 1)
                out.append(lit(cp))
        else:
            out.append(re.escape(c))
            i += 1
    if len(stack) != 1:
        raise bad("missing closing )")
    out.append(")" * len(stack[0][1]))
    return "".join(out)


def go_regex(pat):
    r = _re_cache.get(pat)
    if r is None:
        try:
            r = re.compile(_go_to_python(pat))
        except re.error as e:
            raise BuiltinError("bad regex %r: %s" % (pat, e))
        _re_cache[pat] = r
    return r


# ---------------------------------------------------------------- sprintf (Go fmt)
# fmt's doPrintf: flags, width, .precision, then the verb = the next rune WHATEVER it is ("%!" is the bad verb '!'); a format that ends
# before the verb prints %!(NOVERB).  Not modelled (by the product either; no template of the corpus uses them): explicit argument
# indexes %[n]d and the '*' width / precision -- '[' and '*' are read as (bad) verbs.
_VERB_RE = re.compile(r"%([-+# 0]*)(\d+)?(?:\.(\d*))?(.)?", re.S)


def _go_arg(v):
    """OPA topdown builtinSprintf argument conversion."""
    if _isnum(v):
        if isinstance(v, int):
            return ("int", v)
        if v.is_integer() and abs(v) < 1e21:      # number text without exponent: Number.Int(), else big.Int.SetString
            return ("int", int(v))
        return ("float64", v)
    if isinstance(v, str):
        return ("string", v)
    return ("string", to_string(v))


def _bad(verb, kind, val):
    if kind == "string":
        return "%%!%s(string=%s)" % (verb, val)
    if kind == "int":
        return "%%!%s(int=%d)" % (verb, val)
    return "%%!%s(float64=%s)" % (verb, go_float_v(val))


def _pad(s, flags, width):
    if width is None or len(s) >= width:
        return s
    if "-" in flags:
        return s + " " * (width - len(s))
    if "0" in flags and s and (s[0].isdigit() or s[0] in "+-"):
        sign = ""
        if s[0] in "+-":
            sign, s = s[0], s[1:]
        return sign + "0" * (width - len(s) - len(sign)) + s
    return " " * (width - len(s)) + s


_DIGITS = "0123456789abcdef"


def _fmt_integer(val, base, flags, width, prec, upper=False, o_prefix=False):
    """fmt/format.go fmtInteger: precision = minimum number of digits (%.0d of 0 prints nothing); without one the 0 flag turns the
    width into that minimum; '#' adds the base prefix; then the sign ('-', '+' flag, ' ' flag); the rest of the width is spaces"""
    neg, u = val < 0, abs(val)
    digits = ""
    while u:
        digits = _DIGITS[u % base] + digits
        u //= base
    digits = digits or "0"
    if upper:
        digits = digits.upper()
    zero = "0" in flags and "-" not in flags
    if prec is not None:
        if prec == 0 and val == 0:
            digits = ""
        digits = digits.rjust(prec, "0")
    elif zero and width is not None:
        digits = digits.rjust(width - (1 if neg or "+" in flags or " " in flags else 0), "0")
    if "#" in flags:
        if base == 2:
            digits = "0b" + digits
        elif base == 8 and not digits.startswith("0"):
            digits = "0" + digits
        elif base == 16:
            digits = ("0X" if upper else "0x") + digits
    if o_prefix:
        digits = "0o" + digits
    digits = ("-" if neg else "+" if "+" in flags else " " if " " in flags else "") + digits
    if width is not None and len(digits) < width:
        digits = digits.ljust(width) if "-" in flags else digits.rjust(width)
    return digits


def _fmt_float(num, flags, width):
    """fmt/format.go fmtFloat: sign '-', or '+' / ' ' by flag; the 0 flag pads between the sign and the digits"""
    if not num.startswith("-"):
        num = ("+" if "+" in flags else " " if " " in flags else "") + num
    if width is None or len(num) >= width:
        return num
    if "-" in flags:
        return num.ljust(width)
    if "0" not in flags:
        return num.rjust(width)
    sign = num[0] if num[0] in "+- " else ""
    return sign + num[len(sign):].rjust(width - len(sign), "0")


def _as_v(flags, width, prec, kind, val):
    """the operand under %v with the directive's flags, width and precision (printArg(arg, 'v')): fmtS cuts a string to the precision,
    fmtInteger takes it as the minimum number of digits, fmtFloat prints %g -- the shortest text, or `prec` significant digits"""
    if kind == "string":
        return _pad(val if prec is None else val[:prec], flags, width)
    if kind == "int":
        return _fmt_integer(val, 10, flags, width, prec)
    return _fmt_float(go_float_v(val) if prec is None else ("%." + str(prec) + "g") % val, flags, width)


def _format_operand(verb, flags, width, prec, kind, val):
    """one verb applied to one operand (fmt/print.go printArg -> fmtInteger / fmtFloat / fmtString / badVerb).  badVerb writes
    %!verb(type=value) with the value printed as %v UNDER THE SAME flags, width and precision.  Not modelled (no template of the
    corpus uses them): %q of an integer (quoted rune), %x / %b of a float64, '#' on v / q / floats, '+' on q, %t (operands are never
    Go bools here: builtinSprintf stringifies them)"""
    if verb == "T":   # printArg: the operand's Go type through fmtS
        return _pad(kind if prec is None else kind[:prec], flags, width)
    if verb == "v":
        return _as_v(flags, width, prec, kind, val)
    if kind == "int":
        if verb == "d":
            return _fmt_integer(val, 10, flags, width, prec)
        if verb in "xX":
            return _fmt_integer(val, 16, flags, width, prec, upper=verb == "X")
        if verb in "oO":
            return _fmt_integer(val, 8, flags, width, prec, o_prefix=verb == "O")
        if verb == "b":
            return _fmt_integer(val, 2, flags, width, prec)
        if verb == "c":   # fmtC: no valid code point -> U+FFFD
            return _pad(chr(val) if 0 <= val <= 0x10FFFF and not 0xD800 <= val <= 0xDFFF else "\ufffd", flags, width)
        if verb == "U" and val >= 0:
            return _pad("U+%04X" % val, flags, width)
    elif kind == "float64":
        if verb in "gG" and prec is None:   # %g without a precision: the shortest text that round-trips
            num = go_float_v(val)
            return _fmt_float(num.replace("e", "E") if verb == "G" else num, flags, width)
        if verb in "feEgGF":
            num = ("%." + str(6 if prec is None else prec) + ("f" if verb == "F" else verb)) % val
            return _fmt_float(num, flags, width)
    else:
        if verb == "s":
            return _as_v(flags, width, prec, kind, val)
        if verb == "q":   # fmtQ: cut to the precision first, then quote
            return _pad(quote(val if prec is None else val[:prec]), flags, width)
        if verb in "xX":   # fmtSbx: precision in bytes, ' ' between bytes, '#' prefix (on every byte when separated)
            raw = val.encode()
            raw = raw if prec is None else raw[:prec]
            pre = ("0X" if verb == "X" else "0x") if "#" in flags else ""
            hx = [("%02X" if verb == "X" else "%02x") % c for c in raw]
            text = " ".join(pre + h for h in hx) if " " in flags else (pre if hx else "") + "".join(hx)
            return _pad(text, flags, width)
    return "%%!%s(%s=%s)" % (verb, kind, _as_v(flags, width, prec, kind, val))


def _too_large(digits):
    """parsenum's overflow rule: tooLarge(num) is asked before every further digit"""
    num = 0
    for ch in digits or "":
        if num > 1000000:
            return True
        num = num * 10 + ord(ch) - 48
    return False


def go_sprintf(fmt, args):
    args = [_go_arg(a) for a in args]
    out = []
    pos = 0
    ai = 0
    for m in _VERB_RE.finditer(fmt):
        out.append(fmt[pos:m.start()])
        pos = m.end()
        flags, width, prec, verb = m.group(1), m.group(2), m.group(3), m.group(4)
        if _too_large(width) or _too_large(prec):
            # fmt/print.go parsenum: a number already beyond 1e6 with a further digit to come gives up the directive AND the rest
            # of the format (it returns the format's end as the next position): %!(NOVERB), then the operands are all EXTRA
            out.append("%!(NOVERB)")
            pos = len(fmt)
            break
        if verb is None:
            out.append("%!(NOVERB)")
            continue
        if verb == "%":
            out.append("%")
            continue
        if ai >= len(args):
            out.append("%%!%s(MISSING)" % verb)
            continue
        kind, val = args[ai]
        ai += 1
        width = int(width) if width else None
        prec = (int(prec) if prec else 0) if prec is not None else None
        s = _format_operand(verb, flags, width, prec, kind, val)
        out.append(s)
    out.append(fmt[pos:])
    if ai < len(args):
        extra = ", ".join("%s=%s" % (k, v if k != "float64" else go_float_v(v)) for k, v in args[ai:])
        out.append("%!(EXTRA " + extra + ")")
    return "".join(out)


# ---------------------------------------------------------------- json.marshal
_JSON_SHORT = {'"': '\\"', "\\": "\\\\", "\n": "\\n", "\r": "\\r", "\t": "\\t", "\b": "\\b", "\f": "\\f"}


def _go_json_string(s):
    """encoding/json's string encoder with EscapeHTML on (json.Marshal's default): the short escapes, \\u00NN for the other controls,
    < > & as \\u003c \\u003e \\u0026, U+2028 / U+2029 escaped; everything else as it is"""
    out = ['"']
    for ch in s:
        if ch in _JSON_SHORT:
            out.append(_JSON_SHORT[ch])
        elif ord(ch) < 0x20 or ch in "<>&\u2028\u2029":
            out.append("\\u%04x" % ord(ch))
        else:
            out.append(ch)
    out.append('"')
    return "".join(out)


def _go_json_marshal(x):
    """builtinJSONMarshal (topdown/encoding.go): json.Marshal(ast.JSON(x)) -- compact, object keys sorted, numbers as their own text"""
    if x is None:
        return "null"
    if x is True or x is False:
        return "true" if x else "false"
    if isinstance(x, (int, float)):
        return num_to_string(x)
    if isinstance(x, str):
        return _go_json_string(x)
    if isinstance(x, (list, tuple)):
        return "[" + ",".join(_go_json_marshal(v) for v in x) + "]"
    return "{" + ",".join(_go_json_string(k) + ":" + _go_json_marshal(x[k]) for k in sorted(x)) + "}"


# ---------------------------------------------------------------- builtins
def b_count(x):
    if isinstance(x, str):
        return len(x)
    if isinstance(x, (tuple, RObj, RSet)):
        return len(x)
    raise BuiltinError("count: bad operand")


def _numbers(c, name):
    if isinstance(c, tuple):
        it = c
    elif isinstance(c, RSet):
        it = list(c.elems())
    else:
        raise BuiltinError(name + ": operand must be array or set")
    for v in it:
        _need(v, "number")
    return it


def b_sum(c):
    return _norm(sum(_numbers(c, "sum")))


def b_product(c):
    r = 1
    for v in _numbers(c, "product"):
        r *= v
    return _norm(r)


def _coll(c, name):
    if isinstance(c, tuple):
        return list(c)
    if isinstance(c, RSet):
        return list(c.elems())
    raise BuiltinError(name + ": operand must be array or set")


def b_max(c):
    it = _coll(c, "max")
    if not it:
        raise BuiltinError("max of empty")
    return sorted_values(it)[-1]


def b_min(c):
    it = _coll(c, "min")
    if not it:
        raise BuiltinError("min of empty")
    return sorted_values(it)[0]


def b_sort(c):
    return tuple(sorted_values(_coll(c, "sort")))


def b_any(c):
    return any(v is True for v in _coll(c, "any"))


def b_all(c):
    return all(v is True for v in _coll(c, "all"))


def b_concat(delim, c):
    _need(delim, "string")
    it = _coll(c, "concat")
    if isinstance(c, RSet):
        it = sorted_values(it)
    for v in it:
        _need(v, "string")
    return delim.join(it)


def b_substring(s, off, ln):
    _need(s, "string")
    _need(off, "number")
    _need(ln, "number")
    off, ln = int(off), int(ln)
    if off < 0:
        raise BuiltinError("negative offset")
    if off >= len(s):
        return ""
    if ln < 0:
        return s[off:]
    return s[off:off + ln]


def b_split(s, d):
    _need(s, "string")
    _need(d, "string")
    if d == "":
        return tuple(s)
    return tuple(s.split(d))


def b_trim(s, cut):
    return _need(s, "string").strip(_need(cut, "string")) if cut else s


def b_replace(s, old, new):
    return _need(s, "string").replace(_need(old, "string"), _need(new, "string"))


def b_to_number(x):
    if x is None:
        return 0
    if isinstance(x, bool):
        return 1 if x else 0
    if _isnum(x):
        return x
    if isinstance(x, str):
        # builtinToNumber: strconv.ParseFloat's decimal syntax (ASCII digits only); Inf / Infinity / NaN, which ParseFloat would
        # take, are refused by the builtin itself
        if not re.fullmatch(r"[+-]?([0-9]+\.?[0-9]*([eE][+-]?[0-9]+)?|\.[0-9]+([eE][+-]?[0-9]+)?)", x):
            raise BuiltinError("to_number: invalid syntax")
        try:
            return int(x)
        except ValueError:
            try:
                return _norm(float(x))
            except ValueError:
                raise BuiltinError("to_number: invalid syntax")
    raise BuiltinError("to_number: bad operand")


def b_object_get(obj, key, default):
    _need(obj, "object")
    if isinstance(key, tuple):
        cur = obj
        for k in key:
            if isinstance(cur, RObj) and cur.has(k):
                cur = cur.get(k)
            elif isinstance(cur, tuple) and _isnum(k) and 0 <= int(k) < len(cur) and float(k).is_integer():
                cur = cur[int(k)]
            else:
                return default
        return cur
    return obj.get(key) if obj.has(key) else default


def _strs(x, name):
    if isinstance(x, str):
        return [x]
    it = _coll(x, name)
    for v in it:
        _need(v, "string")
    return it


def b_any_prefix_match(search, base):
    ss, bs = _strs(search, "strings.any_prefix_match"), _strs(base, "strings.any_prefix_match")
    return any(s.startswith(b) for s in ss for b in bs)


def b_any_suffix_match(search, base):
    ss, bs = _strs(search, "strings.any_suffix_match"), _strs(base, "strings.any_suffix_match")
    return any(s.endswith(b) for s in ss for b in bs)


def b_re_match(pat, val):
    _need(pat, "string")
    _need(val, "string")
    return go_regex(pat).search(val) is not None


def b_format_int(n, base):
    _need(n, "number")
    _need(base, "number")
    n = int(math.floor(n)) if n >= 0 else -int(math.floor(-n))
    digs = "0123456789abcdef"
    if base not in (2, 8, 10, 16):
        raise BuiltinError("format_int: bad base")
    if n == 0:
        return "0"
    s, m = "", abs(n)
    while m:
        s = digs[m % base] + s
        m //= base
    return ("-" if n < 0 else "") + s


def b_indexof(s, sub):
    return _need(s, "string").find(_need(sub, "string"))


def b_array_slice(a, lo, hi):
    _need(a, "array")
    lo = max(0, int(_need(lo, "number")))
    hi = min(len(a), int(_need(hi, "number")))
    return a[lo:hi] if lo < hi else ()


def b_union(s):
    r = RSet()
    for x in _need(s, "set").elems():
        r = r.union(_need(x, "set"))
    return r


def b_intersection(s):
    it = list(_need(s, "set").elems())
    if not it:
        return RSet()
    r = _need(it[0], "set")
    for x in it[1:]:
        r = r.intersect(_need(x, "set"))
    return r


def b_object_keys(o):
    return RSet(_need(o, "object").keys())


def b_object_remove(o, ks):
    _need(o, "object")
    kk = RSet(_coll(ks, "object.remove") if not isinstance(ks, RObj) else ks.keys())
    return RObj((k, v) for k, v in o.items() if not kk.has(k))


def b_object_union(a, b):
    _need(a, "object")
    _need(b, "object")
    d = {}
    for k, v in a.items():
        d[id(k)] = (k, v)
    out = RObj(list(a.items()))
    for k, v in b.items():
        if out.has(k) and isinstance(out.get(k), RObj) and isinstance(v, RObj):
            v = b_object_union(out.get(k), v)
        out = RObj([(kk, vv) for kk, vv in out.items() if not equal(kk, k)] + [(k, v)])
    return out


BUILTINS = {
    "count": b_count, "sum": b_sum, "product": b_product, "max": b_max, "min": b_min, "sort": b_sort,
    "any": b_any, "all": b_all,
    "abs": lambda x: abs(_need(x, "number")),
    "round": lambda x: int(math.floor(x + 0.5)) if _need(x, "number") >= 0 else -int(math.floor(-x + 0.5)),
    "ceil": lambda x: int(math.ceil(_need(x, "number"))),
    "floor": lambda x: int(math.floor(_need(x, "number"))),
    "sprintf": lambda f, a: go_sprintf(_need(f, "string"), _need(a, "array")),
    "concat": b_concat,
    "contains": lambda s, sub: _need(sub, "string") in _need(s, "string"),
    "startswith": lambda s, p: _need(s, "string").startswith(_need(p, "string")),
    "endswith": lambda s, p: _need(s, "string").endswith(_need(p, "string")),
    "lower": lambda s: _go_case(_need(s, "string"), str.lower),
    "upper": lambda s: _go_case(_need(s, "string"), str.upper),
    "trim": b_trim,
    "trim_left": lambda s, c: _need(s, "string").lstrip(_need(c, "string")) if c else s,
    "trim_right": lambda s, c: _need(s, "string").rstrip(_need(c, "string")) if c else s,
    "trim_prefix": lambda s, p: s[len(p):] if _need(s, "string").startswith(_need(p, "string")) else s,
    "trim_suffix": lambda s, p: s[:len(s) - len(p)] if p and _need(s, "string").endswith(_need(p, "string")) else s,
    "trim_space": lambda s: _need(s, "string").strip(_GO_SPACES),
    "split": b_split, "replace": b_replace, "substring": b_substring, "indexof": b_indexof,
    "format_int": b_format_int,
    "strings.reverse": lambda s: _need(s, "string")[::-1],
    "strings.any_prefix_match": b_any_prefix_match, "strings.any_suffix_match": b_any_suffix_match,
    "re_match": b_re_match, "regex.match": b_re_match,
    "regex.is_valid": lambda p: _regex_valid(p),
    "is_string": lambda x: isinstance(x, str), "is_number": _isnum,
    "is_boolean": lambda x: isinstance(x, bool), "is_array": lambda x: isinstance(x, tuple),
    "is_object": lambda x: isinstance(x, RObj), "is_set": lambda x: isinstance(x, RSet),
    "is_null": lambda x: x is None, "type_name": type_name,
    "to_number": b_to_number,
    "object.get": b_object_get, "object.keys": b_object_keys, "object.remove": b_object_remove,
    "object.union": b_object_union,
    "array.concat": lambda a, b: _need(a, "array") + _need(b, "array"),
    "array.slice": b_array_slice, "array.reverse": lambda a: _need(a, "array")[::-1],
    "union": b_union, "intersection": b_intersection,
    "json.marshal": lambda x: _go_json_marshal(to_json(x)),
    "json.unmarshal": lambda s: _json_unmarshal(s),
    "print": lambda *a: True, "trace": lambda s: True,
}


def _regex_valid(p):
    if not isinstance(p, str):
        return False
    try:
        go_regex(p)
        return True
    except BuiltinError:
        return False


def _json_unmarshal(s):
    try:
        return from_json(json.loads(_need(s, "string")))
    except ValueError as e:
        raise BuiltinError(str(e))


def arith(op, a, b):
    if op == "-" and isinstance(a, RSet) and isinstance(b, RSet):
        return a.diff(b)
    if op == "&":
        return _need(a, "set").intersect(_need(b, "set"))
    if op == "|":
        return _need(a, "set").union(_need(b, "set"))
    _need(a, "number")
    _need(b, "number")
    if op == "+":
        return _norm(a + b)
    if op == "-":
        return _norm(a - b)
    if op == "*":
        return _norm(a * b)
    if op == "/":
        if b == 0:
            raise BuiltinError("divide by zero")
        if isinstance(a, int) and isinstance(b, int) and a % b == 0:
            return a // b
        return _norm(a / b)
    if op == "%":
        # topdown/arithmetic.go builtinRem (OPA, restated): both operands through builtins.NumberToInt -- a number whose
        # big.Float value is integral (2.0, -0.0, 1e21) IS an integer there -- then big.Int.Rem: exact, truncated division
        # (the sign of the dividend)
        if not (float(a).is_integer() and float(b).is_integer()) if isinstance(a, float) or isinstance(b, float) else False:
            raise BuiltinError("modulo on floating-point number")
        a, b = int(a), int(b)
        if b == 0:
            raise BuiltinError("modulo by zero")
        r = abs(a) % abs(b)
        return -r if a < 0 else r
    raise BuiltinError("bad operator " + op)
