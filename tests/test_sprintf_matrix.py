"""sprintf (topdown/strings.go builtinSprintf -> Go's fmt.Sprintf) three ways: the Python oracle against known answers of fmt's
documented behaviour (fmt/format.go fmtInteger, fmtFloat, fmtS, fmtSbx, fmtC; print.go badVerb -- the operand as %v under the directive's own flags -- / NOVERB / MISSING / EXTRA), then the
product's renderer and the independent compiled checker against the oracle on a verb x flag x operand matrix.

The known answers are written from fmt's documented rules, not from a Go run (no Go toolchain here).  Not modelled by any of the
three (no template of the reference's library uses them): %[n] argument indexes and '*' widths, %q of an integer, %x / %b of a
float64, '#' on v / q / floats, '+' on q."""
import json

import pytest

from gatekeeper_amd import driver as D
from oracle import client as OC
from oracle import target as OT
from oracle.indep_check import IndepChecker
from oracle.rego_builtins import go_sprintf

KNOWN = [("%g", [1234567.25], "1.23456725e+06"), ("%.3d", [7], "007"), ("%8.3d", [-7], "    -007"), ("%08d", [-7], "-0000007"), ("%+d", [5], "+5"),
         ("% d", [5], " 5"), ("%#x", [255], "0xff"), ("%#o", [8], "010"), ("%O", [8], "0o10"), ("%U", [255], "U+00FF"), ("%x", [-255], "-ff"),
         ("%#b", [5], "0b101"), ("%+.1f", [1.5], "+1.5"), ("%08.2f", [-1.5], "-0001.50"), ("% x", ["hi"], "68 69"), ("%# x", ["hi"], "0x68 0x69"),
         ("%.0d", [0], ""), ("%5.0d", [0], "     "), ("%c", [128512], "\U0001F600"), ("%c", [-1], "�"), ("%6.2f", [3.14159], "  3.14"),
         ("%e", [1500.5], "1.500500e+03"), ("%G", [1e-7], "1E-07"), ("%-5d|", [42], "42   |"), ("%05d", [42], "00042"), ("%+v", [1.5], "+1.5"),
         ("%8v|", ["é"], "       é|"), ("%.2s", ["héllo"], "hé"), ("%20d", ["s"], "%!d(string=                   s)"), ("%5s", [5], "%!s(int=    5)"), ("%08d", [1.5], "%!d(float64=000001.5)"), ("%.2v", ["héllo"], "hé"),
         ("%.3v", [1234567.25], "1.23e+06"), ("%.2q", ["héllo"], '"hé"'), ("%8.3d", ["hello"], "%!d(string=     hel)"), ("%.3v", [7], "007"), ("%-6x|", [1.5], "%!x(float64=1.5   )|"), ("%", [1], "%!(NOVERB)%!(EXTRA int=1)"),
         ("%!", [1], "%!!(int=1)"), ("%d %d", [1], "1 %!d(MISSING)"), ("%x", ["hi"], "6869"), ("%X", [255], "FF"), ("%q", ["a\"b"], '"a\\"b"'),
         ("%5%", [], "%"), ("%-08d|", [42], "42      |"), ("%+08d", [42], "+0000042"), ("%08v", [1.5], "000001.5"), ("%b", [-5], "-101"),
         ("%o", [64], "100"), ("%#X", [255], "0XFF"), ("%.2x", ["hello"], "6865"), ("%v", [1e21], "1e+21"), ("%v", [1e6], "1000000"),
         ("%v", [1234567.25], "1.23456725e+06"), ("%s", [1], "%!s(int=1)"), ("%d", [1.5], "%!d(float64=1.5)"), ("%v", [2.0], "2"),
         ("%v %v", [1, 2, "x"], "1 2%!(EXTRA string=x)"), ("%é", [1], "%!é(int=1)"), ("100%%", [], "100%"), ("%v", [[1, "a"]], '[1, "a"]'), ("%T %T %T", [1, 1.5, "s"], "int float64 string"), ("%p", [1], "%!p(int=1)"),
         # fmt/print.go parsenum gives a width / precision up once it is beyond 1e6 with a further digit to come -- the directive AND the rest of the
         # format (no allocation of gigabytes): round-4 advisor finding
         ("a%99999999dz", [1], "a%!(NOVERB)%!(EXTRA int=1)"), ("%.20000000000f|", [1.5], "%!(NOVERB)%!(EXTRA float64=1.5)"), ("%7d|", [1], "      1|")]


@pytest.mark.parametrize("fmt,args,want", KNOWN)
def test_oracle_sprintf_known_answers(fmt, args, want):
    from oracle.values import from_json
    assert go_sprintf(fmt, [from_json(a) for a in args]) == want


TEMPLATE = {"apiVersion": "templates.gatekeeper.sh/v1", "kind": "ConstraintTemplate", "metadata": {"name": "k8sfmtmatrix"},
            "spec": {"crd": {"spec": {"names": {"kind": "K8sFmtMatrix"}, "validation": {"openAPIV3Schema": {
                "type": "object", "properties": {"fmts": {"type": "array", "items": {"type": "string"}}}}}}},
                "targets": [{"target": "admission.k8s.gatekeeper.sh", "rego": """
package k8sfmtmatrix
violation[{"msg": msg}] {
  f := input.parameters.fmts[i]
  x := input.review.object.spec.x
  msg := sprintf("%d:%v => [%s]", [i, f, sprintf(f, [x, x])])
}
"""}]}}
MODS = ("", "8", "-8", "08", "+", ".2", "8.3", "#", " ", "+.1", "-12.4", "+08", ".0", "# ")
MALFORMED = ["%", "%%", "100%", "%!", "% d", "%5%", "%z", "%v %v %v", "é%3vé", "%é", "%-", "%8", "%.", "%.3", "x%123456789dy", "%.99999999999s tail %v"]
OPERANDS = [0, 1, -7, 255, 65, 1.5, -3.25, 1e21, 1e-7, 0.000123, 123456789012, 1234567.25, 100000.5, 2.5, 0.5, "str", "", "héllo wörld", "a\"b\\c\n",
            True, False, None, [1, 2.5, "a"], {"a": 1}, 9007199254740993, 5e-324, 1.7976931348623157e308, 0x1F600, 128, -1]


def _objs():
    return [{"apiVersion": "v1", "kind": "Pod", "metadata": {"name": "p%d" % i}, "spec": {"x": x}} for i, x in enumerate(OPERANDS)]


def _oracle_messages(constraint):
    oc = OC.Client()
    oc.add_template(TEMPLATE)
    oc.add_constraint(constraint)
    return [sorted(r.msg for r in oc.review(OT.AugmentedUnstructured(OT.Unstructured(o), None, "Original"), OC.AUDIT_EP)) for o in _objs()]


def _constraint(verbs):
    fmts = ["%" + m + v for v in verbs for m in MODS] + MALFORMED
    return {"apiVersion": "constraints.gatekeeper.sh/v1beta1", "kind": "K8sFmtMatrix", "metadata": {"name": "f"}, "spec": {"parameters": {"fmts": fmts}}}, len(fmts)


def test_product_sprintf_equals_the_oracle_on_the_verb_matrix():
    c, n = _constraint("vsdqxXoObcfFeEgGtTUp")
    want = _oracle_messages(c)
    drv = D.Driver(device=0, hostemu=True)
    cl = D.Client(drv)
    cl.AddTemplate(TEMPLATE)
    cl.AddConstraint(c)
    got = cl.ReviewBatch([D.AugmentedUnstructured(D.Unstructured(o), None, "Original") for o in _objs()])
    assert all(len(w) == n for w in want)
    for x, g, w in zip(OPERANDS, got, want):
        g = sorted(r.msg for r in g)
        bad = [(a, b) for a, b in zip(g, w) if a != b]
        assert len(g) == len(w) and not bad, (x, bad[:3])


def test_checker_sprintf_equals_the_oracle_on_its_verbs():
    c, n = _constraint("vsdqtTp")
    want = _oracle_messages(c)
    ck = IndepChecker([TEMPLATE], [c])
    for x, o, w in zip(OPERANDS, _objs(), want):
        g = sorted(ck.messages(json.dumps(o)).get(0, []))
        bad = [(a, b) for a, b in zip(g, w) if a != b]
        assert len(g) == len(w) == n and not bad, (x, bad[:3])


QUOTE_TEMPLATE = {"apiVersion": "templates.gatekeeper.sh/v1", "kind": "ConstraintTemplate", "metadata": {"name": "k8squote"},
                  "spec": {"crd": {"spec": {"names": {"kind": "K8sQuote"}}},
                           "targets": [{"target": "admission.k8s.gatekeeper.sh", "rego": """
package k8squote
violation[{"msg": msg}] {
  x := input.review.object.spec.x
  msg := sprintf("q=%q v=%v arr=%v obj=%v set=%v", [x, x, [x], {"k": x}, {x}])
}
"""}]}}
# control characters, DEL, NBSP and other non-ASCII spaces, format characters (ZWSP, BOM, soft hyphen, word joiner), the replacement
# character, unassigned and private-use code points, combining marks, supplementary planes: strconv.Quote / IsPrint decide per code point
CODE_POINTS = [0x9, 0xA, 0xD, 0x7, 0x7F, 0x0, 0x1B, 0xA0, 0x200B, 0xFFFD, 0x1F600, 0x378, 0x22, 0x5C, 0xE9, 0x2028, 0xFEFF, 0xAD, 0xE000, 0x4E2D, 0x301, 0x85,
               0xE0041, 0x10000, 0x1, 0x1F, 0x27, 0xB, 0xC, 0x8, 0x3000, 0x2003, 0x600, 0x61C, 0xFFF9, 0x10FFFF, 0xD7FF, 0x2060, 0x180E, 0x1D173]


def test_string_quoting_three_ways():
    """%q and the term text of strings inside arrays / objects / sets (ast.String.String() = strconv.Quote): product, oracle, checker"""
    con = {"apiVersion": "constraints.gatekeeper.sh/v1beta1", "kind": "K8sQuote", "metadata": {"name": "q"}, "spec": {}}
    objs = [{"apiVersion": "v1", "kind": "Pod", "metadata": {"name": "p%d" % i}, "spec": {"x": "a" + chr(c) + "z"}} for i, c in enumerate(CODE_POINTS)]
    oc = OC.Client()
    oc.add_template(QUOTE_TEMPLATE)
    oc.add_constraint(con)
    cl = D.Client(D.Driver(device=0, hostemu=True))
    cl.AddTemplate(QUOTE_TEMPLATE)
    cl.AddConstraint(con)
    got = cl.ReviewBatch([D.AugmentedUnstructured(D.Unstructured(o), None, "Original") for o in objs])
    ck = IndepChecker([QUOTE_TEMPLATE], [con])
    for c, o, g in zip(CODE_POINTS, objs, got):
        want = [r.msg for r in oc.review(OT.AugmentedUnstructured(OT.Unstructured(o), None, "Original"), OC.AUDIT_EP)]
        assert len(want) == 1 and [r.msg for r in g] == want == ck.messages(json.dumps(o)).get(0), hex(c)
    assert oc.review(OT.AugmentedUnstructured(OT.Unstructured(objs[7]), None, "Original"), OC.AUDIT_EP)[0].msg.startswith('q="a\\u00a0z" v=a z arr=["a\\u00a0z"]')


MARSHAL_TEMPLATE = {"apiVersion": "templates.gatekeeper.sh/v1", "kind": "ConstraintTemplate", "metadata": {"name": "k8smarshal"},
                    "spec": {"crd": {"spec": {"names": {"kind": "K8sMarshal"}}},
                             "targets": [{"target": "admission.k8s.gatekeeper.sh", "rego": """
package k8smarshal
violation[{"msg": msg}] {
  msg := json.marshal({"k": input.parameters.xs, "n": input.parameters.ns, "s": {"b", "a"}})
}
"""}]}}


def test_json_marshal_writes_strings_as_encoding_json():
    """json.marshal = encoding/json.Marshal(ast.JSON(x)) (topdown/encoding.go): EscapeHTML is on (< > & escaped), U+2028 / U+2029 escaped,
    the short escapes, \\u00NN for other controls, DEL and non-ASCII raw; keys sorted, sets as arrays, numbers as their JSON text"""
    cps = [0x3C, 0x3E, 0x26, 0x2028, 0x2029, 0x7F, 0x1, 0x1F, 0x9, 0xA, 0xD, 0x8, 0xC, 0x22, 0x5C, 0x2F, 0xA0, 0xE9, 0x1F600]
    con = {"apiVersion": "constraints.gatekeeper.sh/v1beta1", "kind": "K8sMarshal", "metadata": {"name": "m"},
           "spec": {"parameters": {"xs": ["a" + chr(c) + "z" for c in cps], "ns": [1.5, 1e21, 2.0, 100000000000000000000, 0.000001, 1e-7, -0.5]}}}
    obj = {"apiVersion": "v1", "kind": "Pod", "metadata": {"name": "p"}}
    oc = OC.Client()
    oc.add_template(MARSHAL_TEMPLATE)
    oc.add_constraint(con)
    cl = D.Client(D.Driver(device=0, hostemu=True))
    cl.AddTemplate(MARSHAL_TEMPLATE)
    cl.AddConstraint(con)
    want = [r.msg for r in oc.review(OT.AugmentedUnstructured(OT.Unstructured(obj), None, "Original"), OC.AUDIT_EP)]
    got = [r.msg for r in cl.ReviewBatch([D.AugmentedUnstructured(D.Unstructured(obj), None, "Original")])[0]]
    text = ('{"k":["a\\\\u003cz","a\\\\u003ez","a\\\\u0026z","a\\\\u2028z","a\\\\u2029z","a\\x7fz","a\\\\u0001z","a\\\\u001fz","a\\\\tz","a\\\\nz","a\\\\rz","a\\\\bz","a\\\\fz",'
            '"a\\\\"z","a\\\\\\\\z","a/z","a\\xa0z","a\\xe9z","a\\U0001f600z"],"n":[1.5,1e+21,2,100000000000000000000,0.000001,1e-7,-0.5],"s":["a","b"]}')
    assert got == want == [text.encode().decode("unicode_escape")], (got, want)
