"""String builtins on awkward inputs -- empty strings, number-like text, every kind of Unicode space, format characters, special case
mappings, combining marks, supplementary planes -- product (constant folding and rendering share one builtin table) against the Python
oracle.  Found at the end of round 4: strings.reverse reversed BYTES (invalid UTF-8 out), to_number took "1e"; and in the oracle:
to_number took Inf / NaN / full-width digits."""
import pytest

from gatekeeper_amd import driver as D
from oracle import client as OC
from oracle import target as OT

CALLS = [("trim_space", 'trim_space(s)'), ("upper", 'upper(s)'), ("lower", 'lower(s)'), ("count", 'count(s)'),
         ("trim", 'trim(s, "a ")'), ("trim_left", 'trim_left(s, "a ")'), ("trim_right", 'trim_right(s, "z ")'), ("trim_prefix", 'trim_prefix(s, "a")'),
         ("trim_suffix", 'trim_suffix(s, "z")'), ("split_empty", 'split(s, "")'), ("split_a", 'split(s, "a")'), ("replace_empty", 'replace(s, "", "-")'),
         ("replace", 'replace(s, "a", "bb")'), ("substring", 'substring(s, 1, 2)'), ("substring_end", 'substring(s, 1, -1)'), ("substring_big", 'substring(s, 5, 100)'),
         ("indexof", 'indexof(s, "z")'), ("indexof_empty", 'indexof(s, "")'), ("to_number", 'to_number(s)'), ("contains_empty", 'contains(s, "")'),
         ("startswith", 'startswith(s, "a")'), ("endswith", 'endswith(s, "z")'), ("format_int", 'format_int(count(s) - 3, 2)'), ("concat", 'concat(s, ["x", "y", "z"])'),
         ("reverse", 'strings.reverse(s)'), ("sprintf", 'sprintf("%d|%5v|%-5v|%q", [count(s), s, s, s])'), ("json", 'json.marshal(s)'),
         ("any_prefix", 'strings.any_prefix_match(s, ["a", "b"])'), ("lowerupper", 'lower(upper(s))')]
CODE_POINTS = [0x20, 0x9, 0xA0, 0x2028, 0x3000, 0x85, 0x1C, 0x1F, 0xB, 0xC, 0xFEFF, 0x200B, 0xDF, 0x130, 0x1C5, 0x3A3, 0x3C2, 0x1F600, 0x301, 0x180E, 0x1680, 0x2000,
               0x202F, 0x205F, 0x0]
INPUTS = ["", "a", "az", "aaz z", " a z ", "AbC", "123", "0x10", "1e3", " 1", "+1", "1_000", "Inf", "NaN", "-0", "1.50", ".5", "5.", "1e", "1e+", "--1", "0b11", "1E+2",
          chr(0xFF19), "Infinity", "nan"] + [chr(c) + "a" + chr(c) + "z" + chr(c) for c in CODE_POINTS]


@pytest.mark.parametrize("name,expr", CALLS)
def test_string_builtin_on_awkward_inputs(name, expr):
    kind = "K8sB" + "".join(ch for ch in name.title() if ch.isalnum())
    template = {"apiVersion": "templates.gatekeeper.sh/v1", "kind": "ConstraintTemplate", "metadata": {"name": kind.lower()},
                "spec": {"crd": {"spec": {"names": {"kind": kind}}}, "targets": [{"target": "admission.k8s.gatekeeper.sh", "rego": """
package k
violation[{"msg": msg}] {
  s := input.parameters.xs[i]
  r := %s
  msg := sprintf("%%d: %%v", [i, r])
}
""" % expr}]}}
    con = {"apiVersion": "constraints.gatekeeper.sh/v1beta1", "kind": kind, "metadata": {"name": "c"}, "spec": {"parameters": {"xs": INPUTS}}}
    obj = {"apiVersion": "v1", "kind": "Pod", "metadata": {"name": "p"}}
    oc = OC.Client()
    oc.add_template(template)
    oc.add_constraint(con)
    want = sorted(r.msg for r in oc.review(OT.AugmentedUnstructured(OT.Unstructured(obj), None, "Original"), OC.AUDIT_EP))
    cl = D.Client(D.Driver(device=0, hostemu=True))
    cl.AddTemplate(template)
    cl.AddConstraint(con)
    got = sorted(r.msg for r in cl.ReviewBatch([D.AugmentedUnstructured(D.Unstructured(obj), None, "Original")])[0])
    assert got == want, [(g, w) for g, w in zip(got, want) if g != w][:3]
    if name == "to_number":   # the builtin's own rules: ParseFloat's decimal syntax, no Inf / NaN, ASCII digits, no blanks
        accepted = {int(m.split(":")[0]) for m in want}
        assert accepted == {INPUTS.index(x) for x in ("123", "1e3", "+1", "-0", "1.50", ".5", "5.", "1E+2")}
    if name == "reverse":
        assert "%d: %s" % (INPUTS.index("aaz z"), '"z zaa"'.strip('"')) in want


NUM_CALLS = [("fi10", "format_int(x, 10)"), ("fi16", "format_int(x, 16)"), ("fi2", "format_int(x, 2)"), ("fi8", "format_int(x, 8)"), ("abs", "abs(x)"), ("round", "round(x)"),
             ("mod", "x % 3"), ("div", "x / 2"), ("mulf", "x * 0.1"), ("spf", 'sprintf("%f|%.2f|%8.3f|%e|%g|%.0f", [x, x, x, x, x, x])'), ("json", "json.marshal([x, {\"k\": x}])"),
             ("slice", "array.slice([1, 2, 3, 4], x, 3)"), ("substr", 'substring("abcdef", 1, x)'), ("oget", 'object.get({"1": "a", "k": 2}, x, "dflt")'), ("set", "count({x, 1, 1.0})")]
NUM_INPUTS = [0, 1, -1, 2, 3, 7, -7, 2.5, -2.5, 0.5, -0.5, 1.5, 3.5, 1000.0, 1e-7, 0.1, 2147483648, 1.0, 100.0, -0.0, 1e6, 12345.678, 4294967296, -3, 255, 0.30000000000000004]


@pytest.mark.parametrize("name,expr", NUM_CALLS)
def test_numeric_builtin_on_awkward_inputs(name, expr):
    """Numbers either side of zero, halves, integral floats, 2^31 / 2^32: product against the Python oracle.  Found with it:
    format_int FLOORED a negative fraction (OPA's builtinFormatInt truncates: big.Float.Int), printed 0 for a float beyond
    128 bits, and integer + - * wrapped beyond 128 bits (now continues as a float)."""
    kind = "K8sN" + name.title()
    template = {"apiVersion": "templates.gatekeeper.sh/v1", "kind": "ConstraintTemplate", "metadata": {"name": kind.lower()},
                "spec": {"crd": {"spec": {"names": {"kind": kind}}}, "targets": [{"target": "admission.k8s.gatekeeper.sh", "rego": """
package k
violation[{"msg": msg}] {
  x := input.parameters.xs[i]
  r := %s
  msg := sprintf("%%d: %%v", [i, r])
}
""" % expr}]}}
    xs = NUM_INPUTS + ([1.7976931348623157e308] if name.startswith("fi") else [])
    con = {"apiVersion": "constraints.gatekeeper.sh/v1beta1", "kind": kind, "metadata": {"name": "c"}, "spec": {"parameters": {"xs": xs}}}
    obj = {"apiVersion": "v1", "kind": "Pod", "metadata": {"name": "p"}}
    oc = OC.Client()
    oc.add_template(template)
    oc.add_constraint(con)
    want = sorted(r.msg for r in oc.review(OT.AugmentedUnstructured(OT.Unstructured(obj), None, "Original"), OC.AUDIT_EP))
    cl = D.Client(D.Driver(device=0, hostemu=True))
    cl.AddTemplate(template)
    cl.AddConstraint(con)
    got = sorted(r.msg for r in cl.ReviewBatch([D.AugmentedUnstructured(D.Unstructured(obj), None, "Original")])[0])
    assert got == want, [(g, w) for g, w in zip(got, want) if g != w][:3]
    if name == "fi10":
        assert "%d: -2" % xs.index(-2.5) in want and "%d: 0" % xs.index(-0.5) in want and "%d: 12345" % xs.index(12345.678) in want
        assert any(m.startswith("%d: 17976931348623157081452742373170435679807056752584499659891747680315726078" % (len(xs) - 1)) for m in want)
