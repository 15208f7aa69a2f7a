#!/usr/bin/env python3
"""Generate tests/golden/reference_fixtures.json from the reference checkout (run in the build container only).

The GPU box has no /root/reference, so the policy/test fixtures the reference's own tests use for the hot path
 are bundled here as data: every YAML document is loaded and stored as JSON, keyed by its
path relative to /root/reference.  Go string constants holding YAML fixtures (pkg/gator/fixtures/fixtures.go,
pkg/target/target_integration_test.go:24-44) are extracted with a regex over the Go source.

Hand-transcribed truth tables (Go table tests that cannot be parsed mechanically) live in reference_tables.py.

Usage:  python tests/golden/make_golden.py [/root/reference]
"""
import glob
import json
import os
import re
import sys

import yaml

REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "reference_fixtures.json")

YAML_GLOBS = [
    "pkg/webhook/testdata/psp-all-violations/*/*.yaml",
    "demo/basic/*/*.yaml",
    "demo/agilebank/*/*.yaml",
    "demo/agilebank/*/*/*.yaml",
    "example/*/*.yaml",
    "test/gator/test/fixtures/**/*.yaml",
    "test/gator/test/fixtures/**/*.yml",
    "test/gator/test/fixtures/**/*.json",
    "test/gator/verify/*.yaml",
    "test/gator/bench/*/*.yaml",
    "test/gator/oci-artifacts/*/*.yaml",
    "test/gator/policy/testdata/templates/*/*.yaml",
    "test/bats/tests/templates/*.yaml",
    "test/bats/tests/constraints/*.yaml",
    "test/bats/tests/good/*.yaml",
    "test/bats/tests/bad/*.yaml",
    "pkg/readiness/testdata/*.yaml",
]

GO_CONST_FILES = [
    "pkg/gator/fixtures/fixtures.go",
    "pkg/target/target_integration_test.go",
    "pkg/gator/verify/runner_integer_test.go",   # round 6: TestRunner_Run_Integer (K8sReplicaLimits, three template flavours, 0 / 1 violations)
]


def load_docs(path):
    text = open(path, encoding="utf-8").read()
    try:
        docs = [d for d in yaml.safe_load_all(text) if d is not None]
        json.dumps(docs)
        return {"docs": docs}
    except Exception as e:  # deliberately invalid fixtures are kept as text
        return {"raw": text, "error": str(e).splitlines()[0]}


def go_consts(path):
    src = open(path, encoding="utf-8").read()
    out = {}
    for m in re.finditer(r"^\s*(?:const\s+)?(\w+)\s*=\s*`([^`]*)`", src, re.M):
        name, text = m.group(1), m.group(2)
        try:
            docs = [d for d in yaml.safe_load_all(text) if d is not None]
            json.dumps(docs)
            out[name] = {"docs": docs}
        except Exception as e:
            out[name] = {"raw": text, "error": str(e).splitlines()[0]}
    return out


def validate_constraint_cases(path):
    """pkg/target/target_test.go:42-399 TestValidateConstraint: (Name, Constraint JSON, ErrorExpected) rows."""
    text = open(path, encoding="utf-8").read()
    body = text[text.index("func TestValidateConstraint"):text.index("func TestProcessData")]
    rows = []
    for m in re.finditer(r'Name:\s*"([^"]*)",\s*Constraint:\s*`([^`]*)`,\s*ErrorExpected:\s*(true|false)', body):
        rows.append({"name": m.group(1), "constraint": json.loads(m.group(2)), "error_expected": m.group(3) == "true"})
    return rows


def main():
    bundle = {"yaml": {}, "go_consts": {}}
    for g in YAML_GLOBS:
        for p in sorted(glob.glob(os.path.join(REF, g), recursive=True)):
            bundle["yaml"][os.path.relpath(p, REF)] = load_docs(p)
    for f in GO_CONST_FILES:
        bundle["go_consts"][f] = go_consts(os.path.join(REF, f))
    bundle["validate_constraint_cases"] = validate_constraint_cases(os.path.join(REF, "pkg/target/target_test.go"))
    with open(OUT, "w", encoding="utf-8") as fh:
        json.dump(bundle, fh, indent=1, sort_keys=True)
    # the policy templates bench.py / smoke() load at run time (policy INPUTS, not oracle code) live with the package, so
    # that the product bench does not read a test directory
    keep = ("demo/agilebank/templates/k8srequiredlabels_template.yaml", "demo/agilebank/templates/k8sallowedrepos_template.yaml",
            "demo/agilebank/templates/k8scontainterlimits_template.yaml", "demo/agilebank/templates/k8srequiredprobes_template.yaml",
            "demo/agilebank/remediation/k8sbannedimagetags_template.yaml")
    pol = {p: bundle["yaml"][p] for p in sorted(bundle["yaml"])
           if p.startswith("pkg/webhook/testdata/psp-all-violations/psp-templates/") or p in keep}
    pol_out = os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "gatekeeper_amd", "data", "policy_templates.json")
    os.makedirs(os.path.dirname(pol_out), exist_ok=True)
    with open(pol_out, "w", encoding="utf-8") as fh:
        json.dump({"yaml": pol}, fh, indent=1, sort_keys=True)
    print("wrote %s: %d policy templates" % (pol_out, len(pol)))
    print("wrote %s: %d yaml files, %d go const files, %d ValidateConstraint rows" % (
        OUT, len(bundle["yaml"]), len(bundle["go_consts"]), len(bundle["validate_constraint_cases"])))


if __name__ == "__main__":
    main()
