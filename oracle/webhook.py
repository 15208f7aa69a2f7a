"""ORACLE (test infrastructure only -- never imported by the product path).

CPU restatement of what the validating webhook does with the hot path's results:
  pkg/webhook/policy.go:265-399   processValidationResults (deny / warn message lists; logging, events and export omitted)
  pkg/webhook/policy.go:478-502   validatedEnforcementActions
  pkg/util/enforcement_action.go:61-71  ValidateEnforcementAction (deny, dryrun, warn are valid unscoped actions)
Pinned by pkg/webhook/policy_test.go:1303-1339 and :1395-1536 (tests/test_oracle_match.py).
"""
from __future__ import annotations

VALID_ACTIONS = ("deny", "dryrun", "warn")


def validated_enforcement_actions(result):
    """-> (actions, valid).  policy.go:478-502."""
    if result is None or result.constraint is None:
        return None, False
    if result.enforcement_action == "scoped":
        actions = [a for a in (result.scoped_enforcement_actions or []) if a in VALID_ACTIONS]
        return actions, len(actions) > 0
    if result.enforcement_action not in VALID_ACTIONS:
        return None, False
    return None, True


def process_validation_results(results):
    """-> (deny_msgs, warn_msgs), each "[<constraint name>] <msg>" (policy.go:390,394)."""
    deny, warn = [], []
    for r in results:
        actions, valid = validated_enforcement_actions(r)
        if not valid:
            continue
        if not actions:
            actions = [r.enforcement_action]
        name = ((r.constraint or {}).get("metadata") or {}).get("name", "")
        for a in actions:
            if a == "deny":
                deny.append("[%s] %s" % (name, r.msg))
            if a == "warn":
                warn.append("[%s] %s" % (name, r.msg))
    return deny, warn
