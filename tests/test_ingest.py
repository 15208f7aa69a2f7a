"""Fast ingest (one-pass JSON text -> rows, Flattener::add_json) produces exactly the
table the general path (parse_json -> HandleReview normalisation -> Flattener::add) produces: content digests of the
two tables agree, review by review shape, on synthetic objects and on adversarial JSON."""
import json
import os

import pytest

from gatekeeper_amd import _lib as L
from gatekeeper_amd import driver as D
from gatekeeper_amd import synth


def _digest(engine, rins, slow, threads=None):
    os.environ["GK_TABLE_DIGEST"] = "1"
    if slow:
        os.environ["GK_SLOW_INGEST"] = "1"
    if threads:
        os.environ["GK_HOST_THREADS"] = str(threads)
    try:
        t = engine.create_table(rins, keep_docs=False)
        st = t.stats()
        t.free()
        return st
    finally:
        for k in ("GK_TABLE_DIGEST", "GK_SLOW_INGEST", "GK_HOST_THREADS"):
            os.environ.pop(k, None)


def _raw(text, ns=None, nsobj=None, op="", source="Original"):
    r = D.ReviewIn(L.GK_REVIEW_OBJECT, text.encode() if isinstance(text, str) else text, ns, nsobj, source, op)
    return r


@pytest.fixture(params=["index", "bytes", "index-pruned", "bytes-pruned"])
def tokens(request, monkeypatch):
    """both scanners of the fast path -- the structural index (AVX-512 hosts) and the byte-at-a-time one every other host runs --
    each on full tables and on PRUNED ones (GK_FORCE_PRUNE: every table of the test; the general path prunes the same way, so the
    digests must still agree -- a non-string label value under a key nothing reads is still a bad label map)"""
    if request.param.startswith("bytes"):
        monkeypatch.setenv("GK_NO_INDEX", "1")
    if request.param.endswith("pruned"):
        monkeypatch.setenv("GK_FORCE_PRUNE", "1")
    return request.param


@pytest.mark.parametrize("mixed", [False, True])
def test_fast_ingest_equals_general_path_on_synthetic(mixed, tokens):
    eng = D.Engine(hostemu=True)
    nss = synth.gen_namespaces()
    objs = synth.gen_objects(700, seed=77, mixed=mixed)
    rins = [D.to_review_in(D.AugmentedUnstructured(D.Unstructured(o), synth.namespace_for(o, nss), "Original")) for o in objs]
    fast, slow = _digest(eng, rins, False), _digest(eng, rins, True)
    assert fast["fast_reviews"] == len(rins) and slow["fast_reviews"] == 0
    assert fast["digest"] == slow["digest"] != 0 and fast["n_rows"] == slow["n_rows"] and fast["heap_bytes"] == slow["heap_bytes"]
    assert _digest(eng, rins, False, threads=3)["digest"] == fast["digest"] == _digest(eng, rins, False, threads=1)["digest"]


def test_fast_ingest_adversarial_documents(tokens):
    """escapes, surrogate pairs, numbers at the int64 / float boundaries, empty containers, deep nesting, arrays of arrays,
    > 255 elements, DELETE, namespaceObject (incl. null), wrong-typed metadata, missing kind, whitespace, nsCache fallback"""
    eng = D.Engine(hostemu=True)
    eng.put_data(["cluster", "v1", "Namespace", "cached"], {"apiVersion": "v1", "kind": "Namespace", "metadata": {"name": "cached", "labels": {"env": "x"}}})
    ns = {"apiVersion": "v1", "kind": "Namespace", "metadata": {"name": "prod", "labels": {"env": "prod", "n": 5}}}
    docs = [
        '{"apiVersion":"v1","kind":"Pod","metadata":{"name":"a\\u00e9\\n\\"q\\"","namespace":"cached","labels":{"k":"v\\/x","long-key-with-many-bytes":"\\ud83d\\ude00 long value over twelve"}}}',
        '{ "apiVersion" : "apps/v1" ,\n "kind":"Deployment", "metadata": {"name":"d","generateName":"gen-","labels":{"a":1}} , "spec":{"replicas":3,"x":[[1,2],[3,[4,5]],[]],"e":{},"f":[]}}',
        '{"apiVersion":"v1","kind":"Pod","metadata":{"name":"n","labels":"notamap"},"nums":[0,-0,1.0,1.5,1e3,1E-2,-7,9223372036854775807,9223372036854775808,-9223372036854775808,-9223372036854775809,123456789012345678901234567890,0.1,2e400,true,false,null]}',
        '{"apiVersion":"a/b/c","kind":"","metadata":[1,2]}',
        '{"apiVersion":"/","metadata":{"name":5,"namespace":null}}',
        json.dumps({"apiVersion": "v1", "kind": "Pod", "metadata": {"name": "many"}, "spec": {"containers": [{"name": "c%d" % i, "ports": [{"p": i}]} for i in range(300)]}}),
        json.dumps({"apiVersion": "v1", "kind": "Namespace", "metadata": {"name": "self", "labels": {"a": "b"}}}),
        json.dumps({"apiVersion": "v1", "kind": "Pod", "metadata": {"name": "deep"}, "d": {"a": {"b": {"c": {"d": {"e": [[[["x"]]]]}}}}}}),
        '{"apiVersion":"v1","kind":"Pod","metadata":{"name":"k\\u0000z","namespace":"ns\\u0001"},"s":"' + "y" * 5000 + '"}',
    ]
    rins = []
    for i, d in enumerate(docs):
        rins.append(_raw(d, ns if i % 2 == 0 else None, None, "", ["Original", "Generated", "", "All", "bogus"][i % 5]))
        rins.append(_raw(d, None, {"metadata": {"name": "nsobj", "labels": {"z": "1"}}} if i % 3 else None, "DELETE" if i % 2 else "UPDATE"))
    rins.append(_raw(docs[0], None, None))
    rins[-1].ns_object = b"null"
    fast, slow = _digest(eng, rins, False), _digest(eng, rins, True)
    assert fast["digest"] == slow["digest"] != 0 and fast["n_rows"] == slow["n_rows"]
    assert fast["fast_reviews"] == len(rins)
    # documents the fast path must DECLINE (and the general path then handles identically): duplicate keys, malformed
    # JSON, non-object documents, nesting beyond its depth -- statuses and tables agree
    odd = ['{"apiVersion":"v1","kind":"Pod","kind":"Service","metadata":{"name":"dup","name":"dup2"}}', '{"apiVersion":"v1",', '[1,2,3]', '"str"',
           '{"a":' * 120 + '1' + '}' * 120, '{"apiVersion":"v1","kind":"Pod"} trailing', '{"apiVersion":"v1","kind":"P\\x"}', '']
    rins2 = [_raw(d) for d in odd] + [_raw(docs[1])]
    os.environ["GK_TABLE_DIGEST"] = "1"
    try:
        tf = eng.create_table(rins2, keep_docs=False)
        os.environ["GK_SLOW_INGEST"] = "1"
        ts = eng.create_table(rins2, keep_docs=False)
    finally:
        os.environ.pop("GK_TABLE_DIGEST", None); os.environ.pop("GK_SLOW_INGEST", None)
    assert list(tf.statuses) == list(ts.statuses) and tf.statuses[0] == L.GK_OK and tf.statuses[1] == L.GK_ERR_REVIEW
    sf, ss = tf.stats(), ts.stats()
    assert sf["digest"] == ss["digest"] and sf["fast_reviews"] == 1


def _req(text, ns=None, nsobj=None, source="Original"):
    return D.ReviewIn(L.GK_REVIEW_ADMISSION_REQUEST, text.encode() if isinstance(text, str) else text, ns, nsobj, source, "")


def test_fast_ingest_of_admission_requests(tokens):
    """AdmissionRequest documents (the webhook's wire shape, pkg/target/review.go:16-21) through the one-pass parser:
    CREATE / UPDATE / DELETE, missing and wrong-typed envelope members, unknown members (dropped after a syntax check),
    requestKind / dryRun / options / userInfo subtrees, namespaceObject, nsCache fallback on the REQUEST namespace --
    digests equal the general path's; what it cannot take is declined and handled identically by the general path."""
    eng = D.Engine(hostemu=True)
    eng.put_data(["cluster", "v1", "Namespace", "cached"], {"apiVersion": "v1", "kind": "Namespace", "metadata": {"name": "cached", "labels": {"env": "x"}}})
    nss = synth.gen_namespaces()
    pods = synth.gen_objects(120, seed=91, mixed=True)
    rins = []
    for i, o in enumerate(pods):
        api = o.get("apiVersion", "v1")
        g, _, ver = api.rpartition("/")
        req = {"uid": "u-%d" % i, "kind": {"group": g, "version": ver, "kind": o["kind"]}, "resource": {"group": g, "version": ver, "resource": o["kind"].lower() + "s"},
               "name": o["metadata"]["name"], "operation": ["CREATE", "UPDATE", "DELETE", "CONNECT"][i % 4],
               "userInfo": {"username": "alice", "groups": ["system:authenticated", "g%d" % (i % 3)], "extra": {"k": ["v"]}}}
        if o["metadata"].get("namespace"):
            req["namespace"] = o["metadata"]["namespace"] if i % 5 else "cached"
        if i % 4 == 2:
            req["oldObject"] = o
            if i % 8 == 2:
                req["object"] = None
        elif i % 4 == 1:
            old = json.loads(json.dumps(o)); old["metadata"]["labels"] = {"was": "old"}
            req["object"], req["oldObject"] = o, old
        else:
            req["object"] = o
        if i % 3 == 0:
            req.update({"requestKind": {"group": g, "version": ver, "kind": o["kind"]}, "requestResource": None, "dryRun": bool(i % 2), "options": {"kind": "CreateOptions", "fieldManager": "kubectl"}})
        if i % 7 == 0:
            req.update({"subResource": "status", "requestSubResource": "", "somethingUnknown": {"deep": [1, {"x": "y"}]}, "another": "\\u00e9"})
        text = json.dumps(req) if i % 2 else json.dumps(req, indent=1)
        ns = synth.namespace_for(o, nss) if i % 3 else None
        rins.append(_req(text, ns, {"metadata": {"name": "nsobj"}} if i % 4 == 0 else None, ["Original", "Generated", ""][i % 3]))
    weird = [
        '{}', '{"operation":"CREATE"}', '{"uid":5,"kind":"notanobject","resource":null,"operation":7,"userInfo":"x","object":[1],"oldObject":"s","options":3,"name":{},"namespace":["a"]}',
        '{"kind":{"group":1,"version":null,"kind":"Pod","extra":[{}]},"object":{"metadata":{"name":"nokind"}},"operation":"UPDATE","oldObject":{"kind":"Pod","apiVersion":"v1","metadata":{"labels":{"a":1}}}}',
        '{"object":{"apiVersion":"v1","kind":"Namespace","metadata":{"name":"itself"}},"namespace":"","name":""}',
    ]
    rins += [_req(w) for w in weird]
    fast, slow = _digest(eng, rins, False), _digest(eng, rins, True)
    assert fast["digest"] == slow["digest"] != 0 and fast["n_rows"] == slow["n_rows"] and fast["heap_bytes"] == slow["heap_bytes"]
    assert fast["fast_reviews"] == len(rins) and slow["fast_reviews"] == 0
    # declined (general path words the outcome): DELETE without oldObject, duplicate / escaped envelope members, malformed
    # unknown members, escapes in envelope strings, non-object documents
    odd = ['{"operation":"DELETE","object":{"kind":"Pod"}}', '{"uid":"a","uid":"b"}', '{"na\\u006de":"x","object":{"kind":"Pod"}}', '{"object":{"kind":"Pod"},"junk":tru}',
           '{"uid":"a\\nb","object":{"kind":"Pod","apiVersion":"v1"}}', '[{"object":{}}]', '{"object":{"kind":"Pod"}} x', '{"object":{"kind":"Pod","kind":"Dup"}}']
    rins2 = [_req(d) for d in odd] + [rins[0]]
    os.environ["GK_TABLE_DIGEST"] = "1"
    try:
        tf = eng.create_table(rins2, keep_docs=False)
        os.environ["GK_SLOW_INGEST"] = "1"
        ts = eng.create_table(rins2, keep_docs=False)
    finally:
        os.environ.pop("GK_TABLE_DIGEST", None); os.environ.pop("GK_SLOW_INGEST", None)
    assert list(tf.statuses) == list(ts.statuses) and tf.statuses[0] == L.GK_ERR_REVIEW and tf.statuses[-1] == L.GK_OK
    sf, ss = tf.stats(), ts.stats()
    assert sf["digest"] == ss["digest"] and sf["fast_reviews"] == 1


@pytest.mark.parametrize("policy", ["psp", "corpus"])
def test_pruned_ingest_with_a_policy_loaded_equals_the_general_path(policy, fixtures, monkeypatch):
    """With a policy set loaded the read set is not empty: a pruned table keeps some rows, walks past the rest, and objects whose
    own rows are not kept look only for the members some pattern names.  Synthetic objects plus broken ones (non-string label values
    under keys nothing reads, wrong-typed metadata, duplicate names inside a walked-past subtree): the structural-index scanner, the
    byte scanner and the general path build the same table."""
    drv = D.Driver(device=0, hostemu=True)
    client = D.Client(drv)
    if policy == "psp":
        ts, cs = synth.psp_templates(fixtures), synth.audit_constraints()
    else:
        ts, cs = synth.corpus(fixtures)
        ts = ts[::5]
        kinds = {t["spec"]["crd"]["spec"]["names"]["kind"] for t in ts}
        cs = [c for c in cs if c["kind"] in kinds]
    for t in ts:
        client.AddTemplate(t)
    for c in cs:
        client.AddConstraint(c)
    nss = synth.gen_namespaces()
    objs = synth.gen_objects(300, seed=91, mixed=True)
    rins = [D.to_review_in(D.AugmentedUnstructured(D.Unstructured(o), synth.namespace_for(o, nss), "Original")) for o in objs]
    odd = [
        '{"apiVersion":"v1","kind":"Pod","metadata":{"name":"l1","namespace":"dev-00","labels":{"zz-unread":5,"app":"x"}},"spec":{"containers":[{"name":"c","image":"nginx"}]}}',
        '{"apiVersion":"v1","kind":"Pod","metadata":{"name":"l2","labels":{"zz-unread":{"a":1}}},"spec":{"containers":[{"name":"c","image":"nginx","env":[{"name":"A","name":"B","value":1}]}]}}',
        '{"apiVersion":"v1","kind":"Pod","metadata":{"name":"l3","annotations":{"k":"v\\"q\\\\"}},"spec":{"containers":[{"name":"c\\u00e9","image":"gcr.io/a/b:latest","securityContext":{"privileged":true,"zz":[1,[2,{"y":null}]]},"ports":[{"hostPort":80,"zz":"q"}]}],"volumes":[{"name":"v","hostPath":{"path":"/etc"}}],"hostNetwork":true}}',
        '{"apiVersion":"apps/v1","kind":"Deployment","metadata":{"name":"d","labels":{"a":"b"}},"spec":{"template":{"spec":{"containers":[{"name":"x","image":"busybox","resources":{"limits":{"cpu":"2","memory":"1Gi","zz":1e3}}}]}}}}',
    ]
    rins += [_raw(d, None, None) for d in odd]
    monkeypatch.setenv("GK_FORCE_PRUNE", "1")
    slow = _digest(drv.engine, rins, True)
    fast = _digest(drv.engine, rins, False)
    monkeypatch.setenv("GK_NO_INDEX", "1")
    fast_bytes = _digest(drv.engine, rins, False)
    assert fast["digest"] == slow["digest"] == fast_bytes["digest"] != 0 and fast["n_rows"] == slow["n_rows"] == fast_bytes["n_rows"] > 1000
    assert fast["fast_reviews"] == len(rins) == fast_bytes["fast_reviews"]


def test_structural_index_at_every_block_alignment():
    """Stage 1 of the structural index works on 64-byte blocks with carries between them (is the first byte escaped? inside a
    string? the continuation of a scalar?).  Documents full of the carried states -- runs of backslashes of every parity before
    quotes, escaped quotes, strings and numbers longer than a block, unicode escapes, scalars glued to brackets, tabs and CRs --
    shifted through every alignment by leading white space: the index scanner, the byte scanner and the general path agree."""
    eng = D.Engine(hostemu=True)
    runs = "".join('"b%d":"x%s\\"y","c%d":"%s",' % (k, "\\\\" * k, k, "\\\\" * k) for k in range(0, 9))
    body = ('{"apiVersion":"v1","kind":"Pod","metadata":{"name":"n","labels":{"a":"' + "L" * 70 + '","e":"\\u00e9\\ud83d\\ude00\\n\\t"}},'
            '"spec":{' + runs + '"big":123456789012345678,"neg":-0.5e-3,"t":true,"f":false,"z":null,\t"arr":[1,[2,[3,[]]],{"k":{}}],\r\n'
            '"s":"' + 'q\\"' * 40 + '","containers":[{"name":"c","image":"' + "i" * 130 + '","args":["--x=\\\\","\\\\\\"","end"]}]}}')
    json.loads(body)   # (the test's own document must be JSON)
    rins = [_raw(" " * k + body + ("\n" * (k % 3)), None, None) for k in range(0, 140)]
    slow = _digest(eng, rins, True)
    fast = _digest(eng, rins, False)
    os.environ["GK_NO_INDEX"] = "1"
    try:
        fast_bytes = _digest(eng, rins, False)
    finally:
        os.environ.pop("GK_NO_INDEX", None)
    assert fast["fast_reviews"] == len(rins) == fast_bytes["fast_reviews"]
    assert fast["digest"] == slow["digest"] == fast_bytes["digest"] != 0 and fast["n_rows"] == slow["n_rows"] == fast_bytes["n_rows"]
    # ... and text that is NOT JSON at the places the carries decide: an unterminated escape at the end, a quote escaped by an odd run
    # that closes nothing, a scalar glued to a string -- declined by both scanners, rejected by the general path, at every alignment
    bad = ['{"a":"x\\\\\\"}', '{"a":"x\\', '{"a":"x"y}', '{"a":tru e}', '{"a":1 2}', '{"a":"b"}}', '{"a" "b"}']
    for doc in bad:
        rb = [_raw(" " * k + doc, None, None) for k in (0, 1, 57, 58, 59, 60, 61, 62, 63, 64, 65)]
        t1 = eng.create_table(rb, keep_docs=False)
        os.environ["GK_NO_INDEX"] = "1"
        try:
            t2 = eng.create_table(rb, keep_docs=False)
        finally:
            os.environ.pop("GK_NO_INDEX", None)
        assert list(t1.statuses) == list(t2.statuses) == [L.GK_ERR_REVIEW] * len(rb), doc
        t1.free(); t2.free()


def test_high_cardinality_batch_takes_the_same_rows_on_every_ingest_path(fixtures, monkeypatch):
    """include/gksynth.h `mixed | 16`: every container's image tag and name unique in the stream -- the dictionary expressions of the
    200-template corpus (80 on containers[].image: repos, banned tags through split components) are then evaluated per VALUE by the
    compiled string program (csrc/dexpr.hpp DxStrProg), never out of a memo.  One-pass ingest == general path (which walks the parsed
    document and evaluates the same expressions) row for row, and the bitmaps equal those of the default batch wherever the policies do
    not look at tags or names (the objects are the default ones with longer strings in those two places)."""
    import ctypes as C
    import json
    from parity_util import make_client
    monkeypatch.setenv("GK_TABLE_DIGEST", "1")
    c = make_client("hostemu")
    templates, constraints = synth.corpus(fixtures, 200)
    for t in templates:
        c.AddTemplate(t)
    for k in constraints:
        c.AddConstraint(k)
    eng = c.driver.engine
    n = 1200
    nss = synth.gen_namespaces()
    hc = synth.NativeBatch(eng.lib, n, seed=synth.SEED, mixed=True, namespaces=nss, high_cardinality=True)
    base = synth.NativeBatch(eng.lib, n, seed=synth.SEED, mixed=True, namespaces=nss)
    images, names = set(), set()
    n_containers = 0
    for i in range(n):
        o = json.loads(hc.json_text(i))
        for cont in ((o.get("spec") or {}).get("containers") or []) if o.get("kind") == "Pod" else []:
            images.add(cont["image"]); names.add(cont["name"]); n_containers += 1
    assert n_containers > 1000 and len(images) == n_containers and len(names) == n_containers
    digests = {}
    for mode in ("index", "text", "general"):
        for k in ("GK_NO_INDEX", "GK_SLOW_INGEST"):
            monkeypatch.delenv(k, raising=False)
        if mode == "text":
            monkeypatch.setenv("GK_NO_INDEX", "1")
        if mode == "general":
            monkeypatch.setenv("GK_SLOW_INGEST", "1")
        t = eng.create_table_native(hc.reviews, n, resident=True)
        st = t.stats()
        digests[mode] = (st["digest"], st["n_rows"])
        if mode == "index":
            ev_hc = t.eval()
        t.free()
    for k in ("GK_NO_INDEX", "GK_SLOW_INGEST"):
        monkeypatch.delenv(k, raising=False)
    assert digests["index"] == digests["text"] == digests["general"] and digests["index"][0] != 0
    tb = eng.create_table_native(base.reviews, n, resident=True)
    ev_b = tb.eval()
    tb.free()
    kinds = [k["kind"] for k in constraints]
    same = [i for i, cid in enumerate(ev_b.constraint_ids) if not any(w in c.constraints[next(key for key in c.constraints if c.driver.constraint_id(c.constraints[key]) == int(cid))]["kind"]
                                                                          for w in ("Image", "Repo", "Tag", "Digest"))]
    assert len(same) > 100 and all((ev_b.viol[i] == ev_hc.viol[i]).all() for i in same)
    assert int(ev_hc.counts.sum()) > 1000 and kinds
