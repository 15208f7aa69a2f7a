"""The native synthetic-workload generator (csrc/synth.cpp, include/gksynth.h) produces exactly the objects of
gatekeeper_amd/synth.py -- bench.py builds its 1M-object audit set with the former and checks a sample with the oracle on
objects from the latter, so the two must be the same stream."""
import json

import pytest

from gatekeeper_amd import _lib as L
from gatekeeper_amd import synth


@pytest.mark.parametrize("mixed", [False, True])
@pytest.mark.parametrize("seed,start,n", [(synth.SEED, 0, 1500), (7, 123456, 300), (synth.SEED + 3, 999000, 300)])
def test_native_generator_matches_python(mixed, seed, start, n):
    lib = L.load(hostemu=True)   # same host code as the product library; loadable without a GPU
    nss = synth.gen_namespaces()
    b = synth.NativeBatch(lib, n, seed=seed, mixed=mixed, start=start, namespaces=nss)
    want = synth.gen_objects(n, seed=seed, mixed=mixed, start=start)
    kinds = set()
    for i, o in enumerate(want):
        assert json.loads(b.json_text(i)) == o, i
        ns = synth.namespace_for(o, nss)
        got_ns = b.namespace_text(i)
        assert (json.loads(got_ns) if got_ns else None) == ns, i
        kinds.add(o["kind"])
    assert b.json_bytes == sum(len(b.json_text(i)) for i in range(n))
    assert kinds == ({"Pod", "Deployment", "Namespace", "Service", "ConfigMap"} if mixed and n >= 1000 else kinds)
    b.free()


def test_native_admission_requests_match_python():
    lib = L.load(hostemu=True)
    nss = synth.gen_namespaces()
    n, start = 200, 5000
    b = synth.NativeBatch(lib, n, seed=9, start=start, namespaces=nss, requests=True)
    for i, o in enumerate(synth.gen_objects(n, seed=9, start=start)):
        assert json.loads(b.json_text(i)) == synth.admission_request_for(o, start + i), i
        assert b.reviews[i].kind == L.GK_REVIEW_ADMISSION_REQUEST
    b.free()
