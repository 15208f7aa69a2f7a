from gatekeeper_amd import synth, driver as D
c = D.Client()
for t in synth.psp_templates(): c.AddTemplate(t)
for k in synth.psp_constraints(): c.AddConstraint(k)
nss = synth.gen_namespaces()
for ns in nss.values(): c.AddData(ns)
import sys
objs = synth.gen_objects(int(sys.argv[1]) if len(sys.argv) > 1 else 64, 1)
r = c.ReviewBatch([D.AugmentedUnstructured(D.Unstructured(o), synth.namespace_for(o, nss), "Original") for o in objs], D.AUDIT_EP)
