"""Policy patterns of the public gatekeeper-library, written out here as sixteen templates of the common shapes (allowed
repos, disallowed tags, image digests, replica ranges, required annotations / resources, PSP capabilities / host
namespaces / host ports / read-only root fs / volume types / users, NodePort, wildcard ingress, external IPs, https-only
ingress): comprehension + any / all, set difference, helper rules with several bodies, function rules matched on
constants, object.get, partial-set helper rules, regular expressions, the string builtins startswith / endswith /
contains / concat / lower / sprintf.  The library itself is not part of /root/reference, so these are NOT reference-held
vectors: the test widens the product-vs-oracle comparison (rendered results AND raw device bitmaps, parity_util) to the
template shapes users actually load, all constraints at once in ONE plan.

Also pins a parser rule both sides got wrong: `contains` is a keyword only in a rule head (`violation contains x if`);
anywhere else it is the builtin contains(s, sub) (OPA's parser: "contains anywhere BUT in rule heads gets no special
treatment") -- used as a statement, under `not`, in an assignment and inside a helper rule below."""
import pytest

from gatekeeper_amd import driver as D
from parity_util import BACKENDS, assert_parity, load_both


def tmpl(kind, rego):
    return {"apiVersion": "templates.gatekeeper.sh/v1", "kind": "ConstraintTemplate", "metadata": {"name": kind.lower()},
            "spec": {"crd": {"spec": {"names": {"kind": kind}}}, "targets": [{"target": "admission.k8s.gatekeeper.sh", "rego": rego}]}}


T = {}
T["K8sAllowedRepos"] = ('''package k8sallowedrepos
violation[{"msg": msg}] {
  container := input.review.object.spec.containers[_]
  satisfied := [good | repo = input.parameters.repos[_] ; good = startswith(container.image, repo)]
  not any(satisfied)
  msg := sprintf("container <%v> has an invalid image repo <%v>, allowed repos are %v", [container.name, container.image, input.parameters.repos])
}
violation[{"msg": msg}] {
  container := input.review.object.spec.initContainers[_]
  satisfied := [good | repo = input.parameters.repos[_] ; good = startswith(container.image, repo)]
  not any(satisfied)
  msg := sprintf("initContainer <%v> has an invalid image repo <%v>, allowed repos are %v", [container.name, container.image, input.parameters.repos])
}
''', {"repos": ["gcr.io/good/", "docker.io/library/"]})
T["K8sBlockNodePort"] = ('''package k8sblocknodeport
violation[{"msg": msg}] {
  input.review.kind.kind == "Service"
  input.review.object.spec.type == "NodePort"
  msg := "User is not allowed to create service of type NodePort"
}
''', {})
T["K8sBlockWildcardIngress"] = ('''package K8sBlockWildcardIngress
contains_wildcard(hostname) = true {
  hostname == ""
}
contains_wildcard(hostname) = true {
  contains(hostname, "*")
}
violation[{"msg": msg}] {
  input.review.kind.kind == "Ingress"
  hostnames := {object.get(rule, "host", "") | rule := input.review.object.spec.rules[_]}
  contains_wildcard(hostnames[_])
  msg := sprintf("Hostname '%v' is not allowed since it counts as a wildcard, which can be used to intercept traffic from other applications.", [hostnames[_]])
}
''', {})
T["K8sDisallowedTags"] = ('''package k8sdisallowedtags
violation[{"msg": msg}] {
  container := input_containers[_]
  tags := [forbid | tag = input.parameters.tags[_] ; forbid = endswith(container.image, concat(":", ["", tag]))]
  any(tags)
  msg := sprintf("container <%v> uses a disallowed tag <%v>; disallowed tags are %v", [container.name, container.image, input.parameters.tags])
}
violation[{"msg": msg}] {
  container := input_containers[_]
  tag := [contains(container.image, ":")]
  not all(tag)
  msg := sprintf("container <%v> didn't specify an image tag <%v>", [container.name, container.image])
}
input_containers[c] {
  c := input.review.object.spec.containers[_]
}
input_containers[c] {
  c := input.review.object.spec.initContainers[_]
}
''', {"tags": ["latest"]})
T["K8sImageDigests"] = ('''package k8simagedigests
violation[{"msg": msg}] {
  container := input.review.object.spec.containers[_]
  satisfied := [re_match("@[a-z0-9]+([+._-][a-z0-9]+)*:[a-zA-Z0-9=_-]+", container.image)]
  not all(satisfied)
  msg := sprintf("container <%v> uses an image without a digest <%v>", [container.name, container.image])
}
''', {})
T["K8sReplicaLimits"] = ('''package k8sreplicalimits
deployment_name = input.review.object.metadata.name
violation[{"msg": msg}] {
  spec := input.review.object.spec
  not input_replica_limit(spec)
  msg := sprintf("The provided number of replicas is not allowed for deployment: %v. Allowed ranges: %v", [deployment_name, input.parameters])
}
input_replica_limit(spec) {
  provided := input.review.object.spec.replicas
  count(input.parameters.ranges) > 0
  range := input.parameters.ranges[_]
  value_within_range(range, provided)
}
value_within_range(range, value) {
  range.min_replicas <= value
  range.max_replicas >= value
}
''', {"ranges": [{"min_replicas": 2, "max_replicas": 5}]})
T["K8sRequiredAnnotations"] = ('''package k8srequiredannotations
violation[{"msg": msg, "details": {"missing_annotations": missing}}] {
  provided := {annotation | input.review.object.metadata.annotations[annotation]}
  required := {annotation | annotation := input.parameters.annotations[_].key}
  missing := required - provided
  count(missing) > 0
  msg := sprintf("you must provide annotation(s): %v", [missing])
}
violation[{"msg": msg}] {
  value := input.review.object.metadata.annotations[key]
  expected := input.parameters.annotations[_]
  expected.key == key
  expected.allowedRegex != ""
  not re_match(expected.allowedRegex, value)
  msg := sprintf("Annotation <%v: %v> does not satisfy allowed regex: %v", [key, value, expected.allowedRegex])
}
''', {"annotations": [{"key": "owner", "allowedRegex": "^[a-z]+$"}]})
T["K8sPSPCapabilities"] = ('''package capabilities
violation[{"msg": msg}] {
  container := input.review.object.spec.containers[_]
  has_disallowed_capabilities(container)
  msg := sprintf("container <%v> has a disallowed capability. Allowed capabilities are %v", [container.name, get_default(input.parameters, "allowedCapabilities", "NONE")])
}
violation[{"msg": msg}] {
  container := input.review.object.spec.containers[_]
  missing_drop_capabilities(container)
  msg := sprintf("container <%v> is not dropping all required capabilities. Container must drop all of %v or \\"ALL\\"", [container.name, input.parameters.requiredDropCapabilities])
}
has_disallowed_capabilities(container) {
  allowed := {c | c := lower(input.parameters.allowedCapabilities[_])}
  not allowed["*"]
  capabilities := {c | c := lower(container.securityContext.capabilities.add[_])}
  count(capabilities - allowed) > 0
}
missing_drop_capabilities(container) {
  must_drop := {c | c := lower(input.parameters.requiredDropCapabilities[_])}
  all := {"all"}
  dropped := {c | c := lower(container.securityContext.capabilities.drop[_])}
  count(must_drop - dropped) > 0
  count(all - dropped) > 0
}
get_default(obj, param, _default) = out {
  out = obj[param]
}
get_default(obj, param, _default) = out {
  not obj[param]
  not obj[param] == false
  out = _default
}
''', {"allowedCapabilities": ["NET_BIND_SERVICE"], "requiredDropCapabilities": ["NET_RAW"]})
T["K8sPSPHostNamespace"] = ('''package k8spsphostnamespace
violation[{"msg": msg, "details": {}}] {
  input_share_hostnamespace(input.review.object)
  msg := sprintf("Sharing the host namespace is not allowed: %v", [input.review.object.metadata.name])
}
input_share_hostnamespace(o) {
  o.spec.hostPID
}
input_share_hostnamespace(o) {
  o.spec.hostIPC
}
''', {})
T["K8sPSPHostNetworkingPorts"] = ('''package k8spsphostnetworkingports
violation[{"msg": msg, "details": {}}] {
  input_share_hostnetwork(input.review.object)
  msg := sprintf("The specified hostNetwork and hostPort are not allowed, pod: %v. Allowed values: %v", [input.review.object.metadata.name, input.parameters])
}
input_share_hostnetwork(o) {
  not input.parameters.hostNetwork
  o.spec.hostNetwork
}
input_share_hostnetwork(o) {
  hostPort := input_containers[_].ports[_].hostPort
  hostPort < input.parameters.min
}
input_share_hostnetwork(o) {
  hostPort := input_containers[_].ports[_].hostPort
  hostPort > input.parameters.max
}
input_containers[c] {
  c := input.review.object.spec.containers[_]
}
input_containers[c] {
  c := input.review.object.spec.initContainers[_]
}
''', {"hostNetwork": False, "min": 80, "max": 9000})
T["K8sPSPReadOnlyRootFilesystem"] = ('''package k8spspreadonlyrootfilesystem
violation[{"msg": msg, "details": {}}] {
  c := input_containers[_]
  input_read_only_root_fs(c)
  msg := sprintf("only read-only root filesystem container is allowed: %v", [c.name])
}
input_read_only_root_fs(c) {
  not has_field(c, "securityContext")
}
input_read_only_root_fs(c) {
  not c.securityContext.readOnlyRootFilesystem == true
}
input_containers[c] {
  c := input.review.object.spec.containers[_]
}
has_field(object, field) = true {
  object[field]
}
''', {})
T["K8sPSPVolumeTypes"] = ('''package k8spspvolumetypes
violation[{"msg": msg, "details": {}}] {
  volume_fields := {x | input.review.object.spec.volumes[_][x]; x != "name"}
  field := volume_fields[_]
  not input_volume_type_allowed(field)
  msg := sprintf("The volume type %v is not allowed, pod: %v. Allowed volume types: %v", [field, input.review.object.metadata.name, input.parameters.volumes])
}
input_volume_type_allowed(field) {
  input.parameters.volumes[_] == "*"
}
input_volume_type_allowed(field) {
  field == input.parameters.volumes[_]
}
''', {"volumes": ["configMap", "emptyDir", "secret"]})
T["K8sExternalIPs"] = ('''package k8sexternalips
violation[{"msg": msg}] {
  input.review.kind.kind == "Service"
  input.review.kind.group == ""
  allowedIPs := {ip | ip := input.parameters.allowedIPs[_]}
  externalIPs := {ip | ip := input.review.object.spec.externalIPs[_]}
  forbiddenIPs := externalIPs - allowedIPs
  count(forbiddenIPs) > 0
  msg := sprintf("service has forbidden external IPs: %v", [forbiddenIPs])
}
''', {"allowedIPs": ["203.0.113.0"]})
T["K8sHttpsOnly"] = ('''package k8shttpsonly
violation[{"msg": msg}] {
  input.review.object.kind == "Ingress"
  re_match("^(extensions|networking.k8s.io)/", input.review.object.apiVersion)
  ingress := input.review.object
  not https_complete(ingress)
  msg := sprintf("Ingress should be https. tls configuration and allow-http=false annotation are required for %v", [ingress.metadata.name])
}
https_complete(ingress) = true {
  ingress.spec["tls"]
  count(ingress.spec.tls) > 0
  ingress.metadata.annotations["kubernetes.io/ingress.allow-http"] == "false"
}
''', {})
T["K8sRequiredResources"] = ('''package k8srequiredresources
violation[{"msg": msg}] {
  container := input.review.object.spec.containers[_]
  provided := {resource_type | container.resources.limits[resource_type]}
  required := {resource_type | resource_type := input.parameters.limits[_]}
  missing := required - provided
  count(missing) > 0
  msg := sprintf("container <%v> does not have <%v> limits defined", [container.name, missing])
}
''', {"limits": ["cpu", "memory"]})
T["K8sPSPAllowedUsers"] = ('''package k8spspallowedusers
violation[{"msg": msg}] {
  rule := input.parameters.runAsUser.rule
  container := input.review.object.spec.containers[_]
  provided_user := get_user(container)
  not accept_users(rule, provided_user)
  msg := sprintf("Container %v is attempting to run as disallowed user %v", [container.name, provided_user])
}
get_user(c) = u { u := c.securityContext.runAsUser }
accept_users("RunAsAny", provided_user) {true}
accept_users("MustRunAsNonRoot", provided_user) = res {res := provided_user != 0}
accept_users("MustRunAs", provided_user) = res  {
  ranges := input.parameters.runAsUser.ranges
  matching := {1 | provided_user >= ranges[j].min; provided_user <= ranges[j].max}
  res := count(matching) > 0
}
''', {"runAsUser": {"rule": "MustRunAs", "ranges": [{"min": 100, "max": 200}]}})

def pod(name, containers, **spec):
    s = {"containers": containers}; s.update(spec)
    return {"apiVersion": "v1", "kind": "Pod", "metadata": {"name": name, "namespace": "default", "annotations": {"owner": "Bob1"}}, "spec": s}
OBJS = [
 pod("p1", [{"name": "a", "image": "gcr.io/good/app:1.0", "securityContext": {"runAsUser": 150, "readOnlyRootFilesystem": True, "capabilities": {"add": ["NET_BIND_SERVICE"], "drop": ["ALL"]}}, "resources": {"limits": {"cpu": "1", "memory": "1Gi"}}}]),
 pod("p2", [{"name": "a", "image": "evil.io/app:latest", "securityContext": {"runAsUser": 0, "capabilities": {"add": ["SYS_ADMIN"]}}, "ports": [{"hostPort": 22}]}, {"name": "b", "image": "docker.io/library/nginx", "resources": {"limits": {"cpu": "1"}}}], hostPID=True, hostNetwork=True,
     volumes=[{"name": "v", "hostPath": {"path": "/"}}, {"name": "w", "emptyDir": {}}], initContainers=[{"name": "i", "image": "evil.io/init@sha256:abcdef0123"}]),
 pod("p3", [{"name": "a", "image": "gcr.io/good/app@sha256:0123abcd", "securityContext": {"runAsUser": 500}}], hostIPC=False),
 {"apiVersion": "v1", "kind": "Service", "metadata": {"name": "s1", "namespace": "default"}, "spec": {"type": "NodePort", "externalIPs": ["1.2.3.4", "203.0.113.0"]}},
 {"apiVersion": "v1", "kind": "Service", "metadata": {"name": "s2", "namespace": "default"}, "spec": {"type": "ClusterIP"}},
 {"apiVersion": "networking.k8s.io/v1", "kind": "Ingress", "metadata": {"name": "i1", "namespace": "default", "annotations": {"kubernetes.io/ingress.allow-http": "false"}}, "spec": {"tls": [{"hosts": ["a.com"]}], "rules": [{"host": "*.a.com"}, {"host": "b.com"}, {}]}},
 {"apiVersion": "networking.k8s.io/v1", "kind": "Ingress", "metadata": {"name": "i2", "namespace": "default"}, "spec": {"rules": [{"host": "c.com"}]}},
 {"apiVersion": "apps/v1", "kind": "Deployment", "metadata": {"name": "d1", "namespace": "default"}, "spec": {"replicas": 1, "template": {"spec": {"containers": [{"name": "x", "image": "gcr.io/good/x:1"}]}}}},
 {"apiVersion": "apps/v1", "kind": "Deployment", "metadata": {"name": "d2", "namespace": "default", "annotations": {"owner": "alice"}}, "spec": {"replicas": 3}},
]


# ---- a second batch: PSP AppArmor / sysctls / fsGroup / SELinux / procMount, RBAC subjects, tty / stdin, service-account
# token mounts, service-account updates (object vs oldObject: tests/test_root_scope.py exercises it with UPDATE requests),
# deprecated APIs, `some .. in` / `every`, and two shapes the device plan REFUSES (reported, never approximated)
T2 = {}
T2["K8sPSPAppArmor"] = ('''package k8spspapparmor
violation[{"msg": msg, "details": {}}] {
  metadata := input.review.object.metadata
  container := input_containers[_]
  not input_apparmor_allowed(container, metadata)
  msg := sprintf("AppArmor profile is not allowed, pod: %v, container: %v. Allowed profiles: %v", [input.review.object.metadata.name, container.name, input.parameters.allowedProfiles])
}
input_apparmor_allowed(container, metadata) {
  get_annotation_for(container, metadata) == input.parameters.allowedProfiles[_]
}
input_containers[c] {
  c := input.review.object.spec.containers[_]
}
get_annotation_for(container, metadata) = out {
  out = metadata.annotations[sprintf("container.apparmor.security.beta.kubernetes.io/%v", [container.name])]
}
get_annotation_for(container, metadata) = out {
  not metadata.annotations[sprintf("container.apparmor.security.beta.kubernetes.io/%v", [container.name])]
  out = "runtime/default"
}
''', {"allowedProfiles": ["runtime/default"]})
T2["K8sPSPForbiddenSysctls"] = ('''package k8spspforbiddensysctls
violation[{"msg": msg, "details": {}}] {
  sysctl := input.review.object.spec.securityContext.sysctls[_].name
  forbidden_sysctl(sysctl)
  msg := sprintf("The sysctl %v is not allowed, pod: %v. Forbidden sysctls: %v", [sysctl, input.review.object.metadata.name, input.parameters.forbiddenSysctls])
}
forbidden_sysctl(sysctl) {
  input.parameters.forbiddenSysctls[_] == "*"
}
forbidden_sysctl(sysctl) {
  input.parameters.forbiddenSysctls[_] == sysctl
}
forbidden_sysctl(sysctl) {
  forbidden := input.parameters.forbiddenSysctls[_]
  endswith(forbidden, "*")
  startswith(sysctl, trim_suffix(forbidden, "*"))
}
''', {"forbiddenSysctls": ["kernel.*", "net.core.somaxconn"]})
T2["K8sPSPFSGroup"] = ('''package k8spspfsgroup
violation[{"msg": msg, "details": {}}] {
  spec := input.review.object.spec
  not input_fsGroup_allowed(spec)
  msg := sprintf("The provided pod spec fsGroup is not allowed, pod: %v. Allowed fsGroup: %v", [input.review.object.metadata.name, input.parameters])
}
input_fsGroup_allowed(spec) {
  input.parameters.rule == "RunAsAny"
}
input_fsGroup_allowed(spec) {
  input.parameters.rule == "MustRunAs"
  fg := spec.securityContext.fsGroup
  count(input.parameters.ranges) > 0
  range := input.parameters.ranges[_]
  value_within_range(range, fg)
}
input_fsGroup_allowed(spec) {
  input.parameters.rule == "MayRunAs"
  not has_field(spec, "securityContext")
}
input_fsGroup_allowed(spec) {
  input.parameters.rule == "MayRunAs"
  not spec.securityContext.fsGroup
}
input_fsGroup_allowed(spec) {
  input.parameters.rule == "MayRunAs"
  fg := spec.securityContext.fsGroup
  count(input.parameters.ranges) > 0
  range := input.parameters.ranges[_]
  value_within_range(range, fg)
}
value_within_range(range, value) {
  range.min <= value
  range.max >= value
}
has_field(object, field) = true {
  object[field]
}
''', {"rule": "MayRunAs", "ranges": [{"min": 1, "max": 1000}]})
T2["K8sPSPSELinuxV2"] = ('''package k8spspselinux
violation[{"msg": msg, "details": {}}] {
  has_field(input.review.object.spec, "securityContext")
  has_field(input.review.object.spec.securityContext, "seLinuxOptions")
  not input_seLinuxOptions_allowed(input.review.object.spec.securityContext.seLinuxOptions)
  msg := sprintf("SELinux options is not allowed, pod: %v. Allowed options: %v", [input.review.object.metadata.name, input.parameters.allowedSELinuxOptions])
}
violation[{"msg": msg, "details": {}}] {
  c := input.review.object.spec.containers[_]
  has_field(c.securityContext, "seLinuxOptions")
  not input_seLinuxOptions_allowed(c.securityContext.seLinuxOptions)
  msg := sprintf("SELinux options is not allowed, pod: %v, container %v. Allowed options: %v", [input.review.object.metadata.name, c.name, input.parameters.allowedSELinuxOptions])
}
input_seLinuxOptions_allowed(options) {
  params := input.parameters.allowedSELinuxOptions[_]
  field_allowed("level", options, params)
  field_allowed("role", options, params)
  field_allowed("type", options, params)
  field_allowed("user", options, params)
}
field_allowed(field, options, params) {
  params[field] == options[field]
}
field_allowed(field, options, params) {
  not has_field(options, field)
}
has_field(object, field) = true {
  object[field]
}
''', {"allowedSELinuxOptions": [{"level": "s0:c123,c456", "role": "object_r", "type": "svirt_sandbox_file_t", "user": "system_u"}]})
T2["K8sPSPProcMount"] = ('''package k8spspprocmount
violation[{"msg": msg, "details": {}}] {
  c := input.review.object.spec.containers[_]
  allowedProcMount := get_allowed_proc_mount(input)
  not input_proc_mount_type_allowed(allowedProcMount, c)
  msg := sprintf("ProcMount type is not allowed, container: %v. Allowed procMount types: %v", [c.name, allowedProcMount])
}
input_proc_mount_type_allowed(allowedProcMount, c) {
  allowedProcMount == "default"
  lower(c.securityContext.procMount) == "default"
}
input_proc_mount_type_allowed(allowedProcMount, c) {
  allowedProcMount == "unmasked"
}
input_proc_mount_type_allowed(allowedProcMount, c) {
  not c.securityContext.procMount
}
get_allowed_proc_mount(arg) = out {
  not arg.parameters
  out = "default"
}
get_allowed_proc_mount(arg) = out {
  not arg.parameters.procMount
  out = "default"
}
get_allowed_proc_mount(arg) = out {
  arg.parameters.procMount
  not valid_proc_mount(arg.parameters.procMount)
  out = "default"
}
get_allowed_proc_mount(arg) = out {
  valid_proc_mount(arg.parameters.procMount)
  out = lower(arg.parameters.procMount)
}
valid_proc_mount(str) {
  lower(str) == "default"
}
valid_proc_mount(str) {
  lower(str) == "unmasked"
}
''', {"procMount": "Default"})
T2["K8sDisallowAnonymous"] = ('''package k8sdisallowanonymous
violation[{"msg": msg}] {
  not is_allowed(input.review.object.roleRef, input.parameters.allowedRoles)
  review(input.review.object.subjects[_])
  msg := sprintf("Unauthenticated user reference is not allowed in %v %v ", [input.review.object.kind, input.review.object.metadata.name])
}
is_allowed(role, allowedRoles) {
  role.name == allowedRoles[_]
}
review(subject) = true {
  subject.name == "system:unauthenticated"
}
review(subject) = true {
  subject.name == "system:anonymous"
}
''', {"allowedRoles": ["cluster-role-1"]})
T2["K8sDisallowInteractiveTTY"] = ('''package k8sdisallowinteractivetty
violation[{"msg": msg, "details": {}}] {
  c := input_containers[_]
  input_allow_interactive_fields(c)
  msg := sprintf("Containers using tty or stdin (%v) are not allowed running image: %v", [c.name, c.image])
}
input_allow_interactive_fields(c) {
  has_field(c, "stdin")
  not c.stdin == false
}
input_allow_interactive_fields(c) {
  has_field(c, "tty")
  not c.tty == false
}
input_containers[c] {
  c := input.review.object.spec.containers[_]
}
input_containers[c] {
  c := input.review.object.spec.ephemeralContainers[_]
}
has_field(object, field) = true {
  object[field]
}
has_field(object, field) = true {
  object[field] == false
}
''', {})
T2["K8sAutomountToken"] = ('''package k8sautomountserviceaccounttoken
violation[{"msg": msg}] {
  obj := input.review.object
  mountServiceAccountToken(obj.spec)
  msg := sprintf("Automounting service account token is disallowed, pod: %v", [obj.metadata.name])
}
mountServiceAccountToken(spec) {
  spec.automountServiceAccountToken == true
}
mountServiceAccountToken(spec) {
  not has_key(spec, "automountServiceAccountToken")
  "/var/run/secrets/kubernetes.io/serviceaccount" == input_containers[_].volumeMounts[_].mountPath
}
input_containers[c] {
  c := input.review.object.spec.containers[_]
}
has_key(x, k) {
  _ = x[k]
}
''', {})
T2["K8sNoUpdateServiceAccount"] = ('''package noupdateserviceaccount
violation[{"msg": msg}] {
  input.review.operation == "UPDATE"
  new := input.review.object.spec.serviceAccountName
  old := input.review.oldObject.spec.serviceAccountName
  new != old
  msg := sprintf("cannot update serviceAccountName from %v to %v", [old, new])
}
''', {})
T2["K8sVerifyDeprecatedAPI"] = ('''package verifydeprecatedapi
violation[{"msg": msg}] {
  kvs := input.parameters.kvs[_]
  kvs.deprecatedAPI == input.review.object.apiVersion
  k := kvs.kinds[_]
  k == input.review.object.kind
  msg := get_message(input.review.object.kind, input.review.object.apiVersion, input.parameters.k8sVersion, kvs.targetAPI)
}
get_message(kind, apiVersion, k8sVersion, targetAPI) = msg {
  not match(targetAPI)
  msg := sprintf("API %v for %v is deprecated in Kubernetes version %v, please use %v instead", [kind, apiVersion, k8sVersion, targetAPI])
}
get_message(kind, apiVersion, k8sVersion, targetAPI) = msg {
  match(targetAPI)
  msg := sprintf("API %v for %v is deprecated in Kubernetes version %v, please see Kubernetes API deprecation guide", [kind, apiVersion, k8sVersion])
}
match(api) {
  api == "None"
}
''', {"kvs": [{"deprecatedAPI": "apps/v1beta1", "kinds": ["Deployment"], "targetAPI": "apps/v1"}, {"deprecatedAPI": "networking.k8s.io/v1beta1", "kinds": ["Ingress"], "targetAPI": "None"}], "k8sVersion": 1.22})
T2["K8sBlockLoadBalancer"] = ('''package k8sblockloadbalancer
violation[{"msg": msg}] {
  input.review.kind.kind == "Service"
  input.review.object.spec.type == "LoadBalancer"
  msg := "User is not allowed to create service of type LoadBalancer"
}
''', {})
T2["K8sSomeInEvery"] = ('''package k8ssomeinevery
import future.keywords.in
import future.keywords.every
violation[{"msg": msg}] {
  some c in input.review.object.spec.containers
  not c.image in input.parameters.images
  msg := sprintf("image %v of %v not in list", [c.image, c.name])
}
violation[{"msg": msg}] {
  count(input.review.object.spec.containers) > 0
  every c in input.review.object.spec.containers {
    startswith(c.image, "evil.io/")
  }
  msg := "all containers are evil"
}
''', {"images": ["gcr.io/good/app:1.0", "docker.io/library/nginx"]})
T2["K8sStringOps"] = ('''package k8sstringops
violation[{"msg": msg}] {
  c := input.review.object.spec.containers[_]
  parts := split(c.image, "/")
  count(parts) > 2
  host := parts[0]
  not glob.match("*.io", [], host)
  msg := sprintf("registry %v of %v", [host, upper(c.name)])
}
violation[{"msg": msg}] {
  c := input.review.object.spec.containers[_]
  img := replace(c.image, "evil", "good")
  img != c.image
  n := count(c.image)
  msg := sprintf("%v -> %v (%d bytes, tag %v)", [c.image, img, n, substring(c.image, indexof(c.image, ":"), -1)])
}
''', {})

OBJS2 = list(OBJS)
OBJS2 = OBJS2 + [
 {"apiVersion": "rbac.authorization.k8s.io/v1", "kind": "ClusterRoleBinding", "metadata": {"name": "crb1"}, "roleRef": {"name": "cluster-admin"}, "subjects": [{"name": "system:anonymous"}, {"name": "bob"}]},
 {"apiVersion": "rbac.authorization.k8s.io/v1", "kind": "ClusterRoleBinding", "metadata": {"name": "crb2"}, "roleRef": {"name": "cluster-role-1"}, "subjects": [{"name": "system:unauthenticated"}]},
 {"apiVersion": "apps/v1beta1", "kind": "Deployment", "metadata": {"name": "old", "namespace": "default"}, "spec": {"replicas": 2}},
 {"apiVersion": "networking.k8s.io/v1beta1", "kind": "Ingress", "metadata": {"name": "oldi", "namespace": "default"}, "spec": {}},
 {"apiVersion": "v1", "kind": "Service", "metadata": {"name": "lb", "namespace": "default"}, "spec": {"type": "LoadBalancer"}},
 pod("p4", [{"name": "t", "image": "evil.io/x/y:1", "tty": True, "stdin": False, "securityContext": {"procMount": "Unmasked", "seLinuxOptions": {"level": "s0:c1", "role": "object_r"}}, "volumeMounts": [{"mountPath": "/var/run/secrets/kubernetes.io/serviceaccount"}]}],
     securityContext={"fsGroup": 2000, "sysctls": [{"name": "kernel.shm_rmid_forced"}, {"name": "net.core.somaxconn"}, {"name": "vm.swappiness"}], "seLinuxOptions": {"level": "s0:c123,c456", "role": "object_r", "type": "svirt_sandbox_file_t", "user": "system_u"}}),
 pod("p5", [{"name": "u", "image": "quay.io/a/b/c:2"}], automountServiceAccountToken=True, securityContext={"fsGroup": 500}),
]
OBJS2[-2]["metadata"]["annotations"] = {"container.apparmor.security.beta.kubernetes.io/t": "unconfined"}
UNSUPPORTED2 = {"K8sPSPAppArmor": "review data indexed by a symbolic key",                      # annotations[sprintf(.., [container.name])]
                "K8sStringOps": "builtin glob.match is not implemented by this engine"}       # valid Rego: refused, not a type error


CONTAINS = {
    "K8sContainsStmt": 'package k\nviolation[{"msg": msg}] {\n  contains(input.review.object.metadata.name, "bad")\n  msg := "bad name"\n}\n',
    "K8sContainsAssign": 'package k\nviolation[{"msg": msg}] {\n  x := contains(input.review.object.metadata.name, "bad")\n  x == true\n  msg := "bad name"\n}\n',
    "K8sContainsNot": 'package k\nviolation[{"msg": msg}] {\n  not contains(input.review.object.metadata.name, "good")\n  msg := "not good"\n}\n',
    "K8sContainsHelper": 'package k\nviolation[{"msg": msg}] {\n  bad(input.review.object.metadata.name)\n  msg := "bad name"\n}\nbad(n) {\n  contains(n, "bad")\n}\n',
}


def _constraint(kind, params):
    return {"apiVersion": "constraints.gatekeeper.sh/v1beta1", "kind": kind, "metadata": {"name": kind.lower() + "-1"}, "spec": {"parameters": params}}


@pytest.mark.parametrize("backend", BACKENDS)
def test_library_patterns_one_plan(backend):
    templates = [tmpl(k, rego) for k, (rego, _) in T.items()]
    constraints = [_constraint(k, params) for k, (_, params) in T.items()]
    c, oc = load_both(backend, templates, constraints)
    reviews = [D.AugmentedUnstructured(D.Unstructured(o), None, "Original") for o in OBJS]
    n = assert_parity(c, oc, reviews, D.GATOR_EP)
    assert n == 44   # violations of the nine objects under the sixteen constraints (counted by the oracle; pinned so that both sides cannot drift together unnoticed)


@pytest.mark.parametrize("backend", BACKENDS)
def test_contains_is_a_builtin_outside_rule_heads(backend):
    templates = [tmpl(k, rego) for k, rego in CONTAINS.items()]
    constraints = [_constraint(k, {}) for k in CONTAINS]
    c, oc = load_both(backend, templates, constraints)
    objs = [{"apiVersion": "v1", "kind": "Pod", "metadata": {"name": n, "namespace": "d"}} for n in ("a-bad-pod", "good-pod", "verylongname-with-bad-inside-it")]
    reviews = [D.AugmentedUnstructured(D.Unstructured(o), None, "Original") for o in objs]
    assert assert_parity(c, oc, reviews, D.GATOR_EP) == 8
    got = c.ReviewBatch(reviews, D.GATOR_EP)
    assert [sorted(r.constraint["kind"] for r in g) for g in got] == [
        ["K8sContainsAssign", "K8sContainsHelper", "K8sContainsNot", "K8sContainsStmt"], [],
        ["K8sContainsAssign", "K8sContainsHelper", "K8sContainsNot", "K8sContainsStmt"]]


@pytest.mark.parametrize("backend", BACKENDS)
def test_gator_bench_fixture_pairs(backend, fixtures):
    """test/gator/bench/{basic,both}: the reference's own benchmark inputs -- K8sRequiredLabels and K8sAllowedRepos
    (`strings.any_prefix_match`; `both` also carries a CEL source, the Rego one is the Rego driver's) with one valid and
    one invalid Pod each: the valid one yields nothing, the invalid one exactly one violation."""
    want_msgs = {"basic": [[], ['Missing required labels: {"team"}']],
                 "both": [[], ['container <app> has an invalid image repo <quay.io/unauthorized/app:latest>, allowed repos are '
                               '["gcr.io/myproject/", "docker.io/library/"]']]}
    for d, want in want_msgs.items():
        y = lambda f: fixtures["yaml"]["test/gator/bench/%s/%s.yaml" % (d, f)]["docs"]
        c, oc = load_both(backend, y("template"), y("constraint"))
        reviews = [D.AugmentedUnstructured(D.Unstructured(o), None, "Original") for o in y("resources")]
        assert_parity(c, oc, reviews, D.GATOR_EP)
        assert [sorted(r.msg for r in g) for g in c.ReviewBatch(reviews, D.GATOR_EP)] == want


@pytest.mark.parametrize("backend", BACKENDS)
def test_library_patterns_second_batch(backend):
    good = {k: v for k, v in T2.items() if k not in UNSUPPORTED2}
    c, oc = load_both(backend, [tmpl(k, rego) for k, (rego, _) in good.items()], [_constraint(k, params) for k, (_, params) in good.items()])
    reviews = [D.AugmentedUnstructured(D.Unstructured(o), None, "Original") for o in OBJS2]
    assert assert_parity(c, oc, reviews, D.GATOR_EP) == 17
    # what the engine cannot compile exactly is an ERROR at AddTemplate / AddConstraint -- the cgo shim keeps such a
    # template on the stock driver -- never a different answer
    for kind, why in UNSUPPORTED2.items():
        rego, params = T2[kind]
        with pytest.raises((D.UnsupportedError, D.ClientError, D.EngineError), match=why):
            c2, _ = load_both(backend, [tmpl(kind, rego)], [_constraint(kind, params)])


# ---- a third batch: userInfo / namespaceObject, object.get with a key path, set algebra, type tests, arithmetic, default /
# else rules, string builtins in messages, nested parameters, a regular expression on a split() component -- bare objects AND AdmissionRequests; two shapes are refused
T3 = {}
T3["K8sUserInfo"] = ('''package k
violation[{"msg": msg}] {
  input.review.userInfo.groups[_] == "system:masters"
  not startswith(input.review.userInfo.username, "system:")
  msg := sprintf("user %v may not act as cluster admin on %v", [input.review.userInfo.username, input.review.object.metadata.name])
}
''', {})
T3["K8sNamespaceObjectLabels"] = ('''package k
violation[{"msg": msg}] {
  input.review.namespaceObject.metadata.labels.env == "prod"
  not input.review.object.metadata.labels.owner
  msg := sprintf("objects in prod namespace %v need an owner label", [input.review.namespaceObject.metadata.name])
}
''', {})
T3["K8sObjectGetPath"] = ('''package k
violation[{"msg": msg}] {
  policy := object.get(input.review.object, ["spec", "dnsPolicy"], "ClusterFirst")
  not allowed(policy)
  msg := sprintf("dnsPolicy %v is not allowed", [policy])
}
allowed(p) { p == input.parameters.allowed[_] }
''', {"allowed": ["ClusterFirst", "Default"]})
T3["K8sSetOps"] = ('''package k
violation[{"msg": msg, "details": {"extra": extra, "common": common}}] {
  have := {l | input.review.object.metadata.labels[l]}
  allowed := {l | l := input.parameters.allowed[_]}
  required := {l | l := input.parameters.required[_]}
  extra := have - (allowed | required)
  common := have & required
  count(extra) > 0
  msg := sprintf("labels %v are not allowed (%d required present)", [extra, count(common)])
}
''', {"allowed": ["app", "owner"], "required": ["team"]})
T3["K8sTypeChecks"] = ('''package k
violation[{"msg": msg}] {
  v := input.review.object.spec.replicas
  not is_number(v)
  msg := sprintf("replicas must be a number, got %v", [type_name(v)])
}
violation[{"msg": msg}] {
  is_string(input.review.object.spec.hostNetwork)
  msg := "hostNetwork must be boolean"
}
''', {})
T3["K8sArithmetic"] = ('''package k
violation[{"msg": msg}] {
  r := input.review.object.spec.replicas
  r * 2 > input.parameters.max + 1
  r % 2 == 1
  msg := sprintf("%d replicas: too many and odd (limit %v)", [r, (input.parameters.max + 1) / 2])
}
''', {"max": 5})
T3["K8sCountAndSum"] = ('''package k
violation[{"msg": msg}] {
  ports := [p | p := input.review.object.spec.containers[_].ports[_].containerPort]
  count(ports) > input.parameters.maxPorts
  msg := sprintf("%d ports exposed (sum %d, max %d), more than %d", [count(ports), sum(ports), max(ports), input.parameters.maxPorts])
}
''', {"maxPorts": 1})
T3["K8sElseAndDefault"] = ('''package k
default tier = "none"
tier = "gold" { input.review.object.metadata.labels.tier == "gold" }
tier = "silver" { input.review.object.metadata.labels.tier == "silver" }
limit = 10 { tier == "gold" } else = 5 { tier == "silver" } else = 1
violation[{"msg": msg}] {
  count(input.review.object.spec.containers) > limit
  msg := sprintf("tier %v allows %d containers", [tier, limit])
}
''', {})
T3["K8sStringBuiltins"] = ('''package k
violation[{"msg": msg}] {
  n := input.review.object.metadata.name
  trim_prefix(n, "tmp-") != n
  msg := sprintf("%s (%q) is a temporary name; use %v", [upper(n), n, concat("-", ["perm", trim_prefix(n, "tmp-")])])
}
violation[{"msg": msg}] {
  img := input.review.object.spec.containers[_].image
  strings.any_suffix_match(img, input.parameters.badSuffixes)
  msg := sprintf("image %v has a forbidden suffix", [img])
}
''', {"badSuffixes": [":latest", ":dev"]})
T3["K8sObjectComprehension"] = ('''package k
violation[{"msg": msg, "details": {"limits": limits}}] {
  limits := {c.name: c.resources.limits.cpu | c := input.review.object.spec.containers[_]; c.resources.limits.cpu}
  count(limits) < count(input.review.object.spec.containers)
  msg := sprintf("only %d of %d containers have cpu limits", [count(limits), count(input.review.object.spec.containers)])
}
''', {})
T3["K8sRegexAndSplit"] = ('''package k
violation[{"msg": msg}] {
  c := input.review.object.spec.containers[_]
  parts := split(c.image, ":")
  count(parts) == 2
  not regex.match("^v?[0-9]+\\\\.[0-9]+(\\\\.[0-9]+)?$", parts[1])
  msg := sprintf("tag %v of %v is not a version", [parts[1], parts[0]])
}
''', {})
T3["K8sNestedParams"] = ('''package k
violation[{"msg": msg}] {
  rule := input.parameters.rules[_]
  rule.kind == input.review.object.kind
  lbl := rule.labels[_]
  not input.review.object.metadata.labels[lbl.key]
  msg := sprintf("%v needs label %v (%v)", [rule.kind, lbl.key, lbl.why])
}
''', {"rules": [{"kind": "Pod", "labels": [{"key": "app", "why": "routing"}, {"key": "team", "why": "billing"}]}, {"kind": "Service", "labels": [{"key": "app", "why": "routing"}]}]})
def pod3(name, containers, labels=None, **spec):
    s = {"containers": containers}; s.update(spec)
    md = {"name": name, "namespace": "default"}
    if labels is not None: md["labels"] = labels
    return {"apiVersion": "v1", "kind": "Pod", "metadata": md, "spec": s}
nsobj = {"apiVersion": "v1", "kind": "Namespace", "metadata": {"name": "default", "labels": {"env": "prod"}}}
OBJ3 = [
 pod3("tmp-job", [{"name": "a", "image": "r/a:latest", "ports": [{"containerPort": 80}, {"containerPort": 443}], "resources": {"limits": {"cpu": "1"}}}, {"name": "b", "image": "r/b:v1.2.3", "ports": [{"containerPort": 8080}]}], {"app": "x", "foo": "bar", "team": "t", "tier": "silver"}, dnsPolicy="None", replicas=7),
 pod3("web", [{"name": "a", "image": "r/a:1.0"}], {"owner": "me", "tier": "gold"}, replicas="3", hostNetwork="true"),
 pod3("solo", [{"name": "a", "image": "r/a:dev"}, {"name": "b", "image": "r/b"}], None, replicas=4),
 {"apiVersion": "v1", "kind": "Service", "metadata": {"name": "svc", "namespace": "default", "labels": {"zzz": "1"}}, "spec": {}},
]
def reviews3(wrap):
    out = []
    for o in OBJ3:
        out.append(wrap.AugmentedUnstructured(wrap.Unstructured(o), nsobj, "Original"))
        req = {"uid": "u", "kind": {"group": "", "version": "v1", "kind": o["kind"]}, "operation": "CREATE", "name": o["metadata"]["name"], "namespace": "default",
               "userInfo": {"username": "alice", "groups": ["dev", "system:masters"]}, "object": o}
        out.append(wrap.AugmentedReview(wrap.AdmissionRequest(req), nsobj, "Original"))
    return out
UNSUPPORTED3 = {"K8sCountAndSum": "count\\(\\) of a set built from review data compared with a constant other than 0",
                "K8sObjectComprehension": "object comprehension over review data"}


@pytest.mark.parametrize("backend", BACKENDS)
def test_library_patterns_third_batch(backend):
    good = {k: v for k, v in T3.items() if k not in UNSUPPORTED3}
    c, oc = load_both(backend, [tmpl(k, rego) for k, (rego, _) in good.items()], [_constraint(k, params) for k, (_, params) in good.items()])
    rv = reviews3(D)
    assert assert_parity(c, oc, rv, D.GATOR_EP, namespaces=[nsobj] * len(rv)) == 46
    for kind, why in UNSUPPORTED3.items():
        rego, params = T3[kind]
        with pytest.raises((D.UnsupportedError, D.ClientError, D.EngineError), match=why):
            load_both(backend, [tmpl(kind, rego)], [_constraint(kind, params)])
