"""Writes tests/golden/pod_manifests.json: the pods of the reference's Online Boutique example
(manifests/appgroup/deploy-onlineBoutique-with-networkAware-scheduler.yaml: one Deployment per workload, labelled with the
AppGroup and workload-selector keys the network-aware plugins read) rendered as the v1.Pod JSON their ReplicaSets would create —
the pod template's metadata and spec, untouched.  Run where /root/reference is mounted."""
import json
from pathlib import Path

import yaml

SRC = Path("/root/reference/manifests/appgroup/deploy-onlineBoutique-with-networkAware-scheduler.yaml")
pods = []
for d in yaml.safe_load_all(SRC.read_text()):
    if d and d.get("kind") == "Deployment":
        t = d["spec"]["template"]
        meta = dict(t.get("metadata") or {})
        meta.setdefault("namespace", d["metadata"].get("namespace", "default"))
        meta["name"] = d["metadata"]["name"] + "-0"
        pods.append({"apiVersion": "v1", "kind": "Pod", "metadata": meta, "spec": t["spec"]})
Path(__file__).with_name("pod_manifests.json").write_text(json.dumps({"source": str(SRC).replace("/root/reference/", ""), "items": pods}, indent=1))
print(len(pods), "pods")
