"""Writes tests/golden/nettopo_manifests.json: the reference's example NetworkTopology CR
(manifests/networktopology/networkTopology-example.yaml) as JSON.  The example is written against an older schema (`costList` /
`originCosts` / `costs`); the CRD the reference ships (manifests/networktopology/crd.yaml:49-128, the Go types the decoder follows)
names those levels `topologyList` / `originList` / `costList`.  The values are untouched, the three keys are renamed, and
`configMapName` becomes the CRD's `configmapName`.  Run where /root/reference is mounted."""
import json
from pathlib import Path

import yaml

SRC = Path("/root/reference/manifests/networktopology/networkTopology-example.yaml")
doc = yaml.safe_load(SRC.read_text())
spec = doc["spec"]
if "configMapName" in spec:
    spec["configmapName"] = spec.pop("configMapName")
for w in spec["weights"]:
    w["topologyList"] = w.pop("costList")
    for t in w["topologyList"]:
        t["originList"] = t.pop("originCosts")
        for o in t["originList"]:
            o["costList"] = o.pop("costs")
doc["_source"] = "manifests/networktopology/networkTopology-example.yaml (keys renamed to the CRD's: see gen_nettopo_manifests.py)"
Path(__file__).with_name("nettopo_manifests.json").write_text(json.dumps([doc], indent=1) + "\n")
print(json.dumps(doc)[:300])
