"""Writes tests/golden/appgroup_manifests.json: the reference's example AppGroup CRs (manifests/appgroup/appGroup-example.yaml,
onlineBoutique-appGroup-example.yaml, redis-appGroup-example.yaml) as JSON, values untouched.  Run where /root/reference is mounted."""
import json
from pathlib import Path

import yaml

REF = Path("/root/reference/manifests/appgroup")
docs = []
for name in ("appGroup-example.yaml", "onlineBoutique-appGroup-example.yaml", "redis-appGroup-example.yaml"):
    try:
        loaded = list(yaml.safe_load_all((REF / name).read_text()))
    except yaml.YAMLError as ex:  # one of the reference's examples is not valid YAML (a mis-indented `kind:`)
        print("skipped", name, "-", str(ex).splitlines()[0])
        continue
    for d in loaded:
        if d and d.get("kind") == "AppGroup":
            d["_source"] = f"manifests/appgroup/{name}"
            docs.append(d)
Path(__file__).with_name("appgroup_manifests.json").write_text(json.dumps(docs, indent=1))
print(len(docs), "AppGroups", [d["metadata"]["name"] for d in docs])
