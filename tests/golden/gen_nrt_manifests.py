"""Writes tests/golden/nrt_manifests.json: the reference's example NodeResourceTopology manifests
(manifests/noderesourcetopology/worker-node-A.yaml, worker-node-B.yaml) as the JSON documents the API server would serve
(YAML -> JSON, values untouched).  Run where /root/reference is mounted:  python tests/golden/gen_nrt_manifests.py"""
import json
from pathlib import Path

import yaml

REF = Path("/root/reference/manifests/noderesourcetopology")
docs = []
for name in ("worker-node-A.yaml", "worker-node-B.yaml"):
    for d in yaml.safe_load_all((REF / name).read_text()):
        if d and d.get("kind") == "NodeResourceTopology":
            d["_source"] = f"manifests/noderesourcetopology/{name}"
            docs.append(d)
Path(__file__).with_name("nrt_manifests.json").write_text(json.dumps(docs, indent=1))
print(len(docs), "objects")
