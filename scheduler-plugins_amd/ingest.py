"""ctypes wrapper of the wire-format ingester (host/ingest.cc): NodeResourceTopology JSON -> spx_nrt_objects."""
from __future__ import annotations

import ctypes as C
from typing import Optional, Sequence

from . import header, lib


class _Borrowed:
    """A table owned by the ingest handle, exposing the .ref() / .struct interface of _abi.Table."""

    def __init__(self, ptr, owner):
        self._ptr, self._owner = ptr, owner
        self.struct = ptr.contents

    def ref(self):
        return self._ptr


class NrtIngest:
    def __init__(self, node_names: Sequence[str], resource_names: Sequence[str] = ()):
        self._lib = lib()
        self._h = C.POINTER(header().opaque["spx_ingest"])()
        names = (C.c_char_p * len(node_names))(*[n.encode() for n in node_names])
        rnames = (C.c_char_p * max(len(resource_names), 1))(*[r.encode() for r in resource_names])
        pp = C.POINTER(C.POINTER(C.c_char))
        rc = self._lib.spx_ingest_create(C.cast(names, pp), len(node_names), C.cast(rnames, pp), len(resource_names), C.byref(self._h))
        if rc != 0:
            raise RuntimeError(f"spx_ingest_create failed: {rc}")

    def close(self) -> None:
        if self._h:
            self._lib.spx_ingest_destroy(self._h)
            self._h = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def feed(self, json_bytes: bytes):
        """-> (objects decoded, objects whose name is not a node of the snapshot)"""
        n, unk = C.c_int64(), C.c_int64()
        rc = self._lib.spx_ingest_nrt_json(self._h, json_bytes, len(json_bytes), C.byref(n), C.byref(unk))
        if rc != 0:
            raise ValueError(self._lib.spx_ingest_error(self._h).decode())
        return n.value, unk.value

    def nrt_objects(self) -> _Borrowed:
        return _Borrowed(self._lib.spx_ingest_nrt_objects(self._h), self)

    def resource_classes(self) -> _Borrowed:
        return _Borrowed(self._lib.spx_ingest_resource_classes(self._h), self)

    def resource_id(self, name: str) -> int:
        return self._lib.spx_ingest_resource_id(self._h, name.encode())

    # ------------------------------------------------------------------ v1.Node / v1.Pod
    KINDS = {"region": 0, "zone": 1, "namespace": 2, "appgroup": 3, "selector": 4}

    def seed(self, kind: str, names: Sequence[str]) -> None:
        arr = (C.c_char_p * max(len(names), 1))(*[n.encode() for n in names])
        rc = self._lib.spx_ingest_seed_names(self._h, self.KINDS[kind], C.cast(arr, C.POINTER(C.POINTER(C.c_char))), len(names))
        if rc != 0:
            raise RuntimeError(f"spx_ingest_seed_names failed: {rc}")

    def name_id(self, kind: str, name: str) -> int:
        return self._lib.spx_ingest_name_id(self._h, self.KINDS[kind], name.encode())

    def feed_nodes(self, json_bytes: bytes):
        n, unk = C.c_int64(), C.c_int64()
        if self._lib.spx_ingest_nodes_json(self._h, json_bytes, len(json_bytes), C.byref(n), C.byref(unk)) != 0:
            raise ValueError(self._lib.spx_ingest_error(self._h).decode())
        return n.value, unk.value

    def feed_pods(self, json_bytes: bytes) -> int:
        n = C.c_int64()
        if self._lib.spx_ingest_pods_json(self._h, json_bytes, len(json_bytes), C.byref(n)) != 0:
            raise ValueError(self._lib.spx_ingest_error(self._h).decode())
        return n.value

    def reset_pods(self) -> None:
        self._lib.spx_ingest_pods_reset(self._h)

    def feed_appgroups(self, json_bytes: bytes) -> int:
        n = C.c_int64()
        if self._lib.spx_ingest_appgroups_json(self._h, json_bytes, len(json_bytes), C.byref(n)) != 0:
            raise ValueError(self._lib.spx_ingest_error(self._h).decode())
        return n.value

    def feed_nettopo(self, json_bytes: bytes, weights_name: str) -> None:
        if self._lib.spx_ingest_nettopo_json(self._h, json_bytes, len(json_bytes), weights_name.encode()) != 0:
            raise ValueError(self._lib.spx_ingest_error(self._h).decode())

    def feed_quotas(self, json_bytes: bytes, namespaces: Sequence[str]):
        arr = (C.c_char_p * len(namespaces))(*[n.encode() for n in namespaces])
        n, unk = C.c_int64(), C.c_int64()
        rc = self._lib.spx_ingest_quota_json(self._h, json_bytes, len(json_bytes), C.cast(arr, C.POINTER(C.POINTER(C.c_char))), len(namespaces),
                                             C.byref(n), C.byref(unk))
        if rc != 0:
            raise ValueError(self._lib.spx_ingest_error(self._h).decode())
        return n.value, unk.value

    def feed_metrics(self, json_bytes: bytes):
        """load-watcher response -> (entries of data.NodeMetricsMap, entries naming a node outside the snapshot)"""
        n, unk = C.c_int64(), C.c_int64()
        if self._lib.spx_ingest_metrics_json(self._h, json_bytes, len(json_bytes), C.byref(n), C.byref(unk)) != 0:
            raise ValueError(self._lib.spx_ingest_error(self._h).decode())
        return n.value, unk.value

    def metrics_objects(self) -> Optional[_Borrowed]:
        p = self._lib.spx_ingest_metrics_objects(self._h)
        return _Borrowed(p, self) if p else None

    def quota_objects(self) -> _Borrowed:
        return _Borrowed(self._lib.spx_ingest_quota_objects(self._h), self)

    def appgroup_objects(self) -> _Borrowed:
        return _Borrowed(self._lib.spx_ingest_appgroup_objects(self._h), self)

    def nettopo_objects(self) -> _Borrowed:
        return _Borrowed(self._lib.spx_ingest_nettopo_objects(self._h), self)

    def node_objects(self) -> _Borrowed:
        return _Borrowed(self._lib.spx_ingest_node_objects(self._h), self)

    def pod_objects(self) -> _Borrowed:
        return _Borrowed(self._lib.spx_ingest_pod_objects(self._h), self)


def quantity(text: str, milli: bool) -> Optional[int]:
    out = C.c_int64()
    return out.value if lib().spx_ingest_quantity(text.encode(), 1 if milli else 0, C.byref(out)) == 0 else None
