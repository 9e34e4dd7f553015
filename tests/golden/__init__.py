"""Known-answer vectors transcribed AS DATA from the reference's own unit-test tables.

The reference is Go and cannot be executed in this image (no toolchain), so these
tables — each citing the reference test file:line it was read from — are what pins the oracle.
Nothing here is generated; tests/golden/README.md explains the provenance of every file.
"""
