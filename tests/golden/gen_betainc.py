"""Generates tests/golden/betainc.json: values of the regularized incomplete beta function I_x(a,b) from an independent
implementation (scipy.special.betainc, Boost-based in scipy >= 1.14) over the parameter region LowRiskOverCommitment
reaches (alpha = mu*t, beta = (1-mu)*t with t = mu(1-mu)/variance - 1; beta.go:107-117) and a broad log-uniform region.

gonum v0.12.0 mathext.RegIncBeta — what the reference calls (beta.go:158-171) — is not vendored under /root/reference and
there is no Go toolchain here, so the function is pinned by its mathematical definition instead.  The region is bounded
to a + b <= 1e5, where the Cephes algorithm gonum ports agrees with the definition to ~1e-10.

    python tests/golden/gen_betainc.py        (deterministic: fixed seed)"""
import json
from pathlib import Path

import numpy as np
from scipy.special import betainc

rng = np.random.default_rng(20260921)
rows = []
for i in range(400):
    mu = rng.uniform(0.002, 0.998)
    var = mu * (1 - mu) * 10 ** rng.uniform(-4.5, -0.005)
    t = mu * (1 - mu) / var - 1
    a, b = mu * t, (1 - mu) * t
    x = rng.uniform(0.001, 0.999) if i & 1 else float(np.clip(mu + rng.normal() * var ** 0.5, 1e-6, 1 - 1e-6))
    rows.append((a, b, x))
for i in range(200):
    rows.append((10 ** rng.uniform(-2, 3), 10 ** rng.uniform(-2, 3), rng.uniform(0.001, 0.999)))
out = [{"a": float(a), "b": float(b), "x": float(x), "value": float(betainc(a, b, x))} for a, b, x in rows]
Path(__file__).with_name("betainc.json").write_text(json.dumps(out, indent=0))
print(len(out), "points")
