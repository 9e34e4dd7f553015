"""The exactness claims the fast kernels rest on, checked on the CPU with exact rational arithmetic
and with numpy's IEEE float32/float64 — independent of any GPU run:

  * k_nrt_fast / k_alloc_masked: floor(num / c) computed as floor of ONE float64 product with a biased multiplier,
    for integers below 2^42 and quotients <= 100 (LeastAllocated, MostAllocated, the weighted mean, the masked normalisation);
  * k_nrt_fast MostAllocated: "request <= capacity" read off the same kind of product;
  * k_tlp_fast2 / k_commit_trimaran: the float32 TLP formula with its ambiguity test never disagrees with the float64
    reference sequence on a cell it does not flag;
  * k_lvrb_fast: the same for the float32 LVRB formula.
"""
import math
from fractions import Fraction

import numpy as np

LIM = 1 << 42


def rn(x: Fraction) -> float:
    """round-to-nearest-even float64 of an exact rational (CPython's Fraction -> float conversion is correctly rounded)"""
    return float(x)


def fma(a: float, b: float, c: float) -> float:
    return rn(Fraction(a) * Fraction(b) + Fraction(c))


def _capacities(rng, n):
    c = np.concatenate([rng.integers(1, 200, n // 4), rng.integers(1, 1 << 20, n // 4), (1 << rng.integers(1, 42, n // 4)) + rng.integers(-3, 4, n // 4),
                        LIM - 1 - rng.integers(0, 1000, n - 3 * (n // 4))])
    return [int(x) for x in np.clip(c, 1, LIM - 1)]


def _requests_near_boundaries(c, rng):
    """requests whose quotient (c - v) * 100 / c or v * 100 / c sits on or next to an integer, plus the ends and randoms"""
    out = {0, 1, c - 1, c, c + 1, c + 2, 2 * c}
    for q in list(range(0, 101, 7)) + [99, 100]:
        base = (q * c) // 100
        out.update({base - 1, base, base + 1, -(-q * c // 100)})
    out.update(int(x) for x in rng.integers(0, c + 1, 6))
    return sorted(v for v in out if 0 <= v < LIM)


def test_least_allocated_single_product_floor():
    rng = np.random.default_rng(1)
    bias = 100.0 + 2.0 ** -43
    for c in _capacities(rng, 1600):
        b = rn(Fraction(100, c))
        for v in _requests_near_boundaries(c, rng):
            t = fma(-float(v), b, bias)
            got = max(math.floor(t), 0)
            want = ((c - v) * 100) // c if v <= c else 0  # least_allocated.go:46-54: request > capacity scores 0
            assert got == want, (c, v, t)


def test_most_allocated_single_product_floor_and_fit_test():
    rng = np.random.default_rng(2)
    thr = 100.0 * (1.0 + 2.0 ** -48)
    one_eps = 1.0 + 2.0 ** -49
    for c in _capacities(rng, 1600):
        b = rn(Fraction(100, c))
        for v in _requests_near_boundaries(c, rng):
            tp = float(v) * one_eps  # one rounding
            tp = tp * b              # second rounding
            assert (tp <= thr) == (v <= c), (c, v, tp)
            if v <= c:
                assert math.floor(tp) == (v * 100) // c, (c, v, tp)


def test_weighted_mean_and_masked_normalisation_single_product_floor():
    rng = np.random.default_rng(3)
    one_eps = 1.0 + 2.0 ** -49
    for w in [1, 2, 3, 7, 10, 1000, (1 << 30) + 1, (LIM // 128) - 1] + [int(x) for x in rng.integers(1, LIM // 128, 200)]:
        wrc = (1.0 / w) * one_eps  # nrt_biased_rcp: RN(RN(1/w) * (1 + 2^-49))
        for acc in {0, w - 1, w, w + 1, 50 * w, 100 * w - 1, 100 * w} | {int(x) for x in rng.integers(0, 100 * w + 1, 8)}:
            assert math.floor(float(acc) * wrc) == acc // w, (w, acc)
    for rng_ in [1, 2, 99, 100, 101, (1 << 32) - 1, LIM - 1] + [int(x) for x in rng.integers(1, LIM, 400)]:
        b = (100.0 / rng_) * one_eps  # k_alloc_masked: (100.0 / range) * (1 + 2^-49)
        for d in {0, 1, rng_ - 1, rng_} | {(q * rng_) // 100 + k for q in range(0, 101, 9) for k in (-1, 0, 1)} | {int(x) for x in rng.integers(0, rng_ + 1, 6)}:
            if 0 <= d <= rng_:
                assert math.floor(float(d) * b) == (d * 100) // rng_, (rng_, d)


def _f32(x):
    return np.asarray(x, dtype=np.float32)


def test_tlp_float32_formula_never_disagrees_unflagged():
    """numpy model of k_tlp_fast2's per-cell arithmetic (float32 adds / fma / rint, the same constants) against the
    reference's float64 sequence (targetloadpacking.go:170-184) on 4e6 cells, continuous and integer-valued inputs"""
    rng = np.random.default_rng(4)
    n = 4_000_000
    t = 40.0
    c1, c2 = t / (100.0 - t), (100.0 - t) / t
    cap = rng.choice(np.array([2000, 8000, 16000, 64000, 128000], dtype=np.float64), n)
    util = np.where(rng.random(n) < 0.3, rng.integers(0, 100, n).astype(np.float64), rng.uniform(0, 100, n))
    missing = np.where(rng.random(n) < 0.5, 0.0, rng.integers(0, 4000, n).astype(np.float64))
    pod = rng.integers(0, 12000, n).astype(np.float64)
    # boundary stress: pods that land exactly on the target line of their node
    on_line = rng.random(n) < 0.05
    pod = np.where(on_line, np.floor(np.maximum(t * cap / 100.0 - (util / 100.0) * cap - missing, 0.0)), pod)
    um = (util / 100.0) * cap
    # ---- reference (float64, operation for operation)
    pred = 100.0 * ((um + pod) + missing) / cap
    x = np.where(pred > t, t * (100.0 - pred) / (100.0 - t), (100.0 - t) * pred / t + t)
    want = np.where(pred > 100.0, 0, np.floor(np.abs(x) + 0.5) * np.sign(x)).astype(np.int64)  # math.Round: half away from zero
    want = np.clip(want, 0, 255)
    # ---- fast path (k_tlp_prepare_fast + k_tlp_fast2)
    k = 100.0 / cap
    b = (um + missing) - t * cap / 100.0
    bh = np.rint(b)
    b2h, b2l = _f32(bh), _f32(b - bh)
    kc1, kc2 = _f32(-c1 * k), _f32(c2 * k)
    pod_f = _f32(pod)
    u = (pod_f + b2h) + b2l  # float32 adds
    gt = u > 0
    coef, off = np.where(gt, kc1, kc2), np.where(gt, _f32(t), _f32(100.0))
    xf = _f32(coef.astype(np.float64) * u.astype(np.float64) + off.astype(np.float64))  # fma: exact product and sum in f64, one f32 rounding
    rr = np.rint(xf)
    amb = ~(np.abs(xf - rr) < np.float32(0.5) - np.float32(4e-5)) | ~(np.abs(u) > np.float32(1e-6))
    got = np.clip(rr, 0, 255).astype(np.int64)  # v_cvt_pk_u8_f32 saturates
    bad = (~amb) & (got != want)
    assert not bad.any(), (int(bad.sum()), np.flatnonzero(bad)[:5])
    assert amb.mean() < 0.08  # the stress inputs make ties common; on continuous inputs it is ~1e-4
    cont = ~on_line & (util != np.floor(util))
    assert amb[cont].mean() < 5e-4


def test_lvrb_float32_formula_never_disagrees_unflagged():
    """numpy model of k_lvrb_fast's per-cell arithmetic against lv_total's float64 sequence (analysis.go:34-60,
    loadvariationriskbalancing.go:104-118) for regular resources (margin 1, sensitivity 1)"""
    rng = np.random.default_rng(5)
    n = 2_000_000
    cap_c = rng.choice(np.array([2000, 8000, 64000, 128000], dtype=np.float64), n)
    cap_m = rng.choice(np.array([16, 64, 256, 1024], dtype=np.float64), n) * 1024.0  # MiB
    avg_c, sd_c = rng.uniform(0, 100, n), rng.uniform(0, 40, n)
    avg_m, sd_m = np.where(rng.random(n) < 0.3, rng.integers(0, 100, n).astype(np.float64), rng.uniform(0, 100, n)), rng.uniform(0, 40, n)
    req_c = rng.integers(0, 9000, n).astype(np.float64)
    req_m = rng.integers(0, 40000, n).astype(np.float64)

    def exact(cap, avg, sd, req):
        used_avg = np.clip(avg * cap / 100.0, 0.0, cap)
        sigma = np.clip(np.clip(sd * cap / 100.0, 0.0, cap) / cap, 0.0, 1.0)  # pow(sigma, 1/1) == sigma, margin 1
        mu = np.clip((used_avg + req) / cap, 0.0, 1.0)
        return (1.0 - (mu + sigma) / 2.0) * 100.0, used_avg, sigma

    xc, ua_c, sg_c = exact(cap_c, avg_c, sd_c, req_c)
    xm, ua_m, sg_m = exact(cap_m, avg_m, sd_m, req_m)
    xr = np.minimum(xm, xc)  # both resources valid -> min
    want = np.clip(np.floor(xr + 0.5), 0, 255).astype(np.int64)

    def fast(cap, used_avg, sigma, req):
        # round 6: slope and offset of t / 50, the clamp 0..1 is the fma's output modifier, then one fma with -50 (k_lvrb_prepare_fast / k_lvrb_fast)
        bq = 1.0 / cap
        fa, fb, fc = _f32(100.0 - 50.0 * sigma), _f32(bq), _f32(bq * used_avg)
        tt = _f32(fb.astype(np.float64) * _f32(req).astype(np.float64) + fc.astype(np.float64))
        cl = np.clip(tt, np.float32(0), np.float32(1))
        return _f32(cl.astype(np.float64) * -50.0 + fa.astype(np.float64))  # y = s*(A - 50 clamp) with s = -1 folded below

    # s = -1 (both valid): y_r = -(A_r - clamp_r); x = -max(y_c, y_m)
    yc = -fast(cap_c, ua_c, sg_c, req_c)
    ym = -fast(cap_m, ua_m, sg_m, req_m)
    y = np.maximum(yc, ym)
    ry = np.rint(y)
    amb = ~(np.abs(y - ry) < np.float32(0.5) - np.float32(6e-5))
    got = np.clip(-ry, 0, 255).astype(np.int64)
    bad = (~amb) & (got != want)
    assert not bad.any(), (int(bad.sum()), np.flatnonzero(bad)[:5])
    assert amb.mean() < 0.02


def test_commit_loop_cell_arrangement_never_disagrees_unflagged():
    """numpy model of tlp_cell32 (kernels_commit_trimaran.hip) — the same real number as k_tlp_fast2's cell, arranged for the issue
    rates: branch picked on u's sign bit, coefficients and offsets scaled by 1/256 so that the fma's clamp modifier bounds the score
    to [0, 256], rounding by adding 1.5 * 2^23 — against the reference's float64 sequence (targetloadpacking.go:170-184)"""
    rng = np.random.default_rng(14)
    n = 4_000_000
    t = 40.0
    c1, c2 = t / (100.0 - t), (100.0 - t) / t
    cap = rng.choice(np.array([2000, 8000, 16000, 64000, 128000], dtype=np.float64), n)
    util = np.where(rng.random(n) < 0.3, rng.integers(0, 100, n).astype(np.float64), rng.uniform(0, 100, n))
    missing = np.where(rng.random(n) < 0.5, 0.0, rng.integers(0, 4000, n).astype(np.float64))
    pod = rng.integers(0, 12000, n).astype(np.float64)
    on_line = rng.random(n) < 0.05
    pod = np.where(on_line, np.floor(np.maximum(t * cap / 100.0 - (util / 100.0) * cap - missing, 0.0)), pod)
    overload = rng.random(n) < 0.1   # far beyond capacity: the unclamped value is very negative
    pod = np.where(overload, pod + 3 * cap, pod)
    um = (util / 100.0) * cap
    pred = 100.0 * ((um + pod) + missing) / cap
    x = np.where(pred > t, t * (100.0 - pred) / (100.0 - t), (100.0 - t) * pred / t + t)
    want = np.where(pred > 100.0, 0, np.floor(np.abs(x) + 0.5) * np.sign(x)).astype(np.int64)
    want = np.clip(want, 0, 255)
    k = 100.0 / cap
    b = (um + missing) - t * cap / 100.0
    bh = np.rint(b)
    b2h, b2l = _f32(bh), _f32(b - bh)
    s256 = np.float32(1.0 / 256.0)
    kc1, kc2 = _f32(-c1 * k) * s256, _f32(c2 * k) * s256   # the prologue scales the float32 constants (exact: a power of two)
    assert np.array_equal(kc1.astype(np.float64) * 256.0, _f32(-c1 * k).astype(np.float64))
    tfs, hs = np.float32(t) * s256, np.float32(100.0) / np.float32(256.0)
    u = (_f32(pod) + b2h) + b2l
    neg = np.signbit(u)
    coef, off = np.where(neg, kc2, kc1), np.where(neg, hs, tfs)
    xs = _f32(coef.astype(np.float64) * u.astype(np.float64) + off.astype(np.float64))
    xs = np.clip(xs, np.float32(0), np.float32(1))
    magic = np.float32(12582912.0)
    y = _f32(xs.astype(np.float64) * 256.0 + np.float64(magic))
    rr = y - magic
    d = np.abs(_f32(xs.astype(np.float64) * 256.0 - rr.astype(np.float64)))
    tb = y.view(np.uint32) & np.uint32(0x1ff)
    assert np.array_equal(tb.astype(np.float32), rr)
    amb = ~(d < np.float32(0.5) - np.float32(4e-5)) | ~(np.abs(u) > np.float32(1e-6))
    got = tb.astype(np.int64)
    bad = (~amb) & (got != want)
    assert not bad.any(), (int(bad.sum()), np.flatnonzero(bad)[:5])
    cont = ~on_line & (util != np.floor(util))
    assert amb[cont].mean() < 5e-4


def test_commit_loop_incremental_constant_tracks_the_float64_value():
    """k_commit_trimaran shifts a node's b2h by each bound pod's integer millicores instead of rebuilding b from the
    float64 columns: the represented real number b2h + b2l must stay within ~1e-9 of the rebuilt b (far inside the 4e-5
    ambiguity band), for as long as |b2h| < 2^23"""
    rng = np.random.default_rng(6)
    t = 40.0
    for _ in range(300):
        cap = float(rng.choice([2000, 8000, 64000, 128000]))
        util = float(rng.uniform(0, 100))
        missing = float(rng.integers(0, 3000))
        um = (util / 100.0) * cap
        b = (um + missing) - t * cap / 100.0
        bh = np.rint(b)
        b2h, b2l = np.float32(bh), np.float32(b - bh)
        for _ in range(40):
            pod = int(rng.integers(1, 4000))
            missing += pod
            nb = np.float32(b2h + np.float32(pod))
            if not abs(float(nb)) < 8388607.0:
                break
            b2h = nb
            rebuilt = (um + missing) - t * cap / 100.0
            assert abs((float(b2h) + float(b2l)) - rebuilt) < 1e-6 + abs(float(np.float32(b - bh)) - (b - bh)), (cap, util, missing)


def test_lroc_float32_quotient_is_exact_outside_the_ambiguity_band():
    """k_lroc_fast (kernels_lroc.hip), round 6: riskLimit = clamp(over / (D + d), 0, 1) with over = A + podLimit formed from the
    two-float32 images of A and podLimit (exact below 2^47; high parts, low parts, then both), D + d in float32, ONE 1-ulp
    reciprocal of the two denominators' product for both resources, w*q + (1-w)*riskLoad as one fma, and 100*(1 - max) in units
    of 2^-16 rounded by the sum with 2^23.  Claim: that value is within 8.7e-5 of the reference's float64 value, so a cell
    whose fraction is farther than 8 units (1.22e-4) from k + 0.5 rounds to the reference's score.  The reciprocal is emulated
    pessimistically: correctly rounded, then pushed one ulp either way."""
    rng = np.random.default_rng(12)
    n = 400_000
    f32, f64 = np.float32, np.float64
    cap_c = rng.choice(np.array([8, 16, 64, 128], dtype=np.int64), n) * 1000 - rng.integers(500, 2001, n)
    cap_m = rng.choice(np.array([32, 128, 512, 1024], dtype=np.int64), n) * (1 << 30)
    nreq_c = (rng.uniform(0, 1.5, n) * cap_c).astype(np.int64)
    nlim_c = nreq_c + (rng.uniform(0, 1.5, n) * cap_c * (rng.random(n) < 0.7)).astype(np.int64)
    nreq_m = (rng.uniform(0, 1.3, n) * cap_m).astype(np.int64)
    nlim_m = nreq_m + (rng.uniform(0, 0.6, n) * cap_m * (rng.random(n) < 0.5)).astype(np.int64)
    preq_c = rng.integers(0, 9000, n)
    plim_c = preq_c + rng.integers(0, 9000, n) * (rng.random(n) < 0.5)
    preq_m = rng.integers(0, 64 << 30, n)
    plim_m = preq_m + rng.integers(0, 32 << 30, n) * (rng.random(n) < 0.5)
    # adversarial slices: limits that exceed the capacity by a hair (tiny numerators: the sum A + podLimit cancels), denominators of
    # 0 and 1 (Guaranteed pods on nodes of Guaranteed pods), and the same for memory with odd byte counts near 2^46
    k = n // 20
    nlim_c[:k] = cap_c[:k] - plim_c[:k] + rng.integers(-3, 4, k)
    nreq_c[:k] = np.minimum(nreq_c[:k], nlim_c[:k].clip(0))
    nlim_c[:k] = np.maximum(nlim_c[:k], nreq_c[:k])
    cap_m[k:2 * k] = (1 << 46) + rng.integers(-(1 << 30), 1 << 30, k) * 2 + 1
    plim_m[k:2 * k] = rng.integers(1, 1 << 36, k) * 2 + 1
    preq_m[k:2 * k] = plim_m[k:2 * k] - rng.integers(0, 3, k)
    nlim_m[k:2 * k] = cap_m[k:2 * k] - plim_m[k:2 * k] + rng.integers(-3, 4, k)
    nreq_m[k:2 * k] = nlim_m[k:2 * k] - rng.integers(0, 3, k)
    load_c = rng.choice([0.0, 1.0, 0.5], n, p=[0.3, 0.2, 0.5]) * np.where(rng.random(n) < 0.5, 1.0, rng.random(n))
    load_m = rng.choice([0.0, 1.0, 0.5], n, p=[0.3, 0.2, 0.5]) * np.where(rng.random(n) < 0.5, 1.0, rng.random(n))

    def two(v):  # an integer column as the sum of two float32
        hi = v.astype(f64).astype(f32)
        lo = (v.astype(f64) - hi.astype(f64)).astype(f32)
        assert np.array_equal(hi.astype(f64) + lo.astype(f64), v.astype(f64))  # below 2^47: exact
        return hi, lo

    for w_c, w_m in [(0.5, 0.5), (0.9, 0.2), (0.0, 1.0), (1.0, 0.0), (0.37, 0.63)]:
        kl_c, kl_m = (1 - w_c) * load_c, (1 - w_m) * load_m

        def ref(w, kl, nreq, nlim, cap, preq, plim):  # lowriskovercommitment.go:205-208, :250-253 in float64 / int64
            limit = nlim + plim
            request = np.minimum(nreq + preq, cap)
            rl = np.where(limit > cap, (limit - cap).astype(f64) / np.maximum(limit - request, 1).astype(f64), 0.0)
            return np.clip(w * rl + kl, 0.0, 1.0)

        s64 = (1 - np.maximum(ref(w_c, kl_c, nreq_c, nlim_c, cap_c, preq_c, plim_c), ref(w_m, kl_m, nreq_m, nlim_m, cap_m, preq_m, plim_m))) * 100.0
        want = np.floor(s64 + 0.5).astype(np.int64)  # math.Round on a non-negative value

        def parts(nreq, nlim, cap, preq, plim):
            ah, al = two(nlim - cap)
            ph, pl = two(plim)
            ov = (ah + ph) + (al + pl)                                   # three float32 additions
            dd = np.maximum((nlim - nreq).astype(f64).astype(f32), f32(2.0 ** -30)) + (plim - preq).astype(f64).astype(f32)
            return ov, dd

        ov_c, dd_c = parts(nreq_c, nlim_c, cap_c, preq_c, plim_c)
        ov_m, dd_m = parts(nreq_m, nlim_m, cap_m, preq_m, plim_m)
        assert ov_c.dtype == f32 and dd_m.dtype == f32
        for bump in (0, 1, -1):
            with np.errstate(divide="ignore", invalid="ignore", over="ignore"):
                rr = f32(1.0) / (dd_c * dd_m)
                if bump:
                    rr = np.nextafter(rr, f32(np.inf) * f32(bump))
                q_c = np.clip((ov_c * dd_m) * rr, f32(0), f32(1))
                q_m = np.clip((ov_m * dd_c) * rr, f32(0), f32(1))
            assert not np.isnan(q_c).any() and not np.isnan(q_m).any()
            t_c = (f32(w_c).astype(f64) * q_c.astype(f64) + kl_c.astype(f32).astype(f64)).astype(f32)  # one fma: single rounding
            t_m = (f32(w_m).astype(f64) * q_m.astype(f64) + kl_m.astype(f32).astype(f64)).astype(f32)
            m = np.clip(np.maximum(t_c, t_m), f32(0), f32(1))
            kk = (m.astype(f64) * -6553600.0 + 14974984.0).astype(f32)     # fma(m, -100 * 2^16, 2^23 + 100 * 2^16 + 2^15 + 8)
            nn = kk.astype(np.int64) - (1 << 23)
            assert (nn >= 0).all() and (nn < (1 << 23)).all()
            s32 = (nn - 32776) / 65536.0
            assert np.abs(s32 - s64).max() < 8.7e-5, float(np.abs(s32 - s64).max())
            amb = (nn & 0xfff0) == 0
            bad = (~amb) & ((nn >> 16) != want)
            assert not bad.any(), (w_c, w_m, bump, int(bad.sum()))
            generic = np.abs(s64 - np.floor(s64) - 0.5) > 1e-9   # cells that do not sit on a boundary by construction (e.g. 0.63 * 0.5)
            assert amb[generic].mean() < 5e-3, (w_c, w_m, bump, float(amb[generic].mean()))


def test_reciprocal_division_is_correctly_rounded():
    """k_peaks (kernels_peaks.hip: div_rn) divides by a per-node / per-pod constant through its correctly rounded reciprocal:
    q = RN(a*y), r = a - b*q (one fma, exact), result RN(q + r*y).  Markstein's theorem says this is RN(a/b); replayed here
    against true division with exact rationals on the two shapes the kernel uses (predicted utilisation, min-max rescale)."""
    import random
    from fractions import Fraction
    rnd = random.Random(5)

    def div_rn(a, b):
        y = 1.0 / b
        q = a * y
        r = float(Fraction(a) - Fraction(b) * Fraction(q))   # fma(-b, q, a): float(Fraction) rounds correctly
        return float(Fraction(q) + Fraction(r) * Fraction(y))

    for _ in range(30000):
        cap = float(rnd.choice([8, 16, 32, 64, 96, 128]) * 1000 - rnd.randint(0, 2000))
        a = 100 * ((rnd.uniform(0, 100) / 100) * cap + float(rnd.randint(0, 9000)))
        assert div_rn(a, cap) == a / cap
    for _ in range(30000):
        mx = float(rnd.randint(1, 10 ** 17))
        span = mx - float(rnd.randint(0, int(mx)))
        if span == 0:
            continue
        d = float(rnd.randint(0, int(span)))
        assert div_rn(100.0 * d, span) == 100.0 * d / span
        assert div_rn(100.0 * span, span) == 100.0 * span / span   # the row maximum: 100 or 99.99999999999999, as in the reference


def test_balanced_allocation_divisions_are_correctly_rounded():
    """k_nrt_fast<., BalancedAllocation> (kernels_nrt_fast.hip: div_rn) computes fractionOfCapacity = request / capacity, the mean
    sum / n and stat.Variance's two divisions through correctly rounded reciprocals (Markstein).  Replayed with exact rationals on
    the operand shapes that occur: integer request / integer capacity below 2^42 (quotients above 1 included), sums and sums of
    squares of such fractions divided by n and n - 1 for n = 2..8."""
    import random
    from fractions import Fraction
    rnd = random.Random(11)

    def div_rn(a, b):
        y = 1.0 / b
        q = a * y
        r = float(Fraction(a) - Fraction(b) * Fraction(q))
        return float(Fraction(q) + Fraction(r) * Fraction(y))

    def rn_div(a, b):  # the correctly rounded quotient, independent of the platform's division
        return float(Fraction(a) / Fraction(b))

    caps = [float(c) for c in (1, 2, 3, 7, 96, 1000, 1023, 4095, 2 ** 20 - 1, 2 ** 30 + 1, 2 ** 41 + 12345, 2 ** 42 - 1)]
    for _ in range(40000):
        cap = rnd.choice(caps) if rnd.random() < 0.3 else float(rnd.randint(1, 2 ** rnd.randint(1, 42) - 1))
        req = float(rnd.randint(0, int(cap) * 2)) if rnd.random() < 0.7 else float(rnd.randint(0, 2 ** 42 - 1))
        assert div_rn(req, cap) == rn_div(req, cap) == req / cap
    for _ in range(40000):
        n = rnd.randint(2, 8)
        fr = [rnd.randint(0, 2 ** 30) / rnd.randint(1, 2 ** 30) for _ in range(n)]
        s = 0.0
        for f in fr:
            s += f
        assert div_rn(s, float(n)) == rn_div(s, float(n))
        mean = s / n
        ss = comp = 0.0
        for f in fr:
            d = f - mean
            ss += d * d
            comp += d
        c2 = comp * comp
        assert div_rn(c2, float(n)) == rn_div(c2, float(n))
        num = ss - c2 / n
        assert div_rn(num, float(n - 1)) == rn_div(num, float(n - 1))


def test_balanced_allocation_float32_score_is_exact_outside_the_band():
    """numpy model of score_balanced_f32 (kernels_nrt_fast.hip): fractions as float32 products with the float32 image of
    RN64(1/capacity), variance as (sum f^2 - (sum f)^2 / n) / (n - 1), score fma(-var, 100, 100) — against the reference's float64
    sequence (balanced_allocation.go:32-54 + gonum stat.Variance) on integer requests / capacities up to 2^40, n = 2..8 resources.
    Outside the band (kBalBand around integers; request within 2^-22 of the capacity) the truncated float32 score equals the reference's,
    and the float32 value itself stays within 2.3e-4 of the float64 one (the bound DESIGN.md 3.4 derives)."""
    rng = np.random.default_rng(21)
    cells = 400_000
    band = np.float32(3e-4)
    worst = worst_walk = 0.0
    undecided = plain_cells = 0
    for n in range(2, 9):
        mag = rng.integers(1, 41, (cells, n))
        cap = np.floor(rng.random((cells, n)) * (2.0 ** mag)) + 1.0                      # capacities 1 .. 2^40
        kind = rng.random((cells, n))
        req = np.where(kind < 0.95, np.floor(rng.random((cells, n)) * cap),               # 0 <= request < capacity
                       np.where(kind < 0.97, cap,                                         # exactly full
                                np.where(kind < 0.98, cap + 1.0, np.floor(cap * 0.5))))    # just over; exactly half
        plain = (kind < 0.95).all(axis=1)
        nocap = rng.random((cells, n)) < 0.05                                              # capacity <= 0: the reference's f = 1
        # ---- reference, float64, operation for operation
        f = np.where(nocap, 1.0, req / cap)
        over = (f > 1.0).any(axis=1)
        s = np.zeros(cells)
        for i in range(n):
            s = s + f[:, i]
        mean = s / n
        ss = np.zeros(cells)
        comp = np.zeros(cells)
        for i in range(n):
            d = f[:, i] - mean
            ss = ss + d * d
            comp = comp + d
        var = (ss - comp * comp / n) / (n - 1)
        score64 = (1.0 - var) * 100.0
        want = np.where(over, 0, np.trunc(score64)).astype(np.int64)
        # ---- float32 form
        rcp = np.where(nocap, np.float32(0), _f32(1.0 / cap))
        one = np.where(nocap, np.float32(1), np.float32(0)).astype(np.float32)
        ff = _f32(_f32(req).astype(np.float64) * rcp.astype(np.float64) + one.astype(np.float64))   # fma
        capf = np.where(nocap, np.float32(1e38), _f32(cap)).astype(np.float32)
        dd = (_f32(req) - capf).astype(np.float32)
        mxd = dd.max(axis=1)
        # slots not known to stay below 2^24: undecided when request and capacity lie within 2^-22 of each other
        nr = (_f32(capf.astype(np.float64) * 2.4e-7 - np.abs(dd).astype(np.float64))).max(axis=1)
        sm = np.zeros(cells, np.float32)
        sq = np.zeros(cells, np.float32)
        for i in range(n):
            sm = (sm + ff[:, i]).astype(np.float32)
            sq = _f32(ff[:, i].astype(np.float64) * ff[:, i].astype(np.float64) + sq.astype(np.float64))       # fma
        rn_, rm_ = np.float32(1.0) / np.float32(n), np.float32(1.0) / np.float32(n - 1)
        ssq = (sm * sm).astype(np.float32)
        v32 = (_f32(-(ssq.astype(np.float64)) * np.float64(rn_) + sq.astype(np.float64)) * rm_).astype(np.float32)
        sc = _f32(-(v32.astype(np.float64)) * 100.0 + 100.0)
        near_one = nr >= 0
        valid = ~(mxd > 0)
        # where the float32 comparison is taken as decided it is the exact one (the reference's f > 1.0 <=> request > capacity)
        assert (valid[~near_one] == ~over[~near_one]).all()
        fl = np.floor(sc)
        frac = sc - fl
        redo = near_one | (valid & ((frac < band) | (frac > np.float32(1) - band)))
        got = np.where(valid, fl, 0).astype(np.int64)
        ok = ~redo
        assert (got[ok] == want[ok]).all(), (n, np.flatnonzero(ok & (got != want))[:5])
        both = valid & ~over
        worst = max(worst, float(np.abs(sc[both].astype(np.float64) - score64[both]).max()))
        undecided += int((redo & plain).sum())
        plain_cells += int(plain.sum())
        # ---- the walk's form (fz_score_item_bal, kernels_nrt_fused.hip): the fraction as a clamped product (+inf for "no capacity"),
        # "request > capacity" exact (the Filter's rank-space bits; whole cores for cpu), the variance's last two steps as one fma with
        # RN32(-100 / (n - 1))
        reqz = np.where(req == 0, np.float32(1e-30), _f32(req)).astype(np.float32)  # (a zero request travels as 1e-30: 0 * inf would be NaN)
        fz = np.clip(reqz * np.where(nocap, np.float32(np.inf), _f32(1.0 / cap)).astype(np.float32), np.float32(0), np.float32(1))
        assert not np.isnan(fz).any()
        sm = np.zeros(cells, np.float32)
        sq = np.zeros(cells, np.float32)
        for i in range(n):
            sm = (sm + fz[:, i]).astype(np.float32)
            sq = _f32(fz[:, i].astype(np.float64) * fz[:, i].astype(np.float64) + sq.astype(np.float64))
        m100rm = np.float32(-100.0) * rm_
        t2 = _f32(-((sm * sm).astype(np.float32).astype(np.float64)) * np.float64(rn_) + sq.astype(np.float64))
        scz = _f32(t2.astype(np.float64) * np.float64(m100rm) + 100.0)
        flz = np.floor(scz)
        fracz = scz - flz
        redoz = ~over & ((fracz < band) | (fracz > np.float32(1) - band))
        gotz = np.where(~over, flz, 0).astype(np.int64)
        okz = ~redoz
        assert (gotz[okz] == want[okz]).all(), (n, np.flatnonzero(okz & (gotz != want))[:5])
        sel = ~over
        worst_walk = max(worst_walk, float(np.abs(scz[sel].astype(np.float64) - score64[sel]).max()))
    assert worst < 2.3e-4, worst
    assert worst_walk < 2.3e-4, worst_walk
    # 2 * kBalBand of the cells, the rare fraction next to 1, and the exactly integer scores small capacities produce (all fractions 0 or equal)
    assert undecided < 5e-3 * plain_cells, (undecided, plain_cells)


# ---------------------------------------------------------------------------------------------------------------------------
# k_peaks_minmax_est / k_peaks_write_est (kernels_peaks.hip + peaks_est.h, SPX_OPT_PEAKS_ESTIMATE): the float32 interval of a cell's raw score
# contains the value of the float64 sequence; the cells the two passes leave to raw_score are a superset of the cells that matter.
# The replay below is the kernels' arithmetic in numpy float32 (fma through float64: the product of two float32 is exact there),
# with v_exp_f32 modelled as the correctly rounded 2^y perturbed by up to 3 ulp either way.

def _peaks_consts():
    import re
    from pathlib import Path
    src = (Path(__file__).resolve().parent.parent / "scheduler-plugins_amd" / "csrc" / "peaks_est.h").read_text()

    def c(name):
        return re.search(r"\b" + name + r" = ([^,;]+)[,;]", src).group(1).strip()

    assert (c("kEstBeta"), c("kEstAlpha")) == ("5.0f * 0x1p-24f", "12.0f * 0x1p-24f")
    assert (float(c("kEstHuge").rstrip("f")), float(c("kEstBpInv").rstrip("f")), float(c("kEstGc").rstrip("f"))) == (1e38, 1024.0, 102400.5)
    assert (float(c("kEstUtilMax")), float(c("kEstYMax")), float(c("kEstMagMin")), float(c("kEstMagMax")), c("kEstYClamp")) == (400.0, 40.0, 1e7, 1e24, "41.0f")
    f = np.float32
    return f(5.0 * 2.0 ** -24), f(12.0 * 2.0 ** -24), f(1e38), f(1024.0), f(102400.5)


def _peaks_fma32(a, b, c):
    """fmaf on float32 arrays, correctly rounded: the product of two float32 is exact in float64; the sum is taken with its rounding error
    (TwoSum), and where the float64 sum sits exactly half-way between two float32 the error decides the direction (the float64 -> float32
    conversion alone would round such a tie to even, whichever side the exact value is on)"""
    f = np.float32
    with np.errstate(over="ignore", invalid="ignore"):  # (an interval end beyond float32 becomes inf, as on the device)
        pr = np.asarray(a, np.float64) * np.asarray(b, np.float64)
        cc = np.broadcast_to(np.asarray(c, np.float64), pr.shape)
        sm = pr + cc
        bb = sm - pr
        err = (pr - (sm - bb)) + (cc - bb)
        r = sm.astype(f)
        d = sm - r.astype(np.float64)
        tie = np.isfinite(r) & (np.abs(d) * 2 == np.spacing(np.abs(r)).astype(np.float64)) & (err != 0) & np.isfinite(err)
        if tie.any():
            up = np.where(d > 0, np.nextafter(r, f(np.inf)), r)
            dn = np.where(d > 0, r, np.nextafter(r, f(-np.inf)))
            r = np.where(tie, np.where(err > 0, up, dn), r)
        return r


def _peaks_node_consts(cap, util, k1, k2, valid):
    """est_node_compute: (c0, c1, ql, ke, sigma) per node and which nodes are outside the preconditions"""
    cap = cap.astype(np.float64)
    with np.errstate(all="ignore"):
        util_m = (util / 100) * cap
        c0 = np.where(cap != 0, 100 * util_m / np.where(cap != 0, cap, 1), 0.0)
        c1 = np.where(cap != 0, 100 / np.where(cap != 0, cap, 1), 0.0)
        ql = k2 * c1 * 1.4426950408889634
        ke = k1 * 1e15 * np.exp(k2 * util)
        dmax = 101.0 + np.abs(util)
        ymax = np.abs(k2) * 1.4426950408889634 * dmax
        sigma = np.where(k1 == 0, 0.0, 2.0)
        fin = np.isfinite(c0) & np.isfinite(c1) & np.isfinite(ql) & np.isfinite(ke) & np.isfinite(sigma)
        mag = (k1 == 0) | ((np.abs(ke) >= 1e7) & (np.abs(ke) <= 1e24))
        tame = (cap > 0) & fin & (np.abs(util) <= 400.0) & (ymax <= 40.0) & mag
    z = ~valid | ~tame
    f = np.float32
    with np.errstate(all="ignore"):
        return (np.where(~valid, 0.0, np.where(tame, c0, 100.0)).astype(f), np.where(z, 0.0, c1).astype(f), np.where(z, 0.0, ql).astype(f),
                np.where(z | (k1 == 0), 0.0, ke).astype(f), np.where(z, 0.0, sigma).astype(f)), valid & ~tame


def _peaks_interval_from(consts, pod32, y, e):
    beta, alpha, huge, bpinv, gc = _peaks_consts()
    c0, c1, ql, ke, sigma = consts
    f = np.float32
    p = _peaks_fma32(c1, pod32, c0)
    est = (ke * (e - f(1))).astype(f)
    w = _peaks_fma32(np.abs(ke), e, np.abs(ke))
    b = _peaks_fma32(w, _peaks_fma32(np.abs(y), beta, alpha), sigma)
    with np.errstate(invalid="ignore"):
        g = np.fmin(np.fmax(_peaks_fma32(p, -bpinv, np.broadcast_to(gc, p.shape)), f(0)), f(1)).astype(f)
    am = _peaks_fma32(-g, g, g)
    bg = _peaks_fma32(b, g, (am * huge).astype(f))
    eg = (est * g).astype(f)
    return (eg - bg).astype(f), (eg + bg).astype(f)


def _peaks_interval(consts, pod32, rng, ulps=3):
    f = np.float32
    with np.errstate(all="ignore"):
        y = np.fmax(np.fmin((consts[2] * pod32).astype(f), f(41.0)), f(-200.0))
    e = np.exp2(y.astype(np.float64)).astype(f)
    k = np.where(e > f(1e-30), rng.integers(-ulps, ulps + 1, size=e.shape), 0).astype(np.int32)  # (v_exp_f32 flushes below 2^-126: exactly 0 there)
    e = (e.view(np.int32) + k).view(f)
    return _peaks_interval_from(consts, pod32, y, e)


def _peaks_exact(cap, util, k1, k2, valid, pod):
    cap = cap.astype(np.float64)
    with np.errstate(all="ignore"):
        util_m = (util / 100) * cap
        pred = np.where(cap != 0, 100 * (util_m + pod) / np.where(cap != 0, cap, 1), 0.0)
        v = np.trunc(k1 * (np.exp(k2 * pred) - np.exp(k2 * util)) * 1e15)
    return np.where(valid & ~(pred > 100), v, 0.0)


def _peaks_snapshot(rng, n_nodes, n_pods, regime):
    cap = rng.choice([1000, 2000, 4000, 8000, 16000, 64000, 96000, 128000, 3, 250], n_nodes).astype(np.int64)
    util = rng.uniform(0, 100, n_nodes)
    k1 = -rng.uniform(40, 160, n_nodes)
    k2 = -rng.uniform(0.02, 0.12, n_nodes)
    valid = rng.random(n_nodes) > 0.03
    k1[rng.random(n_nodes) < 0.05] = 0.0  # nodes without a power model (peaks.go:190-196)
    k2[k1 == 0] = 0.0
    pod = rng.integers(0, 4000, n_pods).astype(np.int64)
    if regime == "near100":  # predicted on / next to 100 for one of the pods
        i = rng.integers(0, n_pods, n_nodes)
        util = 100 - 100.0 * pod[i] / cap + rng.choice([0, 1e-9, -1e-9, 1e-6, -1e-6, 1e-4, -1e-4, 3e-4, -3e-4, 1e-3], n_nodes)
    elif regime == "wild":
        k1 = rng.choice([0.0, 1e-12, -1e-9, 5.0, -3000.0, 1e6, -1e12, np.nan, np.inf], n_nodes)
        k2 = rng.choice([0.0, -0.07, 0.07, -3.0, 2.0, 1e-9, -40.0, np.nan], n_nodes)
        util = rng.choice([0.0, 50.0, 100.0, 150.0, -20.0, 1e6, np.nan], n_nodes)
        cap = rng.choice([0, 1, 1000, 64000, 10 ** 9], n_nodes).astype(np.int64)
        pod = rng.choice([0, 1, 500, 10 ** 5, 10 ** 8, 3 * 10 ** 7 + 1], n_pods).astype(np.int64)
    elif regime == "steep":  # exponents up to the precondition's limit, both signs
        k1 = -rng.uniform(40, 160, n_nodes) * rng.choice([1, -1, 1e-3, 1e3], n_nodes)
        k2 = rng.uniform(-0.27, 0.27, n_nodes)
        cap = rng.choice([1000, 4000, 64000, 128000], n_nodes).astype(np.int64)
    elif regime == "extreme_pod":  # requests up to int64's end against tame nodes: |y| unclamped would carry the bound B past FLT_MAX (advisor, round 5)
        pod = rng.choice([0, 1, 4000, 2 ** 40, 2 ** 55, 2 ** 62, 2 ** 63 - 1], n_pods).astype(np.int64)
        cap = rng.choice([1, 3, 1000, 64000], n_nodes).astype(np.int64)
        k2 = rng.choice([-0.27, -0.07, 0.07, 0.27], n_nodes)
    elif regime == "zero_cpu":
        pod[:] = 0
    elif regime == "identical":
        cap[:], util[:], k1[:], k2[:] = 64000, 37.5, -91.5, -0.0718
    return cap, util, k1, k2, valid, pod


def test_peaks_interval_contains_the_float64_score():
    worst = 0.0
    for regime in ("plain", "near100", "wild", "steep", "zero_cpu", "identical", "extreme_pod"):
        for seed in range(2):
            rng = np.random.default_rng(100 + seed)
            cap, util, k1, k2, valid, pod = _peaks_snapshot(rng, 2500, 500, regime)
            consts, forced = _peaks_node_consts(cap, util, k1, k2, valid)
            lo, hi = _peaks_interval(tuple(x[None, :] for x in consts), pod.astype(np.float32)[:, None], rng)
            v = _peaks_exact(cap[None, :], util[None, :], k1[None, :], k2[None, :], valid[None, :], pod[:, None].astype(np.float64))
            forced = np.broadcast_to(forced[None, :], v.shape)
            # a node outside the preconditions is undecided in every row: the widest interval the kernel ever forms
            assert (hi[forced] > np.float32(1e37)).all() and (lo[forced] < np.float32(-1e37)).all()
            ok = ~forced
            assert np.isfinite(v[ok]).all(), regime
            assert np.isfinite(lo[ok]).all() and np.isfinite(hi[ok]).all(), regime  # a NaN interval would read as "outside the table", not "undecided"
            assert ((lo.astype(np.float64) <= v) & (v <= hi.astype(np.float64)))[ok].all(), regime
            known = (lo == hi) & ok
            assert (v[known] == 0).all() and (lo[known] == 0).all(), regime  # lo == hi happens only at 0 and then IS the score
            sel = ok & ~known & (hi < np.float32(1e30))
            if sel.any():
                half = (hi.astype(np.float64) - lo.astype(np.float64))[sel] / 2
                mid = (hi.astype(np.float64) + lo.astype(np.float64))[sel] / 2
                worst = max(worst, float((np.abs(v[sel] - mid) / half).max()))
    assert worst < 0.5, worst  # the bound has a factor of two to spare against a 3-ulp exp2


def test_peaks_estimate_passes_decide_what_the_float64_passes_decide():
    """pass 1: the extremes of every (row, 1024-node tile) are among the undecided cells + the known zeros; pass 2: a cell the interval
    decides gets the byte of the float64 NormalizeScore"""
    f = np.float32
    for regime in ("plain", "near100", "steep", "zero_cpu", "identical"):
        rng = np.random.default_rng(7)
        cap, util, k1, k2, valid, pod = _peaks_snapshot(rng, 2048, 300, regime)
        consts, forced = _peaks_node_consts(cap, util, k1, k2, valid)
        lo, hi = _peaks_interval(tuple(x[None, :] for x in consts), pod.astype(f)[:, None], rng)
        v = _peaks_exact(cap[None, :], util[None, :], k1[None, :], k2[None, :], valid[None, :], pod[:, None].astype(np.float64))
        feas = rng.random(v.shape) > (0.3 if regime != "identical" else 0.0)
        feas[:, :5] = True
        lo_m, hi_m = np.where(feas, lo, f(np.nan)), np.where(feas, hi, f(np.nan))
        n_und = 0
        for t0 in range(0, v.shape[1], 1024):
            sl = slice(t0, t0 + 1024)
            with np.errstate(all="ignore"):
                maxlo = np.fmax.reduce(lo_m[:, sl], axis=1, initial=f(-np.inf))[:, None]
                minhi = np.fmin.reduce(hi_m[:, sl], axis=1, initial=f(np.inf))[:, None]
                und = ((hi_m[:, sl] >= maxlo) | (lo_m[:, sl] <= minhi)) & (hi_m[:, sl] > lo_m[:, sl])
                zero = (lo_m[:, sl] == hi_m[:, sl])
            vv = v[:, sl]
            # (a node outside the preconditions may score NaN — exp overflow; fmin / fmax skip it in the kernels, old and new)
            got_mn = np.fmin(np.fmin.reduce(np.where(und, vv, np.inf), axis=1), np.where(zero.any(axis=1), 0.0, np.inf))
            got_mx = np.fmax(np.fmax.reduce(np.where(und, vv, -np.inf), axis=1), np.where(zero.any(axis=1), 0.0, -np.inf))
            want_mn, want_mx = np.fmin.reduce(np.where(feas[:, sl], vv, np.inf), axis=1), np.fmax.reduce(np.where(feas[:, sl], vv, -np.inf), axis=1)
            assert (got_mn == want_mn).all() and (got_mx == want_mx).all(), regime
            n_und += int((und & ~forced[None, sl]).sum())
        if regime == "plain":
            assert n_und < 0.01 * feas.sum(), n_und  # a few cells per (row, tile) next to the nodes outside the preconditions, not the tile
        # ---- pass 2
        mn = np.fmin.reduce(np.where(feas, v, np.inf), axis=1)
        mx = np.fmax.reduce(np.where(feas, v, -np.inf), axis=1)
        gen = ~((mn == 0) & (mx == 0)) & (mn != mx) & (np.abs(mn) < 2.0 ** 62) & (np.abs(mx) < 2.0 ** 62)  # (the row statistic is an int64 pair)
        span = np.where(gen, mx - mn, 1.0)
        with np.errstate(all="ignore"):
            want = 100 - np.trunc(100.0 * (v - mn[:, None]) / span[:, None])
        r = 100.0 / span
        mr = mn * r
        en = (6.0 * 2.0 ** -24 * (np.abs(mr) + 101.0)).astype(f)
        clo, chi = (-mr).astype(f) - en, (-mr).astype(f) + en
        nlo = _peaks_fma32(lo, r.astype(f)[:, None], clo[:, None])
        nhi = _peaks_fma32(hi, r.astype(f)[:, None], chi[:, None])
        with np.errstate(all="ignore"):
            klo = np.minimum(np.fmax(nlo, f(0)), f(1e6)).astype(np.uint32)
            khi = np.minimum(np.fmax(nhi, f(0)), f(1e6)).astype(np.uint32)
        decided = (klo == khi) & feas & gen[:, None]
        got = 100 - np.minimum(khi, 100).astype(np.int64)
        assert (got[decided] == want[decided]).all(), regime
        if regime == "plain":
            assert decided[:, ~forced].sum() > 0.97 * (feas & gen[:, None])[:, ~forced].sum()


# ---- the same arithmetic from the product's own source: csrc/peaks_est.h compiled for the host (tests/cpp/peaks_est_check.cc)
def _peaks_header_lib():
    import ctypes as C
    import subprocess
    from pathlib import Path
    root = Path(__file__).resolve().parent.parent
    src, hdr = root / "tests" / "cpp" / "peaks_est_check.cc", root / "scheduler-plugins_amd" / "csrc" / "peaks_est.h"
    lib = root / "tests" / "cpp" / "_build" / "libpeaks_est_check.so"
    if not lib.exists() or lib.stat().st_mtime < max(src.stat().st_mtime, hdr.stat().st_mtime):
        lib.parent.mkdir(exist_ok=True)
        subprocess.run(["g++", "-std=c++17", "-O2", "-fPIC", "-shared", "-ffp-contract=off", "-Wall", str(src), "-o", str(lib), "-lm"], check=True)
    L = C.CDLL(str(lib))
    dp, fp, bp, i64 = C.POINTER(C.c_double), C.POINTER(C.c_float), C.POINTER(C.c_uint8), C.c_int64
    L.peaks_est_check_nodes.argtypes = [i64, dp, dp, dp, dp, dp, bp, fp]
    L.peaks_est_check_exponents.argtypes = [i64, i64, fp, fp, fp]
    L.peaks_est_check_intervals.argtypes = [i64, i64, fp, fp, fp, fp, fp]
    L.peaks_est_check_intervals_host_exp.argtypes = [i64, i64, fp, fp, fp, fp]
    return L, (dp, fp, bp)


def test_peaks_header_computes_what_the_replay_computes():
    """peaks_est.h — the source the kernels are compiled from — against the numpy replay above, bit for bit: the per-node constants
    (incl. which nodes are outside the preconditions), the exponent handed to the exponential, and the interval for a given e; so what
    the two tests above establish for the replay holds for the device code's arithmetic, v_exp_f32's 3 ulp aside"""
    L, (dp, fp, bp) = _peaks_header_lib()
    f = np.float32
    cells = 0
    for regime in ("plain", "near100", "wild", "steep", "zero_cpu", "identical", "extreme_pod"):
        rng = np.random.default_rng(11)
        cap, util, k1, k2, valid, pod = _peaks_snapshot(rng, 1500, 300, regime)
        consts, forced = _peaks_node_consts(cap, util, k1, k2, valid)
        n, m = len(cap), len(pod)
        with np.errstate(all="ignore"):
            e_now = np.exp(k2 * util)
        capd, v8 = cap.astype(np.float64), valid.astype(np.uint8)
        got = np.zeros((n, 5), f)
        L.peaks_est_check_nodes(n, capd.ctypes.data_as(dp), np.ascontiguousarray(util).ctypes.data_as(dp), e_now.ctypes.data_as(dp),
                                np.ascontiguousarray(k1, np.float64).ctypes.data_as(dp), np.ascontiguousarray(k2, np.float64).ctypes.data_as(dp), v8.ctypes.data_as(bp), got.ctypes.data_as(fp))
        want = np.stack(consts, axis=1)
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), (regime, np.nonzero((got.view(np.uint32) != want.view(np.uint32)).any(axis=1))[0][:5])
        assert np.array_equal((got[:, 0] == 100) & (got[:, 1:] == 0).all(axis=1) & valid, forced | ((want[:, 0] == 100) & (want[:, 1:] == 0).all(axis=1) & valid))
        pod32 = pod.astype(f)
        y = np.zeros((m, n), f)
        L.peaks_est_check_exponents(m, n, got.ctypes.data_as(fp), pod32.ctypes.data_as(fp), y.ctypes.data_as(fp))
        with np.errstate(all="ignore"):
            y_want = np.fmax(np.fmin((consts[2][None, :] * pod32[:, None]).astype(f), f(41.0)), f(-200.0))
        assert np.array_equal(y.view(np.uint32), y_want.view(np.uint32)), regime
        with np.errstate(all="ignore"):
            e = np.exp2(y.astype(np.float64)).astype(f)
        k = np.where(e > f(1e-30), rng.integers(-3, 4, size=e.shape), 0).astype(np.int32)
        e = (e.view(np.int32) + k).view(f)
        lo, hi = np.zeros((m, n), f), np.zeros((m, n), f)
        L.peaks_est_check_intervals(m, n, got.ctypes.data_as(fp), pod32.ctypes.data_as(fp), e.ctypes.data_as(fp), lo.ctypes.data_as(fp), hi.ctypes.data_as(fp))
        lo_w, hi_w = _peaks_interval_from(tuple(x[None, :] for x in consts), pod32[:, None], y, e)
        for a, b in ((lo, lo_w), (hi, hi_w)):
            same = (a.view(np.uint32) == b.view(np.uint32)) | (np.isnan(a) & np.isnan(b))
            assert same.all(), (regime, int((~same).sum()))
        cells += m * n
    assert cells > 2_000_000
