"""The quantised KNN body's certificate (csrc/knn.hip: refine_q8_body) is float32 arithmetic "pushed outwards by 4e-6 at every use"
(docs/knn.md) — padding chosen by hand.  This CPU test restates those few float32 expressions in NumPy, in the
kernel's operation order, and holds them — over a few million random states covering every magnitude the kernel admits — to the
bounds the MATHEMATICS asks for, evaluated in float64 with the true rho = 2^-19 and with one float64 ulp of slack:

    select    R_f32    >= (sqrt U + e)(1 + rho) + eta                the answer's second distance is <= R            (units of s)
              Dlim_f32 >= ((R + eta)(1 + 2 rho) + e)^2               rows of the answer have D <= Dlim
              thr      :  every record key with acc <= (Dlim - cq) / 2 - base is listed
    certify   lowb_f32 <= (sqrt Dlow - e)(1 - rho) - eta             what a stream's discarded rows are at least
              and the comparison  lowb > d2 / s * (1 + 1e-6)  never accepts when the exact one would not

so that the hand padding (kQ8Rho = 4e-6 for rho = 1.9e-6, the 2e-7 / 1e-6 factors) provably dominates the roundings of the bound
arithmetic itself.  Test infrastructure: nothing here is imported by the product; the expressions are copied by reading
knn.hip:2860-2925 and :3055-3061 and must be kept in step with it (the constants are parsed from the source)."""
import os
import re

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
f32 = np.float32


def _constants():
    src = open(os.path.join(ROOT, "sfm_mvs_amd", "csrc", "knn.hip")).read()
    m = re.search(r"constexpr float kQ8Rho = ([0-9.e+-]+)f, kQ8Eta = ([0-9.e+-]+)f;", src)
    assert m, "kQ8Rho / kQ8Eta not found in knn.hip"
    # the expressions this file restates must still be there, verbatim
    for frag in ("const float r = (x + eta_s) * (1.f + 2.f * kQ8Rho) + e_s;", "return r * r * (1.f + 1e-6f) + 2.f;",
                 "* (1.f + 2e-7f) + e_s) * (1.f + kQ8Rho) + eta_s;", "* (1.f - 2e-7f) - e_s) * (1.f - kQ8Rho) - eta_s;",
                 "return lowb > d2c * inv_s * (1.f + 1e-6f);", "const long long al = (((long long)dlim - cq) >> 1) - base + 1;",
                 "const float E = (sqrtf(qe2) + sqrtf(te2)) * (1.f + 8e-6f) + 24.f * 5.9604645e-08f * mabs * (1.f + 1e-6f);",
                 "e_s = E * inv_s * (1.f + 1e-6f);"):
        assert frag in src, f"refine_q8_body changed: {frag!r} — update tests/test_q8_bounds.py"
    return f32(m.group(1)), f32(m.group(2))


def test_float32_bounds_dominate_the_exact_ones():
    rho32, eta32 = _constants()
    # rho: <= 14 roundings of 2^-24 in the direct-form sum and the square root; eta: what a float32 sum that underflows can lose —
    # sqrt(FLT_MIN) = 1.1e-19; the kernel carries 1e-17, the requirement checked here is 1e-18 (float32 cannot hold 1e-17 / s exactly)
    rho, eta = 2.0 ** -19, 1e-18
    rng = np.random.default_rng(1)
    n = 2_000_000
    # states: the pair's grid step s over 40 decades, residual sums from "exact u8" (0) to a few hundred s^2, integer scores up to 2^23
    s = (10.0 ** rng.uniform(-11, 29, n)).astype(f32)
    lo = (rng.uniform(-1, 1, n) * 10.0 ** rng.uniform(-3, 3, n) * s.astype(np.float64) * 255).astype(f32)
    with np.errstate(over="ignore"):
        qe2 = (rng.uniform(0, 1, n) ** 2 * 300 * s.astype(np.float64) ** 2).astype(f32)
    with np.errstate(over="ignore"):
        te2 = (rng.uniform(0, 1, n) ** 2 * 300 * s.astype(np.float64) ** 2).astype(f32)
    zero = rng.random(n) < 0.1
    qe2[zero] = 0; te2[zero] = 0
    U = np.floor(2.0 ** rng.uniform(0, 23.2, n)).astype(np.int64)
    U[rng.random(n) < 0.02] = 0
    with np.errstate(over="ignore", invalid="ignore"):
        inv_s = f32(1) / s
        eta_s = eta32 * inv_s
        mabs = np.maximum(np.abs(lo), np.abs(lo + f32(255) * s))
        E = (np.sqrt(qe2) + np.sqrt(te2)) * (f32(1) + f32(8e-6)) + f32(24) * f32(5.9604645e-08) * mabs * (f32(1) + f32(1e-6))
        e_s = E * inv_s * (f32(1) + f32(1e-6))
        e_s = np.where(e_s < f32(1e30), e_s, f32(np.inf))
    ok = np.isfinite(e_s) & np.isfinite(eta_s) & (e_s < 5e3)          # (larger slacks: nothing is skipped or certified — no claim to check)
    # the EXACT slack in units of s from the same measured inputs (what the mathematics calls e): float64
    s64, inv64 = s.astype(np.float64), 1.0 / s.astype(np.float64)
    e_true = (np.sqrt(qe2.astype(np.float64)) + np.sqrt(te2.astype(np.float64)) + 24 * 2.0 ** -24 * mabs.astype(np.float64)) * inv64
    eta_true = eta * inv64
    assert np.all(e_s.astype(np.float64)[ok] >= e_true[ok]), "e_s (float32) below the exact slack"
    tiny = 1e-37      # below float32's normal range (eta / s for s > 1e19): an absolute slack no float32 distance in units of s can feel
    assert np.all(eta_s.astype(np.float64)[ok] + tiny >= eta_true[ok])
    # ---- select
    sqU = np.sqrt(U.astype(f32))
    R = (sqU * (f32(1) + f32(2e-7)) + e_s) * (f32(1) + rho32) + eta_s
    R_true = (np.sqrt(U.astype(np.float64)) + e_true) * (1 + rho) + eta_true
    assert np.all(R.astype(np.float64)[ok] + tiny >= R_true[ok]), "R (float32) below the exact bound"
    r = (R + eta_s) * (f32(1) + f32(2) * rho32) + e_s
    dl = r * r * (f32(1) + f32(1e-6)) + f32(2)
    dl_true = ((R_true + eta_true) * (1 + 2 * rho) + e_true) ** 2
    lim = ok & (dl < f32(1.6e7))
    assert np.all(dl.astype(np.float64)[lim] >= dl_true[lim]), "Dlim (float32) below the exact bound"
    dlim = dl[lim].astype(np.int64)                                      # (int) truncation, as the kernel
    assert np.all(dlim + 1 > dl_true[lim]), "the truncated Dlim excludes an integer D the exact bound admits"
    # thr: a row with integer D <= Dlim_true has acc <= (D - cq) / 2 - base (from D >= cq + 2 (acc + base)); the kernel lists acc <= al
    m = int(lim.sum())
    cq = rng.integers(-128, 1 << 22, m)
    base = rng.integers(-(1 << 19), 1 << 19, m)
    al = ((dlim - cq) >> 1) - base + 1
    D = np.floor(dl_true[lim]).astype(np.int64)                          # the largest integer D the exact bound admits
    acc_max = np.floor((D - cq) / 2.0).astype(np.int64) - base           # the largest acc such a row can carry
    assert np.all(al >= acc_max), "a record the exact bound admits is not listed"
    # ---- certify
    dlow = np.floor(2.0 ** rng.uniform(0, 23.2, n)).astype(np.int64)
    lowb = (np.sqrt(dlow.astype(f32)) * (f32(1) - f32(2e-7)) - e_s) * (f32(1) - rho32) - eta_s
    lowb_true = (np.sqrt(dlow.astype(np.float64)) - e_true) * (1 - rho) - eta_true
    assert np.all(lowb.astype(np.float64)[ok] <= lowb_true[ok] + tiny), "lowb (float32) above the exact lower bound"
    # the decision: d2 (a float32 distance, any magnitude) converted to units of s and padded — accept only if the exact test accepts
    with np.errstate(over="ignore", invalid="ignore"):
        d2 = (np.abs(lowb_true) * s64 * (1 + rng.normal(0, 3e-6, n))).astype(f32)
        rhs = d2 * inv_s * (f32(1) + f32(1e-6))
    acc = ok & np.isfinite(rhs) & (lowb > rhs)
    assert np.all(lowb_true[acc] > d2.astype(np.float64)[acc] * inv64[acc]), "a stream certified that the exact test rejects"
    assert acc.sum() > 1000 and (ok & ~acc).sum() > 1000                 # both outcomes occur near the boundary


def test_the_check_above_is_sharp(monkeypatch):
    """... and it is not vacuous: with kQ8Rho below the true rho the same check fails."""
    import pytest
    import sys
    mod = sys.modules[__name__]
    monkeypatch.setattr(mod, "_constants", lambda: (f32(1e-6), f32(1e-17)))
    with pytest.raises(AssertionError, match="below the exact bound"):
        test_float32_bounds_dominate_the_exact_ones()
