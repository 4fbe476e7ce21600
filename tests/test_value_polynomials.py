"""Value polynomials (csrc/mcx_vpoly.hip): the exercise value of a Bermudan swaption — the underlying swap priced from ~35-64
zero-bond requests per exercise date (reference products/bermudan_option.py:40-43, 93-131; products/bond.py:42-68, 115-163) — is one
smooth function of the short rate; the library replaces the per-path term loop by a host-verified polynomial.  Here: the fit and its
bound without a GPU, and on the GPU the collapsed kernels against the exact term loops and the oracle."""
import ctypes as C

import numpy as np
import pytest

import cases


class _VT(C.Structure):
    _fields_ = [(n, C.c_double) for n in ("w", "a", "d", "b", "c0", "c1")]


def _fit(lib, terms, lo, hi, tol=1e-14, max_degree=31):
    arr = (_VT * len(terms))(*[_VT(*t) for t in terms])
    coef = (C.c_double * 32)()
    deg, mid, half, err = C.c_int32(), C.c_double(), C.c_double(), C.c_double()
    rc = lib.mcx_value_poly_fit(arr, len(terms), C.c_double(lo), C.c_double(hi), C.c_double(tol), max_degree, coef, C.byref(deg),
                                C.byref(mid), C.byref(half), C.byref(err))
    return rc, deg.value, np.array(coef[:max(deg.value, 0) + 1]), mid.value, half.value, err.value


def _swap_terms(t, a=0.1, sig=0.01, th=0.05, fixed=0.03, end=16.0):
    """payer swap seen from date t under Vasicek: -fixed * tau * sum P(t, T_i) + P(t, T_0) - P(t, T_n), P = A exp(-B r)"""
    def zcb(tau):
        B = (1 - np.exp(-a * tau)) / a
        return np.exp((th - sig ** 2 / (2 * a * a)) * (B - tau) - sig ** 2 * B * B / (4 * a)), B
    pay = np.arange(np.ceil(t / 0.25) * 0.25 + 0.25, end + 1e-9, 0.25)
    terms = [(-fixed * 0.25, 0.0, 0.0, zcb(T - t)[0], 0.0, -zcb(T - t)[1]) for T in pay]
    A, B = zcb(max(pay[0] - 0.25 - t, 0.0))
    terms.append((1.0, 0.0, 0.0, A, 0.0, -B))
    A, B = zcb(pay[-1] - t)
    terms.append((-1.0, 0.0, 0.0, A, 0.0, -B))
    terms.append((0.5, 0.01, 2.0, 0.0, 0.0, 0.0))                       # an affine atom and a constant ride along
    return terms


def _exact(terms, x):
    x = np.asarray(x, dtype=np.longdouble)
    val = np.zeros_like(x)
    scale = np.zeros_like(x)
    for w, a, d, b, c0, c1 in terms:
        e = np.longdouble(b) * np.exp(np.longdouble(c0) + np.longdouble(c1) * x)
        val += np.longdouble(w) * (np.longdouble(a) + np.longdouble(d) * x + e)
        scale += abs(np.longdouble(w)) * (abs(np.longdouble(a)) + abs(np.longdouble(d) * x) + abs(e))
    return val, scale


@pytest.mark.parametrize("t", [0.125, 4.0, 12.0, 15.5])
@pytest.mark.parametrize("width", [0.1, 0.3, 0.5])
def test_fit_meets_its_bound_on_random_points(t, width):
    from mcx import _native
    lib = _native.load_library()
    terms = _swap_terms(t)
    lo, hi = 0.05 - width / 2, 0.05 + width / 2
    rc, deg, coef, mid, half, err = _fit(lib, terms, lo, hi)
    assert rc == 0 and 1 <= deg <= 24 and err <= 1e-14
    rng = np.random.default_rng(7)
    x = np.concatenate([rng.uniform(lo, hi, 20000), [lo, hi]])
    tt = x * (1.0 / half) + (-mid * (1.0 / half))
    p = np.zeros_like(x)
    for c in coef[::-1]:
        p = p * tt + c                                  # (numpy has no fma: the double rounding is inside the 2e-14 below)
    val, scale = _exact(terms, x)
    assert float(np.max(np.abs(p - val) / scale)) <= 2e-14
    assert deg + 1 < 8 * len(terms)


def test_fit_refuses_what_it_cannot_verify():
    from mcx import _native
    lib = _native.load_library()
    rc, deg, *_ = _fit(lib, [(1.0, 0.0, 0.0, 1.0, 0.0, 60.0)], -1.0, 1.0)           # exp(60 x) on [-1, 1]: not a degree-31 polynomial
    assert rc == 0 and deg == -1
    rc, deg, *_ = _fit(lib, _swap_terms(1.0), 0.0, 0.3, tol=1e-14, max_degree=4)
    assert rc == 0 and deg == -1
    assert _fit(lib, _swap_terms(1.0), 0.3, 0.3)[0] != 0                             # empty range
    rc, deg, coef, *_ = _fit(lib, [(0.0, 1.0, 1.0, 1.0, 0.0, 1.0)], 0.0, 1.0)       # identically zero
    assert rc == 0 and deg == 0 and coef[0] == 0.0


def _run(sc):
    res = sc.run_simulation()
    return [np.array(m, dtype=float) for m in res.results[0]]


def _bermudan(be, n_main=65536, n_pre=16384):
    """BASELINE config 5 at a test size: 16y quarterly payer swap, 120 exercise = exposure dates, one Euler step per date"""
    model = cases.VasicekModel(0.0, 0.03, 0.05, 0.1, 0.01)
    und = cases.InterestRateSwap(0.0, 16.0, 1.0, 0.03, 0.25, 0.25, cases.IRSType.PAYER)
    prod = cases.BermudanOption(und, [0.125 * k for k in range(1, 121)], 0.0, cases.OptionType.CALL)
    tl = np.array([0.125 * k for k in range(0, 121)])
    rm = cases.RiskMetrics([cases.EPEMetric(), cases.PFEMetric(0.95), cases.PVMetric()], exposure_timeline=tl)
    return cases.SimulationController([cases.NettingSet(name="berm", products=[prod])], model, rm, n_main, n_pre, 1, cases.E, backend=be)


@pytest.mark.gpu
@pytest.mark.parametrize("fused", [True, False], ids=["fused", "unfused"])
def test_collapsed_exercise_values_match_the_term_loops_and_the_oracle(fused, hip, oracle):
    """Bermudan swaption, Philox mode: the polynomial path (LSM roll + main pass) against the exact term loops on the GPU (the
    exercise decisions may differ only where |immediate - continuation| < 1e-13: none at these sizes) and against the oracle"""
    out = {}
    for name, collapse in (("poly", True), ("terms", False)):
        sc = _bermudan(hip)
        sc.collapse_values, sc.allow_fused = collapse, fused
        out[name] = _run(sc)
        assert (sc.n_collapsed_events >= 100) == collapse
        coeffs = sc.products[0].regression_coeffs.numpy().copy()
        out[name + "_coeffs"] = coeffs
    ref = _run(_bermudan(oracle))
    assert np.allclose(out["poly_coeffs"], out["terms_coeffs"], rtol=1e-9, atol=1e-12)
    for a, b, c in zip(out["poly"], out["terms"], ref):
        assert np.allclose(a[:, 0], b[:, 0], rtol=1e-11, atol=1e-13), np.max(np.abs(a[:, 0] - b[:, 0]))
        assert np.allclose(a[:, 0], c[:, 0], rtol=1e-8, atol=1e-10)


@pytest.mark.gpu
def test_paths_outside_the_verified_range_take_the_exact_loop(hip):
    """a negative pad verifies the polynomials on a range NARROWER than the data: most waves then hold a path outside it and run
    the term loop, the others the polynomial — the results must not depend on where the line falls"""
    vals = []
    for pad in (0.3, -0.35, -0.45):
        sc = _bermudan(hip)
        sc.collapse_pad = pad
        vals.append(_run(sc))
        assert sc.n_collapsed_events > 0
    for v in vals[1:]:
        for a, b in zip(vals[0], v):
            assert np.allclose(a[:, 0], b[:, 0], rtol=1e-11, atol=1e-13)
