"""High-precision (mpmath, 50 digits) restatement of the floating-point-sensitive pieces of the path -- TEST
INFRASTRUCTURE ONLY, like the rest of oracle/: the checker of the checker.  The reference is Julia and cannot be run here,
and its own tests hold no stored vectors for this path, so the double-precision oracle (oracle/qp_oracle.py: scipy `jv`,
NumPy arithmetic) and the library's host numerics (glibc `jn`, own QR) are pinned against arithmetic that is independent
of both and exact to ~1e-45:

* ``cheby_coeffs``            src/cheby.jl:25-39      (SpecialFunctions.besselj -> mpmath.besselj)
* ``exp(-i H dt) psi``        what test/test_cheby.jl:24-47 and test/test_newton.jl:53-65 compare with (dense `exp`)
* ``extend_leja``             src/newton.jl:97-148    (the product chain in exact-enough arithmetic)
* ``extend_newton_coeffs``    src/newton.jl:176-214   (divided differences)
* ``newton_polynomial``       the value of the interpolation polynomial those coefficients define

Nothing under quantumpropagators.jl_amd/ imports this file.
"""
import mpmath as mp

mp.mp.dps = 50


def cheby_coeffs(Delta, dt, limit=1e-12):
    """src/cheby.jl:25-39 in 50-digit arithmetic: a_1 = J_0(alpha), a_k = 2 J_{k-1}(alpha), appended until the coefficient
    just appended is <= limit in magnitude (that one is kept).  Delta, dt are taken as the doubles they are."""
    alpha = abs(mp.mpf(0.5) * mp.mpf(float(Delta)) * mp.mpf(float(dt)))
    lim = mp.mpf(float(limit))
    coeffs = [mp.besselj(0, alpha)]
    eps = abs(coeffs[0])
    i = 1
    while eps > lim:
        a = 2 * mp.besselj(i, alpha)
        coeffs.append(a)
        eps = abs(a)
        i += 1
    return coeffs


def expm_apply(H, psi, dt, func="expmi"):
    """exp(-i H dt) psi (or exp(H dt) psi for func="exp") for a small dense H, 50 digits (mpmath.expm, Pade + squaring in
    working precision)."""
    n = len(psi)
    A = mp.matrix(n, n)
    s = mp.mpc(0, -1) * mp.mpf(float(dt)) if func == "expmi" else mp.mpf(float(dt))
    for i in range(n):
        for j in range(n):
            z = complex(H[i, j])
            A[i, j] = s * mp.mpc(z.real, z.imag)
    E = mp.expm(A)
    v = mp.matrix(n, 1)
    for i in range(n):
        z = complex(psi[i])
        v[i] = mp.mpc(z.real, z.imag)
    out = E * v
    return [out[i] for i in range(n)]


def extend_leja(leja, n, newpoints, n_use):
    """src/newton.jl:97-148 with every product in 50-digit arithmetic.  `leja`: list of the n points chosen so far;
    `newpoints`: candidates (copied).  Returns (the n_use chosen points in order, for each pick the margin
    1 - p_second / p_max by which it won)."""
    to = lambda z: mp.mpc(complex(z).real, complex(z).imag)      # noqa: E731
    leja = [to(z) for z in leja[:n]]
    pts = [to(z) for z in newpoints]
    u = len(pts) - 1
    chosen, margins = [], []
    i_add_start = 0
    if n == 0:                                                   # :113-126 -- the point of largest magnitude first
        z_last = pts[u]
        for i in range(0, u):
            if abs(pts[i]) > abs(z_last):
                pts[u], pts[i] = pts[i], z_last
                z_last = pts[u]
        leja.append(pts[u])
        chosen.append(pts[u])
        margins.append(None)
        i_add_start = 1
    exponent = mp.mpf(1) / (n + n_use)                           # :127
    for i_add in range(i_add_start, n_use):
        ps = []
        for i in range(0, u - i_add + 1):
            p = mp.mpf(1)
            for j in range(0, n + i_add):
                p = p * abs(pts[i] - leja[j]) ** exponent
            ps.append(p)
        i_max, p_max = 0, mp.mpf(0)
        for i, p in enumerate(ps):                               # strict >, first maximum wins (:137-140)
            if p > p_max:
                p_max, i_max = p, i
        rest = [p for i, p in enumerate(ps) if i != i_max]
        margins.append(float(1 - max(rest) / p_max) if rest and p_max > 0 else None)
        leja.append(pts[i_max])
        chosen.append(pts[i_max])
        pts[i_max] = pts[u - i_add]
    return chosen, margins


def extend_newton_coeffs(leja, radius, func="expmi"):
    """src/newton.jl:176-214 from n_a = 0: divided differences of f on the Leja points, normalised by `radius`."""
    to = lambda z: mp.mpc(complex(z).real, complex(z).imag)      # noqa: E731
    z = [to(x) for x in leja]
    r = mp.mpf(float(radius))
    f = (lambda x: mp.exp(mp.mpc(0, -1) * x)) if func == "expmi" else mp.exp
    a = [f(z[0])]
    for k in range(1, len(z)):
        d = mp.mpc(1)
        pn = mp.mpc(0)
        for n in range(1, k):
            d = d * (z[k] - z[n - 1]) / r
            pn = pn + a[n] * d
        d = d * (z[k] - z[k - 1]) / r
        a.append((f(z[k]) - a[0] - pn) / d)
    return a


def newton_polynomial(a, leja, radius, x):
    """p(x) = sum_k a_k prod_{j<k} (x - z_j) / radius in 50 digits; a may be doubles or mp numbers."""
    to = lambda z: z if isinstance(z, (mp.mpf, mp.mpc)) else mp.mpc(complex(z).real, complex(z).imag)      # noqa: E731
    x = to(x)
    r = mp.mpf(float(radius))
    s, w = mp.mpc(0), mp.mpc(1)
    for k in range(len(a)):
        s += to(a[k]) * w
        w = w * (x - to(leja[k])) / r
    return s
