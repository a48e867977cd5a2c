"""Host-side conventions the propagators depend on (no GPU): the midpoint discretisation of
controls (src/controls.jl:43-124, :189-208), restating test/test_discretization.jl, and the
uniform-time-grid check of src/propagator.jl:267-280 (test/test_prop_interfaces.jl:401-412)."""
import os
import sys
import warnings

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import qprop_amd.propagator as P  # noqa: E402


def blackman(t, t0, T, a=0.16):
    """src/shapes.jl:100-107 (test input only)."""
    if not (t0 <= t <= T):
        return 0.0
    dT = T - t0
    return 0.5 * (1.0 - a - np.cos(2 * np.pi * (t - t0) / dT) + a * np.cos(4 * np.pi * (t - t0) / dT))


def test_get_tlist_midpoints():
    """test/test_discretization.jl:8-34."""
    tlist = [1.0, 3.0, 5.0, 6.0, 7.0]
    assert np.array_equal(P.get_tlist_midpoints(tlist), [1.0, 4.0, 5.5, 7.0])
    assert np.array_equal(P.get_tlist_midpoints(tlist, preserve_start=False), [2.0, 4.0, 5.5, 7.0])
    assert np.array_equal(P.get_tlist_midpoints(tlist, preserve_end=False), [1.0, 4.0, 5.5, 6.5])
    assert np.array_equal(P.get_tlist_midpoints(tlist, preserve_start=False, preserve_end=False), [2.0, 4.0, 5.5, 6.5])
    with pytest.raises(ValueError):
        P.get_tlist_midpoints([1.0, 2.0])
    with pytest.raises(AssertionError):
        P.get_tlist_midpoints([0.0, 0.0, 0.0], preserve_start=False, preserve_end=False)
    with pytest.raises(AssertionError):
        P.get_tlist_midpoints([0.0, 1.0, 1.0, 0.0])


def test_discretize_and_midpoints_round_trip():
    """test/test_discretization.jl:37-76."""
    tlist = np.linspace(0, 10, 20)
    f = lambda t: blackman(t, 0.0, 10.0)        # noqa: E731
    c1 = P.discretize(f, tlist, via_midpoints=True)
    p1 = P.discretize_on_midpoints(f, tlist)
    c2 = P.discretize(p1, tlist)
    p2 = P.discretize_on_midpoints(c1, tlist)
    c3 = P.discretize(f, tlist, via_midpoints=False)
    c4 = P.discretize(c1, tlist)
    p3 = P.discretize_on_midpoints(p1, tlist)
    assert len(c1) == len(c2) == len(tlist) and len(p1) == len(p2) == len(tlist) - 1
    assert np.max(np.abs(c1 - c2)) < 1e-14 and np.max(np.abs(p1 - p2)) < 1e-14
    assert 1e-3 < np.max(np.abs(c1 - c3)) < 1e-1
    assert c3[12] == f(tlist[12]) and c1[12] != f(tlist[12])
    assert 1e-3 < abs(c1[12] - f(tlist[12])) < 1e-1
    assert c4 is not c1 and np.max(np.abs(c4 - c1)) < 1e-14
    assert p3 is not p1 and np.max(np.abs(p3 - p1)) < 1e-14
    with pytest.raises(ValueError):
        P.discretize(np.zeros(5), tlist)
    with pytest.raises(ValueError):
        P.discretize_on_midpoints(np.zeros(5), tlist)


def test_get_uniform_dt():
    """test/test_prop_interfaces.jl:401-412."""
    assert P._get_uniform_dt(np.array([0.0, 1.0, 3.0, 6.0])) is None
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        assert P._get_uniform_dt(np.array([0.0, 1.0, 3.0, 6.0]), warn=True) is None
        assert any("Non-uniform" in str(x.message) for x in w)
    assert abs(P._get_uniform_dt(np.linspace(0, 1, 11)) - 0.1) < 1e-15


def test_hamiltonian_collects_drift_terms():
    """hamiltonian(H0, H0', (H1, eps)) sums the drift terms into one operator listed first
    (src/generators.jl:388-469); a single static term is returned as is."""
    A, B, C = np.eye(2), 2 * np.eye(2), 3 * np.eye(2)
    eps = lambda t: t       # noqa: E731
    G = P.hamiltonian(A, (C, eps), B)
    assert isinstance(G, P.Generator) and len(G.ops) == 2 and len(G.amplitudes) == 1
    assert np.array_equal(G.ops[0], A + B) and G.ops[1] is C and G.amplitudes[0] is eps
    assert P.hamiltonian(A) is A
    # terms with the same amplitude are merged (src/generators.jl:408-424)
    pulse = np.linspace(0, 1, 5)
    G = P.hamiltonian(A, (B, eps), (C, eps), (B, pulse), (C, pulse.copy()))
    assert len(G.ops) == 3 and np.array_equal(G.ops[1], B + C) and np.array_equal(G.ops[2], B + C)
    with pytest.raises(ValueError, match="2-tuple"):
        P.hamiltonian(A, (B, eps, 1.0))
    with pytest.raises(ValueError, match="no terms"):
        P.hamiltonian()
