"""The usage example of README.md, executed (small random inputs)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import qprop_amd  # noqa: F401,E402  (registers the import name of quantumpropagators.jl_amd)
import qprop_amd.propagator as P  # noqa: E402
import qprop_amd.synth as synth  # noqa: E402

rng = np.random.default_rng(0)
n = 12
H0, H1 = synth.dense_hermitian(n, rho=2.0, rng=rng), synth.dense_hermitian(n, rho=0.5, rng=rng)
psi0 = rng.standard_normal(n) + 1j * rng.standard_normal(n)
psi0 /= np.linalg.norm(psi0)
tlist = np.linspace(0, 1, 21)
eps = lambda t: np.sin(t)      # noqa: E731
c_ops = [0.2 * rng.standard_normal((n, n))]
O = H0

H = P.hamiltonian(H0, (H1, lambda t: np.sin(t)))
psi_T = P.propagate(psi0, H, tlist, method="cheby")
psi_T2, ev = P.propagate(psi0, H, tlist, method="newton", storage=True, observables=[O])
p = P.init_prop(psi0, H, tlist, "cheby")
while P.prop_step(p) is not None:
    pass
assert np.linalg.norm(p.state.numpy() - psi_T) < 1e-10 and np.linalg.norm(psi_T2 - psi_T) < 1e-9
L_mf = P.liouvillian((H0, (H1, eps)), c_ops, convention="TDSE", matrix_free=True)
rho0_vec = np.ascontiguousarray(np.outer(psi0, psi0.conj()).T).reshape(-1)
rho_T = P.propagate(rho0_vec, L_mf, tlist, method="newton")
assert abs(np.trace(rho_T.reshape(n, n).T) - 1) < 1e-10
print("README example ok:", abs(np.vdot(psi_T, psi_T)), ev.shape)
