#!/usr/bin/env python3
"""Fuzz the Pauli-string operator (csrc/engine_pauli.hip) against the NumPy oracle on the Kronecker-built matrix: random registers of
6-13 qubits, 1-3 terms of the lazy sum with random real coefficients and a scale, 1-60 random strings per term (any mix of I / X / Y / Z,
so that groups with one string, groups with many, a diagonal group with one or many strings or none all occur), forward and backward
cheby! steps with coefficients changing in between, a mul! with random alpha / beta.  Test infrastructure: oracle/ is the checker.

    python tools/fuzz_pauli.py [n_cases] [seed]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import qp_oracle as qo  # noqa: E402
import qprop_amd.lib as L  # noqa: E402
import qprop_amd.synth as synth  # noqa: E402


def main():
    ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    rng = np.random.default_rng(seed)
    ctx = L.Context(0)
    bad, worst = 0, 0.0
    for case in range(ncases):
        n = int(rng.integers(6, 14))
        N = 1 << n
        nops = int(rng.integers(1, 4))
        ncoeffs = int(rng.integers(0, nops + 1))
        p_id = float(rng.choice([0.3, 0.6, 0.85]))
        terms = []
        for _ in range(nops):
            strings = []
            for _ in range(int(rng.integers(1, 61 // nops + 1))):
                lab = "".join(rng.choice(list("IXYZ"), size=n, p=[p_id] + [(1 - p_id) / 3] * 3))
                if rng.integers(0, 4) == 0:          # a diagonal string
                    lab = lab.replace("X", "Z").replace("Y", "I")
                strings.append((float(rng.uniform(-1, 1)), L.pauli_masks(lab)))
            terms.append(strings)
        mats = [synth.pauli_sum_matrix(n, s_) for s_ in terms]
        op = L.PauliOperator(ctx, n, terms, ncoeffs=ncoeffs)
        scale = float(rng.choice([1.0, -0.7, 2.0]))
        op.set_scale(scale)
        psi0 = synth.random_state(N, seed=case)
        psi = L.State(ctx, data=psi0)
        ref = psi0.copy()
        err = 0.0
        bound = abs(scale) * sum(1.5 * sum(abs(a) for a, _ in s_) for s_ in terms) + 1e-3
        dt = float(rng.uniform(2.0, 12.0)) / bound
        wrk = L.ChebyWrk(ctx, N, 2.1 * bound, -1.05 * bound, dt)
        owrk = qo.ChebyWrk(psi0, 2.1 * bound, -1.05 * bound, dt)
        for step in range(3):
            coeffs = rng.uniform(-1.5, 1.5, ncoeffs)
            if ncoeffs:
                op.set_coeffs(coeffs)
            H = None
            for l, M in enumerate(mats):
                c = 1.0 if l < nops - ncoeffs else coeffs[l - (nops - ncoeffs)]
                H = scale * c * M if H is None else H + scale * c * M
            H = H.tocsr()
            sg = 1 if rng.integers(0, 3) else -1
            L.cheby(psi, op, sg * dt, wrk)
            qo.cheby(ref, H, sg * dt, owrk)
            err = max(err, float(np.linalg.norm(psi.numpy() - ref)))
        x0, y0 = synth.random_state(N, seed=1000 + case), synth.random_state(N, seed=2000 + case)
        x, y = L.State(ctx, data=x0), L.State(ctx, data=y0)
        al, be = complex(rng.normal(), rng.normal()), complex(rng.normal(), rng.normal()) * float(rng.integers(0, 2))
        op.mul(x, y, alpha=al, beta=be)
        err = max(err, float(np.linalg.norm(y.numpy() - (be * y0 + al * (H @ x0)))) / max(1.0, bound))
        worst = max(worst, err)
        if not err < 1e-10:
            bad += 1
            print(f"case {case}: n={n} nops={nops} ncoeffs={ncoeffs} strings={[len(s_) for s_ in terms]} err={err:.3e}  BAD", flush=True)
        for h in (x, y, psi, wrk, op):
            h.close()
    print(f"{ncases} cases (seed {seed}), {bad} bad, worst error {worst:.3e}")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
