#!/usr/bin/env python3
"""Workload for the PMC passes of the value-dictionary mirror (tools/pmc_value_dict.sh): the fused Chebyshev term of spin chains with
the mirror on (rbcsr_coded_spmv_kernel) and off (rbcsr_spmv_kernel) in one process, 2 steps each."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import qprop_amd.lib as L  # noqa: E402
import qprop_amd.synth as synth  # noqa: E402

ctx = L.Context(0)
for name, gen, n in (("tfim", synth.tfim_csr, 20), ("xxz", synth.xxz_csr, 20)):
    N = 1 << n
    rp, col, vals = gen(n)
    psi0 = synth.random_state(N)
    for knob in (1, 0):
        ctx.tuning_set("value_dict", knob)
        op = L.Operator(ctx, [L.Matrix(ctx, N, N, rp, col, vals)])
        b = 2.2 * n
        wrk = L.ChebyWrk(ctx, N, 2.2 * b, -1.1 * b, 10.0 / b)
        psi = L.State(ctx, data=psi0)
        for _ in range(2):
            L.cheby(psi, op, 10.0 / b, wrk)
        ctx.sync()
        print(name, n, "value_dict", knob, op.value_encoding_info(), flush=True)
        for h in (psi, wrk, op):
            h.close()
