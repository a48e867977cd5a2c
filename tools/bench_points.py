"""Measured points shared by bench.py (the extra keys of its JSON line) and the tools (kbench.py, point.py): byte models of the SHIPPED device layouts and short
single-GPU measurements of the other BASELINE configs / patterns.  Everything here calls the product
path through the C ABI (qprop_amd.lib); nothing imports the oracle."""
import gc
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
import qprop_amd.lib as L  # noqa: E402
import qprop_amd.synth as synth  # noqa: E402

HBM_PEAK_GBS = 8000.0     # MI355X HBM3E spec peak (MI355X_MICROARCH.md: 8 TB/s; 6.3 TB/s measured copy)
FMT_NAME = {1: "csr", 2: "rbcsr", 3: "hrb (Hermitian-packed row blocks)", 4: "matrix-free", 5: "dense (row-major, no index bytes)"}


def _cpu_throttled_ms():
    """Milliseconds this process's control group has been throttled by the CPU quota so far (cgroup v2 cpu.stat, else v1), or
    None: a region during which this number grows lost its host thread to the container's scheduler, not to the library."""
    for path, key, scale in (("/sys/fs/cgroup/cpu.stat", "throttled_usec", 1e-3),
                             ("/sys/fs/cgroup/cpu/cpu.stat", "throttled_time", 1e-6),
                             ("/sys/fs/cgroup/cpu,cpuacct/cpu.stat", "throttled_time", 1e-6)):
        try:
            with open(path) as f:
                for ln in f:
                    if ln.startswith(key + " "):
                        return float(ln.split()[1]) * scale
        except OSError:
            continue
    return None


def timed_regions(ctx, fn, steps, repeats=3):
    """`repeats` timed regions of `steps` calls of fn, each bracketed by HIP events on the kernels' stream; next to every
    region's event time what the HOST did meanwhile -- wall time of the enqueue loop and the longest single call -- so that a
    host stall (the device drains its queue and idles inside the event bracket) can be told from a device slow mode.
    -> [(event_ms, enqueue_ms, longest_call_ms, ms the control group was throttled by its CPU quota meanwhile or None)]

    The interpreter's cyclic garbage collector is switched off inside a region (as `timeit` does): in a process that has
    imported torch a full collection takes 50-80 ms, it is triggered by the allocation count -- i.e. by the ctypes wrappers'
    temporaries, in the middle of an enqueue loop -- and the device then idles inside the event bracket.  That is what the
    one 455 us-per-term sample of round 3 was (profiles/r04/n22_outlier.txt); QP_BENCH_GC=1 leaves the collector on."""
    keep_gc = os.environ.get("QP_BENCH_GC") == "1"
    # The host phases before a point (synthetic generation and operator build on several threads, the CPU baseline's OpenMP
    # team) can exhaust the container's CPU quota for the current scheduler period: the whole control group is then frozen
    # until the period ends -- up to 100 ms, landing in the first timed region.  If the throttle counter moved lately, let
    # the period pass before timing.
    thr = _cpu_throttled_ms()
    if thr is not None:
        time.sleep(0.02)
        if _cpu_throttled_ms() != thr or thr != getattr(timed_regions, "_last_thr", None):
            time.sleep(0.12)
        timed_regions._last_thr = _cpu_throttled_ms()
    out = Regions()
    retries = 0
    while len(out) < repeats:
        ctx.sync()
        longest = 0.0
        was_on = gc.isenabled()
        if not keep_gc:
            gc.disable()
        try:
            thr0 = _cpu_throttled_ms()
            cpu0 = time.process_time()
            ctx.timer_begin()
            t0 = time.perf_counter()
            for _ in range(steps):
                t1 = time.perf_counter()
                fn()
                longest = max(longest, time.perf_counter() - t1)
            enq = time.perf_counter() - t0
            ev = ctx.timer_end()
            thr1 = _cpu_throttled_ms()
            region = (ev, 1e3 * enq, 1e3 * longest, (thr1 - thr0) if (thr0 is not None and thr1 is not None) else None,
                      1e3 * (time.process_time() - cpu0))
        finally:
            if was_on:
                gc.enable()
        # A region in which ONE enqueue call took more than half of the whole region's event time did not measure the device:
        # the host thread was held (the container's CPU quota, profiles/r04/n22_outlier.txt) and the device sat idle inside the
        # event bracket.  It is measured again -- at most twice per point -- and kept on record (`discarded`).
        if longest * 1e3 > 0.5 * ev and retries < 2 and os.environ.get("QP_BENCH_KEEP_STALLS") != "1":
            retries += 1
            out.discarded.append(region)
            time.sleep(0.15)
            continue
        out.append(region)
    return out


class Regions(list):
    """the timed regions of a point (tuples, see timed_regions) + the ones that were measured again after a host stall"""

    def __init__(self, *a):
        super().__init__(*a)
        self.discarded = []


def spread(values, regions=None):
    """median / min / max of repeated measurements; `unstable` when max / min > 1.3 (one 455 us sample among
    122-136 us ones became a first-class key of the line).  With the host-side record of the regions: whether the slowest
    region is explained by the host (its enqueue loop took at least 80 % of the event time: the device was waiting)."""
    v = sorted(float(x) for x in values)
    out = {"median": float(np.median(v)), "min": v[0], "max": v[-1], "repeats": len(v), "unstable": bool(v[-1] > 1.3 * v[0])}
    if regions is not None and out["unstable"]:
        slow = max(regions, key=lambda r: r[0])
        out["slowest_region"] = {"event_ms": slow[0], "host_enqueue_ms": slow[1], "longest_single_call_ms": slow[2],
                                 "host_stall_suspected": bool(slow[2] > 0.5 * slow[0] or slow[1] > 0.8 * slow[0]),
                                 "cpu_quota_throttled_ms": slow[3] if len(slow) > 3 else None}
    disc = getattr(regions, "discarded", None)
    if disc:      # regions measured again because one enqueue call held the host for more than half of the region (timed_regions)
        out["regions_remeasured_after_host_stall"] = [
            {"event_ms": r[0], "longest_single_call_ms": r[2], "cpu_quota_throttled_ms": r[3], "process_cpu_ms": r[4] if len(r) > 4 else None}
            for r in disc]
    return out


def cheby_layout_bytes(op, rows, ncols, nnz, coeffs, real_copy=False, whole_step=True):
    """HBM bytes ONE cheby! step must move with the operator laid out as it is on the device, split by
    stream, and the same per fused term (averages over the step's terms).

    matrix  stored values (16 B, or 8 B when the all-real copy is streamed; the Hermitian-packed
            format stores the upper triangle only: the lower entries re-read those values through L2)
            + index bytes (column sections, transpose positions; stencil blocks store none per entry)
            + per-block pointers / metadata (or the CSR row pointers)
    vectors per term: the gathered vector once (16 B per column), v_{m-2} read and v_m written in place
            (16 B per row each; not read by term 1, not written by the last), and the Psi accumulator
            read + written only by the terms of the deferred schedule (qp_acc_schedule_host)
    This is what `roofline.frac` prices: it cannot exceed the chip's peak.  The contract's CSR figure
    ((20 z + 84) N per term) is reported next to it as `csr_equivalent`."""
    lay = op.layout_info()
    fmt = op.format
    nterms = len(coeffs) - 1
    vbytes = 8.0 if real_copy else 16.0
    walk = op.walk_info() if fmt == L.FMT_HRB else {"valid": 0}
    lay["strip_walk"] = walk
    cb = op.colblock_info() if fmt in (L.FMT_CSR, L.FMT_RBCSR) else {"valid": 0}
    cb_on = bool(cb["valid"]) and op.ctx.tuning_get("colblock") != 0
    lay["column_blocked_mirror"] = cb if cb_on else None
    if cb_on:
        # the column-blocked mirror (kernels_colblock.hip): value + column per entry, a 16-bit offset per (row, column block)
        # and one more per (tile, block), the segment pointers
        tr = cb["rows_per_tile"]
        matrix = (vbytes + 4.0) * nnz + 2.0 * cb["tiles"] * cb["column_blocks"] * (tr + 1) + 4.0 * cb["tiles"] * cb["column_blocks"]
    elif fmt == L.FMT_CSR:
        matrix = vbytes * nnz + 4.0 * nnz + 8.0 * (rows + 1)
    else:
        per_block = 32.0 if fmt == L.FMT_HRB else 16.0     # bptr + cmeta (+ lptr + lcmeta)
        index = lay["index_bytes"] + per_block * lay["blocks"]
        stored = lay["stored"]
        if walk["valid"]:      # the strip walk computes positions by formula: index bytes only for its edge blocks ...
            share = walk["edge_blocks"] / max(lay["blocks"], 1)
            index *= share
            # ... and reads the slots that carry entries, not the pad slots of the quad-padded sections
            slots = walk.get("upper_slots") or (walk["diag"] + walk["near"] + walk["far"] + len(walk.get("long_distances", [])))
            stored = 64.0 * slots * (walk["end_block"] - walk["first_block"]) + lay["stored"] * share
        matrix = vbytes * stored + index
        ve = op.value_encoding_info() if fmt == L.FMT_RBCSR else {"valid": 0}
        lay["value_dictionary"] = ve if ve["valid"] else None
        if ve["valid"]:      # the value-dictionary mirror: a byte per stored position + the blocks' (shared) tables
            matrix = ve["coded_bytes"] + index
    sched = L.acc_schedule(coeffs)
    # the two-term strip walk (csrc/kernels_walk2.hip): terms 2 .. in PAIRS, each pair one pass over the values.  What a pair loads is
    # more than the rows it forms: a chunk's 64 lanes form z on W = 64 - 2 d_max of them (ceil(g / W) chunks per strip step), and a
    # segment of L steps runs in 2 K steps before its first z; the blocks outside the two-term region take two per-block passes.
    # (whole_step = False: the terms are launched one by one -- the row-partitioned step exchanges after every term -- never in pairs)
    w2 = op.walk2_info() if (whole_step and fmt == L.FMT_HRB and walk["valid"]) else {"valid": 0}
    pair = None
    if w2["valid"]:
        g, K = walk["rows_per_step"], walk["far"]
        W, S2 = w2["useful_rows_per_chunk"], w2["chunks_per_strip_step"]
        Jz = -(-(w2["end_block"] - w2["first_block"]) * 64 // g)
        Lz, nseg = w2["steps_per_wavefront"], w2["segments"]
        rows_loaded = 64.0 * S2 * (Jz + 2 * K * nseg)          # lanes x Y steps of the walk
        rows_region = 64.0 * (w2["end_block"] - w2["first_block"])
        slots = walk.get("upper_slots") or (walk["diag"] + walk["near"] + walk["far"])
        edge_share = w2["edge_blocks"] / max(lay["blocks"], 1)
        pair = {"rows_loaded_per_pair": rows_loaded, "rows_of_the_region": rows_region, "chunks_per_strip_step": S2, "useful_rows_per_chunk": W,
                "steps_per_wavefront": Lz, "segments": nseg,
                # values once per pair over the lanes that load them + the edge blocks' two per-block passes (values + index bytes)
                "matrix_per_pair": vbytes * slots * rows_loaded + 2.0 * (vbytes * lay["stored"] + lay["index_bytes"] + 32.0 * lay["blocks"]) * edge_share}
        lay["two_term_walk"] = pair
    vec = 0.0
    updated = False
    npairs = 0
    m = 1
    while m <= nterms:
        if pair is not None and m >= 2 and m + 1 <= nterms:
            # x and p over the loaded lanes, y and z over the rows; the accumulator by the schedule (never both terms of a pair)
            vec += 2.0 * 16.0 * pair["rows_loaded_per_pair"] + 16.0 * rows + (16.0 * rows if m + 1 < nterms else 0.0)
            for mm in (m, m + 1):
                if not sched[mm - 1].skip:
                    vec += (32.0 if updated else 16.0) * rows
                    updated = True
            npairs += 1
            m += 2
            continue
        vec += 16.0 * ncols * (8.0 if cb_on else 1.0)    # gathered vector (the mirror: every XCD's L2 loads every window once)
        if m >= 2:
            vec += 16.0 * rows                   # v_{m-2}
        if m < nterms:
            vec += 16.0 * rows                   # v_m
        if not sched[m - 1].skip:
            vec += (32.0 if updated else 16.0) * rows
            updated = True
        m += 1
    if pair is not None:
        vec += 32.0 * rows                       # the result's copy into Psi's vector (the pairs rotate four vectors)
        step = (nterms - 2 * npairs) * matrix + npairs * pair["matrix_per_pair"] + vec
        matrix_per_term = ((nterms - 2 * npairs) * matrix + npairs * pair["matrix_per_pair"]) / nterms
    else:
        step = nterms * matrix + vec
        matrix_per_term = matrix
    return {"per_step": step, "per_term": step / nterms, "matrix_per_term": matrix_per_term, "vectors_per_term": vec / nterms,
            "csr_equivalent_per_term": (12.0 if real_copy else 20.0) * nnz + 4.0 * (rows + 1) + 80.0 * rows,
            "layout": lay}


def cheby_kernel_name(op, whole_step=True):
    """The kernel a whole-operator fused Chebyshev term of `op` launches (substring of its symbol); whole_step: inside cheby!
    (terms in pairs where the two-term walk is taken), else a term launched on its own."""
    if op.format == L.FMT_HRB and op.walk_info()["valid"]:
        return "hrb_walk2_kernel" if (whole_step and op.walk2_info()["valid"]) else "hrb_walk_kernel"
    if op.colblock_info()["valid"] and op.ctx.tuning_get("colblock") != 0:
        return "colblock_spmv_kernel"
    if op.format == L.FMT_RBCSR and op.value_encoding_info()["valid"]:
        return "rbcsr_coded_spmv_kernel"
    return {L.FMT_CSR: "csr_spmv_kernel", L.FMT_RBCSR: "rbcsr_spmv_kernel", L.FMT_HRB: "hrb_spmv_kernel"}[op.format]


def pattern_csr(pattern, N, row_begin=0, row_end=None):
    if pattern == "banded":
        return synth.hermitian_offsets_csr(N, offsets=synth.BANDED_OFFSETS, row_begin=row_begin, row_end=row_end)
    if pattern == "scattered":
        return synth.hermitian_offsets_csr(N, offsets=synth.scattered_offsets(N), row_begin=row_begin, row_end=row_end)
    if row_begin != 0 or (row_end is not None and row_end != N):
        raise ValueError("the random patterns are generated whole (single GPU)")
    if pattern == "random":
        return synth.random_columns_csr(N)
    if pattern == "random-window":
        return synth.random_columns_csr(N, window=4096)
    raise ValueError(pattern)


def measure_pauli(ctx, spins=20, model="tfim", steps=10, warmup=2, repeats=3):
    """Cheby prop_step! of a qubit-register generator applied from its Pauli strings (csrc/engine_pauli.hip: no stored matrix): the
    transverse-field Ising / XXZ chain of `spins` spins, alpha = 10 as the headline.  Bytes per term: the vectors alone."""
    N = 1 << spins
    strings = synth.tfim_pauli_terms(spins) if model == "tfim" else synth.xxz_pauli_terms(spins)
    bound = float(sum(abs(complex(a)) for a, _ in strings))
    window = (2.2 * bound, -1.1 * bound)
    dt = 20.0 / window[0]
    op = L.PauliOperator(ctx, spins, [strings])
    wrk = L.ChebyWrk(ctx, N, window[0], window[1], dt)
    psi = L.State(ctx, data=synth.random_state(N))
    nterms = wrk.n_coeffs - 1
    for _ in range(warmup):
        L.cheby(psi, op, dt, wrk)
    regions = timed_regions(ctx, lambda: L.cheby(psi, op, dt, wrk), steps, repeats)
    sp = spread([1e3 * r[0] / (steps * nterms) for r in regions], regions)
    t_term = sp["median"] * 1e-6
    sched = L.acc_schedule(wrk.coeffs)
    vec, updated = 0.0, False
    for m in range(1, nterms + 1):      # x once (its partners are permuted lines of the same vector), v_{m-2}, v_m, the accumulator by the schedule
        vec += 16.0 * N + (16.0 * N if m >= 2 else 0.0) + (16.0 * N if m < nterms else 0.0)
        if not sched[m - 1].skip:
            vec += (32.0 if updated else 16.0) * N
            updated = True
    per_term = vec / nterms
    out = {"pattern": f"{'transverse-field Ising' if model == 'tfim' else 'XXZ'} chain, {spins} spins, from its {len(strings)} Pauli strings (matrix-free)",
           "N": N, "strings": len(strings), "x_mask_groups": len({x for _, (x, _) in strings}), "device_format": FMT_NAME[op.format],
           "kernel": "pauli_spmv_kernel", "n_coeffs": int(wrk.n_coeffs), "us_per_term": t_term * 1e6, "us_per_term_min": sp["min"],
           "us_per_term_max": sp["max"], "repeats": sp["repeats"], "unstable": sp["unstable"], "steps_per_s": 1e6 / (t_term * 1e6 * nterms),
           "layout_bytes_per_term": per_term, "layout_gbs": per_term / t_term / 1e9, "frac": per_term / t_term / 1e9 / HBM_PEAK_GBS,
           "model": "layout", "norm_drift": abs(psi.norm() - 1.0)}
    for h in (psi, wrk, op):
        h.close()
    return out


def measure_cheby(ctx, pattern="banded", log2n=20, fmt="auto", steps=10, warmup=2, real=False, dt=1.0, grid=None, repeats=3,
                  grid_order=2, spins=None, offsets=None, knobs=None):
    saved = {k: ctx.tuning_get(k) for k in (knobs or {})}
    for k, v in (knobs or {}).items():
        ctx.tuning_set(k, v)
    try:
        out = _measure_cheby(ctx, pattern, log2n, fmt, steps, warmup, real, dt, grid, repeats, grid_order, spins, offsets)
    finally:
        for k, v in saved.items():
            ctx.tuning_set(k, v)
    if knobs:
        out["knobs"] = dict(knobs)
    return out


def _measure_cheby(ctx, pattern, log2n, fmt, steps, warmup, real, dt, grid, repeats, grid_order, spins, offsets):
    """Cheby prop_step! on one GPU for a pattern / size / device format; per-term time from HIP events on
    the kernels' stream -- the MEDIAN of `repeats` timed regions of `steps` steps, with min, max and an `unstable` flag --;
    layout-byte and CSR-equivalent rates.  grid = (nx, ny): the finite-difference Hamiltonian of an
    open-boundary grid (synth.grid_hamiltonian_2d) instead of a pattern."""
    window = (20.0, -10.0)        # Delta, E_min of the synthetic patterns (|spectrum| <= 10)
    if spins:                     # transverse-field Ising chain: the qubit-register structure (row XOR 2^i), real couplings
        N = 1 << spins
        rp, col, vals = synth.tfim_csr(spins)
        bound = 1.0 * (spins - 1) + 0.1 * spins + 1.0 * spins
        window = (2.2 * bound, -1.1 * bound)
        dt = 20.0 / window[0]     # alpha = 10 as the headline
        pattern = f"transverse-field Ising chain, {spins} spins (row XOR 2^i), real couplings"
        real = True
    elif grid and len(grid) == 3:
        Hg = synth.grid_hamiltonian_3d(*grid, flux=0.1, order=grid_order)
        N = grid[0] * grid[1] * grid[2]
        pattern = f"{'seven' if grid_order == 2 else 'thirteen'}-point grid {grid[0]} x {grid[1]} x {grid[2]}, open boundaries"
        rp, col, vals = Hg.indptr.astype(np.int64), Hg.indices.astype(np.int32), Hg.data.astype(np.complex128)
        del Hg
    elif grid:
        Hg = synth.grid_hamiltonian_2d(grid[0], grid[1], flux=0.1)
        N = grid[0] * grid[1]
        pattern = f"five-point grid {grid[0]} x {grid[1]}, open boundaries"
        rp, col, vals = Hg.indptr.astype(np.int64), Hg.indices.astype(np.int32), Hg.data.astype(np.complex128)
        del Hg
    elif offsets:                 # a translation-invariant Hermitian lattice with these column distances
        N = 1 << log2n
        rp, col, vals = synth.hermitian_offsets_csr(N, offsets=tuple(offsets))
        pattern = f"lattice with distances +-{tuple(offsets)}"
    else:
        N = 1 << log2n
        rp, col, vals = pattern_csr(pattern, N)
    if real:
        vals = vals.real.astype(np.complex128)
    nnz = int(rp[-1])
    f = {"auto": L.FMT_AUTO, "hrb": L.FMT_HRB, "rbcsr": L.FMT_RBCSR, "csr": L.FMT_CSR}[fmt]
    M = L.Matrix(ctx, N, N, rp, col, vals)
    del rp, col, vals
    op = L.Operator(ctx, [M], 0, f)
    wrk = L.ChebyWrk(ctx, N, window[0], window[1], dt)
    psi = L.State(ctx, data=synth.random_state(N))
    nterms = wrk.n_coeffs - 1
    import warnings
    with warnings.catch_warnings():      # points that are MEANT to leave the strip walk (generic format, scattered columns) say so
        warnings.simplefilter("ignore", L.QPPerformanceWarning)      # in their record ("strip_walk_reason"), not on stderr
        for _ in range(warmup):
            L.cheby(psi, op, dt, wrk)
        regions = timed_regions(ctx, lambda: L.cheby(psi, op, dt, wrk), steps, repeats)
    sp = spread([1e3 * r[0] / (steps * nterms) for r in regions], regions)      # us per term
    ms = sp["median"] * 1e-3 * steps * nterms
    t_term = sp["median"] * 1e-6
    by = cheby_layout_bytes(op, N, N, nnz, wrk.coeffs, real_copy=real)
    lay = by["layout"]
    out = {"pattern": pattern, "N": N, "nnz_per_row": nnz / N, "device_format": FMT_NAME[op.format], "dt": dt,
           "kernel": cheby_kernel_name(op), "operator_build_ms": op.build_info()["build_ms"],
           "n_coeffs": int(wrk.n_coeffs), "values": "real fp64 (f64 variant)" if real else "complex fp64",
           "ms_per_step": ms / steps, "steps_per_s": 1e3 * steps / ms, "us_per_term": t_term * 1e6,
           "us_per_term_min": sp["min"], "us_per_term_max": sp["max"], "repeats": sp["repeats"], "steps_per_repeat": steps,
           "unstable": sp["unstable"], "slowest_region": sp.get("slowest_region"),
           "regions_remeasured_after_host_stall": sp.get("regions_remeasured_after_host_stall"),
           "layout_bytes_per_term": by["per_term"], "layout_gbs": by["per_term"] / t_term / 1e9,
           "frac": by["per_term"] / t_term / 1e9 / HBM_PEAK_GBS, "model": "layout",
           "csr_equivalent_gbs": by["csr_equivalent_per_term"] / t_term / 1e9,
           "encodings": {"row_blocks": lay["blocks"], "stencil_upper_blocks": lay["stencil_upper_blocks"],
                         "stencil_lower_blocks": lay["stencil_lower_blocks"], "index_bytes": lay["index_bytes"]},
           "column_blocked_mirror": op.colblock_info(), "column_encodings": op.encoding_info(),
           "value_dictionary": op.value_encoding_info(),
           "strip_walk_reason": op.walk_reason()[1],      # "ok", or why this operator's term is not the strip walk (qp_operator_walk_reason)
           "norm_drift": abs(psi.norm() - 1.0)}
    if grid:
        out["explicit_zeros_completing_the_lattice"] = op.fill_info()
        out["strip_walk"] = lay.get("strip_walk")
    for h in (psi, wrk, op, M):
        h.close()
    return out


def measure_newton_c3(ctx, n=512, m=20, dt=0.5, steps=10, warmup=60, repeats=3):
    """BASELINE configs[2]: N = n^2 non-Hermitian Liouvillian, Newton / restarted Arnoldi with m_max = m.  The same `steps` steps
    from rho_0 are timed `repeats` times (wall clock around the synchronised loop: the restart loop has host phases); the
    median region is reported, with min / max / `unstable`."""
    Lm = synth.liouvillian_tridiag(n)
    N, nnz = Lm.shape[0], Lm.nnz
    M = L.Matrix.from_scipy(ctx, Lm)
    op = L.Operator(ctx, [M])
    rho0 = synth.random_state(N)
    wrk = L.NewtonWrk(ctx, N, m_max=m)
    psi = L.State(ctx, data=rho0)
    for _ in range(warmup):
        L.newton(psi, op, dt, wrk)
    runs = []
    for _ in range(repeats):
        psi.upload(rho0)          # the open system relaxes: time the same steps every run
        ctx.sync()
        ctx.reset_stats()
        sweeps = matvecs = 0
        exposed = 0.0
        t0 = time.perf_counter()
        for _ in range(steps):
            L.newton(psi, op, dt, wrk)
            sweeps += wrk.restarts + 1
            matvecs += wrk.stats["n_matvec"]
            exposed += wrk.stats["ms_exposed"]
        ctx.sync()
        runs.append((time.perf_counter() - t0, sweeps, matvecs, exposed, ctx.stats()))
    sp = spread([1e3 * r[0] / steps for r in runs])
    el, sweeps, matvecs, exposed, st = sorted(runs, key=lambda r: r[0])[len(runs) // 2]
    z = nnz / N
    # SURVEY 8d model per Arnoldi sweep, as written there: m plain mat-vecs [matrix + read x + write y] + one fused axpy -> dot pass per
    # (i, j) [48 N each] + norm + scale per column [48 N] + Psi += sum P_i q_i [(16 m + 32) N] + v = sum R_i q_i [(16 (m + 1) + 16) N]
    sweep_bytes = m * (20 * z + 36) * N + 48 * N * m * (m + 1) / 2 + 48 * N * m + (16 * m + 32) * N + (16 * (m + 1) + 16) * N
    # bytes this implementation moves per sweep: per column the matrix, the gathered / written vectors of the fused mat-vec
    # (x, w, q_j, the j older basis vectors) and of the projection (w twice, j + 1 basis vectors); then the two combines
    impl_bytes = m * 20 * z * N + 16 * N * m * (m + 5) + 16 * N * (m + 3)
    # the one-pass sweep (csrc/kernels_onepass.hip; taken when basis + matrix exceed the Infinity Cache): per column kernel t the
    # matrix, the row and the gathers of a_t, t basis rows once, the two rows written; then the two combines
    onepass = ctx.tuning_get("arnoldi_onepass") == 2 or (ctx.tuning_get("arnoldi_onepass") == 1 and
                                                         16.0 * N * (m + 3) + 20.0 * z * N > 224.0 * 1024 * 1024)
    if onepass and m <= 20:
        impl_bytes = m * 20 * z * N + 16 * N * (m * (m + 1) / 2 + 4 * (m + 1)) + 16 * N * (m + 3)
    regime = ("working set (basis + matrix) inside the Infinity Cache: a latency / L1-queue figure, not an HBM fraction" if N <= (1 << 19)
              else "basis beyond the Infinity Cache: HBM-bandwidth bound")
    out = {"workload": f"BASELINE configs[2]: Newton prop_step!, N={N} (n={n}) non-Hermitian sparse Liouvillian, m_max={m}",
           "regime": regime,
           "N": N, "nnz_per_row": z, "m_max": m, "dt": dt, "steps": steps, "device_format": FMT_NAME[op.format],
           "ms_per_step": 1e3 * el / steps, "ms_per_step_min": sp["min"], "ms_per_step_max": sp["max"], "repeats": sp["repeats"],
           "unstable": sp["unstable"], "steps_per_s": steps / el,
           "arnoldi_sweeps_per_step": sweeps / steps, "matvecs_per_step": matvecs / steps,
           "kernel_launches_per_step": st["n_kernel_launches"] / steps,
           "launches_per_column": st["n_kernel_launches"] / max(matvecs, 1),
           "ms_per_sweep": 1e3 * el / sweeps, "host_ms_exposed_per_step": exposed / steps,
           "sweep": "one pass over the basis per column" if (onepass and m <= 20) else "two passes over the basis per column (low-synchronisation MGS)",
           # `frac` prices the bytes this implementation moves (physical: cannot exceed 1); the SURVEY 8d sweep model counts
           # the reference's sequential Gram-Schmidt passes, which the low-synchronisation form does not make
           "implementation_bytes_per_sweep": impl_bytes, "implementation_gbs": impl_bytes * sweeps / el / 1e9,
           "frac": impl_bytes * sweeps / el / 1e9 / HBM_PEAK_GBS, "model": "impl",
           "implementation_frac": impl_bytes * sweeps / el / 1e9 / HBM_PEAK_GBS,
           "algorithmic_gbs": sweep_bytes * sweeps / el / 1e9, "frac_survey_8d_model": sweep_bytes * sweeps / el / 1e9 / HBM_PEAK_GBS,
           "norm": psi.norm()}
    for h in (psi, wrk, op, M):
        h.close()
    return out


def panel_kernel(op, batch, nnz, N):
    """Which kernel qp_cheby_step_batched takes for a panel of `batch` states on this operator, its symbol (for rocprof / PMC filters)
    and the MATRIX bytes of the layout it reads per term: the CSR-ordered mirror's values (16 B) and columns (4 B) per entry + row
    pointer (8 B) and walk position (4 B) per row for the row kernels; the LDS tiles read no columns (the operands' places are a
    table) and one tile entry per sixteen rows."""
    if batch <= 32:
        return "csr_spmm_kernel (state-tiled)", "csr_spmm_kernel", 20.0 * nnz + 12.0 * N
    t = op.spmm_tiles(batch)
    if not t["taken"]:
        return "spmm_rows_smem_kernel (wave per row, lane = state)", "spmm_rows_smem_kernel", 20.0 * nnz + 12.0 * N
    ft = 16.0 * t["tiles"] / N
    return ("spmm_tile_kernel (4 x 4 rows per workgroup, operands staged in LDS; wave per row, lane = state)", "spmm_tile_kernel",
            nnz * (16.0 * ft + 20.0 * (1.0 - ft)) + 8.0 * N + 4.0 * t["tiles"] + 4.0 * t["rest_rows"])


def measure_batched_c5(ctx, log2n=18, batch=64, steps=5, warmup=2, repeats=3):
    """BASELINE configs[4]: `batch` states x N = 2^log2n CSR H, Chebyshev on the panel."""
    N = 1 << log2n
    rp, col, vals = synth.hermitian_offsets_csr(N)
    nnz = int(rp[-1])
    M = L.Matrix(ctx, N, N, rp, col, vals)
    op = L.Operator(ctx, [M])
    states = np.stack([synth.random_state(N, seed=500 + s) for s in range(batch)], axis=1)
    panel = L.State(ctx, data=states.reshape(-1))
    del states
    wrk = L.ChebyWrk(ctx, N * batch, 20.0, -10.0, 1.0)
    nterms = wrk.n_coeffs - 1
    for _ in range(warmup):
        L.cheby_batched(panel, op, 1.0, wrk, batch)
    regions = timed_regions(ctx, lambda: L.cheby_batched(panel, op, 1.0, wrk, batch), steps, repeats)
    sp = spread([1e3 * r[0] / (steps * nterms) for r in regions], regions)
    ms = sp["median"] * 1e-3 * steps * nterms
    t_term = sp["median"] * 1e-6
    alg = 20.0 * nnz + 4.0 * (N + 1) + 80.0 * N * batch          # SURVEY 8d batched model per term
    sched = L.acc_schedule(wrk.coeffs)
    nupd = sum(0 if d.skip else 1 for d in sched)
    # what the shipped step must move per term: matrix once (CSR mirror: 20 B per entry + row pointers + walk order),
    # X, v_{m-2} read, v_m written, accumulator read + written by the terms of the deferred schedule
    kname, kshort, mbytes = panel_kernel(op, batch, nnz, N)
    lay = mbytes + 16.0 * N * batch * (3.0 - 2.0 / nterms + (2.0 * nupd - 1.0) / nterms)
    norms = np.linalg.norm(panel.numpy().reshape(N, batch), axis=0)
    out = {"workload": f"BASELINE configs[4]: batched Cheby prop_step!, {batch} states x N=2^{log2n} CSR H (16 nnz/row)"
                       + ("" if batch == 64 else f" [the per-GPU share of the 64-state panel split over {64 // batch} GPUs]"),
           "kernel": kname, "kernel_symbol": kshort, "lds_tiles": op.spmm_tiles(batch),
           "N": N, "batch": batch, "steps": steps, "ms_per_panel_step": ms / steps,
           "state_steps_per_s": batch * steps / (ms * 1e-3), "us_per_term": t_term * 1e6,
           "us_per_term_min": sp["min"], "us_per_term_max": sp["max"], "repeats": sp["repeats"], "unstable": sp["unstable"],
           "slowest_region": sp.get("slowest_region"), "regions_remeasured_after_host_stall": sp.get("regions_remeasured_after_host_stall"),
           "row_walk": dict(zip(("inner_dimension", "strip_width"), op.spmm_walk(batch))),
           "layout_bytes_per_term": lay, "layout_gbs": lay / t_term / 1e9, "frac": lay / t_term / 1e9 / HBM_PEAK_GBS, "model": "layout",
           "algorithmic_gbs": alg / t_term / 1e9, "algorithmic_frac": alg / t_term / 1e9 / HBM_PEAK_GBS,
           "max_norm_drift": float(np.max(np.abs(norms - 1.0)))}
    for h in (panel, wrk, op, M):
        h.close()
    return out


def with_panel(ctx, log2n, batch, fn, dt=1.0):
    """The C5 panel (operator, `batch` states, workspace) handed to `fn(step, set_states, read_states, rowptr, col, vals, N, batch,
    dt)` -- bench.py's CPU-baseline leg compares a sample of the panel with the C port through this; nothing here touches the
    oracle."""
    N = 1 << log2n
    rp, col, vals = synth.hermitian_offsets_csr(N)
    M = L.Matrix(ctx, N, N, rp, col, vals)
    op = L.Operator(ctx, [M])
    panel = L.State(ctx, n=N * batch)
    wrk = L.ChebyWrk(ctx, N * batch, 20.0, -10.0, dt)
    try:
        return fn(lambda: L.cheby_batched(panel, op, dt, wrk, batch),
                  lambda st: panel.upload(np.ascontiguousarray(st).reshape(-1)),
                  lambda: panel.numpy().reshape(N, batch), rp, col, vals, N, batch, dt)
    finally:
        for h in (panel, wrk, op, M):
            h.close()


def measure_dense(ctx, N=4096, batch=1, fmt="dense", mfma=1, steps=None, repeats=3, seed=7):
    """A dense Hermitian generator (the reference's own test operators: test/test_cheby.jl:24-47) through the fused Chebyshev
    term: one state (row-sum kernel: 16 N^2 bytes per term) or a panel of `batch` states (H X on the fp64 matrix cores:
    8 N^2 b flop per term; mfma = 0: the sparse panel kernels on the same operator).  fmt "csr" forces the CSR kernels."""
    rng = np.random.default_rng(seed)
    X = rng.standard_normal((N, N)) + 1j * rng.standard_normal((N, N))
    H = (X + X.conj().T) * (5.0 / np.sqrt(8.0 * N))          # GUE scaled to a spectral radius of about 10
    del X
    f = {"dense": L.FMT_DENSE, "csr": L.FMT_CSR, "auto": L.FMT_AUTO}[fmt]
    rp = np.arange(N + 1, dtype=np.int64) * N
    col = np.tile(np.arange(N, dtype=np.int32), N)
    M = L.Matrix(ctx, N, N, rp, col, H.reshape(-1))
    del H, rp, col
    op = L.Operator(ctx, [M], 0, f)
    dt = 0.5
    wrk = L.ChebyWrk(ctx, N * batch, 24.0, -12.0, dt)
    nterms = wrk.n_coeffs - 1
    states = rng.standard_normal(N * batch) + 1j * rng.standard_normal(N * batch)
    psi = L.State(ctx, data=states / np.linalg.norm(states) * np.sqrt(batch))
    saved = ctx.tuning_get("dense_panel_mfma")
    ctx.tuning_set("dense_panel_mfma", mfma)
    try:
        step = (lambda: L.cheby(psi, op, dt, wrk)) if batch == 1 else (lambda: L.cheby_batched(psi, op, dt, wrk, batch))
        step()
        ctx.sync()
        if steps is None:      # about 20 ms per timed region
            ctx.timer_begin()
            step()
            one = max(ctx.timer_end(), 1e-3)
            steps = int(max(1, min(50, 20.0 / one)))
        # steady-state clocks: the first ~20 ms after an idle gap (the host built a 4096 x 4096 matrix just now) run at the clock the
        # chip idled at -- 146 - 160 us per term instead of 127.5 - 128.5 for the 64-state panel, profiles/r05/dense_modes.txt
        t_warm = time.perf_counter()
        while time.perf_counter() - t_warm < 0.25:
            step()
        ctx.sync()
        regions = timed_regions(ctx, step, steps, repeats)
    finally:
        ctx.tuning_set("dense_panel_mfma", saved)
    sp = spread([1e3 * r[0] / (steps * nterms) for r in regions], regions)
    t = sp["median"] * 1e-6
    byts = 16.0 * N * N + 80.0 * N * batch
    flops = 8.0 * N * N * batch
    out = {"workload": f"dense Hermitian H, N = {N}, complex fp64" + (f", panel of {batch} states" if batch > 1 else ", one state"),
           "N": N, "batch": batch, "device_format": FMT_NAME[op.format], "n_coeffs": int(wrk.n_coeffs),
           "kernel": ("dense_gemv_kernel" if op.format == L.FMT_DENSE else "csr_spmv_kernel") if batch == 1 else
                     ("dense_zgemm_cheby_kernel (MFMA)" if (op.format == L.FMT_DENSE and mfma) else "sparse panel kernels"),
           "us_per_term": sp["median"], "us_per_term_min": sp["min"], "us_per_term_max": sp["max"], "repeats": sp["repeats"],
           "unstable": sp["unstable"], "operator_build_ms": op.build_info()["build_ms"],
           "bytes_per_term": byts, "gbs": byts / t / 1e9, "frac": byts / t / 1e9 / HBM_PEAK_GBS,
           "tflops": flops / t / 1e12, "frac_fp64_matrix_peak": flops / t / 1e12 / 78.6,
           "bound": "mfma" if batch >= 20 else "hbm", "model": "flops" if batch >= 20 else "layout", "norm_drift": abs(psi.norm() / np.sqrt(batch) - 1.0)}
    for h in (psi, wrk, op, M):
        h.close()
    return out


def measure_liouville(ctx, n=512, nc=2, reps=20):
    """one application L rho of the matrix-free Liouvillian (dense H, nc dense Lindblad operators) on the
    fp64 matrix cores -- the hand-written kernel the library picks for this n, and the chain of library GEMMs beside it."""
    rng = np.random.default_rng(0)
    H = synth.dense_hermitian(n, rho=2.0, rng=rng)
    cops = [0.2 * (rng.standard_normal((n, n)) + 1j * rng.standard_normal((n, n))) / np.sqrt(n) for _ in range(nc)]
    Lmf = L.Liouvillian(ctx, [H], cops, convention="TDSE")
    x = L.State(ctx, data=(rng.standard_normal(n * n) + 1j * rng.standard_normal(n * n)))
    y = L.State(ctx, n=n * n)
    flops = 8.0 * n ** 3 * (2 + 2 * nc)
    keys = ("liouville_fused_n", "liouville_tile32_n")
    saved = {k: ctx.tuning_get(k) for k in keys}
    res, sps = {}, {}
    try:
        for name, override in (("hand_written", None), ("library_chain", 0)):
            if override is not None:
                for k in keys:
                    ctx.tuning_set(k, override)
            for _ in range(3):
                Lmf.mul(x, y)
            regions = timed_regions(ctx, lambda: Lmf.mul(x, y), reps, 3)
            sps[name] = spread([1e3 * r[0] / reps for r in regions], regions)
            res[name] = sps[name]["median"]
    finally:
        for k, v in saved.items():
            ctx.tuning_set(k, v)
    out = {"workload": f"matrix-free Liouvillian, one application L rho: n = {n} (N = n^2 = {n * n}), {nc} Lindblad operators, "
                       f"{2 + 2 * nc} complex n x n products", "n": n, "c_ops": nc,
           "kernel": "zgemm_sum32_kernel (32 x 32 tile per workgroup, v_mfma_f64_16x16x4_f64)" if 260 <= n <= 2048
                     else "zgemm_sum_kernel (16 x 16 tile per workgroup)",
           "us_per_apply": res["hand_written"], "us_per_apply_min": sps["hand_written"]["min"], "us_per_apply_max": sps["hand_written"]["max"],
           "repeats": 3, "unstable": sps["hand_written"]["unstable"], "tflops": flops / res["hand_written"] / 1e6,
           "frac_fp64_matrix_peak": flops / res["hand_written"] / 1e6 / 78.6, "bound": "mfma", "model": "flops",
           "us_per_apply_rocblas_chain": res["library_chain"], "tflops_rocblas_chain": flops / res["library_chain"] / 1e6}
    for h in (x, y, Lmf):
        h.close()
    return out
