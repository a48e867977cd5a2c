#!/usr/bin/env python3
"""Headline benchmark: Chebyshev ``prop_step!``/s on BASELINE.json's configs.

    python bench.py                      # = --gpus 1 --steps 100 --warmup 10: config C2, a few minutes with
                                         #   the PMC passes, the CPU baselines and the extra points
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One "step" = one ``prop_step!`` = one pass of the hot path (31 fused SpMV terms at dt = 1) over the state.

* N = 1 (``--config c2``, the default there): BASELINE configs[1], N = 2^20 rows, CSR sparse Hermitian H with
  16 nnz/row, complex fp64, manual spectral range [-10, 10].
* N > 1 (``--config c4``, the default there): BASELINE configs[3], 2^21 rows per GPU (N = 2^24 at 8 GPUs) row-partitioned,
  the needed slices of the term vector exchanged over RCCL after every mat-vec; weak scaling.  ``value`` counts
  2^20-row blocks advanced per second, the unit of the one-GPU line, so the driver's per-N values compare.  The
  STRONG point BASELINE's metric also names (N = 2^20 in total, split over the ranks) is measured after the timed
  region and reported under ``strong_scaling_point``; ``--config c2 --scaling strong`` makes it the headline.

Rank 0 prints ONE compact JSON line LAST (< 4 KB: the contract's keys, the gate scalars of ``roofline`` and ``cpu_baseline``,
one [time, fraction] pair per extra point); the complete record of the run -- every extra point, the prediction table, the
notes -- goes to ``bench_extras.json`` next to this script (and under ``gpurun_out/`` where that directory exists).
``roofline.frac`` prices the bytes the SHIPPED device layout has to move
(tools/bench_points.py: cheby_layout_bytes) and cannot exceed 1; the contract's CSR figure is kept as
``roofline.effective_csr_equiv_gbs``.  ``roofline.traffic`` is measured in this run: two child processes of this
script under ``rocprofv3 --pmc`` (FETCH_SIZE, WRITE_SIZE; one pass each as the pool requires), or, if the profiler
is not usable, the value of the committed profile, labelled as such.
"""
import argparse
import csv
import gc
import glob
import json
import os
import shutil
import signal
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import numpy as np  # noqa: E402
import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0     # MI355X HBM3E spec peak (MI355X_MICROARCH.md: 8 TB/s; ~6.3 TB/s measured copy)
KERNEL_OF_FORMAT = {1: "csr_spmv_kernel", 2: "rbcsr_spmv_kernel", 3: "hrb_spmv_kernel"}
STATIC_PMC = os.path.join("profiles", "r06", "bench_pmc_summary.json")


def pmc_traffic(argv_inner, kernel_substr, timeout_s, how="mean"):
    """HBM bytes of the named kernel(s), measured now: this script re-run as a CHILD process under rocprofv3, one pass per
    counter (the guide's HBM / rocprofv3 recipe: FETCH_SIZE x 2 on gfx950, WRITE_SIZE as is, units KiB).  `kernel_substr`: one
    substring or a tuple of them; how = "mean": bytes per launch (mean over the matching dispatches), "sum": bytes of ALL
    matching dispatches of the child run (the caller divides by the steps the child ran).  Returns (bytes, detail) or
    (None, reason)."""
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return None, "rocprofv3 not found"
    subs = (kernel_substr,) if isinstance(kernel_substr, str) else tuple(kernel_substr)
    env = dict(os.environ, TMPDIR="/tmp")
    vals = {}
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        out = tempfile.mkdtemp(prefix="qp_pmc_", dir="/tmp")
        cmd = [exe, "--pmc", counter, "--kernel-trace", "--output-format", "csv", "-d", out, "--",
               sys.executable, os.path.join(ROOT, "bench.py")] + argv_inner
        try:   # own process group: a pass that hangs is ended with everything it started (the profiled child too)
            proc = subprocess.Popen(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL,
                                    start_new_session=True)
            try:
                proc.wait(timeout=timeout_s)
            except subprocess.TimeoutExpired:
                try:
                    os.killpg(proc.pid, signal.SIGKILL)
                except OSError:
                    pass
                proc.wait()
                shutil.rmtree(out, ignore_errors=True)
                return None, f"{counter} pass failed: timeout after {timeout_s} s"
            r = proc
        except OSError as e:
            shutil.rmtree(out, ignore_errors=True)
            return None, f"{counter} pass failed: {type(e).__name__}"
        got = []
        for f in glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True):
            with open(f) as fh:
                for row in csv.DictReader(fh):
                    if row["Counter_Name"] == counter and any(s in row["Kernel_Name"] for s in subs):
                        got.append(float(row["Counter_Value"]))
        shutil.rmtree(out, ignore_errors=True)
        if r.returncode != 0 or not got:
            return None, f"{counter} pass: rc {r.returncode}, {len(got)} dispatches of {'/'.join(subs)}"
        vals[counter] = ((sum(got) if how == "sum" else sum(got) / len(got)) * 1024.0, len(got))
    fetch, write = 2.0 * vals["FETCH_SIZE"][0], vals["WRITE_SIZE"][0]
    return fetch + write, {"FETCH_SIZE_bytes_x2": fetch, "WRITE_SIZE_bytes": write,
                           "dispatches": [vals["FETCH_SIZE"][1], vals["WRITE_SIZE"][1]]}


# kernels of one newton! step (csrc/kernels_arnoldi.hip, kernels.hip): what a C3 child run's traffic is summed over
NEWTON_KERNELS = ("arnoldi_matvec_dots_kernel", "mgs_update_kernel", "multidot_kernel", "multidot_reduce_kernel", "combine2_vecs_kernel",
                  "combine_vecs_kernel", "norm_guard_scale_kernel", "rbcsr_spmv_kernel", "csr_spmv_kernel", "mgs_pass_kernel",
                  "arnoldi_onepass_kernel", "arnoldi_onepass_solve_kernel", "rbcsr_coded_spmv_kernel")


def cpu_baseline_c5(L, synth, step_panel, set_panel, read_panel, rp, col, vals, N, batch, dt, cpu_steps, first_state=0):
    """`cpu_baseline` of the batched config (BASELINE configs[4]): oracle/cheby_ref.c -- the serial CSC path of the reference,
    one state at a time, as the reference's own loop over states would run it -- on a SAMPLE of the panel's states (the
    first 4), `cpu_steps` (<= 16) prop_steps each: same H, same coefficients; then the same steps on the GPU panel from the same
    initial states and the largest l2 difference over the sampled states.  Checker / reported baseline only."""
    from oracle import ref_c
    S, K = min(4, batch), max(1, min(int(cpu_steps), 16))
    init = np.stack([synth.random_state(N, seed=500 + first_state + s) for s in range(batch)], axis=1)
    set_panel(init)
    for _ in range(K):
        step_panel()
    gpu = read_panel()[:, :S].copy()
    coeffs = L.cheby_coeffs(20.0, dt)
    colptr, rowval, nzval = rp, col.astype(np.int64), np.conj(vals)      # Hermitian: CSC(H) = conj CSR(H)
    ref_c.load()
    worst = 0.0
    tc = 0.0
    for s_ in range(S):
        cpsi = init[:, s_].copy()
        t0 = time.perf_counter()
        for _ in range(K):
            ref_c.cheby_csc(colptr, rowval, nzval, cpsi, coeffs, 20.0, -10.0, dt)
        tc += time.perf_counter() - t0
        worst = max(worst, float(np.linalg.norm(gpu[:, s_] - cpsi)))
    return {"value": S * K / tc, "unit": "state_step/s", "cores": 1, "kind": "port",
            "sample": f"{K} prop_steps of {S} of the {batch} states, N={N} (oracle/cheby_ref.c: serial CSC SpMV + BLAS-1, one state at a "
                      f"time; gcc {ref_c.build_flags()})",
            "ms_per_step": 1e3 * tc / (S * K), "l2_diff_vs_gpu_after_sample": worst, "states_sampled": S, "steps_sampled": K}


def cpu_baseline_c3(ctx, L, synth, n=512, m=20, dt=0.5, steps=2):
    """`cpu_baseline` of the Newton config (BASELINE configs[2]): oracle/newton_ref.c (serial CSC mat-vec, the reference's sequential
    modified Gram-Schmidt and combinations, one core) with the NumPy oracle's host algebra, `steps` newton! steps from rho_0 at
    the point's own size; the same steps on the GPU and the l2 difference, restart counts of both.  Checker / reported baseline only."""
    from oracle import ref_c
    Lm = synth.liouvillian_tridiag(n)
    N = Lm.shape[0]
    rho0 = synth.random_state(N)
    M = L.Matrix.from_scipy(ctx, Lm)
    op = L.Operator(ctx, [M])
    wrk = L.NewtonWrk(ctx, N, m_max=m)
    psi = L.State(ctx, data=rho0)
    r_gpu = []
    for _ in range(steps):
        L.newton(psi, op, dt, wrk)
        r_gpu.append(int(wrk.restarts))
    gpu = psi.numpy()
    for h in (psi, wrk, op, M):
        h.close()
    Lc = Lm.tocsc()
    Lc.sort_indices()
    colptr, rowval, nzval = Lc.indptr.astype(np.int64), Lc.indices.astype(np.int64), Lc.data.astype(np.complex128)
    ref_c.load()
    wc = ref_c.NewtonCscWrk(N, m_max=m)
    cpsi = rho0.copy()
    r_cpu = []
    t0 = time.perf_counter()
    for _ in range(steps):
        ref_c.newton_csc(colptr, rowval, nzval, cpsi, dt, wc)
        r_cpu.append(int(wc.restarts))
    tc = time.perf_counter() - t0
    return {"value": steps / tc, "unit": "prop_step/s", "cores": 1, "kind": "port",
            "sample": f"{steps} newton! steps from rho_0 of the same N={N} Liouvillian, m_max={m} (oracle/newton_ref.c: serial CSC SpMV, "
                      f"sequential MGS, combinations; host algebra by the NumPy oracle; gcc {ref_c.build_flags()})",
            "ms_per_step": 1e3 * tc / steps, "l2_diff_vs_gpu_after_sample": float(np.linalg.norm(gpu - cpsi)),
            "restarts_cpu": r_cpu, "restarts_gpu": r_gpu, "matvecs_cpu": int(wc.n_matvec)}


def run_point(args, L, bp):
    """`--point c3|c5` (used by the PMC child runs of the extras): that one measurement, nothing else, a short JSON line."""
    ctx = L.Context(0)
    if args.point == "c3":
        r = bp.measure_newton_c3(ctx, steps=args.steps, warmup=args.steps, repeats=1)      # 2 x steps identical newton! steps from rho_0
    else:
        r = bp.measure_batched_c5(ctx, batch=args.batch, steps=args.steps, warmup=1, repeats=1)
    print(json.dumps({"point": args.point, "steps": args.steps, "ms": r.get("ms_per_step", r.get("ms_per_panel_step"))}))


XGMI_LINK_GBS = 153.0      # per direction and link (MI355X_MICROARCH.md: 7 links x ~153 GB/s per GPU, point to point)


def exchange_model(sh, world, rows, us_per_term):
    """What the exchange of one fused term should cost on an 8-GPU xGMI node, from the partition alone, so that the first
    real multi-GPU run can be read against a prediction: bytes every rank sends per term, the peers it sends to (one
    point-to-point link each), the transfer time at the link rate plus a fixed start-up, and how much of it the
    boundary / interior overlap can hide behind the interior launch (measured per-term time of this run)."""
    M = int(sh.M)
    if sh.exchange == "halo" and sh.send_to is not None:
        peers = max(len(sh.send_to), 1)
        sent = 16.0 * M                                  # the packed send rows, once per peer that reads any of them
        per_link = sent                                  # worst link: the whole slab to one neighbour
        kind = "neighbour send/recv of the packed halo rows"
    else:
        peers = max(world - 1, 1)
        sent = 16.0 * M * peers                          # all-gather: this rank's slab to every other rank
        per_link = 16.0 * M
        kind = "all-gather of every rank's slab"
    startup_us = 10.0                                    # collective / send-recv launch + first-byte latency (assumed)
    transfer_us = per_link / (XGMI_LINK_GBS * 1e9) * 1e6
    predicted = startup_us + transfer_us
    return {"kind": kind, "rows_sent_per_rank_per_term": M, "bytes_sent_per_rank_per_term": sent, "peers": peers,
            "bytes_on_busiest_link_per_term": per_link, "link_gbs_assumed": XGMI_LINK_GBS, "startup_us_assumed": startup_us,
            "predicted_exchange_us_per_term": predicted, "measured_us_per_term_all_in": us_per_term,
            "predicted_exposed_us_per_term_serial_schedule": predicted,
            "predicted_exposed_us_per_term_overlap_schedule": max(0.0, predicted - 0.9 * us_per_term),
            "note": "prediction from the partition (no measurement): serial schedule = exchange in line after every term; overlap "
                    "schedule = exchange behind the interior launch, exposed only where it is longer than that launch"}


MACHINERY_FACTOR = 74.7 / 68.8      # overlapped native step / plain term at 2^21 rows on one GPU (profiles/r03/sharded_machinery_1gpu.txt)
STATIC_SIZES = os.path.join("profiles", "r06", "single_gpu_sizes.json")


def scaling_prediction(us_per_term_by_log2rows, value_1gpu, nterms, source, one_term=None):
    """What 1 / 2 / 4 / 8 GPUs of one xGMI node should deliver, from per-term times measured on ONE GPU at the row counts a rank
    would own plus the exchange model -- so that the first real multi-GPU run is read against a table, not a guess.

    FIXED problem (N = 2^24, ">= 6x at 8 GPUs vs 1"): G ranks own 2^24 / G rows each.
      halo form (banded H: each rank sends its 2 x 4096 edge rows to its two neighbours, ncclSend/ncclRecv): 10 us start-up +
        131 KB / 153 GB/s; overlapped schedule = machinery factor x compute (the exchange hides behind the interior launch while
        it is shorter than 0.9 of it), serial schedule = compute + exchange;
      all-gather form (the collective north_star names; what a scattered H needs): every rank receives the other G - 1 slices,
        16 N / G bytes over each of G - 1 links in parallel; every row reads remote data, so nothing hides it: compute + exchange
        (compute priced with the banded operator's time: an upper bound on what a scattered operator reaches).
    WEAK default of `bench.py --gpus G` (2^21 rows per GPU): value in 2^20-row blocks per second against this run's 1-GPU value
    (N = 2^20, Infinity-Cache resident) -- the efficiency the driver will compute."""
    t = {int(k): float(v) for k, v in us_per_term_by_log2rows.items()}
    # A rank of the row-partitioned step launches its terms one by one (it exchanges after every term): its compute time is the ONE-TERM
    # walk's, while one GPU alone takes the two-term walk beyond the Infinity Cache (round 6) -- the denominator got faster, the ranks did not.
    t_rank = {int(k): float(v) for k, v in (one_term or {}).items()}
    startup, link = 10.0, XGMI_LINK_GBS * 1e3      # us, bytes per us
    fixed = []
    for G in (1, 2, 4, 8):
        lg = 24 - G.bit_length() + 1
        tc = t.get(lg) if G == 1 else t_rank.get(lg, t.get(lg))
        if tc is None:
            continue
        row = {"gpus": G, "rows_per_gpu": 1 << lg, "us_per_term_compute": tc}
        if G == 1:
            row.update(us_per_term_halo_overlap=tc, us_per_term_halo_serial=tc, us_per_term_allgather=tc, exchange_us_halo=0.0,
                       exchange_us_allgather=0.0)
        else:
            xh = startup + 16.0 * 2 * 4096 / link
            xa = startup + 16.0 * (1 << lg) / link
            row.update(exchange_us_halo=xh, exchange_us_allgather=xa,
                       us_per_term_halo_overlap=MACHINERY_FACTOR * tc + max(0.0, xh - 0.9 * tc),
                       us_per_term_halo_serial=tc + xh, us_per_term_allgather=tc + xa)
        fixed.append(row)
    if fixed and fixed[0]["gpus"] == 1:
        t1 = fixed[0]["us_per_term_compute"]
        for row in fixed:
            for k in ("halo_overlap", "halo_serial", "allgather"):
                row["speedup_" + k] = t1 / row["us_per_term_" + k]
                row["prop_steps_per_s_" + k] = 1e6 / (nterms * row["us_per_term_" + k])
    weak = []
    if t.get(21):
        for G in (1, 2, 4, 8):
            per = MACHINERY_FACTOR * t_rank.get(21, t[21]) if G > 1 else t[21]
            v = G * 2.0 * 1e6 / (nterms * per)
            weak.append({"gpus": G, "rows_per_gpu": 1 << 21, "predicted_value_blocks_per_s": v,
                         "predicted_efficiency_vs_1gpu_value": (v / (G * value_1gpu)) if value_1gpu else None})
    return {"source_of_compute_times": source, "us_per_term_by_log2_rows": t, "us_per_term_one_term_walk_by_log2_rows": t_rank or None, "terms_per_step": nterms,
            "assumptions": {"xgmi_link_gbs": XGMI_LINK_GBS, "exchange_startup_us": startup, "overlap_machinery_factor": MACHINERY_FACTOR,
                            "halo_rows_per_neighbour": 8192},
            "fixed_problem_N_2^24": fixed, "bench_default_weak_2^21_rows_per_gpu": weak,
            "reading": "halo form: >= 6x at 8 GPUs needs only the per-GPU kernel at 2^21 rows; all-gather form: link-bound "
                       "(32 MiB per link and term at 8 GPUs), it cannot reach 6x on 153 GB/s links"}


def prediction_scalars(pred):
    """The few numbers of the prediction table a reader of the printed line needs (the table itself: bench_extras.json)."""
    if not pred:
        return None
    out = {}
    for row in pred.get("fixed_problem_N_2^24") or []:
        if row["gpus"] == 8 and "speedup_halo_overlap" in row:
            out["fixed_n24_speedup_8gpu_halo"] = row["speedup_halo_overlap"]
            out["fixed_n24_speedup_8gpu_allgather"] = row["speedup_allgather"]
    for row in pred.get("bench_default_weak_2^21_rows_per_gpu") or []:
        if row["gpus"] == 8:
            out["weak_8gpu_blocks_per_s"] = row["predicted_value_blocks_per_s"]
            out["weak_8gpu_efficiency"] = row["predicted_efficiency_vs_1gpu_value"]
    out["source"] = "STATIC" if str(pred.get("source_of_compute_times", "")).startswith("STATIC") else "this run"
    return out or None


def scaling_prediction_static(nterms):
    """The same table in a multi-GPU line, from the committed single-GPU measurements (this run cannot measure them)."""
    path = os.path.join(ROOT, STATIC_SIZES)
    if not os.path.exists(path):
        return None
    with open(path) as f:
        d = json.load(f)
    return scaling_prediction(d["us_per_term_by_log2_rows"], d.get("value_1gpu"), nterms,
                              f"STATIC: {STATIC_SIZES} (single-GPU run of bench.py on another box)", one_term=d.get("us_per_term_one_term_walk_by_log2_rows"))


LINE_LIMIT = 4096          # the driver parses the LAST stdout line; r04's 36 KB line defeated it
EXTRAS_FILE = "bench_extras.json"


def _short(s, n=160):
    s = str(s)
    return s if len(s) <= n else s[:n - 3] + "..."


def _num(x, digits=6):
    """Floats at 6 significant digits (the line is a gate record, not an archive); everything else as is."""
    if isinstance(x, float):
        return float(f"{x:.{digits}g}")
    return x


def _pick(d, keys):
    return {k: _num(d[k]) for k in keys if d and k in d and d[k] is not None}


def compact_line(full):
    """The one JSON object the driver reads: few, scalar, early.  `full` is the complete record of the run (it goes to
    bench_extras.json); this keeps the contract's keys, the gate scalars of `roofline` / `cpu_baseline`, and ONE
    (time, fraction) pair per extra point.  No string longer than 160 characters, the whole line below LINE_LIMIT."""
    cfg = full.get("config") or {}
    rf = full.get("roofline") or {}
    line = {k: _num(full.get(k)) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
                                           "scaling", "vs_baseline", "dtype", "data")}
    line["metric"] = _short(line["metric"])
    c = _pick(cfg, ("N_total", "N", "rows_per_gpu", "states_per_gpu", "nnz_per_row", "pattern", "n_coeffs", "matvecs_per_step", "dt",
                    "device_format", "operator_build_ms"))
    c = {"workload": _short(cfg.get("workload_short") or cfg.get("workload")), **c,
         "parallelism": _short(cfg.get("parallelism_short") or cfg.get("parallelism"))}
    line["config"] = c
    r = {"bound": rf.get("bound"), "achieved": _num(rf.get("achieved")), "peak": rf.get("peak"), "unit": rf.get("unit"),
         "frac": _num(rf.get("frac")), "traffic": _num(rf.get("traffic"))}
    r.update(_pick(rf, ("kernel", "avg_launch_us", "bytes_per_launch", "model", "cache_resident", "traffic_over_bytes", "unstable",
                        "hbm_resident_frac", "hbm_resident_us_per_term", "n23_frac", "n23_us_per_term",
                        "fixed_problem_n24_frac", "fixed_problem_n24_us_per_term", "fixed_problem_n24_blocks_per_s",
                        "stream_read_gbs", "stream_walk_mix_gbs", "hbm_resident_frac_of_stream_mix", "fixed_problem_n24_frac_of_stream_mix",
                        "cache_stream_read_gbs", "cache_stream_mix_gbs", "traffic_rate_gbs",
                        "csr_equiv_frac", "algorithmic_frac", "traffic_measured")))
    if "kernel" in r:
        r["kernel"] = _short(r["kernel"], 96)
    line["roofline"] = r
    cb = full.get("cpu_baseline")
    line["cpu_baseline"] = None if not cb else {
        **_pick(cb, ("value", "unit", "cores", "kind")), "sample": _short(cb.get("sample", "")),
        **_pick(cb, ("ms_per_step", "l2_diff_vs_gpu_after_sample"))}
    ca = full.get("cpu_baseline_all_cores")
    if ca:
        line["cpu_baseline_all_cores"] = _pick(ca, ("value", "cores", "ms_per_step"))
    for k in ("strong_point", "allgather_form", "prediction", "conservative_first"):
        if full.get(k):
            line[k] = {kk: (_short(v, 96) if isinstance(v, str) else _num(v)) for kk, v in full[k].items()}
    # one pair per extra point: [time in the point's own unit (us per term / ms per step / us per apply), fraction of its roofline]
    pts = {}
    for name, pt in (full.get("extras") or {}).items():
        if not isinstance(pt, dict):
            continue
        if "error" in pt:
            pts[name] = "error"
            continue
        t = next((pt[k] for k in ("us_per_term", "ms_per_step", "us_per_apply") if pt.get(k) is not None), None)
        f = next((pt[k] for k in ("frac", "frac_fp64_matrix_peak") if pt.get(k) is not None), None)
        if pt.get("bound") == "mfma" and pt.get("frac_fp64_matrix_peak") is not None:
            f = pt["frac_fp64_matrix_peak"]
        pts[name] = [_num(t, 4), _num(f, 3), pt.get("model", "?")]
        cb = pt.get("cpu_baseline")
        if isinstance(cb, dict) and cb.get("ms_per_step"):      # the reference's CPU path beside the point: ms per (state-)step, one core
            pts[name].append(_num(cb["ms_per_step"], 4))
    if pts:
        line["points"] = pts
        line["points_unit"] = ("[us/term | ms/step (c3*) | us/apply (n4*), frac of peak by the named byte model (layout | impl | flops), "
                               "CPU port ms per (state-)step if timed]")
    for k in ("exchange", "rccl_ranks", "max_norm_drift", "degraded", "native_path"):
        if k in full:
            line[k] = _num(full[k])
    line["extras_file"] = EXTRAS_FILE
    return line


def emit(full, tag=""):
    """Rank 0: the complete record to bench_extras.json (next to this script, and under gpurun_out/ where that exists so that
    it travels back from a GPU box; `tag`: "_c5", "_c4", "_gpus8" .. for the runs that are not the default headline), one short
    `EXTRA ` line per extra point, then -- LAST -- the compact line."""
    name = EXTRAS_FILE.replace(".json", f"{tag}.json")
    for d in (ROOT, os.path.join(ROOT, "gpurun_out")):
        if os.path.isdir(d):
            try:
                with open(os.path.join(d, name), "w") as f:
                    json.dump(full, f, indent=1)
            except OSError:
                pass
    line = compact_line(full)
    line["extras_file"] = name
    for name, v in (line.get("points") or {}).items():
        print("EXTRA " + json.dumps({name: v}))
    text = json.dumps(line, separators=(",", ":"))
    if len(text) >= LINE_LIMIT:      # never print a line the driver cannot read: drop the optional parts, largest first
        for k in ("points", "points_unit", "prediction", "conservative_first", "cpu_baseline_all_cores"):
            line.pop(k, None)
            text = json.dumps(line, separators=(",", ":"))
            if len(text) < LINE_LIMIT:
                break
    sys.stdout.flush()
    print(text)
    sys.stdout.flush()


def run_c5(args, world, rank, local_rank, one_gpu, dist, L, synth, bp):
    """BASELINE configs[4]: 64 states x N = 2^18 CSR H, Chebyshev on the panel, the batch split over the GPUs --
    rank r advances states [r b, (r + 1) b), b = 64 / world, with its own copy of H and no communication (SURVEY 8e
    "Batched").  One step = one prop_step! of the whole 64-state panel; `value` = state-steps per second of the job."""
    if args.batch % world:
        raise SystemExit(f"--config c5: {world} ranks do not divide the {args.batch}-state panel")
    b = args.batch // world
    log2n = args.log2n if args.log2n is not None else 18
    N = 1 << log2n
    import qprop_amd.sharded as sharded
    stream = torch.cuda.current_stream().cuda_stream
    ctx = L.Context(local_rank, stream=stream)
    rp, col, vals = synth.hermitian_offsets_csr(N)
    nnz = int(rp[-1])
    bs = sharded.BatchSplitCheby(ctx, rp, col, vals, N, args.batch, 20.0, -10.0, args.dt, rank=rank, world=world)
    bs.set_states(np.stack([synth.random_state(N, seed=500 + s) for s in range(bs.s0, bs.s1)], axis=1))
    op, wrk = bs.impl.op, bs.impl.wrk
    nterms = wrk.n_coeffs - 1

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        bs.step()
    barrier()
    ctx.timer_begin()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        bs.step()
    ev_ms = ctx.timer_end()
    barrier()
    el = time.perf_counter() - t0
    norms = np.linalg.norm(bs.local_states(), axis=0)
    drift = float(np.max(np.abs(norms - 1.0)))
    if dist is not None:
        t = torch.tensor([el, ev_ms, drift], dtype=torch.float64, device="cpu" if one_gpu else "cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        el, ev_ms, drift = float(t[0]), float(t[1]), float(t[2])
    t_term = ev_ms * 1e-3 / (args.steps * nterms)
    sched = L.acc_schedule(wrk.coeffs)
    nupd = sum(0 if d.skip else 1 for d in sched)
    # bytes one GPU must move per fused term: the matrix once (CSR mirror), its share of X, v_{m-2}, v_m, the accumulator
    kname, kshort, mbytes = bp.panel_kernel(op, b, nnz, N)
    lay = mbytes + 16.0 * N * b * (3.0 - 2.0 / nterms + (2.0 * nupd - 1.0) / nterms)
    alg = 20.0 * nnz + 4.0 * (N + 1) + 80.0 * N * b
    out = {"metric": "batched Cheby prop_step!: state-steps/s, 64 states x N=2^18 CSR fp64 (BASELINE configs[4])",
           "value": args.batch * args.steps / el, "unit": "state_step/s", "n_gpus": world, "steps": args.steps,
           "warmup": args.warmup, "ms_per_step": 1e3 * el / args.steps, "higher_is_better": True, "scaling": "strong",
           "vs_baseline": None, "dtype": "c128 (complex fp64)", "data": "synthetic",
           "config": {"workload": f"BASELINE configs[4]: batched Cheby prop_step!, {args.batch} states x N=2^{log2n} CSR H (16 nnz/row), "
                                  f"batch split over {world} GPU(s): {b} states per GPU, H replicated, no communication",
                      "states_per_gpu": b, "N": N, "n_coeffs": int(wrk.n_coeffs), "matvecs_per_step": nterms, "dt": args.dt,
                      "kernel": kname, "lds_tiles": op.spmm_tiles(b),
                      "row_walk": dict(zip(("inner_dimension", "strip_width"), op.spmm_walk(b))),
                      "parallelism": "single GPU" if world == 1 else f"batch-split x{world} (replicas of H, zero communication)"
                                     + (" [TEST MODE: ranks share one GPU]" if one_gpu else "")},
           "roofline": {"bound": "hbm", "achieved": lay / t_term / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": lay / t_term / 1e9 / HBM_PEAK_GBS, "traffic": None,
                        "layout_bytes_per_launch_per_gpu": lay, "avg_launch_us": t_term * 1e6,
                        "algorithmic_bytes_per_launch_per_gpu": alg, "algorithmic_frac": alg / t_term / 1e9 / HBM_PEAK_GBS,
                        "note": "per GPU (slowest rank): bytes the shipped step must move per fused term (matrix once, this GPU's "
                                "share of the panel streams) / average launch duration from HIP events on the kernels' stream"},
           "cpu_baseline": None, "max_norm_drift": drift, "degraded": False}
    out["config"]["workload_short"] = f"BASELINE configs[4]: batched Cheby prop_step!, {args.batch} states x N=2^{log2n} CSR H, 16 nnz/row"
    out["config"]["parallelism_short"] = ("single GPU" if world == 1 else f"batch-split x{world}, H replicated, no communication") + (
        " [TEST MODE: one GPU]" if one_gpu else "")
    out["config"]["nnz_per_row"] = nnz / N
    out["config"]["device_format"] = bp.FMT_NAME[op.format]
    rfl = out["roofline"]
    rfl["kernel"] = kshort
    rfl["bytes_per_launch"] = lay
    rfl["model"] = "layout"
    # the reference's CPU path beside it (rank 0 of a single-GPU run, after the timed region): a sample of the panel's states
    if rank == 0 and world == 1 and args.cpu_steps > 0:
        out["cpu_baseline"] = cpu_baseline_c5(L, synth, bs.step, bs.set_states, bs.local_states, rp, col, vals, N, args.batch, args.dt,
                                              args.cpu_steps, first_state=bs.s0)
    # HBM traffic of the panel kernel (this line carried "traffic": null): the same child-process PMC
    # passes as the headline, on rank 0 of a single-GPU run, for the panel width this run's GPUs see
    if rank == 0 and world == 1 and not args.no_pmc and args.batch == 64 and log2n == 18:
        bs.close()
        bs = None
        tr, det = pmc_traffic(["--point", "c5", "--steps", "2"], kshort, timeout_s=240, how="mean")
        rfl = out["roofline"]
        rfl["traffic"] = tr
        rfl["traffic_source"] = ("measured in this run: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE child runs of `bench.py --point c5`, "
                                 "FETCH_SIZE x 2 (gfx950), mean per launch") if tr is not None else f"not measured: {det}"
        rfl["traffic_over_layout_bytes"] = (tr / lay) if tr else None
        rfl["traffic_over_bytes"] = rfl["traffic_over_layout_bytes"]
        rfl["traffic_measured"] = tr is not None
    if rank == 0:
        emit(out, "_c5" + (f"_gpus{world}" if world > 1 else ""))
    if bs is not None:
        bs.close()
    if dist is not None:
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--config", default="auto", choices=["auto", "c2", "c4", "c5"],
                    help="c2: BASELINE configs[1], 2^20 rows per GPU (default at 1 GPU); c4: configs[3], 2^21 rows per GPU = "
                         "N 2^24 at 8 GPUs (default at more than 1); c5: configs[4], 64 states x N = 2^18, the batch split over "
                         "the GPUs (H replicated, no communication)")
    ap.add_argument("--batch", type=int, default=64, help="--config c5: states in the whole panel")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="weak: rows per GPU fixed; strong: 2^log2n rows in TOTAL, split over the GPUs")
    ap.add_argument("--log2n", type=int, default=None, help="rows per GPU (weak) or in total (strong) = 2^log2n; overrides --config")
    ap.add_argument("--pattern", default="banded", choices=["banded", "scattered", "random", "random-window"])
    ap.add_argument("--format", default="auto", choices=["auto", "hrb", "rbcsr", "csr"])
    ap.add_argument("--exchange", default="auto", choices=["auto", "halo", "allgather"])
    ap.add_argument("--dt", type=float, default=1.0, help="time step; alpha = 10 dt, i.e. 32 coefficients at dt = 1 "
                    "(SURVEY 8d also asks for alpha = 2 and 50: --dt 0.2 / --dt 5)")
    ap.add_argument("--real", action="store_true", help="real-symmetric H (the f64 variant of SURVEY 8d): values "
                    "are streamed as fp64")
    ap.add_argument("--schedule", default="auto", choices=["auto", "overlap", "serial"],
                    help="N > 1: boundary/interior overlap of the exchange, the exchange in line, or whichever a short trial finds faster")
    ap.add_argument("--driver", default="native", choices=["native", "torch"],
                    help="multi-GPU step: one library call with its own RCCL communicator, or the Python loop")
    ap.add_argument("--cpu-steps", type=int, default=16,
                    help="steps of the CPU baseline sample (0 = skip); 16 steps ~ 10 s of one core (c5: per sampled state, 4 states)")
    ap.add_argument("--no-pmc", action="store_true", help="do not measure roofline.traffic with rocprofv3 child runs")
    ap.add_argument("--no-extras", action="store_true", help="headline only: no extra points (formats, patterns, C3, C5)")
    ap.add_argument("--no-strong", action="store_true", help="N > 1: skip the strong-scaling point")
    ap.add_argument("--no-allgather", action="store_true", help="N > 1: skip the 4-step measurement of the all-gather exchange form")
    ap.add_argument("--no-safe", action="store_true", help="N > 1: skip the conservative first measurement and the watchdog")
    ap.add_argument("--point", default=None, choices=["c3", "c5"], help=argparse.SUPPRESS)      # one extras point only (PMC child runs)
    ap.add_argument("--watchdog", type=float, default=float(os.environ.get("QP_BENCH_WATCHDOG", "420")),
                    help="N > 1: seconds the native / overlapped path (set-up, self-check, trial, measurement, strong point) may take")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the prop_step! engine has no CPU fallback")
    # QP_BENCH_ONE_GPU=1 (testing only): all ranks share GPU 0 and the collective is staged through
    # the host with gloo, so that the multi-rank code path of this script can be exercised on a
    # 1-GPU box.  Numbers from that mode are meaningless and are labelled as such.
    one_gpu = os.environ.get("QP_BENCH_ONE_GPU") == "1"
    if one_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        if one_gpu:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    import qprop_amd.lib as L
    import qprop_amd.synth as synth
    import bench_points as bp

    if args.point:
        run_point(args, L, bp)
        return
    if args.config == "c5":
        run_c5(args, world, rank, local_rank, one_gpu, dist, L, synth, bp)
        return
    config = args.config if args.config != "auto" else ("c2" if world == 1 else "c4")
    # c4 on ONE GPU (or --scaling strong) is the FIXED problem of BASELINE.md section 2: N = 2^24 in total -- the denominator
    # of ">= 6x at 8 GPUs vs 1"; with more ranks and weak scaling it is 2^21 rows per GPU (N = 2^24 at 8)
    if config == "c4" and (world == 1 or args.scaling == "strong"):
        default_log2n = 24
    else:
        default_log2n = 20 if config == "c2" else 21
    log2n = args.log2n if args.log2n is not None else default_log2n
    if args.scaling == "strong":
        if (1 << log2n) % world:
            raise SystemExit("--scaling strong needs a rank count that divides 2^log2n")
        N = 1 << log2n
        rows = N // world
    else:
        rows = 1 << log2n
        N = rows * world
    if args.pattern.startswith("random") and world > 1:
        raise SystemExit("the random patterns are single-GPU points")
    r0, r1 = rank * rows, (rank + 1) * rows
    Delta, E_min, dt = 20.0, -10.0, args.dt    # manual range [-10,10], specrange_buffer=0
    fmt = {"auto": L.FMT_AUTO, "hrb": L.FMT_HRB, "rbcsr": L.FMT_RBCSR, "csr": L.FMT_CSR}[args.format]

    stream = torch.cuda.current_stream().cuda_stream
    ctx = L.Context(local_rank, stream=stream)
    rp, col, vals = bp.pattern_csr(args.pattern, N, r0, r1)
    if args.real:
        vals = vals.real.astype(np.complex128)
    psi0_local = synth.random_state(N, row_begin=r0, row_end=r1)
    nnz_local = int(rp[-1])
    coeffs = L.cheby_coeffs(Delta, dt)
    nterms = len(coeffs) - 1

    def barrier():
        torch.cuda.synchronize()     # nothing of ours in flight while the barrier's collective runs
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def dev_tensor(x):
        return torch.tensor(x, dtype=torch.float64, device="cpu" if one_gpu else "cuda")

    def comm_ranks(sh_=None, nat=False):
        """Ranks of the RCCL communicator a measured point exchanged on: ncclCommCount of the library's own communicator
        (qp_comm_info) for the native step, the nccl group's size for the torch.distributed-driven one; 0 in the one-GPU test
        mode (host-staged gloo / callback communicator: no RCCL).  A multi-GPU line whose points did not run on `world` RCCL
        ranks is refused below (exit status 4)."""
        if world == 1 or one_gpu:
            return 0
        if nat and sh_ is not None and getattr(sh_, "comm", None) is not None:
            return int(L.comm_info(sh_.comm)["rccl_ranks"])
        return int(dist.get_world_size()) if str(dist.get_backend()).lower() == "nccl" else 0

    parity = None
    cpu = None
    cpu_omp = None
    if world == 1:
        op = L.Operator(ctx, [L.Matrix(ctx, rows, N, rp, col, vals)], 0, fmt)
        wrk = L.ChebyWrk(ctx, N, Delta, E_min, dt)
        psi = L.State(ctx, data=psi0_local)

        def step():
            L.cheby(psi, op, dt, wrk)

        def cpu_baselines():
            nonlocal parity, cpu, cpu_omp
            if args.cpu_steps > 0:
                # CPU baseline: the oracle's C restatement of the reference's serial CSC path
                # (checker / reported baseline only; never on the product path)
                from oracle import ref_c
                psi.upload(psi0_local)
                for _ in range(args.cpu_steps):
                    step()
                gpu_k = psi.numpy()
                psi.upload(psi0_local)
                cpsi = psi0_local.copy()
                colptr, rowval, nzval = rp, col.astype(np.int64), np.conj(vals)   # Hermitian: CSC(H) = conj CSR(H)
                ref_c.load()
                flags = ref_c.build_flags()
                t0 = time.perf_counter()
                for _ in range(args.cpu_steps):
                    ref_c.cheby_csc(colptr, rowval, nzval, cpsi, coeffs, Delta, E_min, dt)
                tc = time.perf_counter() - t0
                parity = float(np.linalg.norm(gpu_k - cpsi))
                cpu = {"value": args.cpu_steps / tc, "unit": "prop_step/s", "cores": 1, "kind": "port",
                       "sample": f"{args.cpu_steps} prop_steps of the same N=2^{log2n} workload "
                                 f"(oracle/cheby_ref.c: serial CSC SpMV + BLAS-1, reference operation order; gcc {flags})",
                       "ms_per_step": 1e3 * tc / args.cpu_steps,
                       "l2_diff_vs_gpu_after_sample": parity}
                del colptr, rowval, nzval, cpsi
                # the same arithmetic with every host core: row-parallel CSR, passes fused (OpenMP), every
                # buffer first touched by the thread that works on it
                nthreads = ref_c.omp_threads()
                omp = ref_c.ChebyCsrOmp(rp, col.astype(np.int64), vals, psi0_local)
                omp.step(coeffs, Delta, E_min, dt)       # warm-up
                osteps = 2 * args.cpu_steps
                t0 = time.perf_counter()
                for _ in range(osteps):
                    omp.step(coeffs, Delta, E_min, dt)
                to = time.perf_counter() - t0
                omp.close()
                cpu_omp = {"value": osteps / to, "unit": "prop_step/s", "cores": int(nthreads), "kind": "port",
                           "sample": f"{osteps} prop_steps of the same workload (oracle/cheby_ref.c: OpenMP row-parallel CSR "
                                     f"mat-vec with the term's BLAS-1 passes fused into the row loop, buffers first touched "
                                     f"in parallel; gcc {flags})",
                           "ms_per_step": 1e3 * to / osteps}
        exchange_used = "none"
        schedule_note = driver_note = None
        ncols_local = N
        op_for_layout = op
    else:
        import qprop_amd.sharded as sharded
        # native: the whole step is one library call and the exchange runs on an RCCL
        # communicator the library owns (qp_sharded_cheby_step); otherwise the step loop is
        # driven from Python with torch.distributed collectives.  The native path is used only
        # if, on every rank, one step of it reproduces the torch-driven step bit for bit.
        want_native = args.driver == "native"     # (test mode: callback communicator, host-staged)

        def make_stepper(rp_, col_, vals_, N_, r0_, r1_, psi0_, exchange=None):
            """Both schedules of the partitioned step for one problem -> (stepper, native?, exchange, notes, layout op)."""
            def build(overlap):
                sh_ = sharded.ShardedCheby(ctx, rp_, col_, vals_, N_, r0_, r1_, Delta, E_min, dt, fmt=fmt, exchange=exchange or args.exchange,
                                           host_staged=one_gpu, native=want_native, overlap=overlap)
                nat = sh_.native is not None
                if nat:
                    sh_.set_state(psi0_)
                    sh_.step(native=True)
                    torch.cuda.synchronize()
                    got = sh_.local_state()
                    sh_.set_state(psi0_)
                    sh_.step(native=False)
                    torch.cuda.synchronize()
                    same = torch.tensor([1 if np.array_equal(got, sh_.local_state()) else 0], device="cpu" if one_gpu else "cuda")
                    dist.all_reduce(same, op=dist.ReduceOp.MIN)
                    nat = bool(same.item())
                    how = ("exchange handed back through a callback communicator" if one_gpu else
                           "RCCL communicator of the library, " + ("ncclSend/ncclRecv with the neighbours" if sh_.p2p else "ncclAllGather"))
                    note = f"native (library step, {how})" if nat else \
                        "torch.distributed (native step disagreed with it in the self-check)"
                else:
                    note = "torch.distributed (step loop in Python" + (
                        f"; native driver unavailable: {sh_.native_error})" if getattr(sh_, "native_error", None) else ")")
                sh_.set_state(psi0_)
                return sh_, nat, note

            def trial(sh_, nat, k=4):
                """Seconds per step of a schedule, the slowest rank's (every rank sees the same number)."""
                sh_.set_state(psi0_)
                for _ in range(2):
                    sh_.step(native=nat)
                torch.cuda.synchronize()
                dist.barrier()
                t0_ = time.perf_counter()
                for _ in range(k):
                    sh_.step(native=nat)
                torch.cuda.synchronize()
                t_ = dev_tensor([(time.perf_counter() - t0_) / k])
                dist.all_reduce(t_, op=dist.ReduceOp.MAX)
                sh_.set_state(psi0_)
                return float(t_[0])

            # Two schedules of the same arithmetic (bit-identical results): the boundary / interior overlap
            # hides the exchange behind the interior launch but pays for a second stream; the serial one has
            # the exchange in line.  Which is faster depends on the exchange latency of the machine, so both
            # are timed for a few steps before the measurement and the faster one is measured ("auto").
            sh_, nat, dnote = build(args.schedule != "serial")
            snote = args.schedule
            has_split = torch.tensor([1 if sh_.split is not None else 0], device="cpu" if one_gpu else "cuda")
            dist.all_reduce(has_split, op=dist.ReduceOp.MIN)      # the same decision on every rank
            if args.schedule == "auto" and bool(has_split.item()):
                sh2, nat2, note2 = build(False)
                t_overlap, t_serial = trial(sh_, nat), trial(sh2, nat2)
                snote = f"auto: overlap {1e3 * t_overlap:.3f} ms/step, serial {1e3 * t_serial:.3f} ms/step"
                if t_serial < t_overlap:
                    sh_, sh2, nat, dnote = sh2, sh_, nat2, note2
                    snote += " -> serial"
                else:
                    snote += " -> overlap"
                sh2.close()
                del sh2
            return sh_, nat, snote, dnote


    kernel_used = None
    build_ms = None
    if world == 1:
        fmt_used = op_for_layout.format
        model = bp.cheby_layout_bytes(op_for_layout, rows, ncols_local, nnz_local, coeffs, real_copy=args.real)
        kernel_used = bp.cheby_kernel_name(op_for_layout)
        build_ms = op_for_layout.build_info()

    pcie = None
    if world == 1 and os.environ.get("QP_BENCH_PCIE") == "1":
        # what a host-resident caller (a glue layer without a device state type) would see:
        # the state is downloaded after every step.  Reported separately, never as `value`.
        host = L.host_register(np.empty(N, dtype=np.complex128))
        for _ in range(3):
            step()
            psi.download(host)
        t0p = time.perf_counter()
        for _ in range(20):
            step()
            psi.download(host)
        pcie = 20 / (time.perf_counter() - t0p)
        L.host_unregister(host)

    # on-box HBM yardsticks: what this box streams read-only and in the walk's own 8 : 1 read : write
    # mix at 2 GiB working sets -- tools/probe/stream_yardstick (a child process: its own context, nothing of ours running)
    # ... and at a footprint the Infinity Cache holds (128 / 144 MiB: the headline's own working set is 195 MB and is served on-die)
    stream_read_gbs = stream_mix_gbs = cache_read_gbs = cache_mix_gbs = None
    if world == 1 and not args.no_extras:
        exe = os.path.join(ROOT, "tools", "probe", "stream_yardstick")
        try:
            torch.cuda.synchronize()
            r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
            y = json.loads(r.stdout.strip().splitlines()[-1])
            stream_read_gbs, stream_mix_gbs = y.get("stream_read_gbs"), y.get("stream_walk_mix_gbs")
            r = subprocess.run([exe, "20"], capture_output=True, text=True, timeout=60)
            y = json.loads(r.stdout.strip().splitlines()[-1])
            cache_read_gbs, cache_mix_gbs = y.get("stream_read_gbs"), y.get("stream_walk_mix_gbs")
        except Exception as e:  # noqa: BLE001  (a missing probe must not take the headline down)
            sys.stderr.write(f"[bench.py] stream yardstick not measured: {type(e).__name__}: {e}\n")

    def make_out(elapsed, ev_ms, st, fmt_used, model, exchange_used, schedule_note, driver_note, strong,
                 traffic=None, traffic_src=None, traffic_detail=None, extras=None, note=None, xmodel=None, rccl=0):
        """The JSON line for one measured run of args.steps steps (elapsed: wall seconds, slowest rank)."""
        layout = model["layout"]
        steps_per_s = args.steps / elapsed
        n_launch = args.steps * nterms
        avg_launch_s = (ev_ms * 1e-3) / n_launch
        achieved = model["per_term"] / avg_launch_s / 1e9
        csr_equiv = model["csr_equivalent_per_term"] / avg_launch_s / 1e9
        kern = kernel_used + "<ChebyOp>"
        blocks_per_step = N / float(1 << 20)

        workload = {"c2": "BASELINE configs[1]: Cheby prop_step!, N=2^20 CSR sparse Hermitian H, 16 nnz/row",
                    "c4": "BASELINE configs[3]: Cheby prop_step!, CSR sparse H row-partitioned, 2^21 rows per GPU (N=2^24 at 8 GPUs), "
                          "RCCL exchange of psi after each mat-vec"}[config]
        if config == "c4" and world == 1:
            workload = ("BASELINE configs[3] on ONE GPU: Cheby prop_step!, N=2^24 CSR sparse Hermitian H, 16 nnz/row -- the fixed problem whose "
                        "8-GPU row-partitioned run BASELINE.md section 2 compares with (no exchange here)")
        if args.log2n is not None or args.scaling == "strong":
            workload += f" [size overridden: 2^{log2n} rows {'in total (strong scaling)' if args.scaling == 'strong' else 'per GPU'}]"
        seg = laps.get("us") or []
        out = {
            "metric": "Cheby prop_step!/s at N=2^20 CSR fp64 (2^20-row blocks advanced per second)",
            "value": steps_per_s * blocks_per_step,
            "unit": "prop_step/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
            "dtype": "c128 state, f64 matrix values" if args.real else "c128 (complex fp64)", "data": "synthetic",
            "config": {"workload": workload + (", real fp64 values (f64 variant)" if args.real else ", complex fp64 values") + ", int32 indices",
                       "rows_per_gpu": rows, "N_total": N, "blocks_of_2^20_rows_per_step": blocks_per_step,
                       "nnz_per_row": nnz_local / rows, "pattern": args.pattern,
                       "n_coeffs": int(len(coeffs)), "matvecs_per_step": nterms, "dt": dt,
                       "device_format": bp.FMT_NAME[fmt_used],
                       "operator_build_ms": (build_ms or {}).get("build_ms"),
                       "parallelism": "single GPU" if world == 1 else (
                           f"row-partitioned x{world}, exchange={exchange_used}, schedule={schedule_note}, driver={driver_note}"
                           + (" [TEST MODE: ranks share one GPU, host-staged gloo]" if one_gpu else "")),
                       "global_steps_per_s": steps_per_s,
                       "spectral_range": [-10.0, 10.0],
                       "operator_build": build_ms,
                       "device_layout": layout},
            # `roofline`: the gate scalars (compact_line() copies them into the printed line); the explanatory material goes
            # to bench_extras.json only.  hbm_resident_* / n23_* / fixed_problem_* are filled from the extras below.
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "kernel": kern,
                         "avg_launch_us": avg_launch_s * 1e6,
                         "bytes_per_launch": model["per_term"],
                         "traffic_over_bytes": (traffic / model["per_term"]) if traffic else None,
                         "traffic_measured": bool(traffic_src and traffic_src.startswith("measured")) if traffic is not None else None,
                         # the bytes one launch moves fit the 256 MiB Infinity Cache: `frac` is then a cache-assisted rate, not an HBM one
                         "cache_resident": bool(model["per_term"] < 256.0 * 1024 * 1024), "model": "layout",
                         "unstable": bool(seg and max(seg) > 1.3 * min(seg)),
                         "hbm_resident_frac": None, "hbm_resident_us_per_term": None, "n23_frac": None, "n23_us_per_term": None,
                         "fixed_problem_n24_frac": None, "fixed_problem_n24_us_per_term": None, "fixed_problem_n24_blocks_per_s": None,
                         "stream_read_gbs": stream_read_gbs, "stream_walk_mix_gbs": stream_mix_gbs,
                         "cache_stream_read_gbs": cache_read_gbs, "cache_stream_mix_gbs": cache_mix_gbs,
                         "hbm_resident_frac_of_stream_mix": None, "fixed_problem_n24_frac_of_stream_mix": None,
                         "csr_equiv_frac": csr_equiv / HBM_PEAK_GBS,
                         "hbm_resident_frac_2^21_rows": None,
                         "operator_build_ms": (build_ms or {}).get("build_ms"),
                         "launch_us_min_segment": min(seg) if seg else None, "launch_us_max_segment": max(seg) if seg else None,
                         "layout_bytes_per_launch": model["per_term"],
                         "csr_equivalent_bytes_per_launch": model["csr_equivalent_per_term"],
                         "effective_csr_equiv_gbs": csr_equiv,
                         "traffic_rate_gbs": (traffic / avg_launch_s / 1e9) if traffic else None,
                         "layout_bytes_matrix": model["matrix_per_term"], "layout_bytes_vectors": model["vectors_per_term"],
                         "launches_timed": n_launch, "hip_event_ms": ev_ms,
                         "traffic_source": traffic_src,
                         "launch_us_segments": seg, "traffic_detail": traffic_detail,
                         "note": "avg_launch_us = HIP-event time of the timed region on the kernels' stream / fused-term launches (gaps and, "
                                 "multi-GPU, the exchange included).  achieved = bytes the SHIPPED device layout must move per launch / "
                                 "that time; csr_equiv_frac prices the time with SURVEY 8d's CSR bytes (404 B/row), which the "
                                 "Hermitian-packed stencil layout undercuts -- not a physical fraction.  traffic = HBM bytes per launch "
                                 "from PMC counters (FETCH_SIZE counts Infinity-Cache hits: at N = 2^20 the working set is served on-die; "
                                 "hbm_resident_* = the N = 2^22 point beyond it, fixed_problem_n24_* = config C4's N = 2^24 on this one "
                                 "GPU); stream_* = tools/probe/stream_yardstick on this box (read only / the walk's 8:1 mix, 2 GiB); cache_stream_* = the same "
                                 "at 128 / 144 MiB, inside the Infinity Cache like the headline's working set: compare traffic_rate_gbs"},
            "exchange": exchange_used, "rccl_ranks": rccl,
            "cpu_baseline": cpu,
            "cpu_baseline_all_cores": cpu_omp,
            "pcie_inclusive_steps_per_s": pcie,
            "strong_scaling_point": strong,
            "exchange_model": xmodel,
            "scaling_prediction": None,
            "extras": extras,
            "stats": {"n_matvec": st["n_matvec"], "kernel_launches": st["n_kernel_launches"]},
        }
        out["config"]["workload_short"] = {
            "c2": f"BASELINE configs[1]: Cheby prop_step!, N=2^{log2n} CSR sparse Hermitian H, 16 nnz/row, c128, int32 indices",
            "c4": (f"BASELINE configs[3] fixed problem on ONE GPU: Cheby prop_step!, N=2^{log2n} CSR H, 16 nnz/row" if world == 1 else
                   f"BASELINE configs[3]: Cheby prop_step!, CSR H row-partitioned, 2^{(rows - 1).bit_length()} rows/GPU, RCCL exchange per mat-vec")}[config] + (
            " [f64 values]" if args.real else "")
        out["config"]["parallelism_short"] = "single GPU" if world == 1 else (
            f"row-partitioned x{world}, exchange={exchange_used}, schedule={str(schedule_note).split(' -> ')[-1].split(':')[0]}, "
            f"driver={'native (own RCCL communicator)' if str(driver_note).startswith('native') else 'torch.distributed'}"
            + (" [TEST MODE: one GPU, gloo]" if one_gpu else ""))
        if note:
            out["config"]["parallelism"] += " | " + note
        return out

    laps = {}

    def timed_steps(step_fn):
        """W untimed + exactly K timed steps, barrier + synchronize on both sides, MAX over the ranks.  Inside the timed
        region an event is recorded on the kernels' stream after every quarter of the steps (three records: a few
        microseconds in all), so that the line can say whether the region was uniform: `launch_us_segments`, `unstable`."""
        for _ in range(args.warmup):
            step_fn()
        ctx.reset_stats()
        barrier()
        nseg = 4 if args.steps >= 8 else 1
        marks = [torch.cuda.Event(enable_timing=True) for _ in range(nseg + 1)]
        cuts = [round(k * args.steps / nseg) for k in range(nseg + 1)]
        # (the interpreter's cyclic garbage collector stays out of the timed region, as in `timeit`: a full collection of a
        # process that has imported torch takes 50-80 ms, see tools/bench_points.py: timed_regions)
        gc_was_on = gc.isenabled()
        gc.disable()
        ctx.timer_begin()
        marks[0].record()
        t0_ = time.perf_counter()
        for i in range(args.steps):
            step_fn()
            if (i + 1) in cuts[1:]:
                marks[cuts.index(i + 1)].record()
        ev_ = ctx.timer_end()          # HIP events on the kernels' own stream
        torch.cuda.synchronize()
        if gc_was_on:
            gc.enable()
        if dist is not None:
            dist.barrier()
        el_ = time.perf_counter() - t0_
        st_ = ctx.stats()
        seg = [1e3 * marks[k].elapsed_time(marks[k + 1]) / ((cuts[k + 1] - cuts[k]) * nterms) for k in range(nseg)]
        laps["us"] = seg
        if dist is not None:
            t_ = dev_tensor([el_, ev_])
            dist.all_reduce(t_, op=dist.ReduceOp.MAX)
            el_, ev_ = float(t_[0]), float(t_[1])
        return el_, ev_, st_

    # ---- N > 1: a conservative measurement first, the native / overlapped path under a watchdog -------------------
    # The library's own RCCL communicator, the neighbour send / recv exchange and the two-stream overlap cannot be run
    # with more than one rank on the one-GPU boxes this was developed on.  So the first thing measured on a multi-GPU
    # node is the plainest schedule -- step loop in Python, one torch.distributed all-gather per term on the main
    # stream, no second stream: all of it exercised with RCCL at world 1 and with gloo at world 2-3 -- and its complete
    # JSON line is kept.  The faster path then runs under a watchdog: if it does not come back (a hang inside a
    # collective cannot be recovered in-process), rank 0 prints the kept line and every rank leaves.
    fallback = None
    watchdog = None
    ag_form = {}
    ranks_seen = {}
    if world > 1:
        if not args.no_safe:
            shA = sharded.ShardedCheby(ctx, rp, col, vals, N, r0, r1, Delta, E_min, dt, fmt=fmt, exchange=args.exchange,
                                       host_staged=one_gpu, native=False, overlap=False)
            shA.set_state(psi0_local)
            elA, evA, stA = timed_steps(lambda: shA.step(native=False))
            modelA = bp.cheby_layout_bytes(shA.op, rows, rows + world * shA.M, nnz_local, coeffs, real_copy=args.real, whole_step=False)
            kernel_used = KERNEL_OF_FORMAT[shA.op.format]
            build_ms = shA.op.build_info()
            fallback = make_out(elA, evA, stA, shA.op.format, modelA, shA.exchange, "serial", "torch.distributed (step loop in Python)",
                                None, note="conservative schedule (reported because the native / overlapped path did not finish)",
                                xmodel=exchange_model(shA, world, rows, 1e3 * evA / (args.steps * nterms)), rccl=comm_ranks())
            fallback_line = json.dumps(fallback)
            fallback["config"]["parallelism"] = fallback["config"]["parallelism"].split(" | ")[0]
            shA.close()
            del shA
            import threading

            def give_up(reason="hung"):
                # the kept line is printed, marked as degraded, and the process leaves with a NON-ZERO status: a hang or a
                # failure of the native / RCCL / overlapped path must not look like a clean run to whoever gates on rc
                if rank == 0:
                    kept = json.loads(fallback_line)
                    kept["degraded"] = True
                    kept["native_path"] = reason
                    emit(kept, f"_{config}_gpus{world}")
                sys.stderr.write(f"[bench.py rank {rank}] native / overlapped path {reason} (limit {args.watchdog} s): "
                                 f"reporting the conservative measurement, exit status 3\n")
                if reason == "hung":      # where: the Python stacks
                    import faulthandler
                    faulthandler.dump_traceback(file=sys.stderr, all_threads=True)
                sys.stderr.flush()
                os._exit(3)
            watchdog = threading.Timer(args.watchdog, give_up)
            watchdog.daemon = True
            watchdog.start()
        def stage(msg):      # progress on stderr (rank 0): a watchdog exit then says how far the path got
            if rank == 0:
                sys.stderr.write(f"[bench.py] native path: {msg} (+{time.perf_counter() - t_native:.1f} s)\n")
                sys.stderr.flush()
        t_native = time.perf_counter()

        def native_path():
            if os.environ.get("QP_BENCH_TEST_HANG") == "1":      # testing only: what the watchdog is for
                time.sleep(1e6)
            if os.environ.get("QP_BENCH_TEST_HANG") == "raise" and rank == world - 1:
                raise RuntimeError("simulated failure of the native path on one rank")
            stage("set-up, self-check and schedule trial")
            sh_, nat_, snote_, dnote_ = make_stepper(rp, col, vals, N, r0, r1, psi0_local)
            stage("timed steps")
            model_ = bp.cheby_layout_bytes(sh_.op, rows, rows + world * sh_.M, nnz_local, coeffs, real_copy=args.real, whole_step=False)
            el_, ev_, st_ = timed_steps(lambda: sh_.step(native=nat_))
            sh_.check()      # outside the timed region: the overlapped schedule never timed out
            ranks_seen["headline"] = comm_ranks(sh_, nat_)
            strong_ = None
            # the strong-scaling point of BASELINE's metric (N = 2^20 in total)
            if args.scaling == "weak" and not args.no_strong and (1 << 20) % world == 0:
                stage("strong-scaling point")
                Ns = 1 << 20
                rs = Ns // world
                s0, s1 = rank * rs, (rank + 1) * rs
                rps, cols_, valss = bp.pattern_csr(args.pattern, Ns, s0, s1)
                psis = synth.random_state(Ns, row_begin=s0, row_end=s1)
                sh_s, nat_s, snote_s, dnote_s = make_stepper(rps, cols_, valss, Ns, s0, s1, psis)
                ksteps = max(10, args.steps // 2)
                for _ in range(5):
                    sh_s.step(native=nat_s)
                barrier()
                t0s = time.perf_counter()
                for _ in range(ksteps):
                    sh_s.step(native=nat_s)
                torch.cuda.synchronize()
                dist.barrier()
                ts = dev_tensor([time.perf_counter() - t0s])
                dist.all_reduce(ts, op=dist.ReduceOp.MAX)
                sh_s.check()
                strong_ = {"workload": "Cheby prop_step!, N=2^20 CSR fp64 in TOTAL, row-partitioned over the ranks (strong scaling)",
                           "N_total": Ns, "rows_per_gpu": rs, "steps": ksteps, "prop_steps_per_s": ksteps / float(ts[0]),
                           "ms_per_step": 1e3 * float(ts[0]) / ksteps, "exchange": sh_s.exchange, "schedule": snote_s, "driver": dnote_s,
                           "rccl_ranks": comm_ranks(sh_s, nat_s)}
                sh_s.close()
            # the collective north_star NAMES -- an all-gather of the term vector after every mat-vec -- gets its own measured
            # point next to the headline (for a banded H `auto` picks the halo exchange): 4 steps with the whole slice
            # exchanged, the same operator, the same driver
            if not args.no_allgather:
                stage("all-gather form")
                per_term_main = 1e3 * ev_ / (args.steps * nterms)
                ag_form["exchange"] = "allgather"
                if sh_.exchange == "allgather":
                    ag_form["us_per_term"], ag_form["blocks_per_s"] = per_term_main, (N / float(1 << 20)) * args.steps / el_
                    ag_form["rccl_ranks"] = ranks_seen["headline"]
                else:
                    try:
                        # the plainest schedule (step loop in Python, one torch.distributed all-gather per term on the main stream, no
                        # second stream): the form exercised with RCCL at world 1 and with gloo at world 2 - 3 -- a number for the
                        # collective, not another first run of the native machinery
                        # ... on the SCATTERED pattern (synth: eight seeded distances in [1, N / 2)): every row reads remote data, the send
                        # list is every rank's whole slice -- the case the collective exists for (a banded H exchanges halos)
                        rows_t = min(rows, 1 << 13) if one_gpu else rows      # (test mode stages every byte through the host: 128 KiB per rank)
                        Ng, g0, g1 = rows_t * world, rank * rows_t, (rank + 1) * rows_t
                        rpg, colg, valsg = bp.pattern_csr("scattered", Ng, g0, g1)
                        psig = synth.random_state(Ng, row_begin=g0, row_end=g1)
                        if one_gpu:
                            ag_form["test_mode_rows_per_rank"] = rows_t
                        sh_g = sharded.ShardedCheby(ctx, rpg, colg, valsg, Ng, g0, g1, Delta, E_min, dt, fmt=fmt, exchange="allgather",
                                                    host_staged=one_gpu, native=False, overlap=False)
                        sh_g.set_state(psig)
                        nat_g = False
                        kg = 4
                        for _ in range(2):
                            sh_g.step(native=nat_g)
                        barrier()
                        t0g = time.perf_counter()
                        for _ in range(kg):
                            sh_g.step(native=nat_g)
                        torch.cuda.synchronize()
                        dist.barrier()
                        tg = dev_tensor([time.perf_counter() - t0g])
                        dist.all_reduce(tg, op=dist.ReduceOp.MAX)
                        sh_g.check()
                        ag_form["us_per_term"] = 1e6 * float(tg[0]) / (kg * nterms)
                        ag_form["blocks_per_s"] = (Ng / float(1 << 20)) * kg / float(tg[0])
                        ag_form["driver"] = "torch.distributed, serial"
                        ag_form["rccl_ranks"] = comm_ranks()
                        ag_form["pattern"] = "scattered"
                        sh_g.close()
                    except Exception as e:  # noqa: BLE001 -- the extra point must not take the headline down
                        import traceback
                        sys.stderr.write(f"[bench.py rank {rank}] all-gather form failed:\n{traceback.format_exc()}\n")
                        ag_form["error"] = f"{type(e).__name__}: {e}"
            stage("done")
            return sh_, el_, ev_, st_, model_, snote_, dnote_, strong_

        try:
            sh, elapsed, ev_ms, st, model, schedule_note, driver_note, strong = native_path()
        except Exception:      # noqa: BLE001 -- with a complete conservative measurement in hand, report that one
            if fallback is None:
                raise
            import traceback
            traceback.print_exc()
            give_up("failed")
        exchange_used = sh.exchange
        fmt_used = sh.op.format
        kernel_used = KERNEL_OF_FORMAT[fmt_used]
        if sh.split is not None and sh.split.walk_info()["valid"]:
            kernel_used = "hrb_walk_kernel (interior set) + " + kernel_used + " (boundary set)"
        elif sh.split is None:
            kernel_used = bp.cheby_kernel_name(sh.op, whole_step=False)
        build_ms = sh.op.build_info()
    else:
        strong = None
        elapsed, ev_ms, st = timed_steps(step)

    # ---- single GPU: CPU baselines, HBM traffic measured under rocprofv3, the other points ------------
    traffic = traffic_src = traffic_detail = None
    extras = None
    if world == 1:
        if args.cpu_steps > 0:
            cpu_baselines()  # after the timed region: busy OpenMP threads must not sit next to the measurement
        headline = (config == "c2" and log2n == 20 and args.pattern == "banded" and args.format == "auto" and not args.real
                    and args.dt == 1.0)
        if not args.no_pmc:
            inner = ["--steps", "3", "--warmup", "1", "--cpu-steps", "0", "--no-pmc", "--no-extras", "--log2n", str(log2n),
                     "--pattern", args.pattern, "--format", args.format, "--dt", str(args.dt)] + (["--real"] if args.real else [])
            traffic, traffic_detail = pmc_traffic(inner, kernel_used, timeout_s=240)
            if traffic is not None and kernel_used == "hrb_walk2_kernel":
                traffic /= 2.0      # a launch of the two-term walk forms TWO terms: per term, like bytes_per_launch and avg_launch_us
                traffic_detail["note"] = "hrb_walk2_kernel: one launch = two terms; `traffic` is per TERM (half the per-launch mean)"
            if traffic is not None:
                traffic_src = ("measured in this run: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE --kernel-trace on child runs of "
                               "this command (3 steps each); mean per launch of the kernel, FETCH_SIZE x 2 (gfx950), KiB -> bytes")
        if traffic is None and headline and os.path.exists(os.path.join(ROOT, STATIC_PMC)):
            with open(os.path.join(ROOT, STATIC_PMC)) as f:
                traffic = json.load(f)["hbm_traffic_bytes_per_launch"]
            traffic_src = (f"STATIC: from the committed profile {STATIC_PMC} of the same command on another box "
                           f"(not measured in this run: {traffic_detail if isinstance(traffic_detail, str) else 'PMC passes disabled'})")
            traffic_detail = None
        if not args.no_extras and headline:
            psi.close()
            wrk.close()
            op.close()
            extras = {}
            for name, kw in (("c2_rbcsr", dict(pattern="banded", log2n=20, fmt="rbcsr")),
                             ("c2_scattered", dict(pattern="scattered", log2n=20)),
                             ("c2_random", dict(pattern="random", log2n=20)),
                             ("c2_random_window", dict(pattern="random-window", log2n=20)),
                             ("banded_n21", dict(pattern="banded", log2n=21, steps=8)),
                             ("banded_n22", dict(pattern="banded", log2n=22, steps=5)),
                             ("banded_n23", dict(pattern="banded", log2n=23, steps=4, warmup=3)),
                             ("banded_n24", dict(pattern="banded", log2n=24, steps=4, warmup=3)),
                             # the one-term walk at the row counts a rank of 8 / 4 / 2 GPUs owns (the row-partitioned step never pairs terms)
                             ("banded_n21_one_term", dict(pattern="banded", log2n=21, steps=8, knobs={"walk_pair": 0})),
                             ("banded_n22_one_term", dict(pattern="banded", log2n=22, steps=5, knobs={"walk_pair": 0})),
                             ("banded_n23_one_term", dict(pattern="banded", log2n=23, steps=4, warmup=3, knobs={"walk_pair": 0})),
                             ("c2_alpha2", dict(pattern="banded", log2n=20, dt=0.2, steps=20)),
                             ("c2_alpha50", dict(pattern="banded", log2n=20, dt=5.0, steps=5)),
                             ("c2_real_f64", dict(pattern="banded", log2n=20, real=True)),
                             ("grid2d_5pt", dict(grid=(2048, 2048), steps=5)),
                             ("grid3d_7pt", dict(grid=(256, 128, 128), steps=4)),
                             ("grid3d_13pt", dict(grid=(256, 128, 128), grid_order=4, steps=4)),
                             ("lattice_9pt_n22", dict(offsets=(1, 2047, 2048, 2049), log2n=22, steps=5)),
                             ("tfim20", dict(spins=20, steps=5))):
                try:
                    extras[name] = bp.measure_cheby(ctx, **kw)
                except Exception as e:  # noqa: BLE001  (an extra point must not take the headline down)
                    extras[name] = {"error": f"{type(e).__name__}: {e}"}
            for name, fn in (("tfim20_pauli", lambda c: bp.measure_pauli(c, spins=20, steps=5)),
                             ("c3_newton", bp.measure_newton_c3),
                             ("c3_newton_n2048", lambda c: bp.measure_newton_c3(c, n=2048, steps=3, warmup=4)),
                             ("c5_batched", bp.measure_batched_c5),
                             ("c5_batched_8", lambda c: bp.measure_batched_c5(c, batch=8, steps=20)),
                             ("dense_n4096", lambda c: bp.measure_dense(c, N=4096)),
                             ("dense_n4096_b64_mfma", lambda c: bp.measure_dense(c, N=4096, batch=64)),
                             ("n4_liouville_n512", bp.measure_liouville)):
                try:
                    extras[name] = fn(ctx)
                except Exception as e:  # noqa: BLE001
                    extras[name] = {"error": f"{type(e).__name__}: {e}"}

            # the reference's CPU path beside the other two GPU configs: bounded samples, one core
            if args.cpu_steps > 0:
                try:
                    if "error" not in extras.get("c3_newton", {"error": 1}):
                        extras["c3_newton"]["cpu_baseline"] = cpu_baseline_c3(ctx, L, synth, steps=2)
                except Exception as e:  # noqa: BLE001
                    extras["c3_newton"]["cpu_baseline"] = {"error": f"{type(e).__name__}: {e}"}
                try:
                    if "error" not in extras.get("c5_batched", {"error": 1}):
                        extras["c5_batched"]["cpu_baseline"] = bp.with_panel(ctx, 18, 64, lambda *a: cpu_baseline_c5(L, synth, *a, min(args.cpu_steps, 8)))
                except Exception as e:  # noqa: BLE001
                    extras["c5_batched"]["cpu_baseline"] = {"error": f"{type(e).__name__}: {e}"}
            # HBM traffic of the C3 and C5 points, by the same child-process PMC passes as the headline (one counter per pass)
            if not args.no_pmc:
                for name, argv, subs, how, per in (
                        ("c3_newton", ["--point", "c3", "--steps", "3"], NEWTON_KERNELS, "sum", 6.0),       # the child runs 2 x 3 identical steps
                        ("c5_batched", ["--point", "c5", "--steps", "2"], extras.get("c5_batched", {}).get("kernel_symbol", "spmm_rows_smem_kernel"), "mean", 1.0)):
                    pt = extras.get(name)
                    if not pt or "error" in pt:
                        continue
                    tr, det = pmc_traffic(argv, subs, timeout_s=240, how=how)
                    if tr is None:
                        pt["traffic"], pt["traffic_source"] = None, f"not measured: {det}"
                        continue
                    tr /= per
                    impl = pt.get("implementation_bytes_per_sweep", 0) * pt.get("arnoldi_sweeps_per_step", 0) if name == "c3_newton" \
                        else pt.get("layout_bytes_per_term")
                    pt["traffic"] = tr
                    pt["traffic_unit"] = "HBM bytes per newton! step (all kernels of the step)" if name == "c3_newton" else "HBM bytes per fused panel term"
                    pt["traffic_over_implementation_bytes"] = tr / impl if impl else None
                    pt["traffic_source"] = ("measured in this run: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE child runs of `bench.py " + " ".join(argv)
                                            + "`, FETCH_SIZE x 2 (gfx950); " + ("sum over the step's kernels / steps" if how == "sum" else "mean per launch"))
                    pt["traffic_detail"] = det

    out = make_out(elapsed, ev_ms, st, fmt_used, model, exchange_used, schedule_note, driver_note, strong,
                   traffic, traffic_src, traffic_detail, extras,
                   xmodel=(exchange_model(sh, world, rows, 1e3 * ev_ms / (args.steps * nterms)) if world > 1 else None),
                   rccl=ranks_seen.get("headline", 0),
                   note=(None if fallback is None else
                         f"conservative schedule measured first: {fallback['value']:.1f} {fallback['unit']} "
                         f"({fallback['ms_per_step']:.3f} ms/step, torch.distributed all-gather per term, no overlap)"))
    # the fraction beyond the Infinity Cache and the fixed problem of BASELINE.md section 2 as scalar keys of the headline
    # object (at N = 2^20 the 134 MB of values and the vectors are served on-die; config C4 is N = 2^24)
    if extras:
        rf = out["roofline"]
        p22 = extras.get("banded_n22") or {}
        p21 = extras.get("banded_n21") or {}
        p24 = extras.get("banded_n24") or {}
        p23 = extras.get("banded_n23") or {}
        rf["hbm_resident_frac"] = p22.get("frac")
        rf["hbm_resident_us_per_term"] = p22.get("us_per_term")
        rf["hbm_resident_frac_2^21_rows"] = p21.get("frac")
        rf["n23_frac"], rf["n23_us_per_term"] = p23.get("frac"), p23.get("us_per_term")
        rf["fixed_problem_n24_us_per_term"] = p24.get("us_per_term")
        rf["fixed_problem_n24_frac"] = p24.get("frac")
        rf["fixed_problem_n24_blocks_per_s"] = (16.0 * p24["steps_per_s"]) if p24.get("steps_per_s") else None
        # the same two points against what THIS box streams in the walk's own read : write mix (tools/probe/stream_yardstick):
        # a fraction of a measured ceiling of the same mix, so it cannot exceed 1 by construction of the yardstick
        if stream_mix_gbs:
            rf["hbm_resident_frac_of_stream_mix"] = (p22["layout_gbs"] / stream_mix_gbs) if p22.get("layout_gbs") else None
            rf["fixed_problem_n24_frac_of_stream_mix"] = (p24["layout_gbs"] / stream_mix_gbs) if p24.get("layout_gbs") else None
        rf["hbm_resident_point"] = {k: p22.get(k) for k in ("N", "us_per_term", "us_per_term_min", "us_per_term_max", "unstable",
                                                            "layout_bytes_per_term", "layout_gbs", "kernel")}
        rf["fixed_problem_point"] = {k: p24.get(k) for k in ("N", "us_per_term", "us_per_term_min", "us_per_term_max", "unstable", "ms_per_step",
                                                             "steps_per_s", "layout_bytes_per_term", "layout_gbs", "frac", "kernel",
                                                             "operator_build_ms")}
        rf["unstable_extras"] = sorted(k for k, v in extras.items() if isinstance(v, dict) and v.get("unstable"))
        # points one of whose timed regions was measured again because a single enqueue call held the host for more than half
        # of the region (tools/bench_points.py: timed_regions; the discarded regions are kept in the point's record)
        rf["extras_with_a_region_remeasured_after_a_host_stall"] = sorted(
            k for k, v in extras.items() if isinstance(v, dict) and v.get("regions_remeasured_after_host_stall"))
        t_gc = time.perf_counter()
        gc.collect()
        rf["host_gc_full_collection_ms"] = 1e3 * (time.perf_counter() - t_gc)    # what a collection inside a timed region would cost
        rf["note"] = rf.pop("note")        # the long text stays last
        sizes = {}
        for lg, pt in ((20, {"us_per_term": rf["avg_launch_us"]}), (21, p21), (22, p22),
                       (23, extras.get("banded_n23") or {}), (24, p24)):
            if pt.get("us_per_term"):
                sizes[lg] = pt["us_per_term"]
        one_term = {lg: extras[f"banded_n{lg}_one_term"]["us_per_term"] for lg in (21, 22, 23)
                    if (extras.get(f"banded_n{lg}_one_term") or {}).get("us_per_term")}
        out["scaling_prediction"] = scaling_prediction(sizes, out["value"], nterms, "measured in this run on one GPU", one_term=one_term)
    out["degraded"] = False
    out["native_path"] = "ok" if world > 1 else None
    if watchdog is not None:
        # every rank is through the native path before any rank disarms: a rank whose timer fires late finds the others
        # still armed (they leave with it) instead of blocked for good in the teardown collectives
        barrier()
        watchdog.cancel()
    if fallback is not None and fallback["value"] > out["value"]:      # report the faster of the two complete measurements
        fallback["config"]["parallelism"] += (f" | the native / overlapped path (schedule={schedule_note}, driver={driver_note}) measured "
                                              f"{out['value']:.1f} {out['unit']} ({out['ms_per_step']:.3f} ms/step): slower, not reported as `value`")
        fallback["config"]["parallelism_short"] = _short(fallback["config"]["parallelism_short"].replace(" [TEST MODE: one GPU, gloo]", "") +
                                                         " | slower, not `value`: " + out["config"]["parallelism_short"].split(", ", 2)[-1])
        fallback["strong_scaling_point"] = out["strong_scaling_point"]
        fallback["degraded"] = False
        fallback["native_path"] = "ok (slower than the conservative schedule)"
        out = fallback
    if world > 1:      # the 1 / 2 / 4 / 8 table this run is to be read against (from the committed single-GPU measurements)
        out["scaling_prediction"] = scaling_prediction_static(nterms)
        out["allgather_form"] = ag_form or None
        if fallback is not None and out is not fallback:
            out["conservative_first"] = {"value": fallback["value"], "ms_per_step": fallback["ms_per_step"],
                                         "exchange": fallback["exchange"], "rccl_ranks": fallback["rccl_ranks"]}
    sp_ = out.get("strong_scaling_point")
    if sp_:
        out["strong_point"] = {k: sp_.get(k) for k in ("N_total", "prop_steps_per_s", "ms_per_step", "exchange", "rccl_ranks")}
    out["prediction"] = prediction_scalars(out.get("scaling_prediction"))
    # every measured point of a multi-GPU line names its exchange and the RCCL ranks it ran on; a point that did not run on an
    # N-rank RCCL communicator makes the run a failed one (the line is still printed, marked): nobody can mistake a run whose
    # ranks never talked over RCCL for a scaling measurement.  (One-GPU test mode: rccl_ranks = 0 by construction, labelled.)
    bad_rccl = []
    if world > 1 and not one_gpu:
        for name in (None, "strong_point", "allgather_form", "conservative_first"):
            pt = out if name is None else out.get(name)
            if pt and "error" not in pt and pt.get("rccl_ranks") != world:
                bad_rccl.append(f"{name or 'headline'}: rccl_ranks={pt.get('rccl_ranks')}")
        if bad_rccl:
            out["degraded"] = True
            out["native_path"] = _short("rccl_ranks != n_gpus: " + "; ".join(bad_rccl), 96)
    if rank == 0:
        default_run = world == 1 and config == "c2" and args.log2n is None and args.pattern == "banded" and args.format == "auto" and \
            not args.real and args.dt == 1.0 and args.scaling == "weak"
        emit(out, "" if default_run else f"_{config}" + (f"_gpus{world}" if world > 1 else "") + ("" if args.log2n is None else f"_n{log2n}"))
    if dist is not None:
        sh.close()      # the library's communicator goes before the process group it was bootstrapped over
        dist.destroy_process_group()
    if bad_rccl:
        sys.stderr.write(f"[bench.py rank {rank}] {'; '.join(bad_rccl)} with --gpus {world}: exit status 4\n")
        sys.exit(4)


if __name__ == "__main__":
    main()
