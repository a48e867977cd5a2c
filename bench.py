#!/usr/bin/env python3
"""Headline benchmark: Chebyshev ``prop_step!``/s on the BASELINE.json config C2
(N = 2^20 rows per GPU, CSR sparse Hermitian H with 16 nnz/row, complex fp64,
manual spectral range [-10, 10], dt = 1 => 32 coefficients = 31 fused mat-vec terms).

    python bench.py                      # = --gpus 1 --steps 100 --warmup 10, about 20 s with the CPU baselines
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One "step" = one ``prop_step!`` = one pass of the hot path (31 fused SpMV terms) over the
state.  For N > 1 the CSR rows are partitioned across the ranks (2^20 rows per GPU, weak
scaling) and the needed slices of the term vector are exchanged over RCCL after every
mat-vec.  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import numpy as np  # noqa: E402
import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0     # MI355X HBM3E spec peak (MI355X_MICROARCH.md: 8 TB/s; ~6.3 TB/s measured copy)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--log2n", type=int, default=20, help="rows per GPU = 2^log2n")
    ap.add_argument("--pattern", default="banded", choices=["banded", "scattered"])
    ap.add_argument("--format", default="auto", choices=["auto", "hrb", "rbcsr", "csr"])
    ap.add_argument("--exchange", default="auto", choices=["auto", "halo", "allgather"])
    ap.add_argument("--dt", type=float, default=1.0, help="time step; alpha = 10 dt, i.e. 32 coefficients at dt = 1 "
                    "(SURVEY 8d also asks for alpha = 2 and 50: --dt 0.2 / --dt 5)")
    ap.add_argument("--real", action="store_true", help="real-symmetric H (the f64 variant of SURVEY 8d): values "
                    "are streamed as fp64; algorithmic bytes (12 z + 84) N")
    ap.add_argument("--schedule", default="auto", choices=["auto", "overlap", "serial"],
                    help="N > 1: boundary/interior overlap of the exchange, the exchange in line, or whichever a short trial finds faster")
    ap.add_argument("--driver", default="native", choices=["native", "torch"],
                    help="multi-GPU step: one library call with its own RCCL communicator, or the Python loop")
    ap.add_argument("--cpu-steps", type=int, default=16,
                    help="steps of the CPU baseline sample (0 = skip); 16 steps ~ 10 s of one core")
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

    rows = 1 << args.log2n
    N = rows * world
    r0, r1 = rank * rows, (rank + 1) * rows
    offsets = synth.BANDED_OFFSETS if args.pattern == "banded" else synth.scattered_offsets(N)
    Delta, E_min, dt = 20.0, -10.0, args.dt    # manual range [-10,10], specrange_buffer=0
    fmt = {"auto": L.FMT_AUTO, "hrb": L.FMT_HRB, "rbcsr": L.FMT_RBCSR, "csr": L.FMT_CSR}[args.format]

    stream = torch.cuda.current_stream().cuda_stream
    ctx = L.Context(local_rank, stream=stream)
    rp, col, vals = synth.hermitian_offsets_csr(N, offsets=offsets, row_begin=r0, row_end=r1)
    if args.real:
        vals = vals.real.astype(np.complex128)
    psi0_local = synth.random_state(N, row_begin=r0, row_end=r1)
    nnz_local = int(rp[-1])
    coeffs = L.cheby_coeffs(Delta, dt)
    nterms = len(coeffs) - 1

    parity = None
    cpu = None
    cpu_omp = None
    if world == 1:
        op = L.Operator(ctx, [L.Matrix(ctx, rows, N, rp, col, vals)], 0, fmt)
        wrk = L.ChebyWrk(ctx, N, Delta, E_min, dt)
        psi = L.State(ctx, data=psi0_local)
        fmt_used = op.format
        layout = op.layout_info()

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
                t0 = time.perf_counter()
                for _ in range(args.cpu_steps):
                    ref_c.cheby_csc(colptr, rowval, nzval, cpsi, coeffs, Delta, E_min, dt)
                tc = time.perf_counter() - t0
                parity = float(np.linalg.norm(gpu_k - cpsi))
                cpu = {"value": args.cpu_steps / tc, "unit": "prop_step/s", "cores": 1, "kind": "port",
                       "sample": f"{args.cpu_steps} prop_steps of the same N=2^{args.log2n} workload "
                                 f"(oracle/cheby_ref.c: serial CSC SpMV + BLAS-1, reference operation order)",
                       "ms_per_step": 1e3 * tc / args.cpu_steps,
                       "l2_diff_vs_gpu_after_sample": parity}
                del colptr, rowval, nzval, cpsi
                # the same arithmetic with every host core: row-parallel CSR, passes fused (OpenMP)
                nthreads = ref_c.omp_threads()
                opsi = psi0_local.copy()
                col64 = col.astype(np.int64)
                ref_c.cheby_csr_omp(rp, col64, vals, opsi, coeffs, Delta, E_min, dt)       # warm-up, first touch
                opsi = psi0_local.copy()
                osteps = 2 * args.cpu_steps
                t0 = time.perf_counter()
                for _ in range(osteps):
                    ref_c.cheby_csr_omp(rp, col64, vals, opsi, coeffs, Delta, E_min, dt)
                to = time.perf_counter() - t0
                cpu_omp = {"value": osteps / to, "unit": "prop_step/s", "cores": int(nthreads), "kind": "port",
                           "sample": f"{osteps} prop_steps of the same workload (oracle/cheby_ref.c: OpenMP row-parallel CSR "
                                     f"mat-vec with the term's BLAS-1 passes fused into the row loop)",
                           "ms_per_step": 1e3 * to / osteps}
                del col64, opsi
        exchange_used = "none"
    else:
        import qprop_amd.sharded as sharded
        # native: the whole step is one library call and the exchange runs on an RCCL
        # communicator the library owns (qp_sharded_cheby_step); otherwise the step loop is
        # driven from Python with torch.distributed collectives.  The native path is used only
        # if, on every rank, one step of it reproduces the torch-driven step bit for bit.
        want_native = args.driver == "native"     # (test mode: callback communicator, host-staged)

        def build(overlap):
            """One schedule of the partitioned step, ready to run: (stepper, native?, note)."""
            sh_ = sharded.ShardedCheby(ctx, rp, col, vals, N, r0, r1, Delta, E_min, dt, fmt=fmt, exchange=args.exchange,
                                       host_staged=one_gpu, native=want_native, overlap=overlap)
            nat = sh_.native is not None
            if nat:
                sh_.set_state(psi0_local)
                sh_.step(native=True)
                torch.cuda.synchronize()
                got = sh_.local_state()
                sh_.set_state(psi0_local)
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
            sh_.set_state(psi0_local)
            return sh_, nat, note

        def trial(sh_, nat, k=4):
            """Seconds per step of a schedule, the slowest rank's (every rank sees the same number)."""
            sh_.set_state(psi0_local)
            for _ in range(2):
                sh_.step(native=nat)
            torch.cuda.synchronize()
            dist.barrier()
            t0_ = time.perf_counter()
            for _ in range(k):
                sh_.step(native=nat)
            torch.cuda.synchronize()
            t_ = torch.tensor([(time.perf_counter() - t0_) / k], dtype=torch.float64, device="cpu" if one_gpu else "cuda")
            dist.all_reduce(t_, op=dist.ReduceOp.MAX)
            sh_.set_state(psi0_local)
            return float(t_[0])

        # Two schedules of the same arithmetic (bit-identical results): the boundary / interior overlap
        # hides the exchange behind the interior launch but pays for a second stream; the serial one has
        # the exchange in line.  Which is faster depends on the exchange latency of the machine, so both
        # are timed for a few steps before the measurement and the faster one is measured ("auto").
        sh, use_native, driver_note = build(args.schedule != "serial")
        schedule_note = args.schedule
        has_split = torch.tensor([1 if sh.split is not None else 0], device="cpu" if one_gpu else "cuda")
        dist.all_reduce(has_split, op=dist.ReduceOp.MIN)      # the same decision on every rank
        if args.schedule == "auto" and bool(has_split.item()):
            sh2, nat2, note2 = build(False)
            t_overlap, t_serial = trial(sh, use_native), trial(sh2, nat2)
            schedule_note = f"auto: overlap {1e3 * t_overlap:.3f} ms/step, serial {1e3 * t_serial:.3f} ms/step"
            if t_serial < t_overlap:
                sh, sh2, use_native, driver_note = sh2, sh, nat2, note2
                schedule_note += " -> serial"
            else:
                schedule_note += " -> overlap"
            sh2.close()
            del sh2
        fmt_used = sh.op.format
        layout = sh.op.layout_info()
        exchange_used = sh.exchange

        def step():
            sh.step(native=use_native)

    pcie = None
    if world == 1 and os.environ.get("QP_BENCH_PCIE") == "1":
        # what a host-resident caller (the Julia glue without a device state type) would see:
        # the state is downloaded after every step.  Reported separately, never as `value`.
        # The caller's array is registered (pinned) once, as the glue does for the propagator's state.
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

    def barrier():
        torch.cuda.synchronize()     # nothing of ours in flight while the barrier's collective runs
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # on-box streaming ceiling: y += a x over 2^26 complex elements, 48 B per element
    stream_gbs = None
    if world == 1:
        ns = 1 << 26
        sx, sy = L.State(ctx, n=ns), L.State(ctx, n=ns)
        sx.fill(1.0)
        for _ in range(2):
            sy.axpy(0.5, sx)
        ctx.timer_begin()
        for _ in range(10):
            sy.axpy(0.5, sx)
        stream_gbs = 10 * 48.0 * ns / (ctx.timer_end() * 1e-3) / 1e9
        sx.close()
        sy.close()

    for _ in range(args.warmup):
        step()
    ctx.reset_stats()
    barrier()
    ctx.timer_begin()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    t_enq = time.perf_counter() - t0
    ev_ms = ctx.timer_end()          # HIP events on the kernels' own stream
    t_ev = time.perf_counter() - t0
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if os.environ.get("QP_BENCH_DEBUG"):
        print(f"[debug] rank {rank}: enqueue {1e3*t_enq:.2f} ms, +events {1e3*t_ev:.2f} ms, +sync/barrier "
              f"{1e3*elapsed:.2f} ms, hip events {ev_ms:.2f} ms", file=sys.stderr)
    st = ctx.stats()
    if world == 1 and args.cpu_steps > 0:
        cpu_baselines()  # after the timed region: 16 busy OpenMP threads must not sit next to the measurement
    if world > 1:
        sh.check()      # outside the timed region: the overlapped schedule never timed out
    if dist is not None:
        t = torch.tensor([elapsed, ev_ms], dtype=torch.float64, device="cpu" if one_gpu else "cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed, ev_ms = float(t[0]), float(t[1])

    # HBM traffic per launch of the dominant kernel: PMC counters need their own rocprofv3
    # passes (tools/profile.sh), so the value measured for this exact command is read back
    # from the committed summary under profiles/ (null when there is none for this kernel)
    traffic = None
    kern = {1: "csr_spmv_kernel<16,ChebyOp>", 2: "rbcsr_spmv_kernel<ChebyOp,7>", 3: "hrb_spmv_kernel<ChebyOp,7>"}[fmt_used]
    pmc_file = os.path.join(ROOT, "profiles", "r01", "bench_final_pmc_summary.json")
    if (fmt_used == 3 and world == 1 and args.log2n == 20 and args.pattern == "banded" and not args.real and
            os.path.exists(pmc_file)):
        with open(pmc_file) as f:
            traffic = json.load(f)["hbm_traffic_bytes_per_launch"]

    steps_per_s = args.steps / elapsed
    n_launch = args.steps * nterms
    # algorithmic bytes of one fused term on one GPU: (20 z + 84) N + 4
    alg_bytes = (12.0 if args.real else 20.0) * nnz_local + 4.0 * (rows + 1) + 80.0 * rows
    avg_launch_s = (ev_ms * 1e-3) / n_launch
    achieved = alg_bytes / avg_launch_s / 1e9
    out = {
        "metric": "Cheby prop_step!/s at N=2^20 CSR fp64 (2^20-row blocks advanced per second)",
        "value": steps_per_s * world,
        "unit": "prop_step/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / args.steps,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "c128 state, f64 matrix values" if args.real else "c128 (complex fp64)", "data": "synthetic",
        "config": {"workload": "BASELINE configs[1]: Cheby prop_step!, CSR sparse Hermitian H, 16 nnz/row, "
                               + ("real fp64 values (f64 variant), " if args.real else "complex fp64 values, ")
                               + "int32 indices",
                   "rows_per_gpu": rows, "N_total": N, "nnz_per_row": 16, "pattern": args.pattern,
                   "offsets": [int(o) for o in offsets], "n_coeffs": int(len(coeffs)), "matvecs_per_step": nterms,
                   "spectral_range": [-10.0, 10.0], "dt": dt,
                   "device_format": {1: "csr", 2: "rbcsr", 3: "hrb (Hermitian-packed row blocks)"}[fmt_used],
                   "device_layout": layout,
                   "parallelism": "single GPU" if world == 1 else (
                       f"row-partitioned x{world}, exchange={exchange_used}, schedule={schedule_note}, driver={driver_note}"
                       + (" [TEST MODE: ranks share one GPU, host-staged gloo]" if one_gpu else "")),
                   "global_steps_per_s": steps_per_s},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                     "traffic_source": "profiles/r01/bench_final_pmc_summary.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, "
                                       "FETCH x2 gfx950 correction)" if traffic else None,
                     "kernel": kern,
                     "algorithmic_bytes_per_launch": alg_bytes,
                     "avg_launch_us": avg_launch_s * 1e6,
                     "launches_timed": n_launch, "hip_event_ms": ev_ms,
                     "traffic_rate_gbs": (traffic / avg_launch_s / 1e9) if traffic else None,
                     "hbm_stream_measured_gbs": stream_gbs,
                     "traffic_frac_of_stream": (traffic / avg_launch_s / 1e9 / stream_gbs) if (traffic and stream_gbs) else None,
                     "traffic_frac_of_peak": (traffic / avg_launch_s / 1e9 / HBM_PEAK_GBS) if traffic else None,
                     "note": "avg launch duration = HIP-event time of the timed region on the kernels' stream / "
                             "number of fused-term launches (includes launch gaps; multi-GPU: includes exchange). "
                             "`achieved` uses the contract's algorithmic CSR bytes (SURVEY 8d: (20 z + 84) N = 404 B/row); "
                             "the shipped layout moves fewer: Hermitian packing (lower triangle read back from L2 as "
                             "conjugates), stencil row blocks (block-wide column distances instead of per-entry "
                             "indices) and the Psi accumulator touched every third term, about 190 B/row, so "
                             "`achieved` exceeds what the same bytes would allow -- `traffic` is the HBM bytes the "
                             "PMC counters saw per launch, `traffic_rate_gbs` / `traffic_frac_of_peak` the real HBM rate"},
        "cpu_baseline": cpu,
        "cpu_baseline_all_cores": cpu_omp,
        "pcie_inclusive_steps_per_s": pcie,
        "stats": {"n_matvec": st["n_matvec"], "kernel_launches": st["n_kernel_launches"]},
    }
    if rank == 0:
        print(json.dumps(out))
    if dist is not None:
        sh.close()      # the library's communicator goes before the process group it was bootstrapped over
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
