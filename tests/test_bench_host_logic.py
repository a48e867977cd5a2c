"""CPU tests of the measurement plumbing that needs no GPU: the repeat / median / outlier record of tools/bench_points.py, the
prediction table of bench.py (what the first real multi-GPU run is read against), the quota-aware thread count of the
synthetic generator, and that the chunked generator is deterministic across chunkings."""
import importlib.util
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))


def _bench():
    spec = importlib.util.spec_from_file_location("qp_bench_module", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_spread_flags_an_unstable_point_and_blames_the_host_when_it_was_the_host():
    import bench_points as bp
    quiet = bp.spread([134.6, 134.5, 135.7], [(20.9, 0.6, 0.2, 0.0), (20.8, 0.6, 0.2, 0.0), (21.0, 0.7, 0.2, 0.0)])
    assert quiet["median"] == 134.6 and quiet["unstable"] is False and "slowest_region" not in quiet
    # one region with a 70 ms stall inside ONE library call while the control group was throttled
    stalled = bp.spread([134.6, 586.1, 134.5], [(20.9, 0.6, 0.2, 0.0), (90.85, 70.7, 70.2, 1656.3), (20.8, 0.6, 0.2, 0.0)])
    assert stalled["median"] == 134.6 and stalled["min"] == 134.5 and stalled["max"] == 586.1 and stalled["unstable"] is True
    sr = stalled["slowest_region"]
    assert sr["host_stall_suspected"] is True and sr["cpu_quota_throttled_ms"] == 1656.3 and sr["longest_single_call_ms"] == 70.2
    # a slow region the host does NOT explain (short enqueue, long events): a device matter
    device = bp.spread([100.0, 100.0, 180.0], [(15.5, 0.5, 0.2, 0.0), (15.5, 0.5, 0.2, 0.0), (27.9, 0.6, 0.2, 0.0)])
    assert device["unstable"] is True and device["slowest_region"]["host_stall_suspected"] is False


def test_scaling_prediction_table():
    b = _bench()
    sizes = {20: 31.1, 21: 65.5, 22: 127.6, 23: 284.2, 24: 629.6}
    p = b.scaling_prediction(sizes, 1036.5, 31, "test")
    rows = p["fixed_problem_N_2^24"]
    assert [r["gpus"] for r in rows] == [1, 2, 4, 8] and [r["rows_per_gpu"] for r in rows] == [1 << 24, 1 << 23, 1 << 22, 1 << 21]
    assert rows[0]["speedup_halo_overlap"] == 1.0
    # halo form: the exchange (10 us + 131 KB at 153 GB/s) hides behind the interior launch; the machinery costs 8.6 %
    assert abs(rows[3]["exchange_us_halo"] - (10.0 + 16 * 8192 / 153e3)) < 1e-9
    assert abs(rows[3]["us_per_term_halo_overlap"] - b.MACHINERY_FACTOR * 65.5) < 1e-9
    assert rows[3]["speedup_halo_overlap"] > 6.0 and rows[3]["speedup_halo_serial"] > 6.0
    # all-gather form: 16 N / G bytes per link and term -- link-bound, far from 6 x
    assert abs(rows[3]["exchange_us_allgather"] - (10.0 + 16.0 * (1 << 21) / 153e3)) < 1e-9
    assert rows[3]["speedup_allgather"] < 3.0 and rows[1]["speedup_allgather"] < 1.0
    weak = p["bench_default_weak_2^21_rows_per_gpu"]
    assert [w["gpus"] for w in weak] == [1, 2, 4, 8] and 0.8 < weak[3]["predicted_efficiency_vs_1gpu_value"] < 1.0
    # round 6: one GPU alone takes the two-term walk beyond the Infinity Cache, a rank of the row-partitioned step does not (it exchanges
    # after every term): the denominator of the fixed problem is the faster single-GPU time, the ranks keep the one-term walk's times
    p2 = b.scaling_prediction({20: 30.9, 21: 57.0, 22: 103.0, 23: 185.0, 24: 364.0}, 1042.0, 31, "test", one_term={21: 67.0, 22: 136.0, 23: 285.0})
    r2 = p2["fixed_problem_N_2^24"]
    assert r2[0]["us_per_term_compute"] == 364.0 and r2[3]["us_per_term_compute"] == 67.0 and r2[1]["us_per_term_compute"] == 285.0
    assert 4.5 < r2[3]["speedup_halo_overlap"] < 5.5          # 364 / (1.086 x 67): below the 6 x that the one-term denominator gave
    assert p2["us_per_term_one_term_walk_by_log2_rows"] == {21: 67.0, 22: 136.0, 23: 285.0}
    # a run that lacks a size leaves that row out instead of inventing it
    q = b.scaling_prediction({21: 65.5, 24: 629.6}, 1036.5, 31, "test")
    assert [r["gpus"] for r in q["fixed_problem_N_2^24"]] == [1, 8]
    # the committed single-GPU sizes feed a multi-GPU line's table
    st = b.scaling_prediction_static(31)
    assert st is None or st["source_of_compute_times"].startswith("STATIC")


def test_exchange_model_of_a_halo_partition():
    b = _bench()

    class Sh:
        M, exchange, send_to = 8192, "halo", [1, 7]
    xm = b.exchange_model(Sh, 8, 1 << 21, 70.0)
    assert xm["peers"] == 2 and xm["bytes_on_busiest_link_per_term"] == 16 * 8192
    assert xm["predicted_exposed_us_per_term_overlap_schedule"] == 0.0 and xm["predicted_exposed_us_per_term_serial_schedule"] > 10.0

    class Ag:
        M, exchange, send_to = 1 << 21, "allgather", None
    xa = b.exchange_model(Ag, 8, 1 << 21, 70.0)
    assert xa["peers"] == 7 and xa["bytes_sent_per_rank_per_term"] == 16.0 * (1 << 21) * 7
    assert xa["predicted_exposed_us_per_term_overlap_schedule"] > 100.0          # 32 MiB per link: not hidden by a 70 us launch


def test_generator_is_chunking_independent_and_quota_aware():
    import qprop_amd.synth as synth
    assert 1 <= synth._usable_cpus() <= (os.cpu_count() or 1)
    N = 70000 + 1234       # not a multiple of the chunk size; banded offsets wrap at both ends
    rp, col, vals = synth.hermitian_offsets_csr(N)
    a, b, c = synth.hermitian_offsets_csr(N, row_begin=65000, row_end=69000)
    assert np.array_equal(col[65000 * 16:69000 * 16], b) and np.array_equal(vals[65000 * 16:69000 * 16], c)
    # Hermitian, 16 distinct columns per row, ascending
    import scipy.sparse as sp
    H = sp.csr_matrix((vals, col, rp), shape=(N, N))
    assert abs(H - H.getH()).max() == 0.0
    assert np.all(np.diff(col.reshape(N, 16), axis=1) > 0)
    offs = synth.scattered_offsets(1 << 14)
    r2, c2, v2 = synth.hermitian_offsets_csr(1 << 14, offsets=offs)
    H2 = sp.csr_matrix((v2, c2, r2), shape=(1 << 14, 1 << 14))
    assert abs(H2 - H2.getH()).max() == 0.0 and np.all(np.diff(c2.reshape(-1, 16), axis=1) > 0)


def test_a_region_with_a_host_stall_is_measured_again_and_kept_on_record():
    """tools/bench_points.py: timed_regions -- one enqueue call that takes more than half of its region's event time means the
    device idled inside the event bracket; the region is repeated (at most twice per point) and the discarded one reported."""
    import time
    import bench_points as bp

    class FakeCtx:
        def sync(self):
            pass

        def timer_begin(self):
            self.t0 = time.perf_counter()

        def timer_end(self):
            return 1e3 * (time.perf_counter() - self.t0)

    calls = {"n": 0}

    def fn():
        calls["n"] += 1
        time.sleep(0.08 if calls["n"] == 2 else 0.002)      # the second call of the first region is held for 80 ms

    regions = bp.timed_regions(FakeCtx(), fn, steps=4, repeats=3)
    assert len(regions) == 3 and len(regions.discarded) == 1 and calls["n"] == 16
    assert regions.discarded[0][2] > 70.0 and all(r[2] < 0.5 * r[0] for r in regions)
    sp = bp.spread([r[0] for r in regions], regions)
    assert sp["unstable"] is False and len(sp["regions_remeasured_after_host_stall"]) == 1
    assert sp["regions_remeasured_after_host_stall"][0]["longest_single_call_ms"] > 70.0


def test_lattice_completion_knows_the_round_4_stencil_shapes():
    """Host index work, no GPU: `qp_lattice_fill_host` completes an open-boundary grid's edge rows only when the fullest row's
    distance list has a shape the strip walk has a kernel for -- now also the thirteen-point stencil of a three-dimensional grid
    (two long pairs beside two far distances; the reference row is found on small grids too), the nine-point stencil with
    diagonal neighbours, and layers of such planes (diagonal neighbours + one long pair)."""
    import qprop_amd.lib as L
    import qprop_amd.synth as synth
    cases = {"thirteen-point 64x8x24": synth.grid_hamiltonian_3d(64, 8, 24, order=4),
             "thirteen-point 72x10x20": synth.grid_hamiltonian_3d(72, 10, 20, flux=0.1, order=4),
             "nine-point diagonal 96x50": synth.grid_hamiltonian_2d(96, 50, flux=0.2, diagonal=0.3),
             "layers 72x10x20 with t'": synth.grid_hamiltonian_3d(72, 10, 20, flux=0.2, diagonal=0.3),
             "seven-point 64x8x40": synth.grid_hamiltonian_3d(64, 8, 40)}
    for name, H in cases.items():
        N = H.shape[0]
        z = int(np.diff(H.indptr).max())
        rp, col = L.lattice_fill_host(N, N, H.indptr, H.indices, min_blocks=16)
        filled = int(rp[-1]) - H.nnz
        assert 0 < filled < 0.12 * H.nnz, (name, filled)
        lens = np.diff(rp)
        reach = int(np.max(np.abs(H.indices[H.indptr[N // 2 + 7]:H.indptr[N // 2 + 8]] - (N // 2 + 7))))
        reach = max(reach, int(np.max(np.abs(col[rp[N // 2]:rp[N // 2 + 1]] - N // 2))))
        assert np.all(lens[reach:N - reach] == z), name          # every row between the first and last `reach` rows is complete
        # the original entries are all still there, columns ascending
        for r in (0, reach, N // 2, N - 1):
            orig = set(H.indices[H.indptr[r]:H.indptr[r + 1]].tolist())
            got = col[rp[r]:rp[r + 1]]
            assert orig <= set(got.tolist()) and np.all(np.diff(got) > 0), (name, r)
    # a nine-point stencil whose strip step is below 64 rows is not the walk's lattice: nothing is completed
    Hs = synth.grid_hamiltonian_2d(40, 120, diagonal=0.3)
    rp, col = L.lattice_fill_host(Hs.shape[0], Hs.shape[0], Hs.indptr, Hs.indices, min_blocks=16)
    assert int(rp[-1]) == Hs.nnz


def _strings(o):
    if isinstance(o, str):
        yield o
    elif isinstance(o, dict):
        for v in o.values():
            yield from _strings(v)
    elif isinstance(o, (list, tuple)):
        for v in o:
            yield from _strings(v)


def test_printed_line_is_compact_and_carries_the_gate_keys(capsys, tmp_path, monkeypatch):
    """the driver reads the LAST stdout line; round 4's 36 KB line defeated it.  The canned measurement is
    that very record (profiles/r04/bench_line.json, every extras point and the prediction table included)."""
    import json
    b = _bench()
    with open(os.path.join(ROOT, "profiles", "r04", "bench_line.json")) as f:
        full = json.load(f)
    assert len(json.dumps(full)) > 30000
    monkeypatch.setattr(b, "ROOT", str(tmp_path))       # the sidecar goes next to the script: not into the repo from a test
    full["prediction"] = b.prediction_scalars(full["scaling_prediction"])
    # round 6: every point names ITS byte model, the headline says whether its working set is cache
    # resident, and the C3 / C5 points carry the CPU port's time beside the GPU's
    full["roofline"].update(model="layout", cache_resident=True)
    full.update(exchange="none", rccl_ranks=0)
    for name, pt in full["extras"].items():
        pt["model"] = "impl" if name.startswith("c3") else ("flops" if "matrix_cores" in name or name.startswith("n4") else "layout")
    full["extras"]["c3_newton"]["cpu_baseline"] = {"value": 0.4, "ms_per_step": 2500.0, "cores": 1, "kind": "port"}
    b.emit(full)
    lines = capsys.readouterr().out.strip().splitlines()
    text = lines[-1]
    assert len(text) < b.LINE_LIMIT == 4096
    line = json.loads(text)
    assert line["roofline"]["model"] == "layout" and line["roofline"]["cache_resident"] is True
    assert all(len(v) >= 3 and v[2] in ("layout", "impl", "flops") for v in line["points"].values())
    assert line["points"]["c3_newton"][2:] == ["impl", 2500.0] and line["points"]["dense_h_n4096_64_states_on_the_matrix_cores"][2] == "flops"
    assert "exchange" in line and "rccl_ranks" in line
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
              "data", "config", "roofline", "cpu_baseline", "degraded"):
        assert k in line, k
    assert line["value"] == float(f"{full['value']:.6g}") and line["n_gpus"] == 1
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "avg_launch_us", "unstable"):
        assert k in line["roofline"], k
    assert 0.0 < line["roofline"]["frac"] <= 1.0 and line["roofline"]["bound"] == "hbm"
    assert line["cpu_baseline"]["value"] > 0 and line["cpu_baseline"]["cores"] == 1 and line["cpu_baseline"]["kind"] == "port"
    assert "workload" in line["config"] and line["config"]["device_format"]
    assert all(len(s) <= 160 for s in _strings(line)), max(_strings(line), key=len)
    # one pair per extras point; the other stdout lines are the short EXTRA ones
    assert set(line["points"]) == set(full["extras"]) and line["points"]["c3_newton"][0] == float(f"{full['extras']['c3_newton']['ms_per_step']:.4g}")
    assert all(ln.startswith("EXTRA ") and len(ln) < 200 for ln in lines[:-1]) and len(lines) - 1 == len(full["extras"])
    # the complete record went to the sidecar
    with open(tmp_path / b.EXTRAS_FILE) as f:
        assert json.load(f)["extras"].keys() == full["extras"].keys()


def test_printed_line_of_a_multi_gpu_and_a_c5_record_stays_under_the_limit(capsys, tmp_path, monkeypatch):
    import json
    b = _bench()
    monkeypatch.setattr(b, "ROOT", str(tmp_path))
    with open(os.path.join(ROOT, "profiles", "r04", "bench_line.json")) as f:
        full = json.load(f)
    # what a --gpus 8 record adds: long parallelism prose, the strong point, the all-gather form, the prediction, the first measurement
    full.update(n_gpus=8, extras=None, cpu_baseline=None, cpu_baseline_all_cores=None)
    full["config"]["parallelism"] = "row-partitioned x8, " + "x" * 700
    full["config"]["parallelism_short"] = "row-partitioned x8, exchange=halo, schedule=overlap, driver=native (own RCCL communicator)"
    full["strong_point"] = {"N_total": 1 << 20, "prop_steps_per_s": 2100.0, "ms_per_step": 0.476, "exchange": "halo"}
    full["allgather_form"] = {"us_per_term": 310.2, "blocks_per_s": 1664.0, "driver": "native"}
    full["conservative_first"] = {"value": 3000.0, "ms_per_step": 5.3}
    full["prediction"] = b.prediction_scalars(b.scaling_prediction_static(31))
    b.emit(full)
    line = json.loads(capsys.readouterr().out.strip().splitlines()[-1])
    assert len(json.dumps(line, separators=(",", ":"))) < 4096 and line["cpu_baseline"] is None
    assert line["allgather_form"]["us_per_term"] == 310.2 and line["strong_point"]["prop_steps_per_s"] == 2100.0
    assert all(len(s) <= 160 for s in _strings(line))
    with open(os.path.join(ROOT, "profiles", "r04", "bench_line_config_c5.json")) as f:
        c5 = json.load(f)
    b.emit(c5)
    line = json.loads(capsys.readouterr().out.strip().splitlines()[-1])
    assert len(json.dumps(line)) < 4096 and line["roofline"]["frac"] > 0 and all(len(s) <= 160 for s in _strings(line))
    # a record that would still be too long loses its optional parts, never its gate keys
    full["extras"] = {f"point_{i}_{'y' * 60}": {"us_per_term": 1.0, "frac": 0.5} for i in range(80)}
    b.emit(full)
    line = json.loads(capsys.readouterr().out.strip().splitlines()[-1])
    assert "points" not in line and line["roofline"]["frac"] > 0 and len(json.dumps(line, separators=(",", ":"))) < 4096
