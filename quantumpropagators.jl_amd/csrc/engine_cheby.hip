// cheby!: one step, batched states, one fused term (plain and boundary/interior split),
// and the propagate step loop.
#include "engine.h"

// terms whose epilogue updates the Psi accumulator: every third one counted from the last
// (the epilogue of term m holds v_{m-2}, v_{m-1}, v_m of the row).  The first update must
// still see Psi = v_0, which term 2 overwrites in place, so it is forced to term <= 2.
void acc_schedule(const double* a, int n_coeffs, bool defer, qp_acc_defer* out) {
  const int nterms = n_coeffs - 1;
  std::vector<char> upd((size_t)nterms + 1, defer ? 0 : 1);
  if (defer) {
    int m0 = nterms;
    for (; m0 >= 1; m0 -= 3) upd[m0] = 1;
    if (m0 + 3 == 3) upd[1] = 1;
  }
  int last_upd = 0;
  for (int m = 1; m <= nterms; ++m) {
    qp_acc_defer& d = out[m - 1];
    d = qp_acc_defer{0, 0, 0.0, 0.0};
    if (upd[m]) {
      d.n_defer = m - last_upd - 1;
      d.a_d1 = (d.n_defer >= 1) ? a[m - 1] : 0.0;
      d.a_d2 = (d.n_defer == 2) ? a[m - 2] : 0.0;
      last_upd = m;
    } else {
      d.skip = 1;
    }
  }
}

int split_timed_out(const qp_split* sp) {
  if (sp && sp->timeout_host && __atomic_load_n(sp->timeout_host, __ATOMIC_RELAXED) != 0)
    return qp::fail(QP_E_INTERNAL, "an interior launch of an earlier term timed out waiting for its boundary launch: "
                                   "the state of this partitioned cheby! is not valid");
  return QP_OK;
}

void set_defer(qp::ChebyEpi& e, const qp_acc_defer* d) {
  if (!d) return;
  e.acc_skip = d->skip ? 1 : 0;
  e.n_defer = d->skip ? 0 : d->n_defer;
  e.a_d1 = d->a_d1;
  e.a_d2 = d->a_d2;
}

extern "C" {

// ---------------------------------------------------------------------------
// Chebyshev
// ---------------------------------------------------------------------------
int qp_cheby_create(qp_ctx* ctx, int64_t n, qp_cheby** out) {
  QP_TRY
  if (!ctx || !out || n < 0) return qp::fail(QP_E_BAD_ARG, "qp_cheby_create: bad arguments");
  QP_CHECK(use(ctx));
  auto w = std::make_unique<qp_cheby>();
  w->ctx = ctx;
  w->n = n;
  QP_CHECK(dev_alloc(&w->bufA, (size_t)n));
  QP_CHECK(dev_alloc(&w->acc, (size_t)n));
  *out = w.release();
  return QP_OK;
  QP_CATCH
}

int qp_cheby_destroy(qp_cheby* w) {
  QP_TRY
  if (!w) return QP_OK;
  (void)hipSetDevice(w->ctx->device);
  (void)hipStreamSynchronize(w->ctx->stream);
  if (w->bufA) (void)hipFree(w->bufA);
  if (w->bufC) (void)hipFree(w->bufC);
  if (w->bufD) (void)hipFree(w->bufD);
  if (w->acc) (void)hipFree(w->acc);
  if (w->chk_part) (void)hipFree(w->chk_part);
  if (w->chk_out) (void)hipFree(w->chk_out);
  if (w->gexec) (void)hipGraphExecDestroy(w->gexec);
  delete w;
  return QP_OK;
  QP_CATCH
}

int qp_acc_schedule_host(const double* a, int n_coeffs, qp_acc_defer* out) {
  if (!a || !out || n_coeffs < 2) return qp::fail(QP_E_BAD_ARG, "qp_acc_schedule_host: bad arguments");
  acc_schedule(a, n_coeffs, true, out);
  return QP_OK;
}

int qp_cheby_term(qp_operator* op, const qp_state* x, int64_t xoff, const qp_state* v0, qp_state* vout,
                  const qp_state* acc_in, qp_state* acc_out, qp_c128 c, double beta, double a_prev, double a,
                  qp_c128 phase, const qp_acc_defer* defer) {
  QP_TRY
  const bool skip = defer && defer->skip;
  if (!op || !x || (!acc_out && !skip)) return qp::fail(QP_E_BAD_ARG, "qp_cheby_term: NULL argument");
  if (defer && !skip && (defer->n_defer < 0 || defer->n_defer > 2 || (defer->n_defer > 0 && !v0 && (defer->n_defer == 2 || !acc_in))))
    return qp::fail(QP_E_BAD_ARG, "qp_cheby_term: deferred accumulation needs v0");
  const int64_t nr = op->A.nrows;
  if (x->n != op->A.ncols || xoff < 0 || xoff + nr > x->n) return qp::fail(QP_E_BAD_ARG, "qp_cheby_term: x shape / offset mismatch");
  if ((v0 && v0->n != nr) || (vout && vout->n != nr) || (acc_in && acc_in->n != nr) || (acc_out && acc_out->n != nr))
    return qp::fail(QP_E_BAD_ARG, "qp_cheby_term: local vector length mismatch");
  auto overlaps = [&](const qp_state* s) { return s && s->d < x->d + x->n && x->d < s->d + s->n; };
  if (overlaps(vout) || overlaps(acc_out)) return qp::fail(QP_E_BAD_ARG, "qp_cheby_term: outputs must not overlap the gathered x");
  QP_CHECK(use(op->ctx));
  qp::ChebyEpi e;
  e.xloc = x->d + xoff;
  e.v0 = v0 ? v0->d : nullptr;
  e.vout = vout ? vout->d : nullptr;
  e.acc_in = (acc_in && !skip) ? acc_in->d : nullptr;
  e.acc_out = (acc_out && !skip) ? acc_out->d : nullptr;
  e.c = d2(c);
  e.beta = beta;
  e.a_prev = a_prev;
  e.a = a;
  e.phase = d2(phase);
  e.apply_phase = !(phase.re == 1.0 && phase.im == 0.0);
  e.check_partials = nullptr;
  set_defer(e, defer);
  return qp::launch_spmv_cheby(op->ctx->stream, op->A, x->d, e, &op->ctx->stats);
  QP_CATCH
}

// Does a whole-operator cheby! of `op` take the two-term strip walk (kernels_walk2.hip) under the context's knobs?  Beyond the
// Infinity Cache only (inside it the one-term walk is not bound by the value stream), and only when a wavefront's strip column is
// long enough for the 2 K steps a segment runs in before its first z to be a small part of it.
static bool walk2_wanted(const qp_operator* op) {
  const qp::Tuning& tun = op->ctx->tun;
  const qp::WalkPlan& Q = op->walk2;
  const DevMatrix& A = op->A;
  if (!Q.valid || tun.walk_pair == 0 || !tun.hrb_walk || (tun.rbcsr_variant & 31) != 15 || A.walk != &op->walk || !op->walk.valid) return false;
  if (op->walk.R1 - op->walk.W0 < tun.walk_min_blocks) return false;
  if (tun.walk_pair == 1) return true;
  const double footprint = (double)(Q.z0 + Q.nn + Q.K) * kRB * (double)A.nblocks * (A.vals_r ? 8.0 : 16.0) + 64.0 * (double)A.nrows;
  if (footprint <= 230e6) return false;
  const int W = kRB - 2 * Q.near[Q.nn - 1];
  const int64_t S2 = (Q.g + W - 1) / W, Jz = ((Q.R1 - Q.W0) * (int64_t)kRB + Q.g - 1) / Q.g;
  const int64_t waves = tun.walk_waves > 0 ? tun.walk_waves : 4 * (int64_t)qp::device_cu_count();
  return Jz / std::max<int64_t>(1, waves / S2) >= 24;
}

int qp_operator_walk2_info(const qp_operator* op, int64_t out[8]) {
  QP_TRY
  if (!op || !out) return qp::fail(QP_E_BAD_ARG, "qp_operator_walk2_info: NULL argument");
  const bool on = walk2_wanted(op);
  const int W = on ? kRB - 2 * op->walk2.near[op->walk2.nn - 1] : 0;
  out[0] = on ? 1 : 0;
  out[1] = on ? op->walk2.W0 : 0;
  out[2] = on ? op->walk2.R1 : 0;
  out[3] = on ? op->walk2.n_edge : 0;
  out[4] = W;
  out[5] = on ? (op->walk2.g + W - 1) / W : 0;
  out[6] = out[7] = 0;
  if (on) {      // the cut of the walk (kernels_walk2.hip: launch_hrb_walk2_cheby): z steps per wavefront, segments per strip column
    const qp::Tuning& tun = op->ctx->tun;
    const int64_t Jz = ((op->walk2.R1 - op->walk2.W0) * (int64_t)kRB + op->walk2.g - 1) / op->walk2.g;
    const int64_t waves = tun.walk_waves > 0 ? tun.walk_waves : 4 * (int64_t)qp::device_cu_count();
    const int64_t L = (Jz + std::max<int64_t>(1, waves / out[5]) - 1) / std::max<int64_t>(1, waves / out[5]);
    out[6] = L;
    out[7] = (Jz + L - 1) / L;
  }
  return QP_OK;
  QP_CATCH
}

// the launches of one cheby! call (src/cheby.jl:171-211): n_coeffs - 1 fused mat-vec + term
// kernels and, when the result does not land in Psi's buffer, one copy.
// Term vectors: `cur` holds v_{m-1} (gathered by term m), `prev` holds v_{m-2}.  A one-term launch writes v_m over v_{m-2} in
// place; a two-term launch (terms m and m + 1 in one pass over the matrix values, kernels_walk2.hip) writes v_m and v_{m+1} to two
// vectors nobody reads meanwhile -- the neighbouring strip columns still gather from `cur` and `prev` -- so a step that takes pairs
// rotates four vectors (Psi's, bufA, bufC, bufD) instead of two.
static int cheby_step_launches(qp_cheby* w, qp_operator* op, qp_state* psi, const double* a, int n_coeffs, double beta,
                               cplx c, cplx phase, bool check_normalization) {
  qp_ctx* ctx = op->ctx;
  const DevMatrix& A = op->A;
  const int nterms = n_coeffs - 1;
  const int nwg = qp::spmv_grid_size(A);
  double2* P = psi->d;
  double2* B = w->bufA;
  double2* ACC = w->acc;
  double2* result = nullptr;
  std::vector<qp_acc_defer> sched((size_t)nterms);
  acc_schedule(a, n_coeffs, ctx->tun.acc_defer != 0, sched.data());
  bool pairs = !check_normalization && nterms >= 3 && w->bufC && w->bufD && walk2_wanted(op);   // (qp_cheby_step allocated the two vectors)
  double2 *cur = P, *prev = nullptr;
  double2* spare[2] = {w->bufC, w->bufD};      // the two vectors no term of the step is reading
  bool updated = false;   // has any term written the accumulator yet?
  auto fill = [&](qp::ChebyEpi& e, int m, const double2* x) {      // what every term's epilogue carries
    e.a_prev = updated ? 0.0 : a[0];
    set_defer(e, &sched[m - 1]);
    if (sched[m - 1].skip) {
      e.acc_in = nullptr;
      e.acc_out = nullptr;
    } else {
      updated = true;
    }
    e.xloc = x;
    e.c = d2(c);
    e.beta = beta;
    e.a = a[m];
    e.phase = d2(phase);
    e.apply_phase = (m == nterms) ? 1 : 0;
  };
  for (int m = 1; m <= nterms; ++m) {
    const bool last = (m == nterms);
    qp::ChebyEpi e;
    if (m == 1) {
      // v0 = Psi; Psi = a1 v0; v1 = c (H v0 - beta v0); Psi += a2 v1     :171-182
      e.v0 = nullptr;
      e.vout = last ? nullptr : B;
      e.acc_in = nullptr;
      e.acc_out = ACC;
      result = ACC;
    } else if (pairs && m + 1 <= nterms && !(sched[m - 1].skip == 0 && sched[m].skip == 0)) {
      // terms m and m + 1 in one pass over the values: y = v_m -> spare[0], z = v_{m+1} -> spare[1]
      const bool last2 = (m + 1 == nterms);
      const bool updated_before = updated;
      qp::ChebyEpi e1, e2;
      e1.v0 = prev;
      e1.vout = spare[0];
      e1.acc_in = updated ? ACC : nullptr;
      e1.acc_out = ACC;
      e1.check_partials = nullptr;
      fill(e1, m, cur);
      e2.v0 = cur;
      e2.vout = last2 ? nullptr : spare[1];
      e2.acc_in = updated ? ACC : nullptr;
      e2.acc_out = ACC;
      e2.check_partials = nullptr;
      fill(e2, m + 1, spare[0]);
      bool launched = false;
      {
        const qp::ScopedRange mv_range(ctx->tun.roctx != 0 || qp::ranges_enabled_by_env(), "matrix-vector product");
        QP_CHECK(qp::launch_hrb_walk2_cheby(ctx->stream, A, op->walk2, cur, e1, e2, ctx->tun, &launched));
        if (launched) {
          // term m + 1 of the blocks outside the two-term region: they read y of their neighbours, which the launch above wrote
          qp::RowSet rs;
          rs.block_map = op->walk2.edge_map;
          rs.nmap = op->walk2.n_edge;
          rs.count = false;
          QP_CHECK(qp::launch_spmv_cheby(ctx->stream, A, spare[0], e2, &ctx->stats, &rs));
          ctx->stats.n_launch++;
          ctx->stats.n_matvec += 2;
          ctx->stats.spmv_bytes += 2.0 * (20.0 * (double)A.nnz + 4.0 * (double)(A.nrows + 1) + 80.0 * (double)A.nrows);
        }
      }
      if (launched) {
        result = ACC;
        double2* const y = spare[0];
        double2* const z = spare[1];
        spare[0] = cur;
        spare[1] = prev;
        prev = y;
        cur = z;
        ++m;
        continue;
      }
      // not taken after all (the kernel's LDS opt-in was refused): this and every later term as one-term launches
      updated = updated_before;
      pairs = false;
    }
    if (m > 1) {
      // v2 = c (H v1 - beta v1) + v0; Psi += a_i v2; rotate            :186-207
      e.v0 = prev;                      // holds v0, overwritten in place by v2
      e.vout = last ? nullptr : prev;
      e.acc_in = updated ? ACC : nullptr;
      e.acc_out = (last && cur != P && !pairs) ? P : ACC;  // P may be written only while it is not gathered
      result = e.acc_out;
    }
    fill(e, m, cur);
    // the reference checks terms i >= 3 only (inside the loop at :186)
    e.check_partials = (check_normalization && m >= 2) ? w->chk_part : nullptr;
    {
      const qp::ScopedRange mv_range(ctx->tun.roctx != 0 || qp::ranges_enabled_by_env(), "matrix-vector product");   // src/cheby.jl:175, :189
      QP_CHECK(qp::launch_spmv_cheby(ctx->stream, A, cur, e, &ctx->stats));
    }
    if (e.check_partials)
      QP_CHECK(qp::launch_reduce_triples(ctx->stream, w->chk_part, nwg, w->chk_out + 3 * (m - 1), &ctx->stats));
    if (m == 1) {
      c *= 2.0;  // :184
      prev = P;
      cur = B;
    } else {
      std::swap(cur, prev);      // prev's buffer now holds v_m: it is gathered next
    }
  }
  if (result != P)
    QP_HIP(hipMemcpyAsync(P, result, (size_t)psi->n * sizeof(double2), hipMemcpyDeviceToDevice, ctx->stream));
  return QP_OK;
}

int qp_cheby_step(qp_cheby* w, qp_operator* op, qp_state* psi, const double* a, int n_coeffs, double Delta,
                  double E_min, double dt, double wrk_dt, double limit, int check_normalization) {
  QP_TRY
  if (!w || !op || !psi || !a) return qp::fail(QP_E_BAD_ARG, "qp_cheby_step: NULL argument");
  if (op->A.nrows != op->A.ncols || psi->n != op->A.nrows || w->n != psi->n)
    return qp::fail(QP_E_BAD_ARG, "qp_cheby_step: shape mismatch");
  // @assert abs(dt) ~ abs(wrk.dt)   (isapprox, rtol = sqrt(eps))   src/cheby.jl:157
  {
    const double x = std::fabs(dt), y = std::fabs(wrk_dt);
    if (!(std::fabs(x - y) <= 1.4901161193847656e-08 * std::max(x, y)))
      return qp::fail(QP_E_DT_MISMATCH, "wrk was initialized for dt=%g, not dt=abs(%g)", wrk_dt, dt);
  }
  if (n_coeffs < 2) return qp::fail(QP_E_TOO_FEW_COEFFS, "Need at least 2 Chebychev coefficients");
  if (!(Delta > 0)) return qp::fail(QP_E_BAD_ARG, "Delta must be positive");
  qp_ctx* ctx = op->ctx;
  QP_CHECK(use(ctx));
  const qp::ScopedRange step_range(ctx->tun.roctx != 0 || qp::ranges_enabled_by_env(), "prop_step!");   // src/cheby_propagator.jl:349
  const double beta = (Delta / 2) + E_min;                        // :156
  cplx c = (dt > 0) ? cplx(0, -2.0) / Delta : cplx(0, 2.0) / Delta;  // :158-162
  const cplx phase = std::exp(cplx(0, -1) * beta * dt);            // :211
  const int nterms = n_coeffs - 1;
  const DevMatrix& A = op->A;
  const int nwg = qp::spmv_grid_size(A);
  if (check_normalization) {
    if (w->chk_wg < nwg) {
      if (w->chk_part) QP_HIP(hipFree(w->chk_part));
      QP_CHECK(dev_alloc(&w->chk_part, (size_t)3 * nwg));
      w->chk_wg = nwg;
    }
    if (w->chk_terms < nterms) {
      if (w->chk_out) QP_HIP(hipFree(w->chk_out));
      QP_CHECK(dev_alloc(&w->chk_out, (size_t)3 * nterms));
      w->chk_terms = nterms;
    }
  }
  // the two-term strip walk rotates four term vectors: the two beyond Psi's and bufA are allocated on first use (here, not inside a
  // stream capture)
  if (nterms >= 3 && !check_normalization && !w->bufC && walk2_wanted(op)) {
    QP_CHECK(dev_alloc(&w->bufC, (size_t)w->n));
    QP_CHECK(dev_alloc(&w->bufD, (size_t)w->n));
  }
  // launch-bound systems (a term takes less than its launch): replay the step as a hipGraph
  bool done = false;
  if (ctx->tun.cheby_graph && !check_normalization && ctx->stream != nullptr && ctx->stream != hipStreamLegacy) {
    qp_cheby::GraphKey key;
    key.vals = A.vals_r ? (const void*)A.vals_r : (const void*)A.vals;
    key.cols = A.cols;
    key.rowptr = qp::csr_layout(A.format) ? (const void*)A.rowptr : (const void*)A.bptr;
    key.psi = psi->d;
    key.format = A.format;
    // every knob that selects a kernel or a launch shape of the step's terms
    key.variant = ((((ctx->tun.rbcsr_variant * 2) * 16 + 8) * 2 + (ctx->tun.hrb_walk ? 1 : 0)) * 8 +
                   (ctx->tun.walk_nt & 7)) * 4096 + (ctx->tun.walk_waves & 4095);
    key.variant = key.variant * 2 + (ctx->tun.value_dict ? 1 : 0);      // (coded / plain row-block kernel)
    key.variant = key.variant * 2 + (walk2_wanted(op) ? 1 : 0);         // (terms in pairs)
    key.n_coeffs = n_coeffs;
    key.dt = dt;
    key.Delta = Delta;
    key.E_min = E_min;
    uint64_t h = 1469598103934665603ull;   // FNV-1a over the coefficient bits
    for (int i = 0; i < n_coeffs; ++i) {
      uint64_t bits;
      std::memcpy(&bits, &a[i], 8);
      h = (h ^ bits) * 1099511628211ull;
    }
    key.a_hash = h;
    auto replay = [&]() -> int {
      QP_HIP(hipGraphLaunch(w->gexec, ctx->stream));
      ctx->stats.n_graph_launch++;
      ctx->stats.n_matvec += w->gstats.n_matvec;
      ctx->stats.n_launch += w->gstats.n_launch;
      ctx->stats.spmv_bytes += w->gstats.spmv_bytes;
      return QP_OK;
    };
    if (w->gexec && key == w->gkey) {
      QP_CHECK(replay());
      done = true;
    } else if (key == w->gpending && (int64_t)spmv_grid_size(A) <= ctx->tun.cheby_graph) {
      // second identical call in a row: record it
      hipGraph_t graph = nullptr;
      const Stats before = ctx->stats;
      QP_HIP(hipStreamBeginCapture(ctx->stream, hipStreamCaptureModeThreadLocal));
      const int rc = cheby_step_launches(w, op, psi, a, n_coeffs, beta, c, phase, false);
      w->gstats = Stats();
      w->gstats.n_matvec = ctx->stats.n_matvec - before.n_matvec;
      w->gstats.n_launch = ctx->stats.n_launch - before.n_launch;
      w->gstats.spmv_bytes = ctx->stats.spmv_bytes - before.spmv_bytes;
      ctx->stats = before;
      const hipError_t ec = hipStreamEndCapture(ctx->stream, &graph);
      if (rc != QP_OK) {
        if (graph) (void)hipGraphDestroy(graph);
        return rc;
      }
      QP_HIP(ec);
      if (w->gexec) {
        (void)hipGraphExecDestroy(w->gexec);
        w->gexec = nullptr;
      }
      const hipError_t ei = hipGraphInstantiate(&w->gexec, graph, nullptr, nullptr, 0);
      (void)hipGraphDestroy(graph);
      QP_HIP(ei);
      w->gkey = key;
      QP_CHECK(replay());
      done = true;
    } else {
      w->gpending = key;
    }
  }
  if (!done) QP_CHECK(cheby_step_launches(w, op, psi, a, n_coeffs, beta, c, phase, check_normalization != 0));
  ctx->stats.n_cheby_steps++;
  if (check_normalization && nterms >= 2) {
    std::vector<double> h((size_t)3 * nterms);
    QP_HIP(hipMemcpyAsync(h.data(), w->chk_out, h.size() * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    QP_HIP(hipStreamSynchronize(ctx->stream));
    for (int m = 2; m <= nterms; ++m) {
      const double* t = &h[3 * (m - 1)];
      const double map_norm = std::hypot(t[0], t[1]) / (2 * t[2]);   // :195
      if (!(map_norm <= 1.0 + limit))
        return qp::fail(QP_E_NORMALIZATION, "Incorrect normalization (E_min=%g, Delta=%g)", E_min, Delta);
    }
  }
  return QP_OK;
  QP_CATCH
}

// ---------------------------------------------------------------------------
// batched states (BASELINE configs[4]): panel X[i*b + s]
// ---------------------------------------------------------------------------
int qp_cheby_step_batched(qp_cheby* w, qp_operator* op, qp_state* psi, int batch, const double* a, int n_coeffs,
                          double Delta, double E_min, double dt, double wrk_dt) {
  QP_TRY
  if (!w || !op || !psi || !a || batch < 1) return qp::fail(QP_E_BAD_ARG, "qp_cheby_step_batched: bad arguments");
  const int64_t n = op->A.nrows;
  if (op->A.nrows != op->A.ncols || psi->n != n * batch || w->n != psi->n)
    return qp::fail(QP_E_BAD_ARG, "qp_cheby_step_batched: shape mismatch (operator %lld, batch %d, panel %lld)",
                    (long long)n, batch, (long long)psi->n);
  {
    const double x = std::fabs(dt), y = std::fabs(wrk_dt);
    if (!(std::fabs(x - y) <= 1.4901161193847656e-08 * std::max(x, y)))
      return qp::fail(QP_E_DT_MISMATCH, "wrk was initialized for dt=%g, not dt=abs(%g)", wrk_dt, dt);
  }
  if (n_coeffs < 2) return qp::fail(QP_E_TOO_FEW_COEFFS, "Need at least 2 Chebychev coefficients");
  if (!(Delta > 0)) return qp::fail(QP_E_BAD_ARG, "Delta must be positive");
  qp_ctx* ctx = op->ctx;
  QP_CHECK(use(ctx));
  // a dense operator (QP_FMT_DENSE): H X on the fp64 matrix cores (csrc/kernels_dense.hip), straight from the operator's own
  // value array; knob dense_panel_mfma 0 sends it through the sparse panel kernels like any CSR operator (A/B)
  const bool dense = op->A.format == QP_FMT_DENSE && ctx->tun.dense_panel_mfma != 0 && (op->A.vals || op->A.vals_r);
  if (!dense) QP_CHECK(operator_csr_mirror(op));
  // kernel choice (knob spmm_rows): the wave-per-row kernel for wide panels, else the state-tiled kernel
  const bool rows_kernel = !dense && qp::spmm_uses_rows_kernel(ctx->tun, batch);
  const int32_t* order = nullptr;   // row walk of the wave-per-row kernel
  const qp::SpmmTiles* tiles = nullptr;   // LDS-staged tiles of a lattice operator's interior rows (knob spmm_rw -1, the default)
  if (rows_kernel && ctx->tun.spmm_rw < 0) QP_CHECK(operator_spmm_tiles(op, &tiles));
  if (rows_kernel && !tiles) QP_CHECK(operator_spmm_order(op, batch, &order));

  const double beta = (Delta / 2) + E_min;
  cplx c = (dt > 0) ? cplx(0, -2.0) / Delta : cplx(0, 2.0) / Delta;
  const cplx phase = std::exp(cplx(0, -1) * beta * dt);
  const int nterms = n_coeffs - 1;
  double2* P = psi->d;
  double2* B = w->bufA;
  double2* ACC = w->acc;
  double2* result = nullptr;
  std::vector<qp_acc_defer> sched((size_t)nterms);
  acc_schedule(a, n_coeffs, ctx->tun.acc_defer != 0, sched.data());
  bool updated = false;
  for (int m = 1; m <= nterms; ++m) {   // same buffer rotation as qp_cheby_step, element = (row, state)
    const bool last = (m == nterms);
    qp::ChebyEpi e;
    const double2* x;
    if (m == 1) {
      x = P;
      e.v0 = nullptr;
      e.vout = last ? nullptr : B;
      e.acc_in = nullptr;
      e.acc_out = ACC;
      result = ACC;
    } else {
      double2* xb = (m % 2 == 0) ? B : P;
      double2* ob = (m % 2 == 0) ? P : B;
      x = xb;
      e.v0 = ob;
      e.vout = last ? nullptr : ob;
      e.acc_in = updated ? ACC : nullptr;
      e.acc_out = (last && xb == B) ? P : ACC;
      result = e.acc_out;
    }
    e.a_prev = updated ? 0.0 : a[0];
    set_defer(e, &sched[(size_t)m - 1]);
    if (sched[(size_t)m - 1].skip) {
      e.acc_in = nullptr;
      e.acc_out = nullptr;
    } else {
      updated = true;
    }
    e.xloc = x;
    e.c = d2(c);
    e.beta = beta;
    e.a = a[m];
    e.phase = d2(phase);
    e.apply_phase = last ? 1 : 0;
    e.check_partials = nullptr;
    bool walked = false;
    if (dense) {
      QP_CHECK(qp::launch_dense_zgemm_cheby(ctx->stream, op->A, x, batch, e, &ctx->stats));
      walked = true;
    }
    if (!walked && tiles) {
      QP_CHECK(qp::launch_spmm_tile_cheby(ctx->stream, op->m_rowptr, op->m_cols, op->m_vals, x, n, op->A.nnz, batch, e, ctx->tun,
                                          *tiles, &ctx->stats));
      walked = true;
    }
    if (!walked)
      QP_CHECK(qp::launch_spmm_cheby(ctx->stream, op->m_rowptr, op->m_cols, op->m_vals, x, n, op->A.nnz, batch, e,
                                     ctx->tun, rows_kernel, order, &ctx->stats));
    if (m == 1) c *= 2.0;
  }
  if (result != P) QP_HIP(hipMemcpyAsync(P, result, (size_t)psi->n * sizeof(double2), hipMemcpyDeviceToDevice, ctx->stream));
  ctx->stats.n_cheby_steps++;
  return QP_OK;
  QP_CATCH
}

// ---------------------------------------------------------------------------
// boundary / interior split of one fused term (overlap of the multi-GPU exchange)
// ---------------------------------------------------------------------------
int qp_split_create(qp_operator* op, const int64_t* send_rows, int64_t nsend, qp_split** out) {
  QP_TRY
  if (!op || !out || nsend < 0 || (nsend > 0 && !send_rows)) return qp::fail(QP_E_BAD_ARG, "qp_split_create: bad arguments");
  const DevMatrix& A = op->A;
  if (A.format != QP_FMT_RBCSR && A.format != QP_FMT_HRB)
    return qp::fail(QP_E_BAD_ARG, "qp_split_create needs a row-block device format (got %d)", A.format);
  QP_CHECK(use(op->ctx));
  const auto& ur = op->u_rowptr;
  const auto& uc = op->u_col;
  std::vector<char> is_boundary((size_t)A.nblocks, 0);
  std::vector<int32_t> slot_of_row((size_t)A.nrows, -1);
  for (int64_t i = 0; i < nsend; ++i) {
    const int64_t r = send_rows[i];
    if (r < 0 || r >= A.nrows) return qp::fail(QP_E_BAD_ARG, "send row %lld out of range", (long long)r);
    if (slot_of_row[r] >= 0) return qp::fail(QP_E_BAD_ARG, "send row %lld listed twice", (long long)r);
    slot_of_row[r] = (int32_t)i;
    is_boundary[r / kRB] = 1;
  }
  for (int64_t r = 0; r < A.nrows; ++r)   // rows that read a ghost column must wait for the exchange
    if (ur[r + 1] > ur[r] && uc[ur[r + 1] - 1] >= A.nrows) is_boundary[r / kRB] = 1;
  // interior blocks that exchange data with boundary rows inside the local block (they gather
  // from boundary rows, or boundary rows gather from them) are listed last: only they have to
  // wait for the boundary launch of the previous term
  std::vector<char> adjacent((size_t)A.nblocks, 0);
  for (int64_t r = 0; r < A.nrows; ++r) {
    const bool rb_ = is_boundary[r / kRB];
    for (int64_t p = ur[r]; p < ur[r + 1]; ++p) {
      const int64_t c = uc[p];
      if (c >= A.nrows) continue;
      const bool cb = is_boundary[c / kRB];
      if (rb_ && !cb) adjacent[c / kRB] = 1;
      if (!rb_ && cb) adjacent[r / kRB] = 1;
    }
  }
  std::vector<int32_t> bb, bi, bi_adj;
  for (int64_t b = 0; b < A.nblocks; ++b) {
    if (is_boundary[b]) bb.push_back((int32_t)b);
    else if (adjacent[b]) bi_adj.push_back((int32_t)b);
    else bi.push_back((int32_t)b);
  }
  const unsigned wait_from_wg = (unsigned)(bi.size() / (qp::kThreads / 64));
  bi.insert(bi.end(), bi_adj.begin(), bi_adj.end());
  std::vector<int32_t> mirror(bb.size() * kRB + 1, -1);
  for (size_t k = 0; k < bb.size(); ++k)
    for (int l = 0; l < kRB; ++l) {
      const int64_t r = (int64_t)bb[k] * kRB + l;
      if (r < A.nrows) mirror[k * kRB + l] = slot_of_row[r];
    }
  auto sp = std::make_unique<qp_split>();
  sp->op = op;
  sp->device = op->ctx->device;
  sp->n_boundary = (int64_t)bb.size();
  sp->n_interior = (int64_t)bi.size();
  sp->nsend = nsend;
  QP_CHECK(dev_alloc(&sp->bmap_boundary, bb.size()));
  QP_CHECK(dev_alloc(&sp->bmap_interior, bi.size()));
  QP_CHECK(dev_alloc(&sp->mirror, mirror.size()));
  if (!bb.empty()) QP_HIP(hipMemcpy(sp->bmap_boundary, bb.data(), bb.size() * sizeof(int32_t), hipMemcpyHostToDevice));
  if (!bi.empty()) QP_HIP(hipMemcpy(sp->bmap_interior, bi.data(), bi.size() * sizeof(int32_t), hipMemcpyHostToDevice));
  QP_HIP(hipMemcpy(sp->mirror, mirror.data(), mirror.size() * sizeof(int32_t), hipMemcpyHostToDevice));
  QP_HIP(hipEventCreateWithFlags(&sp->ev_b, hipEventDisableTiming));
  QP_HIP(hipEventCreateWithFlags(&sp->ev_i, hipEventDisableTiming));
  QP_CHECK(dev_alloc(&sp->counter, 2));
  QP_HIP(hipMemset(sp->counter, 0, 2 * sizeof(unsigned)));
  QP_HIP(hipHostMalloc((void**)&sp->timeout_host, sizeof(unsigned), hipHostMallocMapped));
  *sp->timeout_host = 0;
  QP_HIP(hipHostGetDevicePointer((void**)&sp->timeout_dev, sp->timeout_host, 0));
  // Interior as a strip walk.  The boundary blocks of a row-partitioned lattice are a prefix and a suffix of the local
  // rows; the walk takes the interior blocks from which no walked row block, with its ring of +-K strip steps and its
  // one-block halo, reaches a boundary block: [max(W0, lo + K S + 1), min(R1, hi - K S - 1)).
  sp->walk = qp::WalkPlan();
  if (op->A.walk && op->A.walk->valid && op->A.format == QP_FMT_HRB) {
    const qp::WalkPlan& P = *op->A.walk;
    int64_t lo = 0, hi = A.nblocks;
    while (lo < A.nblocks && is_boundary[lo]) ++lo;
    while (hi > lo && is_boundary[hi - 1]) --hi;
    bool contiguous = true;
    for (int64_t b = lo; b < hi && contiguous; ++b) contiguous = !is_boundary[b];
    const int64_t reach = (std::max<int64_t>((int64_t)P.K * P.g + P.fd, P.glong) + kRB - 1) / kRB + 1;
    const int64_t w0 = std::max(P.W0, lo + reach), r1 = std::min(P.R1, hi - reach);
    if (contiguous && r1 - w0 >= 8) {
      std::vector<int32_t> edge;
      for (int32_t b : bi)
        if (b < w0 || b >= r1) edge.push_back(b);
      qp::WalkPlan W = P;
      W.W0 = w0;
      W.R1 = r1;
      W.edge_map = nullptr;
      W.n_edge = (int64_t)edge.size();
      QP_CHECK(dev_alloc(&W.edge_map, std::max<size_t>(edge.size(), 1)));
      if (!edge.empty()) QP_HIP(hipMemcpy(W.edge_map, edge.data(), edge.size() * sizeof(int32_t), hipMemcpyHostToDevice));
      sp->walk = W;
    }
  }
  sp->wait_from_wg = wait_from_wg;
  {
    const unsigned total = (unsigned)((bi.size() + qp::kThreads / 64 - 1) / (qp::kThreads / 64));
    sp->n_waiting_wg = total > wait_from_wg ? total - wait_from_wg : 0;
  }
  *out = sp.release();
  return QP_OK;
  QP_CATCH
}

int qp_split_destroy(qp_split* sp) {
  QP_TRY
  if (!sp) return QP_OK;
  (void)hipSetDevice(sp->device);
  (void)hipDeviceSynchronize();
  if (sp->bmap_boundary) (void)hipFree(sp->bmap_boundary);
  if (sp->bmap_interior) (void)hipFree(sp->bmap_interior);
  if (sp->mirror) (void)hipFree(sp->mirror);
  if (sp->walk.edge_map) (void)hipFree(sp->walk.edge_map);
  if (sp->counter) (void)hipFree(sp->counter);
  if (sp->timeout_host) (void)hipHostFree(sp->timeout_host);
  if (sp->ev_b) (void)hipEventDestroy(sp->ev_b);
  if (sp->ev_i) (void)hipEventDestroy(sp->ev_i);
  delete sp;
  return QP_OK;
  QP_CATCH
}

int qp_split_info(const qp_split* sp, int64_t* n_boundary_blocks, int64_t* n_interior_blocks) {
  if (!sp) return qp::fail(QP_E_BAD_ARG, "split is NULL");
  if (n_boundary_blocks) *n_boundary_blocks = sp->n_boundary;
  if (n_interior_blocks) *n_interior_blocks = sp->n_interior;
  return QP_OK;
}

int qp_split_walk_info(const qp_split* sp, int64_t out[4]) {
  if (!sp || !out) return qp::fail(QP_E_BAD_ARG, "qp_split_walk_info: NULL argument");
  out[0] = sp->walk.valid ? 1 : 0;
  out[1] = sp->walk.valid ? sp->walk.W0 : 0;
  out[2] = sp->walk.valid ? sp->walk.R1 : 0;
  out[3] = sp->walk.valid ? sp->walk.n_edge : 0;
  return QP_OK;
}

/* synchronises the device; returns QP_E_INTERNAL if an in-launch wait ever timed out */
int qp_split_check(qp_split* sp) {
  QP_TRY
  if (!sp) return qp::fail(QP_E_BAD_ARG, "split is NULL");
  QP_HIP(hipSetDevice(sp->device));
  QP_HIP(hipDeviceSynchronize());
  return split_timed_out(sp);
  QP_CATCH
}

int qp_cheby_term_split(qp_operator* op, qp_split* sp, void* boundary_stream, int first, const qp_state* x,
                        int64_t xoff, const qp_state* v0, qp_state* vout, const qp_state* acc_in, qp_state* acc_out,
                        qp_state* slab, qp_c128 c, double beta, double a_prev, double a, qp_c128 phase,
                        const qp_acc_defer* defer) {
  QP_TRY
  const bool skip = defer && defer->skip;
  if (!op || !sp || sp->op != op || !boundary_stream || !x || (!acc_out && !skip))
    return qp::fail(QP_E_BAD_ARG, "qp_cheby_term_split: bad arguments");
  if (defer && !skip && (defer->n_defer < 0 || defer->n_defer > 2 || (defer->n_defer > 0 && !v0 && (defer->n_defer == 2 || !acc_in))))
    return qp::fail(QP_E_BAD_ARG, "qp_cheby_term_split: deferred accumulation needs v0");
  const int64_t nr = op->A.nrows;
  if (x->n != op->A.ncols || xoff < 0 || xoff + nr > x->n) return qp::fail(QP_E_BAD_ARG, "qp_cheby_term_split: x shape / offset mismatch");
  if ((v0 && v0->n != nr) || (vout && vout->n != nr) || (acc_in && acc_in->n != nr) || (acc_out && acc_out->n != nr))
    return qp::fail(QP_E_BAD_ARG, "qp_cheby_term_split: local vector length mismatch");
  if (slab && slab->n < sp->nsend) return qp::fail(QP_E_BAD_ARG, "qp_cheby_term_split: slab too small");
  qp_ctx* ctx = op->ctx;
  QP_CHECK(use(ctx));
  hipStream_t S_c = ctx->stream, S_x = (hipStream_t)boundary_stream;
  qp::ChebyEpi e;
  e.xloc = x->d + xoff;
  e.v0 = v0 ? v0->d : nullptr;
  e.vout = vout ? vout->d : nullptr;
  e.acc_in = (acc_in && !skip) ? acc_in->d : nullptr;
  e.acc_out = (acc_out && !skip) ? acc_out->d : nullptr;
  e.c = d2(c);
  e.beta = beta;
  e.a_prev = a_prev;
  e.a = a;
  e.phase = d2(phase);
  e.apply_phase = !(phase.re == 1.0 && phase.im == 0.0);
  e.check_partials = nullptr;
  set_defer(e, defer);
  qp::RowSet rb{sp->bmap_boundary, sp->n_boundary, false};
  qp::RowSet ri{sp->bmap_interior, sp->n_interior, true};
  if (sp->walk.valid) {   // lattice operator: the interior as a strip walk; CUs beyond the edge workgroups' stay free (knob) for
    ri.walk = &sp->walk;  // the boundary launch and the collective's kernel that run beside it
    ri.reserve_cu = qp::kWalkReserveCu;
  }
  QP_CHECK(split_timed_out(sp));
  // knob split_mode: 1 = in-launch counter hand-off, 0 = events on both streams, 2 (default) = the counter
  // where it is safe by construction: the polling workgroups hold their CU slots while the boundary launch
  // they wait for may sit behind a collective that depends on other ranks, so they must be few enough to
  // leave room for that launch and the collective's kernel on every CU (at most one per CU here)
  const int mode = op->ctx->tun.split_mode;
  const bool flag_mode = mode == 1 || (mode == 2 && sp->n_waiting_wg <= 256);
  if (first && flag_mode) {
    // the caller joined both streams: restart the signal counter (keeps it far from wrap)
    QP_HIP(hipMemsetAsync(sp->counter, 0, sizeof(unsigned), S_c));
    sp->signals_issued = 0;
    QP_HIP(hipEventRecord(sp->ev_i, S_c));
    QP_HIP(hipStreamWaitEvent(S_x, sp->ev_i, 0));
  }
  if (!first) {
    // boundary(m) overwrites rows that interior(m-1) gathered from, and vice versa.  The side
    // stream takes a queue-level event wait (its idle time is hidden); the main stream either
    // does the same (mode 0) or lets only the adjacent workgroups of the interior launch poll
    // the boundary launch's completion counter (mode 1: no idle gap between interior launches)
    QP_HIP(hipStreamWaitEvent(S_x, sp->ev_i, 0));
    if (!flag_mode) QP_HIP(hipStreamWaitEvent(S_c, sp->ev_b, 0));
  }
  if (flag_mode) {
    ri.sync.wait = sp->counter;
    ri.sync.wait_target = sp->signals_issued;      // every boundary workgroup launched so far
    ri.sync.wait_from_wg = sp->wait_from_wg;
    ri.sync.timeout_flag = sp->timeout_dev;
    ri.sync.spin_limit = 1u << std::min(std::max(op->ctx->tun.split_spin_log2, 4), 31);
#ifdef QP_DEVELOPER
    rb.sync.signal = (op->ctx->tun.split_dbg & 1) ? nullptr : sp->counter;   // (time-out test: the boundary launch does not signal)
#else
    rb.sync.signal = sp->counter;
#endif
    sp->signals_issued += (unsigned)((sp->n_boundary + qp::kThreads / 64 - 1) / (qp::kThreads / 64));
  }
  qp::ChebyEpi eb = e;
  if (slab && vout) {   // the slab carries the new term vector (what the next term gathers)
    eb.mirror = sp->mirror;
    eb.slab = slab->d;
  }
  if (sp->n_boundary > 0) QP_CHECK(qp::launch_spmv_cheby(S_x, op->A, x->d, eb, &ctx->stats, &rb));
  if (!flag_mode) QP_HIP(hipEventRecord(sp->ev_b, S_x));
  if (sp->n_interior > 0) QP_CHECK(qp::launch_spmv_cheby(S_c, op->A, x->d, e, &ctx->stats, &ri));
  QP_HIP(hipEventRecord(sp->ev_i, S_c));
  return QP_OK;
  QP_CATCH
}

// ---------------------------------------------------------------------------
// propagate step loop (src/propagate.jl:283-344)
// ---------------------------------------------------------------------------
// Small systems: the whole time grid in one persistent single-workgroup launch
// (kernels_small.hip: cheby_propagate_small_kernel).  Same arguments as qp_propagate, method 0.
static int propagate_cheby_small(qp_operator* op, qp_state* psi, const qp_prop_spec* spec, qp::SmallArgs a,
                                 const double* dts,
                                 const qp_c128* coeff_table, int ncoeffs, int nsteps, qp_operator* const* observables,
                                 int nobs, qp_c128* expvals_out, qp_c128* states_out) {
  qp_ctx* ctx = op->ctx;
  qp_cheby* w = spec->cheby;
  const int64_t n = psi->n;
  const size_t rows = (size_t)nsteps + 1;
  if (!w || !spec->a) return qp::fail(QP_E_BAD_ARG, "qp_propagate: NULL Chebychev workspace");
  if (op->A.nrows != op->A.ncols || n != op->A.nrows || w->n != n) return qp::fail(QP_E_BAD_ARG, "qp_propagate: shape mismatch");
  if (spec->n_coeffs < 2) return qp::fail(QP_E_TOO_FEW_COEFFS, "Need at least 2 Chebychev coefficients");
  if (!(spec->Delta > 0)) return qp::fail(QP_E_BAD_ARG, "Delta must be positive");
  for (int k = 0; k < nsteps; ++k) {
    const double x = std::fabs(dts[k]), y = std::fabs(spec->wrk_dt);   // src/cheby.jl:157
    if (!(std::fabs(x - y) <= 1.4901161193847656e-08 * std::max(x, y)))
      return qp::fail(QP_E_DT_MISMATCH, "wrk was initialized for dt=%g, not dt=abs(%g)", spec->wrk_dt, dts[k]);
    if ((dts[k] > 0) != (dts[0] > 0)) return qp::fail(QP_E_BAD_ARG, "qp_propagate: time steps change sign");
  }
  QP_CHECK(operator_csr_mirror(op));
  for (int o = 0; o < nobs; ++o) QP_CHECK(operator_csr_mirror(observables[o]));
  const double dt = dts[0];
  const double beta = (spec->Delta / 2) + spec->E_min;
  const cplx c = (dt > 0) ? cplx(0, -2.0) / spec->Delta : cplx(0, 2.0) / spec->Delta;

  a.n = n;
  a.nnz = op->A.nnz;
  a.rowptr = op->m_rowptr;
  a.cols = op->m_cols;
  a.map = op->m_map;
  a.planes = op->planes_dev;
  a.nops = op->nops;
  a.ncoeffs = ncoeffs;
  a.scale = d2(op->scale);
  a.nsteps = nsteps;
  a.n_coeffs = spec->n_coeffs;
  a.c = d2(c);
  a.beta = beta;
  a.phase = d2(std::exp(cplx(0, -1) * beta * dt));
  a.psi = psi->d;
  a.nobs = nobs;
  a.check = spec->check_normalization ? 1 : 0;
  a.limit = spec->limit;

  // one staging buffer: [table | a | obs descriptors | fail | expvals | work | states]
  std::vector<void*> owned;
  struct Free {
    std::vector<void*>& v;
    ~Free() {
      for (void* p : v) (void)hipFree(p);
    }
  } guard{owned};
  auto upload = [&](const void* src, size_t bytes, void** out) -> int {
    void* d = nullptr;
    QP_HIP(hipMalloc(&d, std::max<size_t>(bytes, 16)));
    owned.push_back(d);
    if (src && bytes) QP_HIP(hipMemcpyAsync(d, src, bytes, hipMemcpyHostToDevice, ctx->stream));
    *out = d;
    return QP_OK;
  };
  void* p = nullptr;
  QP_CHECK(upload(coeff_table, sizeof(qp_c128) * (size_t)nsteps * ncoeffs, &p));
  a.table = static_cast<const double2*>(p);
  QP_CHECK(upload(spec->a, sizeof(double) * (size_t)spec->n_coeffs, &p));
  a.a = static_cast<const double*>(p);
  std::vector<qp::SmallObs> hobs((size_t)nobs);
  for (int o = 0; o < nobs; ++o) hobs[o] = qp::SmallObs{observables[o]->m_rowptr, observables[o]->m_cols, observables[o]->m_vals};
  QP_CHECK(upload(hobs.data(), sizeof(qp::SmallObs) * (size_t)nobs, &p));
  a.obs = static_cast<const qp::SmallObs*>(p);
  QP_CHECK(upload(nullptr, sizeof(int) * 4, &p));
  a.fail = static_cast<int*>(p);
  QP_HIP(hipMemsetAsync(a.fail, 0, sizeof(int) * 4, ctx->stream));
  QP_CHECK(upload(nullptr, sizeof(double2) * rows * (size_t)nobs, &p));
  a.expvals = static_cast<double2*>(p);
  if (states_out) {
    QP_CHECK(upload(nullptr, sizeof(double2) * rows * (size_t)n, &p));
    a.states = static_cast<double2*>(p);
  }
  QP_CHECK(qp::launch_cheby_propagate_small(ctx->stream, a, &ctx->stats));
  ctx->stats.n_cheby_steps += nsteps;
  int fail[4] = {0, 0, 0, 0};
  QP_HIP(hipMemcpyAsync(fail, a.fail, sizeof(fail), hipMemcpyDeviceToHost, ctx->stream));
  if (nobs > 0)
    QP_HIP(hipMemcpyAsync(expvals_out, a.expvals, sizeof(double2) * rows * (size_t)nobs, hipMemcpyDeviceToHost, ctx->stream));
  if (states_out)
    QP_HIP(hipMemcpyAsync(states_out, a.states, sizeof(double2) * rows * (size_t)n, hipMemcpyDeviceToHost, ctx->stream));
  QP_HIP(hipStreamSynchronize(ctx->stream));
  // leave the operator as the step-by-step loop would: holding the last interval's values
  if (ncoeffs > 0) QP_CHECK(qp_operator_set_coeffs(op, coeff_table + (size_t)(nsteps - 1) * ncoeffs, ncoeffs));
  if (fail[0])
    return qp::fail(QP_E_NORMALIZATION, "Incorrect normalization (E_min=%g, Delta=%g) in step %d, term %d", spec->E_min,
                    spec->Delta, fail[1] + 1, fail[2] + 1);
  return QP_OK;
}

int qp_propagate(qp_operator* op, qp_state* psi, const qp_prop_spec* spec, const double* dts,
                 const qp_c128* coeff_table, int ncoeffs, int nsteps, qp_operator* const* observables, int nobs,
                 qp_c128* expvals_out, qp_c128* states_out) {
  QP_TRY
  if (!op || !psi || !spec || !dts || nsteps < 0 || nobs < 0 || (nobs > 0 && (!observables || !expvals_out)))
    return qp::fail(QP_E_BAD_ARG, "qp_propagate: bad arguments");
  if (ncoeffs != op->ncoeffs || (ncoeffs > 0 && nsteps > 0 && !coeff_table))
    return qp::fail(QP_E_BAD_ARG, "qp_propagate: expected %d coefficients per step", op->ncoeffs);
  if (spec->method != 0 && spec->method != 1) return qp::fail(QP_E_BAD_ARG, "qp_propagate: bad method");
  for (int o = 0; o < nobs; ++o)
    if (!observables[o] || observables[o]->A.nrows != psi->n || observables[o]->A.ncols != psi->n)
      return qp::fail(QP_E_BAD_ARG, "qp_propagate: observable %d has the wrong shape", o);
  qp_ctx* ctx = op->ctx;
  QP_CHECK(use(ctx));
  const int64_t n = psi->n;
  const size_t rows = (size_t)nsteps + 1;
  if (spec->method == 0 && nsteps > 0 && ctx->tun.small_nnz > 0 && op->A.nnz <= 2 * (int64_t)ctx->tun.small_nnz && op->nops <= 64) {
    qp::SmallArgs plan;
    int64_t maxrow = 0;
    for (int64_t r = 0; r < n; ++r) maxrow = std::max<int64_t>(maxrow, op->u_rowptr[r + 1] - op->u_rowptr[r]);
    // 16 register slots per lane where that is enough; otherwise 32 (the upper 16 values of a lane
    // live in LDS: 128 KB, which leaves room for vectors of up to 600 rows)
    bool ok = op->A.nnz <= ctx->tun.small_nnz && qp::small_plan(n, maxrow, &plan, qp::kSmallEpt);
    if (!ok && n <= 600) ok = qp::small_plan(n, maxrow, &plan, 2 * qp::kSmallEpt);
    if (ok)
      return propagate_cheby_small(op, psi, spec, plan, dts, coeff_table, ncoeffs, nsteps, observables, nobs,
                                   expvals_out, states_out);
  }
  // device staging, released at the end: observable partials and the state history
  double2* d_part = nullptr;
  double2* d_tmp = nullptr;
  double2* d_states = nullptr;
  struct Free {
    double2 *&a, *&b, *&c;
    ~Free() {
      if (a) (void)hipFree(a);
      if (b) (void)hipFree(b);
      if (c) (void)hipFree(c);
    }
  } guard{d_part, d_tmp, d_states};
  if (nobs > 0) {
    QP_CHECK(dev_alloc(&d_part, rows * nobs * kRedBlocks));
    QP_CHECK(dev_alloc(&d_tmp, (size_t)n));
  }
  if (states_out) QP_CHECK(dev_alloc(&d_states, rows * (size_t)n));
  auto record = [&](size_t row) -> int {
    for (int o = 0; o < nobs; ++o) {   // <psi|O|psi> = dot(psi, O psi)
      qp::PlainEpi e;
      e.y = d_tmp;
      e.alpha = make_double2(1.0, 0.0);
      e.beta = make_double2(0.0, 0.0);
      e.beta_zero = 1;
      QP_CHECK(qp::launch_spmv_plain(ctx->stream, observables[o]->A, psi->d, e, &ctx->stats));
      QP_CHECK(qp::launch_dot_partials(ctx->stream, psi->d, d_tmp, d_part + (row * nobs + o) * kRedBlocks, n, &ctx->stats));
    }
    if (d_states)
      QP_HIP(hipMemcpyAsync(d_states + row * (size_t)n, psi->d, (size_t)n * sizeof(double2), hipMemcpyDeviceToDevice, ctx->stream));
    return QP_OK;
  };
  QP_CHECK(record(0));
  for (int k = 0; k < nsteps; ++k) {
    if (ncoeffs > 0) QP_CHECK(qp_operator_set_coeffs(op, coeff_table + (size_t)k * ncoeffs, ncoeffs));
    if (spec->method == 0) {
      QP_CHECK(qp_cheby_step(spec->cheby, op, psi, spec->a, spec->n_coeffs, spec->Delta, spec->E_min, dts[k],
                             spec->wrk_dt, spec->limit, spec->check_normalization));
    } else {
      QP_CHECK(qp_newton_step(spec->newton, op, psi, dts[k], spec->func_id, spec->cb, spec->user, spec->norm_min,
                              spec->relerr, spec->max_restarts, nullptr));
    }
    QP_CHECK(record((size_t)k + 1));
  }
  if (nobs > 0) {
    std::vector<cplx> hp(rows * nobs * kRedBlocks);
    QP_HIP(hipMemcpyAsync(hp.data(), d_part, hp.size() * sizeof(double2), hipMemcpyDeviceToHost, ctx->stream));
    QP_HIP(hipStreamSynchronize(ctx->stream));
    for (size_t i = 0; i < rows * nobs; ++i) {
      const cplx v = sum_partials(reinterpret_cast<const double2*>(hp.data() + i * kRedBlocks));
      expvals_out[i] = qp_c128{v.real(), v.imag()};
    }
  }
  if (states_out) {
    QP_HIP(hipMemcpyAsync(states_out, d_states, rows * (size_t)n * sizeof(double2), hipMemcpyDeviceToHost, ctx->stream));
    QP_HIP(hipStreamSynchronize(ctx->stream));
  }
  return QP_OK;
  QP_CATCH
}

}  // extern "C"
