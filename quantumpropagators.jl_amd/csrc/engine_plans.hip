// Host index work that decides HOW an operator runs (no device arithmetic): the device format (choose_format), the lattice
// completion of open-boundary grids (lattice_fill), the strip-walk plan of a Hermitian-packed lattice (build_walk_plan: shape
// parser, run detection, positions), the column-blocked mirror of an operator with irregular columns (build_colblock), the batched path's row walk and LDS tiles
// (operator_spmm_order, operator_spmm_tiles; moved here in round 6).
// Split out of engine_core.hip in round 4; called from qp_operator_create / operator_build_device there.
#include <atomic>
#include <numeric>

#include "engine_host.h"

// Strip-walk plan of a Hermitian-packed lattice operator (device.h: WalkPlan; kernels_walk.hip).  Looks, in the union
// pattern itself, for the longest run of row blocks in which every row has the same list of column distances, checks that
// the list has the shape the walk needs -- [-K g .. -g] [-d_nn .. -d_1] [0]? [d_1 .. d_nn] [g .. K g], g a multiple of 64
// rows -- and that the upper values of the run's blocks lie at equal strides.  With one stencil on every row of the run the
// position of every value, and of every conj-transposed value, is a formula; how the column sections of those blocks are
// encoded does not matter (the walk reads none of them).  The remaining blocks are listed for the per-block path.
// Index work only: host, exact.
// The walk's stencil shape read off one row's sorted list of column distances (kernels_walk.hip):
//   [-L_1]? [-L_0]? [-K g .. -g] [-d_nn .. -d_1] [0]? [d_1 .. d_nn] [g .. K g] [L_0]? [L_1]?
// near distances of at most kWalkHalo rows, far distances the multiples of one strip step g >= 64 rows, optionally one or two
// more pairs +-L beyond them (xl: the plane distance of a three-dimensional grid, and its double for a fourth-order stencil;
// the volume distance of a four-dimensional one).  false: not a shape the walk has a kernel for.
struct WalkShape {
  int nn = 0, K = 0, z0 = 0, xl = 0, fd = 0;   // fd = 1: diagonal far neighbours g - 1, g, g + 1 (nine-point stencils)
  int64_t g = 0, glong = 0, glong1 = 0;   // glong: the longest distance; glong1: the shorter long one when there are two
  int near[8] = {0, 0, 0, 0, 0, 0, 0, 0};
};
// 0 = a shape the walk has a kernel for; else the QP_WALK_* code of what broke it (include/qprop.h), with the offending
// numbers in *why
static int walk_shape_reason(const std::vector<int64_t>& dl, WalkShape& w, std::string* why) {
  auto say = [&](const char* fmt, long long a = 0, long long b = 0) {
    if (why) {
      char buf[160];
      std::snprintf(buf, sizeof buf, fmt, a, b);
      *why = buf;
    }
  };
  const int z = (int)dl.size();
  if (z < 3 || z > 19) {
    say("%lld entries per row (the walk takes 3 to 19)", z);
    return QP_WALK_ROW_LENGTH;
  }
  for (int k = 0; k < z; ++k)
    if (dl[(size_t)k] != -dl[(size_t)(z - 1 - k)]) {   // mirror images of each other
      say("column distance %lld has no mirror image %lld in the row", dl[(size_t)k], -dl[(size_t)k]);
      return QP_WALK_NOT_MIRRORED;
    }
  int nbig = 0;
  while (nbig < z && dl[(size_t)nbig] <= -(int64_t)kRB) ++nbig;
  if (nbig < 1) {
    say("no column distance of at least 64 rows: a band of half-width %lld has no strip step", -dl[0]);
    return QP_WALK_NO_FAR;
  }
  // near part first: it decides between "too many / too far near" and the far diagnoses below
  {
    int k = nbig, nnear = 0;
    while (k < z && dl[(size_t)k] < 0) ++k, ++nnear;
    if (nnear >= 1 && -dl[(size_t)nbig] > qp::kWalkHalo) {
      say("near column distance %lld exceeds the %lld-row halo of the walk's window", -dl[(size_t)nbig], qp::kWalkHalo);
      return QP_WALK_NEAR_TOO_FAR;
    }
    if (nnear > 4) {
      say("%lld near column distances per side (the walk takes 1 to 4)", nnear);
      return QP_WALK_TOO_MANY_NEAR;
    }
    if (nnear < 1) {
      say("no near column distance (the walk's kernels take 1 to 4 within %lld rows)", qp::kWalkHalo);
      return QP_WALK_NO_NEAR;
    }
  }
  if (nbig > 6) {
    say("%lld far column distances per side (the walk takes up to 4 multiples of one stride, or up to 2 plus one or two long pairs)", nbig);
    return QP_WALK_TOO_MANY_FAR;
  }
  w = WalkShape();
  w.g = -dl[(size_t)(nbig - 1)];
  auto multiples = [&](int first, int K) {
    for (int m = 1; m <= K; ++m)
      if (dl[(size_t)(first + K - m)] != -(int64_t)m * w.g) return false;
    return true;
  };
  if (nbig == 3 && dl[0] + 1 == dl[1] && dl[1] + 1 == dl[2] && -dl[2] >= (int64_t)kRB) {
    // -(g + 1), -g, -(g - 1): one strip step with its two diagonal neighbours (the nine-point stencil of a two-dimensional grid)
    w.g = -dl[1];
    w.K = 1;
    w.fd = 1;
  } else if (nbig == 4 && dl[1] + 1 == dl[2] && dl[2] + 1 == dl[3] && -dl[3] >= (int64_t)kRB && -dl[0] > -dl[1] + qp::kWalkHalo) {
    // ... and one long pair beyond them: layers of such planes (+-nx ny)
    w.g = -dl[2];
    w.K = 1;
    w.fd = 1;
    w.xl = 1;
    w.glong = -dl[0];
  } else if (nbig <= 4 && multiples(0, nbig)) {
    w.K = nbig;
  } else if (nbig >= 2 && nbig <= 5 && multiples(1, nbig - 1) && -dl[0] > (int64_t)(nbig - 1) * w.g) {
    w.K = nbig - 1;
    w.xl = 1;
    w.glong = -dl[0];
  } else if (nbig >= 3 && multiples(2, nbig - 2) && -dl[1] > (int64_t)(nbig - 2) * w.g) {   // (sorted: -dl[0] > -dl[1])
    w.K = nbig - 2;
    w.xl = 2;
    w.glong = -dl[0];
    w.glong1 = -dl[1];
  } else {
    long long bad = 0;
    for (int i = 0; i < nbig; ++i)
      if ((-dl[(size_t)i]) % w.g != 0) bad = -dl[(size_t)i];
    if (bad) say("far column distance %lld is no multiple of the strip step %lld (two incommensurate strides)", bad, w.g);
    else say("far column distances are multiples of %lld but not the consecutive ones 1 .. K, K <= 4 (largest: %lld)", w.g, -dl[0]);
    return bad ? QP_WALK_INCOMMENSURATE : QP_WALK_TOO_MANY_FAR;
  }
  int k = nbig;
  while (k < z && dl[(size_t)k] < 0) ++k, ++w.nn;
  for (int i = 0; i < w.nn; ++i) w.near[i] = (int)(-dl[(size_t)(nbig + w.nn - 1 - i)]);
  w.z0 = (k < z && dl[(size_t)k] == 0) ? 1 : 0;
  if (z != 2 * (w.nn + w.K * (1 + 2 * w.fd) + w.xl) + w.z0) {
    say("row of %lld entries does not split into diagonal + near + far parts", z);
    return QP_WALK_NOT_MIRRORED;
  }
  if (w.xl >= 1) {
    // a "long" distance right beside the ring's far reach is a DIAGONAL neighbour (nine-point stencil: g - 1, g, g + 1 read as
    // stride g - 1 plus two long pairs): its operands are lane shifts of the ring's elements, which the walk does not keep in
    // a window -- loading them directly makes the walk no faster than the per-block kernel (N = 2^22: 117.9 vs 113.9 us)
    const int64_t first = (w.xl == 2) ? w.glong1 : w.glong;
    if (first - (int64_t)w.K * w.g <= qp::kWalkHalo) {
      say("distance %lld is a diagonal neighbour of the far reach %lld (windows on the ring's far steps are not built; the per-block kernel is as fast)",
          first, (long long)w.K * w.g);
      return QP_WALK_NO_KERNEL;
    }
  }
  if (w.fd && !qp::walk_shape_supported(w.nn, w.K, w.z0, w.xl, w.fd)) {
    say("no kernel instance for %lld near distances beside diagonal far neighbours (they come with at most 2 near and one long pair)", w.nn);
    return QP_WALK_NO_KERNEL;
  }
  if (!w.fd && !qp::walk_shape_supported(w.nn, w.K, w.z0, w.xl)) {
    say("no kernel instance for %lld near and %lld far distances with long pairs (they come with at most 2 near, 2 far)", w.nn, w.K);
    return QP_WALK_NO_KERNEL;
  }
  return QP_WALK_OK;
}
static bool parse_walk_shape(const std::vector<int64_t>& dl, WalkShape& w) { return walk_shape_reason(dl, w, nullptr) == QP_WALK_OK; }

int build_walk_plan(qp_operator* op) {
  qp::WalkPlan& P = op->walk;
  if (P.edge_map) (void)hipFree(P.edge_map);
  P = qp::WalkPlan();
  if (op->walk2.edge_map) (void)hipFree(op->walk2.edge_map);
  op->walk2 = qp::WalkPlan();
  DevMatrix& A = op->A;
  A.walk = nullptr;
  const HostLayout& Lh = op->layout;
  const auto& ur = op->u_rowptr;
  const auto& uc = op->u_col;
  const int64_t nb = A.nblocks;
  auto why = [&](int code, const char* fmt, long long a = 0, long long b = 0) {
    char buf[200];
    std::snprintf(buf, sizeof buf, fmt, a, b);
    op->walk_reason = code;
    op->walk_reason_text = buf;
    return QP_OK;
  };
  op->walk_reason = QP_WALK_OK;
  op->walk_reason_text.clear();
  if (A.format != QP_FMT_HRB) return why(QP_WALK_NOT_PACKED, "device format %lld is not the Hermitian-packed one", A.format);
  if (nb < 8 || A.ncols < A.nrows) return why(QP_WALK_TOO_FEW_BLOCKS, "%lld row blocks (or fewer columns than rows)", nb);   // (more columns than rows: the halo slabs of a row-partitioned operator)
  const int64_t nfull = A.nrows / kRB;   // (a partly filled last block never belongs to the run)
  auto same_row = [&](int64_t r, int64_t ref) {   // same distances as row `ref`?
    const int64_t len = ur[ref + 1] - ur[ref];
    if (ur[r + 1] - ur[r] != len) return false;
    const int32_t* a = uc.data() + ur[r];
    const int32_t* c = uc.data() + ur[ref];
    const int64_t shift = r - ref;
    for (int64_t k = 0; k < len; ++k)
      if ((int64_t)a[k] - (int64_t)c[k] != shift) return false;
    return true;
  };
  // Longest run of row blocks whose rows all carry ONE list of column distances.  "Same distances" is transitive, so a block
  // belongs to the run of the block before it iff (a) its own 64 rows agree with one another and (b) its first row agrees with the
  // first row of the block before: both per block, on a few host threads; the runs are then read off the two flag arrays.
  std::vector<char> uniform((size_t)std::max<int64_t>(nfull, 1), 0), joins((size_t)std::max<int64_t>(nfull, 1), 0);
  parallel_rows(nfull, [&](int64_t b0, int64_t b1) {
    for (int64_t b = b0; b < b1; ++b) {
      const int64_t ref = b * kRB;
      const int64_t len = ur[ref + 1] - ur[ref];
      bool ok = len >= 3 && len <= 19;
      for (int64_t r = ref + 1; r < ref + kRB && ok; ++r) ok = same_row(r, ref);
      uniform[(size_t)b] = ok ? 1 : 0;
      joins[(size_t)b] = (ok && b > 0 && same_row(ref, ref - kRB)) ? 1 : 0;
    }
  }, 1024);
  int64_t best0 = 0, best1 = 0;
  for (int64_t b = 0; b < nfull;) {
    if (!uniform[(size_t)b]) {
      ++b;
      continue;
    }
    int64_t e = b + 1;
    while (e < nfull && uniform[(size_t)e] && joins[(size_t)e]) ++e;
    if (e - b > best1 - best0) best0 = b, best1 = e;
    b = e;
  }
  if (best1 - best0 < 8)
    return why(QP_WALK_NO_UNIFORM_RUN, "the longest run of row blocks whose rows all carry one list of column distances is %lld blocks (of %lld): not a lattice",
               best1 - best0, nb);
  const int64_t R0 = best0, R1 = best1, rref = R0 * kRB;
  const int64_t z = ur[rref + 1] - ur[rref];
  std::vector<int64_t> dl((size_t)z);
  for (int64_t k = 0; k < z; ++k) dl[(size_t)k] = (int64_t)uc[ur[rref] + k] - rref;
  WalkShape ws;
  {
    std::string text;
    const int code = walk_shape_reason(dl, ws, &text);
    if (code != QP_WALK_OK) {
      op->walk_reason = code;
      op->walk_reason_text = text;
      return QP_OK;
    }
  }
  const int nn = ws.nn, K = ws.K, z0 = ws.z0, xl = ws.xl, fd = ws.fd, KF = ws.K * (1 + 2 * ws.fd);
  const int64_t g = ws.g;
  for (int i = 0; i < nn; ++i) P.near[i] = ws.near[i];
  const int S = (int)((g + kRB - 1) / kRB);
  // first block whose rows find their history (K g rows back, L for the long pair) inside the run
  const int64_t W0 = R0 + (std::max<int64_t>((int64_t)K * g + fd, ws.glong) + kRB - 1) / kRB;
  if (R1 - W0 < 8) return why(QP_WALK_TOO_FEW_BLOCKS, "%lld walkable row blocks after the first %lld of the run (whose history lies outside it)", R1 - W0, W0 - R0);
  // the upper section of every block of the run: z0 + nn + K entries per row, padded to a multiple of four, at equal strides
  const int64_t wu = ((z0 + nn + KF + xl + 3) / 4) * 4;
  const int64_t U0 = Lh.bptr[R0], ustride = wu * kRB;
  for (int64_t b = R0; b <= R1; ++b)
    if (Lh.bptr[b] != U0 + (b - R0) * ustride) return why(QP_WALK_LAYOUT, "upper sections of the run are not at equal strides (block %lld)", b);
  for (int64_t r = rref; r < R1 * kRB; r += kRB)
    if (Lh.nlow[r] != nn + KF + xl) return why(QP_WALK_LAYOUT, "row %lld has %lld lower entries", r, Lh.nlow[r]);
  if (U0 + (R1 - R0) * ustride >= (int64_t)INT32_MAX) return why(QP_WALK_LAYOUT, "value positions beyond 2^31");
  std::vector<int32_t> edge;
  for (int64_t b = 0; b < W0; ++b) edge.push_back((int32_t)b);
  for (int64_t b = R1; b < nb; ++b) edge.push_back((int32_t)b);
  P.nn = nn;
  P.K = K;
  P.z0 = z0;
  P.S = S;
  P.g = g;
  P.xl = xl;
  P.glong = ws.glong;
  P.glong1 = ws.glong1;
  P.fd = fd;
  P.R0 = R0;
  P.R1 = R1;
  P.W0 = W0;
  P.U0 = U0;
  P.ustride = (int)ustride;
  P.n_edge = (int64_t)edge.size();
  QP_CHECK(dev_alloc(&P.edge_map, std::max<size_t>(edge.size(), 1)));
  if (!edge.empty()) QP_HIP(hipMemcpy(P.edge_map, edge.data(), edge.size() * sizeof(int32_t), hipMemcpyHostToDevice));
  P.valid = 1;
  A.walk = &P;
  // the two-term walk's plan: the region in which phase Z (term m + 1 of block t - K) finds y of the K blocks on either side formed
  // by the same walk -- [W0 + K S, R1 - K S) -- and, as its edge list, every block outside it (they take two per-block launches)
  if (!xl && !fd && g % kRB == 0 && qp::walk2_shape_supported(nn, K, z0) && (R1 - W0) - 2 * (int64_t)K * S >= 8 * (int64_t)S) {
    qp::WalkPlan& Q = op->walk2;
    Q = P;
    Q.edge_map = nullptr;
    Q.W0 = W0 + (int64_t)K * S;
    Q.R1 = R1 - (int64_t)K * S;
    std::vector<int32_t> edge2;
    for (int64_t b = 0; b < Q.W0; ++b) edge2.push_back((int32_t)b);
    for (int64_t b = Q.R1; b < nb; ++b) edge2.push_back((int32_t)b);
    Q.n_edge = (int64_t)edge2.size();
    QP_CHECK(dev_alloc(&Q.edge_map, std::max<size_t>(edge2.size(), 1)));
    QP_HIP(hipMemcpy(Q.edge_map, edge2.data(), edge2.size() * sizeof(int32_t), hipMemcpyHostToDevice));
    Q.valid = 1;
  }
  return QP_OK;
}

// position of every union-CSR entry in the operator's value array (-(position) - 1: the complex conjugate of that value --
// the lower entries of a Hermitian-packed operator)
void csr_value_map(const qp_operator* op, std::vector<int64_t>& map) {
  const DevMatrix& A = op->A;
  const auto& ur = op->u_rowptr;
  const auto& uc = op->u_col;
  map.assign((size_t)std::max<int64_t>(A.nnz, 1), 0);
  parallel_rows(A.nrows, [&](int64_t r_begin, int64_t r_end) {
    for (int64_t r = r_begin; r < r_end; ++r) {
      const int64_t nl = (A.format == QP_FMT_HRB) ? op->layout.nlow[r] : 0;
      for (int64_t k = 0; k < ur[r + 1] - ur[r]; ++k) {
        int64_t m;
        if (qp::csr_layout(A.format)) {
          m = ur[r] + k;
        } else if (k >= nl) {
          m = rb_val_pos(op->layout.bptr, r, k - nl);
        } else {
          const int64_t c = uc[ur[r] + k];
          const int32_t* b = uc.data() + ur[c];
          const int32_t* e = uc.data() + ur[c + 1];
          const int64_t kk = (std::lower_bound(b, e, (int32_t)r) - b) - op->layout.nlow[c];
          m = -rb_val_pos(op->layout.bptr, c, kk) - 1;
        }
        map[ur[r] + k] = m;
      }
    }
  });
}

// Column-blocked mirror (device.h: ColBlockPlan; kernels_colblock.hip) for an operator whose gathers are irregular.
// Decision (knob colblock = 1): plain row blocks or CSR (a lattice is Hermitian-packed and walked; a dense operator has
// its own kernels), at least 2^20 (kCbMinLog2N) columns (below, the vector sits in the L2 as it is), at most 64 column blocks
// and kCbMaxTilesPerWave tiles per resident wavefront, and -- sampled over the row blocks -- more than half of the gathers
// of a wavefront's load pulling a 128-byte line of their own (a band or a lattice shares each line among 8 lanes: 0.125).
int build_colblock(qp_operator* op) {
  qp_ctx* ctx = op->ctx;
  const qp::Tuning& tun = ctx->tun;
  DevMatrix& A = op->A;
  op->A.cb = nullptr;
  if (tun.colblock == 0 || (A.format != QP_FMT_RBCSR && A.format != QP_FMT_CSR)) return QP_OK;
  const auto& ur = op->u_rowptr;
  const auto& uc = op->u_col;
  const int64_t nrows = A.nrows, ncols = A.ncols, nnz = A.nnz;
  if (nrows < 64 || nnz < 1 || nnz >= (int64_t)INT32_MAX || ncols >= ((int64_t)1 << 32)) return QP_OK;
  // columns per block: about sixteen blocks (measured, 16 random columns per row: N = 2^20 171 / 178 / 195 us per term with
  // 2^16 / 2^17 / 2^18 columns per block, 2^21 416 / 353 / 460, 2^22 1125 / 890 / 829 -- profiles/r04/colblock.txt), a block
  // never larger than half an XCD's L2 share would like (2^18 elements = 4 MB is the whole L2; taken only from 2^22 columns on)
  int log2w = tun.cb_log2w;
  if (log2w <= 0) {
    int lg = 0;
    while (((int64_t)1 << lg) < ncols) ++lg;
    log2w = std::max(16, std::min(lg - 4, 18));
  }
  log2w = std::max(6, std::min(log2w, 24));
  const int64_t W = (int64_t)1 << log2w;
  const int64_t P = (ncols + W - 1) / W;
  if (tun.colblock == 1 && (ncols < ((int64_t)1 << qp::kCbMinLog2N) || P < 2)) return QP_OK;
  if (P > 256) return QP_OK;
  // irregularity: distinct 128-byte lines among the k-th gathers of a 64-row block, over the entries sampled
  {
    const int64_t nblocks = nrows / kRB;
    const int64_t bstride = std::max<int64_t>(1, nblocks / 1024);
    int64_t gathers = 0, lines = 0, local = 0;
    std::vector<int64_t> ln;
    for (int64_t b = 0; b < nblocks; b += bstride) {
      int64_t wmax = 0;
      for (int64_t r = b * kRB; r < (b + 1) * kRB; ++r) wmax = std::max(wmax, ur[r + 1] - ur[r]);
      for (int64_t k = 0; k < wmax; ++k) {
        ln.clear();
        for (int64_t r = b * kRB; r < (b + 1) * kRB; ++r)
          if (k < ur[r + 1] - ur[r]) {
            const int64_t c = uc[ur[r] + k];
            ln.push_back(c >> 3);
            if (std::llabs(c - r) <= ((int64_t)1 << 16)) ++local;
          }
        std::sort(ln.begin(), ln.end());
        gathers += (int64_t)ln.size();
        lines += (int64_t)(std::unique(ln.begin(), ln.end()) - ln.begin());
      }
    }
    op->cb_line_share = gathers > 0 ? (double)lines / (double)gathers : 0.0;
    if (tun.colblock == 1 && op->cb_line_share <= 0.5) return QP_OK;
    // ... and the gathers must really leave the L2: columns drawn near the row (within 2^16 elements = 1 MB either side)
    // stay in the XCD's L2 as the rows stream by -- columns random inside 4096-row windows: 101 us per term on the
    // row-block kernel, 186 through the mirror
    if (tun.colblock == 1 && 4 * local >= 3 * gathers) return QP_OK;
  }
  // tile height: 128 rows unless a segment would outgrow the wavefront's LDS buffer, then 64
  qp::ColBlockPlan& C = op->cb;
  std::vector<int32_t> segcnt;
  int rpt = 0, max_seg = 0;
  int64_t ntiles = 0;
  for (int tryr : {2, 1}) {
    const int64_t TR = 64 * tryr;
    ntiles = (nrows + TR - 1) / TR;
    segcnt.assign((size_t)(ntiles * P + 1), 0);
    parallel_rows(ntiles, [&](int64_t t0, int64_t t1) {
      for (int64_t t = t0; t < t1; ++t)
        for (int64_t r = t * TR; r < std::min(nrows, (t + 1) * TR); ++r)
          for (int64_t p = ur[r]; p < ur[r + 1]; ++p) segcnt[(size_t)(t * P + ((int64_t)uc[p] >> log2w))]++;
    }, 512);
    max_seg = 0;
    for (int64_t sgi = 0; sgi < ntiles * P; ++sgi) max_seg = std::max(max_seg, (int)segcnt[(size_t)sgi]);
    if (max_seg <= qp::kCbMaxSeg) {
      rpt = tryr;
      break;
    }
  }
  if (rpt == 0) return QP_OK;   // a (64-row, 2^log2w-column) cell with more than kCbMaxSeg entries: not this kernel's operator
  {
    const int64_t resident = (int64_t)std::max(qp::device_cu_count(), 1) * qp::kCbWavesPerCu;
    if ((ntiles + resident - 1) / resident > qp::kCbMaxTilesPerWave) return QP_OK;
  }
  const int64_t TR = 64 * rpt;
  std::vector<int32_t> segptr((size_t)(ntiles * P + 1));
  {
    int64_t run = 0;
    for (int64_t sgi = 0; sgi < ntiles * P; ++sgi) {
      segptr[(size_t)sgi] = (int32_t)run;
      run += segcnt[(size_t)sgi];
    }
    segptr[(size_t)(ntiles * P)] = (int32_t)run;
  }
  std::vector<int64_t> vmap;
  csr_value_map(op, vmap);
  std::vector<uint16_t> rowoff((size_t)(ntiles * P) * (size_t)(TR + 1));
  std::vector<uint32_t> cols((size_t)nnz);
  std::vector<int64_t> map((size_t)nnz);
  parallel_rows(ntiles, [&](int64_t t0, int64_t t1) {
    std::vector<int32_t> fill((size_t)P);
    for (int64_t t = t0; t < t1; ++t) {
      for (int64_t c = 0; c < P; ++c) fill[(size_t)c] = 0;
      for (int64_t l = 0; l < TR; ++l) {
        const int64_t r = t * TR + l;
        for (int64_t c = 0; c < P; ++c) rowoff[(size_t)(t * P + c) * (size_t)(TR + 1) + (size_t)l] = (uint16_t)fill[(size_t)c];
        if (r >= nrows) continue;
        for (int64_t p = ur[r]; p < ur[r + 1]; ++p) {
          const int64_t c = (int64_t)uc[p] >> log2w;
          const int64_t e = (int64_t)segptr[(size_t)(t * P + c)] + fill[(size_t)c]++;
          cols[(size_t)e] = (uint32_t)uc[p];
          map[(size_t)e] = vmap[(size_t)p];
        }
      }
      for (int64_t c = 0; c < P; ++c) rowoff[(size_t)(t * P + c) * (size_t)(TR + 1) + (size_t)TR] = (uint16_t)fill[(size_t)c];
    }
    }, 512);
  QP_CHECK(dev_alloc(&C.segptr, segptr.size()));
  QP_CHECK(dev_alloc(&C.rowoff, rowoff.size()));
  QP_CHECK(dev_alloc(&C.cols, cols.size()));
  QP_CHECK(dev_alloc(&C.map, map.size()));
  QP_CHECK(dev_alloc(&C.vals, (size_t)nnz));
  QP_HIP(hipMemcpy(C.segptr, segptr.data(), segptr.size() * sizeof(int32_t), hipMemcpyHostToDevice));
  QP_HIP(hipMemcpy(C.rowoff, rowoff.data(), rowoff.size() * sizeof(uint16_t), hipMemcpyHostToDevice));
  QP_HIP(hipMemcpy(C.cols, cols.data(), cols.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
  QP_HIP(hipMemcpy(C.map, map.data(), map.size() * sizeof(int64_t), hipMemcpyHostToDevice));
  C.log2w = log2w;
  C.P = (int)P;
  C.rpt = rpt;
  C.max_seg = max_seg;
  C.ntiles = ntiles;
  C.nnz = nnz;
  C.valid = 1;
  op->A.cb = &op->cb;
  return QP_OK;
}

int choose_format(qp_operator* op, int requested, bool hermitian) {
  const auto& ur = op->u_rowptr;
  const int64_t nrows = op->A.nrows, nnz = ur[nrows];
  const int64_t nblocks = (nrows + kRB - 1) / kRB;
  int64_t rb_stored = 0;
  for (int64_t b = 0; b < nblocks; ++b) {
    int64_t w = 0;
    for (int64_t r = b * kRB; r < std::min(nrows, (b + 1) * kRB); ++r) w = std::max(w, ur[r + 1] - ur[r]);
    rb_stored += ((w + 3) & ~(int64_t)3) * kRB;
  }
  // the row-block kernels stream ~1.6x faster than the sub-wave CSR kernel at equal bytes
  // (profiles/r01/kbench_*): accept up to 50 % padding before falling back
  const bool rb_ok = (double)rb_stored <= 1.5 * (double)nnz + 1024.0;
  if (requested == QP_FMT_AUTO) {
    // (a Hermitian operator is judged by the padding of its packed form further down: five entries per row -- the
    // five-point lattice -- pad to eight as plain row blocks, but to four upper entries when packed)
    if (!rb_ok && !hermitian) return QP_FMT_CSR;
    // few, long rows (small dense generators: the reference's test and benchmark sizes): a
    // row block gives one wavefront 64 rows to walk entry by entry -- too few wavefronts to
    // hide the latency.  One wavefront per row instead (CSR kernel, 64 lanes per row).
    if (nblocks < 2048 && nnz >= 32 * nrows) return QP_FMT_CSR;
    if (!hermitian) return QP_FMT_RBCSR;
    int64_t hu_stored = 0;   // values the packed form stores: the upper section of every block, padded to a multiple of four
    {
      const auto& ucc = op->u_col;
      std::atomic<int64_t> total{0};
      parallel_rows(nblocks, [&](int64_t b0, int64_t b1) {
        int64_t mine = 0;
        for (int64_t b = b0; b < b1; ++b) {
          int64_t w = 0;
          for (int64_t r = b * kRB; r < std::min(nrows, (b + 1) * kRB); ++r) {
            const int32_t* cb = ucc.data() + ur[r];
            const int32_t* ce = ucc.data() + ur[r + 1];
            w = std::max<int64_t>(w, ce - std::lower_bound(cb, ce, (int32_t)r));
          }
          mine += ((w + 3) & ~(int64_t)3) * kRB;
        }
        total.fetch_add(mine, std::memory_order_relaxed);
      }, 1024);
      hu_stored = total.load();
    }
    const bool hrb_ok = rb_ok || (double)hu_stored <= 0.9 * (double)nnz + 1024.0;
    // Hermitian packing pays only if the transposed values are still in the XCD's L2
    // (4 MiB) when the lower entry is processed: the rows stream in order, so require
    // (row - col) * bytes-per-row <= 2 MiB for at least 85 % of the lower entries.
    // Measured (profiles/r01/kbench): banded 53.6 vs 72.1 us per term, scattered 103.8 vs 93.4.
    const auto& uc = op->u_col;
    const double row_bytes = 14.0 * (double)nnz / (double)std::max<int64_t>(nrows, 1) + 80.0;
    const int64_t maxdist = (int64_t)(2.0 * 1048576.0 / row_bytes);
    int64_t nlow = 0, nnear = 0;
    {
      std::atomic<int64_t> tl{0}, tn{0};
      parallel_rows(nrows, [&](int64_t r0, int64_t r1) {
        int64_t l = 0, nr = 0;
        for (int64_t r = r0; r < r1; ++r)
          for (int64_t p = ur[r]; p < ur[r + 1] && uc[p] < r; ++p) {
            ++l;
            if (r - uc[p] <= maxdist) ++nr;
          }
        tl.fetch_add(l, std::memory_order_relaxed);
        tn.fetch_add(nr, std::memory_order_relaxed);
      });
      nlow = tl.load();
      nnear = tn.load();
    }
    // ... and only if the transposed reads are coalesced: in a block whose 64 rows have their k-th entry at
    // the same distance from the diagonal (stencil-like: lattices, tensor products) the wave reads 64
    // consecutive values; in an irregular block every lane pulls its own L2 line for 16 useful bytes
    // (measured, columns drawn per row inside 4096-row windows: Hermitian-packed 138 us vs 103 us per
    // term, profiles/r02/kbench_random_window.txt).  A sample of the blocks decides.
    int64_t sampled = 0, regular = 0;
    const int64_t bstride = std::max<int64_t>(1, nblocks / 512);
    for (int64_t b = 0; b < nblocks; b += bstride) {
      const int64_t r0 = b * kRB, r1 = std::min(nrows, r0 + kRB);
      if (r1 - r0 < kRB) continue;
      ++sampled;
      // the lower entries (col < row: the ones read through the transposed position) of every row of the
      // block at the same distances from the diagonal
      bool same = true;
      int64_t nl0 = 0;
      while (ur[r0] + nl0 < ur[r0 + 1] && uc[ur[r0] + nl0] < r0) ++nl0;
      for (int64_t r = r0 + 1; r < r1 && same; ++r) {
        int64_t nl = 0;
        while (ur[r] + nl < ur[r + 1] && uc[ur[r] + nl] < r) ++nl;
        if (nl != nl0) same = false;
        for (int64_t k = 0; k < nl0 && same; ++k)
          if ((int64_t)uc[ur[r] + k] - r != (int64_t)uc[ur[r0] + k] - r0) same = false;
      }
      if (same) ++regular;
    }
    const bool coalesced = sampled == 0 || 4 * regular >= 3 * sampled;
    // (the packed format addresses the transposed values with int32 positions)
    // ... unless the operator will take the strip walk, which reads a far transposed value one step ahead like any other
    // stream instead of waiting for it in the row sum (the plane distance of a three-dimensional grid is 32768 rows away:
    // 256 x 128 x 128 grid 136 us per term as plain row blocks, 101 packed and walked)
    bool walkable = false;
    if (op->ctx->tun.hrb_walk && nblocks >= op->ctx->tun.walk_min_blocks) {
      int64_t rm = nrows / 2;
      for (int64_t r = std::max<int64_t>(0, nrows / 2 - 128); r < std::min(nrows, nrows / 2 + 128); ++r)
        if (ur[r + 1] - ur[r] > ur[rm + 1] - ur[rm]) rm = r;
      std::vector<int64_t> D((size_t)(ur[rm + 1] - ur[rm]));
      for (size_t k = 0; k < D.size(); ++k) D[k] = (int64_t)uc[ur[rm] + (int64_t)k] - rm;
      WalkShape wsh;
      walkable = parse_walk_shape(D, wsh) && 4 * regular >= 3 * sampled;
    }
    if (((double)nnear >= 0.85 * (double)nlow || walkable) && coalesced && hrb_ok && rb_stored < (int64_t)INT32_MAX) return QP_FMT_HRB;
    return rb_ok ? QP_FMT_RBCSR : QP_FMT_CSR;
  }
  if (requested == QP_FMT_HRB && !hermitian) return -1;
  return requested;
}

// Lattice completion (knob lattice_fill).  A finite-difference operator on an nx x ny grid with open boundaries is the
// walk's lattice -- distances +-1, +-nx -- except that the rows at x = 0 lack the -1 entry and those at x = nx - 1 the +1
// entry: one row in nx breaks the "same distances on every row" run that the strip walk (and the stencil encoding of the
// row blocks) needs.  If every row between the first and the last K g rows carries a SUBSET of the middle row's distance
// list, that list has the walk's shape, and at most 12 % of the entries are missing (a 64 x 8 x nz grid: 4 %), the missing ones are stored as explicit
// zeros (with their transposes, so that the pattern stays structurally symmetric).  Index work only; 0 * x terms change no
// row sum beyond the order in which the two accumulators of a row take their entries.
void lattice_fill(const qp::Tuning& tun, int64_t n, int64_t ncols, qp::HostVec<int64_t>& ur, qp::HostVec<int32_t>& uc,
                  qp::HostVec<int64_t>* ur_before, qp::HostVec<int32_t>* uc_before) {
  if (!tun.lattice_fill || ncols < n || n / kRB < std::max(tun.walk_min_blocks, 16)) return;
  // (ncols > n: the local rows of a row-partitioned operator; its halo columns appear only in the first / last K g rows)
  // the reference row: the fullest one near the middle (the middle row itself may sit on the grid's edge)
  int64_t rm = n / 2;
  for (int64_t r = std::max<int64_t>(0, n / 2 - 128); r < std::min(n, n / 2 + 128); ++r)
    if (ur[r + 1] - ur[r] > ur[rm + 1] - ur[rm]) rm = r;
  for (int64_t t = -32; t <= 32; ++t) {   // (a whole line of a three-dimensional grid may sit on an edge: look further out too)
    const int64_t r = n / 2 + t * 4099;
    if (r >= 0 && r < n && ur[r + 1] - ur[r] > ur[rm + 1] - ur[rm]) rm = r;
  }
  // (... or a whole band of planes, when the stencil reaches two lines and two planes out and the grid is small: 2048 rows
  // spread over the middle half, at offsets that run through every position inside a line)
  for (int64_t k = 0; k < 2048; ++k) {
    const int64_t r = n / 4 + (k * (n / 2)) / 2048 + (k * 37) % 64;
    if (r >= 0 && r < n && ur[r + 1] - ur[r] > ur[rm + 1] - ur[rm]) rm = r;
  }
  const int z = (int)(ur[rm + 1] - ur[rm]);
  if (z < 3 || z > 19) return;
  std::vector<int64_t> D((size_t)z);
  for (int k = 0; k < z; ++k) D[(size_t)k] = (int64_t)uc[ur[rm] + k] - rm;
  WalkShape ws;
  if (!parse_walk_shape(D, ws)) return;
  const int64_t reach = std::max<int64_t>((int64_t)ws.K * ws.g + ws.fd, ws.glong);
  const int64_t lo = reach, hi = n - reach;
  if (hi - lo < 16 * (int64_t)kRB) return;
  int64_t missing = 0;
  {   // nothing to complete when every row between lo and hi already has z entries (the headline lattice): skip the check
    bool longer = false;
    for (int64_t r = lo; r < hi && !longer; ++r) {
      const int64_t len = ur[r + 1] - ur[r];
      longer = len > z;
      missing += z - len;
    }
    if (longer || missing == 0) return;
    missing = 0;
  }
  for (int64_t r = lo; r < hi; ++r) {
    int d = 0;
    for (int64_t p = ur[r]; p < ur[r + 1]; ++p) {
      if (uc[p] >= n) return;
      const int64_t delta = (int64_t)uc[p] - r;
      while (d < z && D[(size_t)d] < delta) ++d;
      if (d == z || D[(size_t)d] != delta) return;   // an entry outside the lattice's distances: not this kind of operator
      ++d;
    }
    missing += z - (ur[r + 1] - ur[r]);
  }
  if (missing == 0 || (double)missing > 0.12 * (double)ur[n]) return;
  // transposes of filled entries that land in the first / last K g rows
  std::vector<std::pair<int64_t, int32_t>> extra;
  for (int64_t r = lo; r < hi; ++r) {
    if (ur[r + 1] - ur[r] == z) continue;
    int64_t p = ur[r];
    for (int d = 0; d < z; ++d) {
      const int64_t c = r + D[(size_t)d];
      if (p < ur[r + 1] && uc[p] == c) {
        ++p;
        continue;
      }
      if (c < lo || c >= hi) extra.emplace_back(c, (int32_t)r);
    }
  }
  std::sort(extra.begin(), extra.end());
  qp::HostVec<int64_t> nr((size_t)n + 1, 0);
  qp::HostVec<int32_t> nc;
  nc.reserve(uc.size() + (size_t)missing + extra.size());
  size_t ex = 0;
  std::vector<int32_t> row;
  for (int64_t r = 0; r < n; ++r) {
    if (r >= lo && r < hi) {
      for (int d = 0; d < z; ++d) nc.push_back((int32_t)(r + D[(size_t)d]));
    } else {
      row.assign(uc.begin() + ur[r], uc.begin() + ur[r + 1]);
      while (ex < extra.size() && extra[ex].first == r) row.push_back(extra[ex++].second);
      std::sort(row.begin(), row.end());
      row.erase(std::unique(row.begin(), row.end()), row.end());
      nc.insert(nc.end(), row.begin(), row.end());
    }
    nr[(size_t)r + 1] = (int64_t)nc.size();
  }
  ur.swap(nr);
  uc.swap(nc);
  // the pattern as it was (the caller takes the completion back when the operator does not end up walked) -- handed over only
  // when something was completed: no copy of a pattern that stays as it is
  if (ur_before) ur_before->swap(nr);
  if (uc_before) uc_before->swap(nc);
}

// ---- plans of the batched path (qp_cheby_step_batched): the row walk of the row kernel and the LDS-staged tiles ----
// Row walk for the batched kernel (kernels_spmm.hip: spmm_rows_kernel).  The pattern is
// sampled for its offsets d = col - row (folded to (-n/2, n/2]); when the far ones (|d| >= 64) are all
// multiples of one inner dimension g -- H = H_a (x) 1 + 1 (x) H_c, i = a g + c: lattice and tensor-product
// operators -- the rows are listed strip by strip: `sw` consecutive inner indices c, all outer indices a
// in turn.  Then the rows that gather a given row of X (its +-k g and +-near neighbours) are visited
// within a few strip widths of each other instead of 2 a_max g rows apart, and the strip width is chosen
// so that this window, plus the rows in flight, fits half an XCD's L2 at `batch` states per row.
// Anything else (no far offsets, no common inner dimension, strips narrower than four times the near
// reach) keeps the natural order.  Index work only: the arithmetic per row does not change.
static void spmm_walk_host(const qp_operator* op, int batch, int knob, std::vector<int32_t>* order,
                           int64_t* g_out, int64_t* sw_out) {
  order->clear();
  *g_out = *sw_out = 0;
  const int64_t n = op->A.nrows;
  const auto& ur = op->u_rowptr;
  const auto& uc = op->u_col;
  if (knob < 0 || n < 4096 || n > INT32_MAX || op->A.ncols != n || ur.empty()) return;
  int64_t g = 0, far_max = 0, near_max = 0;
  const int64_t nsample = std::min<int64_t>(n, 4096), stride = n / nsample;
  for (int64_t t = 0; t < nsample; ++t) {
    const int64_t r = t * stride;
    for (int64_t p = ur[r]; p < ur[r + 1]; ++p) {
      int64_t d = (int64_t)uc[p] - r;
      if (d > n / 2) d -= n;
      if (d <= -(n + 1) / 2) d += n;
      d = std::llabs(d);
      if (d >= 64) {
        g = std::gcd(g, d);
        far_max = std::max(far_max, d);
      } else {
        near_max = std::max(near_max, d);
      }
    }
  }
  if (g < 256 || far_max / g > 64) return;
  const int64_t amax = far_max / g;
  const int64_t l2_budget = 1280 * 1024;                 // under a third of an XCD's 4 MiB L2 for the gather window (measured at
                                                         // 64 states: strips of 64 beat 128 and 32, profiles/r02/batched_c5_sweep.txt)
  const int64_t row_bytes = (int64_t)std::min(batch, 64) * (int64_t)sizeof(double2);
  int64_t sw;
  if (knob > 0) {
    sw = knob;
  } else {
    const int64_t inflight = 512;                         // rows an XCD has in flight (32 CUs x 16 waves)
    sw = (l2_budget / row_bytes - inflight) / (2 * amax + 1);
  }
  sw = std::min(sw, g);
  while (sw > 1 && g % sw != 0) --sw;                     // every strip the same width
  if (sw < 1) return;
  if (knob == 0 && (sw < 4 * std::max<int64_t>(near_max, 1) || sw < 16)) return;
  if (sw >= g) return;                                    // one strip = the natural order
  order->resize((size_t)n);
  const int64_t na = (n + g - 1) / g;
  size_t k = 0;
  for (int64_t c0 = 0; c0 < g; c0 += sw)
    for (int64_t a = 0; a < na; ++a)
      for (int64_t c = c0; c < c0 + sw; ++c) {
        const int64_t i = a * g + c;
        if (i < n) (*order)[k++] = (int32_t)i;
      }
  if ((int64_t)k != n) {   // cannot happen; fall back to the natural order rather than skip rows
    order->clear();
    return;
  }
  *g_out = g;
  *sw_out = sw;
}

int operator_spmm_order(qp_operator* op, int batch, const int32_t** order_out) {
  qp_ctx* ctx = op->ctx;
  const int knob = ctx->tun.spmm_strip;
  if (op->m_order_valid && op->m_order_batch == batch && op->m_order_knob == knob) {
    *order_out = op->m_order;
    return QP_OK;
  }
  if (op->m_order) (void)hipFree(op->m_order);
  op->m_order = nullptr;
  op->m_order_valid = true;
  op->m_order_batch = batch;
  op->m_order_knob = knob;
  op->m_order_g = op->m_order_sw = 0;
  *order_out = nullptr;
  std::vector<int32_t> order;
  int64_t g = 0, sw = 0;
  spmm_walk_host(op, batch, knob, &order, &g, &sw);
  if (order.empty()) return QP_OK;
  QP_CHECK(dev_alloc(&op->m_order, order.size()));
  QP_HIP(hipMemcpy(op->m_order, order.data(), order.size() * sizeof(int32_t), hipMemcpyHostToDevice));
  op->m_order_g = g;
  op->m_order_sw = sw;
  *order_out = op->m_order;
  return QP_OK;
}

// Tiles of the batched path (device.h: SpmmTiles; kernel: spmm_tile_kernel).  The pattern of the row in the middle of the matrix is
// the candidate: every distance near (|d| <= 4) or a multiple m g of the far distances' gcd with |m| <= 4.  A row is regular when
// its entries are exactly that pattern; a tile is sixteen regular rows r0 + i g + j (i, j < 4, r0 = 4 ta g + 4 tc).  Tiles are
// listed strip by strip like the row walk above (the kernel is not sensitive to the strip width: any order that keeps the tiles of
// neighbouring strip steps close in time serves the far halo from L2); every other row goes to the row kernel.  Index work only.
static void spmm_tiles_host(const qp_operator* op, int knob, qp::SpmmTileShape* shape, std::vector<int32_t>* tiles,
                            std::vector<int32_t>* rest, int64_t* g_out, int64_t* sw_out) {
  tiles->clear();
  rest->clear();
  *g_out = *sw_out = 0;
  const int64_t n = op->A.nrows;
  const auto& ur = op->u_rowptr;
  const auto& uc = op->u_col;
  if (knob < 0 || n < 4096 || n > INT32_MAX || op->A.ncols != n || ur.empty() || op->A.nnz > (int64_t)INT32_MAX) return;
  const int64_t rm = n / 2;
  const int nd = (int)(ur[rm + 1] - ur[rm]);
  if (nd < 2 || nd > qp::kSpmmTileMaxEntries) return;
  std::vector<int64_t> D((size_t)nd);
  int64_t g = 0, far_max = 0, near_max = 0;
  for (int k = 0; k < nd; ++k) {
    const int64_t d = (int64_t)uc[ur[rm] + k] - rm;
    D[(size_t)k] = d;
    const int64_t ad = std::llabs(d);
    if (ad <= 4) near_max = std::max(near_max, ad);
    else if (ad >= 64) {
      g = std::gcd(g, ad);
      far_max = std::max(far_max, ad);
    } else return;
  }
  if (g < 64 || far_max / g > 4) return;
  qp::SpmmTileShape sh;
  sh.nd = nd;
  sh.K = (int)(far_max / g);
  sh.NN = (int)near_max;
  for (int k = 0; k < nd; ++k) {
    const int64_t d = D[(size_t)k];
    const bool nearby = std::llabs(d) <= 4 && d != 0;
    sh.dnear[k] = nearby ? (int)d : 0;
    sh.dfar[k] = nearby ? 0 : (int)(d / g);
  }
  std::vector<uint8_t> regular((size_t)n);
  parallel_rows(n, [&](int64_t r_begin, int64_t r_end) {
    for (int64_t r = r_begin; r < r_end; ++r) {
      bool ok = ur[r + 1] - ur[r] == nd;
      for (int k = 0; ok && k < nd; ++k) ok = (int64_t)uc[ur[r] + k] - r == D[(size_t)k];
      regular[(size_t)r] = ok ? 1 : 0;
    }
  });
  const int64_t na = (n + g - 1) / g;
  int64_t sw = knob > 0 ? knob : 128;
  sw = std::max<int64_t>(4, std::min(sw, g) / 4 * 4);
  std::vector<uint8_t> covered((size_t)n, 0);
  for (int64_t c0 = 0; c0 < g; c0 += sw)
    for (int64_t a0 = 0; a0 + 3 < na; a0 += 4)
      for (int64_t c = c0; c + 3 < std::min(c0 + sw, g); c += 4) {
        const int64_t r0 = a0 * g + c;
        if (r0 + 3 * g + 3 >= n) continue;
        bool ok = true;
        for (int i = 0; ok && i < 4; ++i)
          for (int j = 0; ok && j < 4; ++j) ok = regular[(size_t)(r0 + i * g + j)] != 0;
        if (!ok) continue;
        tiles->push_back((int32_t)r0);
        for (int i = 0; i < 4; ++i)
          for (int j = 0; j < 4; ++j) covered[(size_t)(r0 + i * g + j)] = 1;
      }
  if ((int64_t)tiles->size() * 16 < n / 2) {   // mostly edges: the row kernel alone
    tiles->clear();
    return;
  }
  for (int64_t r = 0; r < n; ++r)
    if (!covered[(size_t)r]) rest->push_back((int32_t)r);
  *shape = sh;
  *g_out = g;
  *sw_out = sw;
}

int operator_spmm_tiles(qp_operator* op, const qp::SpmmTiles** out) {
  qp_ctx* ctx = op->ctx;
  qp::SpmmTiles& P = op->m_tiles;
  const int knob = ctx->tun.spmm_strip;
  *out = nullptr;
  if (!P.built || P.knob != knob) {
    if (P.tiles) (void)hipFree(P.tiles);
    if (P.rest) (void)hipFree(P.rest);
    if (P.tab) (void)hipFree(P.tab);
    P = qp::SpmmTiles();
    P.built = true;
    P.knob = knob;
    std::vector<int32_t> tiles, rest;
    spmm_tiles_host(op, knob, &P.shape, &tiles, &rest, &P.g, &P.sw);
    if (!tiles.empty()) {
      // the kernel's table: where the staged rows lie relative to r0, where every wavefront finds its operands in LDS
      const int K = P.shape.K, NN = P.shape.NN, nfar = (4 + 2 * K) * 4;
      P.T = nfar + 8 * NN;
      std::vector<int32_t> tab((size_t)qp::kSpmmTileTab, 0);
      for (int slot = 0; slot < P.T; ++slot) {
        int64_t d;
        if (slot < nfar) {
          d = (int64_t)(slot / 4 - K) * P.g + slot % 4;
        } else {
          const int s2 = slot - nfar, ii = s2 / (2 * NN), jj = s2 % (2 * NN);
          d = (int64_t)ii * P.g + (jj < NN ? jj - NN : 4 + jj - NN);
        }
        tab[(size_t)slot] = (int32_t)d;
      }
      for (int w = 0; w < 16; ++w) {
        const int i = w / 4, j = w % 4, own = (i + K) * 4 + j;
        for (int k = 0; k < P.shape.nd; ++k) {
          int slot;
          const int dn = P.shape.dnear[k];
          if (dn == 0) {
            slot = own + 4 * P.shape.dfar[k];
          } else {
            const int jj = j + dn;
            slot = (jj >= 0 && jj < 4) ? own + dn : nfar + i * 2 * NN + (jj < 0 ? jj + NN : jj - 4 + NN);
          }
          tab[(size_t)(qp::kSpmmTileSlots + qp::kSpmmTileMaxEntries * w + k)] = slot * 64 * (int32_t)sizeof(double2);
        }
        tab[(size_t)(qp::kSpmmTileSlots + qp::kSpmmTileMaxEntries * 16 + w)] = (int32_t)((int64_t)i * P.g + j);
        tab[(size_t)(qp::kSpmmTileSlots + qp::kSpmmTileMaxEntries * 16 + 16 + w)] = own * 64 * (int32_t)sizeof(double2);
      }
      QP_CHECK(dev_alloc(&P.tab, tab.size()));
      QP_HIP(hipMemcpy(P.tab, tab.data(), tab.size() * sizeof(int32_t), hipMemcpyHostToDevice));
      QP_CHECK(dev_alloc(&P.tiles, tiles.size()));
      QP_HIP(hipMemcpy(P.tiles, tiles.data(), tiles.size() * sizeof(int32_t), hipMemcpyHostToDevice));
      if (!rest.empty()) {
        QP_CHECK(dev_alloc(&P.rest, rest.size()));
        QP_HIP(hipMemcpy(P.rest, rest.data(), rest.size() * sizeof(int32_t), hipMemcpyHostToDevice));
      }
      P.ntiles = (int64_t)tiles.size();
      P.nrest = (int64_t)rest.size();
      P.valid = 1;
    }
  }
  if (P.valid) *out = &P;
  return QP_OK;
}
