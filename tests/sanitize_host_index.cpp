// AddressSanitizer / UBSan harness for the library's HOST INDEX WORK: everything qp_operator_create does on the
// host before a kernel ever runs -- union pattern, Hermitian check, format choice, lattice completion, strip-walk plan, column
// encodings (int32 / int16 / stencil / block map), Hermitian packing with transposed positions, column-blocked mirror, value
// dictionary -- and the decoders of qp_operator_get_csr, compiled from the library's own sources (csrc/engine_core.hip,
// csrc/engine_plans.hip, csrc/host_numerics.cpp) with g++ against tests/hip_host_shim (device memory = heap memory, so the
// sanitizer checks every copy into a "device" array against its allocation) and driven by a fuzz over the shapes that have bitten:
// tall (5681 x 358: the round-4 GPU memory fault), wide, 1 row, fewer than 64 rows, empty rows, lattices, grids, spin chains.
// Checked per case: the get_csr round trip reproduces the input exactly; EVERY stored slot of every row block -- real entries,
// pads, the lanes beyond the last row -- decodes to a column inside the matrix; every pad carries the value zero in every term;
// every transposed position of a Hermitian-packed operator is -1 (pad) or inside the value array.
// Built and run by tests/test_cabi_host.py::test_host_index_work_under_sanitizers (CPU only).
#include <cstdio>
#include <random>

#include "../quantumpropagators.jl_amd/csrc/engine_core.hip"
#include "../quantumpropagators.jl_amd/csrc/engine_plans.hip"

// ---- host stand-ins of the few launches the creation path makes (values only; no kernel runs in this build) ---------------
namespace qp {
static inline void cfma_h(double2& s, double2 a, double2 b) {
  s.x = std::fma(a.x, b.x, s.x);
  s.x = std::fma(-a.y, b.y, s.x);
  s.y = std::fma(a.x, b.y, s.y);
  s.y = std::fma(a.y, b.x, s.y);
}
int launch_combine_planes(hipStream_t, double2* vals, const double2* const* planes, const double2* coefs, int nplanes, int64_t n,
                          double* vals_r, Stats*) {
  for (int64_t p = 0; p < n; ++p) {
    double2 acc{0.0, 0.0};
    for (int l = 0; l < nplanes; ++l) cfma_h(acc, coefs[l], planes[l][p]);
    vals[p] = acc;
    if (vals_r) vals_r[p] = acc.x;
  }
  return QP_OK;
}
int launch_sparse_planes_update(hipStream_t, double2* vals, const double2* base, const int32_t* support, int64_t n_support,
                                const double2* support_vals, int nplanes, const double2* coefs, double* vals_r, Stats*) {
  for (int64_t i = 0; i < n_support; ++i) {
    double2 acc = base[support[i]];
    for (int l = 0; l < nplanes; ++l) cfma_h(acc, coefs[l], support_vals[(size_t)l * n_support + i]);
    vals[support[i]] = acc;
    if (vals_r) vals_r[support[i]] = acc.x;
  }
  return QP_OK;
}
int launch_real_part(hipStream_t, double* out, const double2* v, int64_t n, Stats*) {
  for (int64_t i = 0; i < n; ++i) out[i] = v[i].x;
  return QP_OK;
}
int launch_colblock_gather(hipStream_t, const ColBlockPlan& P, const double2* src, Stats*) {
  for (int64_t p = 0; p < P.nnz; ++p) {
    const int64_t m = P.map[p];
    const double2 v = m >= 0 ? src[m] : make_double2(src[-m - 1].x, -src[-m - 1].y);
    P.vals[p] = v;
    if (P.use_real && P.vals_r) P.vals_r[p] = v.x;
  }
  return QP_OK;
}
int launch_gather_csr_vals(hipStream_t, double2* out, const double2* vals, const int64_t* map, int64_t nnz, Stats*) {
  for (int64_t p = 0; p < nnz; ++p) out[p] = map[p] >= 0 ? vals[map[p]] : make_double2(vals[-map[p] - 1].x, -vals[-map[p] - 1].y);
  return QP_OK;
}
// (never reached by operator creation / get_csr; the linker wants them)
int launch_fill(hipStream_t, double2*, double2, int64_t, Stats*) { std::abort(); }
int launch_scal(hipStream_t, double2*, double2, int64_t, Stats*) { std::abort(); }
int launch_axpy(hipStream_t, double2, const double2*, double2*, int64_t, Stats*) { std::abort(); }
int launch_dot_partials(hipStream_t, const double2*, const double2*, double2*, int64_t, Stats*) { std::abort(); }
int launch_spmv_plain(hipStream_t, const DevMatrix&, const double2*, const PlainEpi&, Stats*) { std::abort(); }
}  // namespace qp

namespace {

struct Csr {
  int64_t nrows = 0, ncols = 0;
  std::vector<int64_t> rp;
  std::vector<int32_t> col;      // (int64 copies are made for qp_matrix_create, whose index arrays are the reference's Int64)
  std::vector<qp_c128> val;
};

#define REQUIRE(cond, ...)                                   \
  do {                                                       \
    if (!(cond)) {                                           \
      std::fprintf(stderr, "FAILED %s:%d: ", __FILE__, __LINE__); \
      std::fprintf(stderr, __VA_ARGS__);                     \
      std::fprintf(stderr, "\n");                            \
      std::exit(1);                                          \
    }                                                        \
  } while (0)

// the kernels' column decoder (csrc/kernel_common.h: ld_cols) restated on the host bytes: column of slot k, lane l of block b
int64_t decode_any(const char* bytes, int64_t meta, int64_t k, int l, int64_t rowc) {
  const char* p = bytes + (meta >> 2);
  const int mode = (int)(meta & 3);
  const int64_t q = k >> 2;
  const int j = (int)(k & 3);
  if (mode == 2) {
    int32_t d[4];
    std::memcpy(d, p + q * 16, 16);
    return rowc + d[j];
  }
  if (mode == 3) {
    const char* pq = p + (size_t)q * (16 + 4 * 64);
    int32_t cb[4];
    std::memcpy(cb, pq, 16);
    uint32_t lb;
    std::memcpy(&lb, pq + 16 + 4 * l, 4);
    return ((int64_t)cb[j] << 6) | ((lb >> (8 * j)) & 63u);
  }
  if (mode == 1) {
    int16_t d[4];
    std::memcpy(d, p + ((size_t)q * 64 + l) * 8, 8);
    return rowc + d[j];
  }
  int32_t c[4];
  std::memcpy(c, p + ((size_t)q * 64 + l) * 16, 16);
  return c[j];
}

int64_t n_cases = 0, n_walked = 0, n_hrb = 0, n_coded = 0, n_cb = 0, n_dense = 0;

void check_operator(qp_ctx* ctx, const std::vector<Csr>& terms, int ncoeffs, int format, const char* what) {
  std::vector<qp_matrix*> ms;
  for (const Csr& t : terms) {
    qp_matrix* m = nullptr;
    std::vector<int64_t> idx(t.col.begin(), t.col.end());
    REQUIRE(qp_matrix_create(ctx, t.nrows, t.ncols, (int64_t)t.col.size(), t.rp.data(), idx.data(), t.val.data(), QP_VAL_C128, QP_LAYOUT_CSR, 0, QP_FMT_AUTO, &m) == QP_OK,
            "%s: qp_matrix_create: %s", what, qp::g_last_error.c_str());
    ms.push_back(m);
  }
  qp_operator* op = nullptr;
  const int rc = qp_operator_create(ctx, ms.data(), (int)ms.size(), ncoeffs, format, &op);
  if (rc != QP_OK) {   // a request the library refuses (e.g. HRB for a non-Hermitian operator) is fine; a crash is not
    for (auto m : ms) qp_matrix_destroy(m);
    return;
  }
  ++n_cases;
  const DevMatrix& A = op->A;
  const int64_t nrows = A.nrows, ncols = A.ncols;
  if (A.format == QP_FMT_HRB) ++n_hrb;
  if (A.format == QP_FMT_DENSE) ++n_dense;
  if (op->walk.valid) ++n_walked;
  if (op->cv.valid) ++n_coded;
  if (op->cb.valid) ++n_cb;
  // ---- round trip: one term, unit coefficient -> the input bit for bit (more terms: the union pattern, values summed)
  {
    const int64_t nnz = A.nnz;
    std::vector<int64_t> rp((size_t)nrows + 1);
    std::vector<int32_t> col((size_t)std::max<int64_t>(nnz, 1));
    std::vector<qp_c128> val((size_t)std::max<int64_t>(nnz, 1));
    REQUIRE(qp_operator_get_csr(op, rp.data(), col.data(), val.data()) == QP_OK, "%s: get_csr: %s", what, qp::g_last_error.c_str());
    for (int64_t r = 0; r < nrows; ++r)
      for (int64_t p = rp[r]; p < rp[r + 1]; ++p) REQUIRE(col[p] >= 0 && col[p] < ncols, "%s: get_csr column %d of row %lld outside [0, %lld)", what, col[p], (long long)r, (long long)ncols);
    if (terms.size() == 1 && op->n_lattice_fill == 0 && A.format != QP_FMT_DENSE) {
      const Csr& t = terms[0];
      REQUIRE(nnz == (int64_t)t.col.size(), "%s: nnz %lld != %zu", what, (long long)nnz, t.col.size());
      for (int64_t r = 0; r <= nrows; ++r) REQUIRE(rp[r] == t.rp[r], "%s: rowptr[%lld]", what, (long long)r);
      for (int64_t p = 0; p < nnz; ++p)
        REQUIRE(col[p] == t.col[p] && val[p].re == t.val[p].re && val[p].im == t.val[p].im,   // (== : a conjugated zero is -0.0)
                "%s: entry %lld differs after the round trip: column %d value (%.17g, %.17g), passed in column %d value (%.17g, %.17g) [format %d, coded %d]", what,
                (long long)p, col[p], val[p].re, val[p].im, t.col[p], t.val[p].re, t.val[p].im, A.format, op->cv.valid);
    }
  }
  // ---- every stored slot of the row-block formats: a column inside the matrix, zero pads, valid transposed positions
  if (A.format == QP_FMT_RBCSR || A.format == QP_FMT_HRB) {
    const HostLayout& Lh = op->layout;
    const bool hrb = A.format == QP_FMT_HRB;
    const char* ub = reinterpret_cast<const char*>(A.cols);
    const char* lb = reinterpret_cast<const char*>(A.lcols);
    for (int64_t b = 0; b < A.nblocks; ++b) {
      const int64_t wu = (Lh.bptr[b + 1] - Lh.bptr[b]) / kRB;
      for (int64_t k = 0; k < wu; ++k)
        for (int l = 0; l < kRB; ++l) {
          const int64_t row = b * kRB + l, rowc = std::min(row, nrows - 1);   // (the kernels clamp the lanes beyond the last row)
          const int64_t c = decode_any(ub, Lh.cmeta[b], k, l, rowc);
          REQUIRE(c >= 0 && c < ncols, "%s: upper slot %lld of block %lld lane %d decodes to column %lld outside [0, %lld) (mode %d)", what,
                  (long long)k, (long long)b, l, (long long)c, (long long)ncols, (int)(Lh.cmeta[b] & 3));
          const int64_t nl = (hrb && row < nrows) ? Lh.nlow[row] : 0;
          const int64_t len = row < nrows ? op->u_rowptr[row + 1] - op->u_rowptr[row] - nl : 0;
          if (k >= len)   // a pad: inert in every term
            for (double2* plane : op->planes) {
              const double2 v = plane[Lh.bptr[b] + k * kRB + l];
              REQUIRE(v.x == 0.0 && v.y == 0.0, "%s: pad slot %lld of block %lld lane %d carries (%g, %g)", what, (long long)k, (long long)b, l, v.x, v.y);
            }
        }
      if (hrb) {
        const int64_t wl = (Lh.lptr[b + 1] - Lh.lptr[b]) / kRB;
        const bool stencil = (Lh.lcmeta[b] & 3) == 2;
        for (int64_t k = 0; k < wl; ++k)
          for (int l = 0; l < kRB; ++l) {
            const int64_t row = b * kRB + l, rowc = std::min(row, nrows - 1);
            int64_t c;
            if (stencil) {
              LowerStencilSlot e;
              std::memcpy(&e, lb + (Lh.lcmeta[b] >> 2) + (size_t)k * sizeof(e), sizeof(e));
              c = rowc + e.delta;
              const int64_t pos = (((c >> 6) == e.cb0) ? e.pb0 : e.pb1) + (c & 63);
              REQUIRE(pos >= 0 && pos < A.stored, "%s: stencil lower slot %lld of block %lld lane %d reads position %lld outside [0, %lld)", what,
                      (long long)k, (long long)b, l, (long long)pos, (long long)A.stored);
            } else {
              c = decode_any(lb, Lh.lcmeta[b], k, l, rowc);
              const int32_t pos = A.lpos[Lh.lptr[b] + (k >> 2) * (4 * kRB) + l * 4 + (k & 3)];
              REQUIRE(pos >= -1 && pos < A.stored, "%s: lower slot %lld of block %lld lane %d: position %d outside [-1, %lld)", what, (long long)k,
                      (long long)b, l, pos, (long long)A.stored);
            }
            REQUIRE(c >= 0 && c < ncols, "%s: lower slot %lld of block %lld lane %d decodes to column %lld outside [0, %lld)", what, (long long)k,
                    (long long)b, l, (long long)c, (long long)ncols);
          }
      }
    }
    if (op->walk.valid) {   // the walk's edge list names blocks of the operator
      for (int64_t i = 0; i < op->walk.n_edge; ++i)
        REQUIRE(op->walk.edge_map[i] >= 0 && op->walk.edge_map[i] < A.nblocks, "%s: edge block %d", what, op->walk.edge_map[i]);
      REQUIRE(op->walk.R0 >= 0 && op->walk.R1 <= A.nblocks && op->walk.W0 >= op->walk.R0, "%s: walk run [%lld, %lld)", what, (long long)op->walk.R0, (long long)op->walk.R1);
    }
    if (op->cv.valid) {     // value dictionary: every code inside its block's table
      for (int64_t b = 0; b < A.nblocks; ++b) {
        const int64_t tp = op->cv.tptr[b], len = tp & 511, first = tp >> 9;
        REQUIRE(first >= 0 && first + len <= op->cv.ntab && len <= 256, "%s: table of block %lld", what, (long long)b);
        for (int64_t p = Lh.bptr[b]; p < Lh.bptr[b + 1]; ++p) REQUIRE(op->cv.codes[p] < len, "%s: code %d beyond the table of block %lld", what, op->cv.codes[p], (long long)b);
      }
    }
  }
  if (op->cb.valid) {       // column-blocked mirror: columns inside the matrix, map inside the value array
    for (int64_t p = 0; p < op->cb.nnz; ++p) {
      REQUIRE((int64_t)op->cb.cols[p] < ncols, "%s: mirror column %u", what, op->cb.cols[p]);
      const int64_t m = op->cb.map[p], pos = m >= 0 ? m : -m - 1;
      REQUIRE(pos < A.stored, "%s: mirror map %lld", what, (long long)m);
    }
  }
  // coefficients change: the refresh paths (combined plane, sparse control terms, tables) stay inside their arrays
  if (ncoeffs > 0) {
    std::vector<qp_c128> c((size_t)ncoeffs);
    for (int rep = 0; rep < 3; ++rep) {
      for (int i = 0; i < ncoeffs; ++i) c[(size_t)i] = qp_c128{0.25 * (rep + 1) - i, rep == 2 ? 0.5 : 0.0};
      REQUIRE(qp_operator_set_coeffs(op, c.data(), ncoeffs) == QP_OK, "%s: set_coeffs: %s", what, qp::g_last_error.c_str());
    }
  }
  qp_operator_destroy(op);
  for (auto m : ms) qp_matrix_destroy(m);
}

Csr random_csr(std::mt19937_64& rng, int64_t nrows, int64_t ncols, int maxlen, int nvals, bool hermitian) {
  std::vector<std::vector<std::pair<int32_t, std::complex<double>>>> rows((size_t)nrows);
  std::vector<std::complex<double>> palette((size_t)nvals);
  std::normal_distribution<double> g;
  for (auto& v : palette) v = {g(rng), (rng() & 1) ? g(rng) : 0.0};
  for (int64_t r = 0; r < nrows; ++r) {
    const int len = (rng() % 7 == 0) ? 0 : (int)(rng() % (uint64_t)(maxlen + 1));
    for (int k = 0; k < len; ++k) {
      const int32_t c = (int32_t)(rng() % (uint64_t)ncols);
      std::complex<double> v = palette[rng() % palette.size()];
      if (hermitian) {
        if (c >= nrows || r >= ncols) continue;
        if (c == r) v = {v.real(), 0.0};
        rows[(size_t)r].push_back({c, v});
        if (c != r) rows[(size_t)c].push_back({(int32_t)r, std::conj(v)});
      } else {
        rows[(size_t)r].push_back({c, v});
      }
    }
  }
  Csr out;
  out.nrows = nrows;
  out.ncols = ncols;
  out.rp.assign((size_t)nrows + 1, 0);
  for (int64_t r = 0; r < nrows; ++r) {
    auto& rr = rows[(size_t)r];
    std::sort(rr.begin(), rr.end(), [](auto& a, auto& b) { return a.first < b.first; });
    // (duplicates: keep the first of a column; for the Hermitian case both triangles dropped the same ones by symmetry of the sort)
    int32_t last = -1;
    for (auto& e : rr)
      if (e.first != last) {
        out.col.push_back(e.first);
        out.val.push_back(qp_c128{e.second.real(), e.second.imag()});
        last = e.first;
      }
    out.rp[(size_t)r + 1] = (int64_t)out.col.size();
  }
  if (hermitian) {   // make the kept entries exactly conjugate-symmetric: rebuild the lower triangle from the upper one
    std::vector<std::vector<std::pair<int32_t, qp_c128>>> sym((size_t)nrows);
    for (int64_t r = 0; r < nrows; ++r)
      for (int64_t p = out.rp[r]; p < out.rp[r + 1]; ++p)
        if (out.col[p] >= r) {
          sym[(size_t)r].push_back({out.col[p], out.val[p]});
          if (out.col[p] != r) sym[(size_t)out.col[p]].push_back({(int32_t)r, qp_c128{out.val[p].re, -out.val[p].im}});
        }
    out.col.clear();
    out.val.clear();
    for (int64_t r = 0; r < nrows; ++r) {
      auto& rr = sym[(size_t)r];
      std::sort(rr.begin(), rr.end(), [](auto& a, auto& b) { return a.first < b.first; });
      for (auto& e : rr) {
        out.col.push_back(e.first);
        out.val.push_back(e.second);
      }
      out.rp[(size_t)r + 1] = (int64_t)out.col.size();
    }
  }
  return out;
}

// a Hermitian lattice: col = (i +- d) mod N for the given distances (the headline's generator), optional open ends
Csr lattice(std::mt19937_64& rng, int64_t N, const std::vector<int64_t>& dist, bool periodic, bool constant) {
  std::vector<std::vector<std::pair<int32_t, qp_c128>>> rows((size_t)N);
  std::normal_distribution<double> g;
  for (int64_t i = 0; i < N; ++i)
    for (int64_t d : dist) {
      int64_t j = i + d;
      if (j >= N) {
        if (!periodic) continue;
        j -= N;
      }
      if (j == i) continue;
      const qp_c128 v = constant ? qp_c128{-1.0, 0.0} : qp_c128{g(rng), g(rng)};
      rows[(size_t)i].push_back({(int32_t)j, v});
      rows[(size_t)j].push_back({(int32_t)i, qp_c128{v.re, -v.im}});
    }
  Csr out;
  out.nrows = out.ncols = N;
  out.rp.assign((size_t)N + 1, 0);
  for (int64_t r = 0; r < N; ++r) {
    auto& rr = rows[(size_t)r];
    std::sort(rr.begin(), rr.end(), [](auto& a, auto& b) { return a.first < b.first; });
    int32_t last = -1;
    for (auto& e : rr)
      if (e.first != last) {
        out.col.push_back(e.first);
        out.val.push_back(e.second);
        last = e.first;
      }
    out.rp[(size_t)r + 1] = (int64_t)out.col.size();
  }
  return out;
}

// transverse-field Ising chain (qubit-register structure: row XOR 2^i), few distinct values
Csr tfim(int n) {
  const int64_t N = (int64_t)1 << n;
  Csr out;
  out.nrows = out.ncols = N;
  out.rp.assign((size_t)N + 1, 0);
  for (int64_t r = 0; r < N; ++r) {
    std::vector<std::pair<int32_t, double>> rr;
    double diag = 0.0;
    for (int i = 0; i + 1 < n; ++i) diag -= (((r >> i) & 1) == ((r >> (i + 1)) & 1)) ? 1.0 : -1.0;
    rr.push_back({(int32_t)r, diag});
    for (int i = 0; i < n; ++i) rr.push_back({(int32_t)(r ^ ((int64_t)1 << i)), -1.0});
    std::sort(rr.begin(), rr.end());
    for (auto& e : rr) {
      out.col.push_back(e.first);
      out.val.push_back(qp_c128{e.second, 0.0});
    }
    out.rp[(size_t)r + 1] = (int64_t)out.col.size();
  }
  return out;
}

}  // namespace

int main() {
  qp_ctx* ctx = nullptr;
  if (qp_ctx_create(0, nullptr, &ctx) != QP_OK) {
    std::fprintf(stderr, "qp_ctx_create: %s\n", qp::g_last_error.c_str());
    return 2;
  }
  std::mt19937_64 rng(20261004);
  const int formats[] = {QP_FMT_AUTO, QP_FMT_RBCSR, QP_FMT_HRB, QP_FMT_CSR};
  // the shapes that have bitten, first
  const std::pair<int64_t, int64_t> shapes[] = {{5681, 358}, {358, 5681}, {1, 1}, {1, 70}, {70, 1}, {5, 5}, {63, 63}, {64, 64}, {65, 65},
                                                {127, 300}, {300, 127}, {1000, 1000}, {4097, 4097}, {200, 9000}};
  for (auto sh : shapes)
    for (int fmt : formats)
      for (int nvals : {2, 100000})
        for (int rep = 0; rep < 2; ++rep) {
          const bool herm = sh.first == sh.second && rep == 1;
          std::vector<Csr> terms{random_csr(rng, sh.first, sh.second, 11, nvals, herm)};
          char what[128];
          std::snprintf(what, sizeof what, "random %lld x %lld fmt %d nvals %d herm %d", (long long)sh.first, (long long)sh.second, fmt, nvals, (int)herm);
          check_operator(ctx, terms, 0, fmt, what);
        }
  // a row with a repeated column makes the stored count reach nrows x ncols although a position is missing -- the
  // duplicate is summed (Julia's sparse() semantics), the operator must not be taken for a complete dense one
  {
    Csr t;
    t.nrows = t.ncols = 3;
    t.rp = {0, 3, 6, 9};
    t.col = {0, 0, 1, 0, 1, 2, 0, 1, 2};
    for (int i = 0; i < 9; ++i) t.val.push_back(qp_c128{1.0 + i, 0.5});
    for (int fmt : {QP_FMT_AUTO, QP_FMT_DENSE, QP_FMT_RBCSR, QP_FMT_CSR}) {
      qp_matrix* m = nullptr;
      std::vector<int64_t> idx(t.col.begin(), t.col.end());
      REQUIRE(qp_matrix_create(ctx, 3, 3, 9, t.rp.data(), idx.data(), t.val.data(), QP_VAL_C128, QP_LAYOUT_CSR, 0, QP_FMT_AUTO, &m) == QP_OK, "dup: matrix");
      qp_operator* op = nullptr;
      REQUIRE(qp_operator_create(ctx, &m, 1, 0, fmt, &op) == QP_OK, "dup: operator (format %d): %s", fmt, qp::g_last_error.c_str());
      const int64_t nnz = op->A.nnz;
      std::vector<int64_t> rp(4);
      std::vector<int32_t> col((size_t)nnz);
      std::vector<qp_c128> val((size_t)nnz);
      REQUIRE(qp_operator_get_csr(op, rp.data(), col.data(), val.data()) == QP_OK, "dup: get_csr");
      // row 0: columns 0 (1 + 2 = 3, 0.5 + 0.5) and 1; a dense layout adds the explicit zero at (0, 2)
      REQUIRE(col[0] == 0 && val[0].re == 3.0 && val[0].im == 1.0 && col[1] == 1 && val[1].re == 3.0, "dup: row 0 is (%d: %g), (%d: %g)", col[0], val[0].re, col[1], val[1].re);
      if (op->A.format == QP_FMT_DENSE) REQUIRE(nnz == 9 && col[2] == 2 && val[2].re == 0.0 && val[2].im == 0.0, "dup: the missing position must be an explicit zero");
      else REQUIRE(nnz == 8, "dup: nnz %lld", (long long)nnz);
      qp_operator_destroy(op);
      qp_matrix_destroy(m);
      ++n_cases;
    }
  }
  // several terms with coefficients (lazy sum, sparse control terms, table recombination)
  for (int rep = 0; rep < 12; ++rep) {
    const int64_t n = 50 + (int64_t)(rng() % 3000);
    const int nt = 2 + (int)(rng() % 2);
    std::vector<Csr> terms;
    for (int t = 0; t < nt; ++t) terms.push_back(random_csr(rng, n, n, t == 0 ? 9 : 2, rep % 2 ? 3 : 100000, rep % 3 == 0));
    check_operator(ctx, terms, 1 + (int)(rng() % (uint64_t)(nt - 1 + 1)) % nt, formats[rep % 4], "several terms");
  }
  // lattices (strip-walk plans, lattice completion with open ends), sizes that walk (>= 3072 blocks needs N >= 196608: use the knob)
  qp_ctx_tuning_set(ctx, "walk_min_blocks", 8);
  const std::vector<std::vector<int64_t>> stencils = {{1, 2, 3, 4, 64, 128, 192, 256}, {1, 64}, {1, 2, 100, 200}, {1, 63, 64, 65}, {1, 32, 1024}, {1, 2, 16, 32, 512, 1024},
                                                      {3, 17}, {1, 2, 3, 4, 5}, {20, 40}};
  for (const auto& st : stencils)
    for (int64_t N : {(int64_t)4096, (int64_t)5000, (int64_t)16384 + 77})
      for (int periodic = 0; periodic < 2; ++periodic)
        for (int constant = 0; constant < 2; ++constant) {
          std::vector<Csr> terms{lattice(rng, N, st, periodic != 0, constant != 0)};
          char what[128];
          std::snprintf(what, sizeof what, "lattice N %lld first distance %lld periodic %d constant %d", (long long)N, (long long)st[0], periodic, constant);
          check_operator(ctx, terms, 0, QP_FMT_AUTO, what);
          check_operator(ctx, terms, 0, QP_FMT_HRB, what);
        }
  // spin chains: block-map encoding + value dictionary; the column-blocked mirror forced on an irregular operator
  for (int n : {5, 7, 10, 13}) {
    std::vector<Csr> terms{tfim(n)};
    check_operator(ctx, terms, 0, QP_FMT_AUTO, "tfim");
    check_operator(ctx, terms, 0, QP_FMT_HRB, "tfim hrb");
  }
  qp_ctx_tuning_set(ctx, "colblock", 2);
  qp_ctx_tuning_set(ctx, "cb_log2w", 8);
  for (int rep = 0; rep < 6; ++rep) {
    const int64_t nr = 300 + (int64_t)(rng() % 4000), nc = rep % 2 ? nr : 100 + (int64_t)(rng() % 6000);
    std::vector<Csr> terms{random_csr(rng, nr, nc, 12, 100000, false)};
    check_operator(ctx, terms, 0, rep % 3 == 0 ? QP_FMT_CSR : QP_FMT_RBCSR, "forced column-blocked mirror");
  }
  qp_ctx_destroy(ctx);
  std::printf("sanitizer run clean: %lld operators (%lld Hermitian-packed, %lld with a strip-walk plan, %lld with a value dictionary, %lld with a "
              "column-blocked mirror, %lld dense)\n",
              (long long)n_cases, (long long)n_hrb, (long long)n_walked, (long long)n_coded, (long long)n_cb, (long long)n_dense);
  return 0;
}
