// AddressSanitizer / UBSan harness for the library's host numerics (csrc/host_numerics.cpp): Bessel
// coefficients, Hessenberg eigenvalues of every leading block, Leja ordering, Newton divided
// differences, CSC -> CSR and row partitions, on random inputs.  Built and run by
// tests/test_cabi_host.py::test_host_numerics_under_sanitizers with g++ (CPU only; GPU sanitizers are
// not available on the test pool).
#include <cstdio>
#include <cstdlib>
#include <random>
#include "qprop_internal.h"
using qp::cplx;
int main() {
  std::mt19937_64 rng(1);
  std::normal_distribution<double> g;
  // cheby coefficients over a range of alpha
  for (double dt : {1e-14, 0.01, 0.5, 3.0, 40.0}) {
    auto a = qp::cheby_coeffs(20.0, dt, 1e-12);
    if (a.empty()) return 1;
  }
  // Hessenberg eigenvalues of all leading blocks, m = 1 .. 40
  for (int m = 1; m <= 40; ++m) {
    std::vector<cplx> H((size_t)(m + 1) * (m + 1));
    for (int c = 0; c < m + 1; ++c)
      for (int r = 0; r < m + 1; ++r) H[(size_t)c * (m + 1) + r] = (r <= c + 1) ? cplx(g(rng), g(rng)) : cplx(0);
    std::vector<cplx> ritz((size_t)m * (m + 1) / 2);
    if (qp::diagonalize_hessenberg(H.data(), m + 1, m, true, ritz.data()) != 0) return 2;
    // Leja ordering + Newton coefficients, several restarts
    std::vector<cplx> leja((size_t)8 * m + 8), a((size_t)8 * m + 8);
    int n = 0, n_a = 0;
    double radius = 0;
    for (auto& z : ritz) radius = std::max(radius, 1.2 * std::abs(z));
    for (int s = 0; s < 4; ++s) {
      std::vector<cplx> cand = ritz;
      qp::extend_leja(leja.data(), n, cand.data(), (int)cand.size(), m);
      n += m;
      int st = qp::extend_newton_coeffs(a.data(), n_a, leja.data(), QP_FUNC_EXPMI, nullptr, nullptr, n, radius);
      if (st != 0 && st != QP_E_DIVDIFF_UNDERFLOW) return 3;
      if (st != 0) break;
      n_a = n;
    }
  }
  // CSC -> CSR and row partitions on random patterns (empty rows / columns included)
  for (int trial = 0; trial < 200; ++trial) {
    const int64_t nr = 1 + rng() % 50, nc = 1 + rng() % 50;
    std::vector<int64_t> colptr{1}, rowval;
    for (int64_t c = 0; c < nc; ++c) {
      std::vector<int64_t> rows;
      for (int64_t r = 0; r < nr; ++r)
        if (rng() % 5 == 0) rows.push_back(r + 1);
      rowval.insert(rowval.end(), rows.begin(), rows.end());
      colptr.push_back(colptr.back() + (int64_t)rows.size());
    }
    std::vector<qp_c128> nz(rowval.size() + 1, qp_c128{1.0, 2.0});
    std::vector<int64_t> rp(nr + 1);
    std::vector<int32_t> col(rowval.size() + 1);
    std::vector<qp_c128> vals(rowval.size() + 1);
    if (qp::csc_to_csr(nr, nc, colptr.data(), rowval.data(), nz.data(), 1, rp.data(), col.data(), vals.data()) != 0) return 4;
    for (int parts = 1; parts <= 9; ++parts)
      for (int bal = 0; bal < 2; ++bal) {
        std::vector<int64_t> b(parts + 1);
        qp::partition_rows(rp.data(), nr, parts, bal, b.data());
        if (b[0] != 0 || b[parts] != nr) return 5;
      }
  }
  std::puts("host numerics: sanitizer run clean");
  return 0;
}
namespace qp {
int fail(int status, const char* fmt, ...) { (void)fmt; return status; }
void set_error(const char*) {}
}
