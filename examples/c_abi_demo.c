/* A plain-C consumer of the C ABI (include/qprop.h): no Python, no torch, no HIP headers.
 * What a foreign-language binding (the Julia package extension of INTEGRATION.md) does, written out in C:
 *   - hand a Julia-style 1-based CSC Hermitian matrix to the library (the index work happens inside),
 *   - advance a state with the Chebychev propagator (cheby!, src/cheby.jl:150-213) forward and back,
 *   - advance the same state with the restarted-Arnoldi Newton propagator (newton!, src/newton.jl:246-385),
 *   - check: norm conserved, forward + backward = identity, Newton == Cheby to 1e-10.
 * Build:  gcc -std=c11 -O2 -Iinclude examples/c_abi_demo.c -o c_abi_demo \
 *             -Lquantumpropagators.jl_amd/lib -lqprop_hip -Wl,-rpath,$PWD/quantumpropagators.jl_amd/lib -lm
 * Exit status 0 = all checks passed. */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include "qprop.h"

#define CK(call)                                                                     \
  do {                                                                               \
    int st_ = (call);                                                                \
    if (st_ != QP_OK) {                                                              \
      fprintf(stderr, "%s -> %s: %s\n", #call, qp_status_name(st_), qp_last_error()); \
      return 1;                                                                      \
    }                                                                                \
  } while (0)

static double urand(uint64_t* s) {   /* splitmix64 -> [0, 1) */
  uint64_t z = (*s += 0x9e3779b97f4a7c15ULL);
  z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ULL;
  z = (z ^ (z >> 27)) * 0x94d049bb133111ebULL;
  z ^= z >> 31;
  return (double)(z >> 11) / 9007199254740992.0;
}

int main(void) {
  const int64_t N = 6000;
  const int offs[4] = {1, 7, 64, 500};
  const double rho = 4.0;              /* Gershgorin: spectrum inside [-rho, rho] */
  /* Hermitian H, periodic offsets, stored the Julia way: 1-based CSC, Int64 indices, ComplexF64 values */
  const int64_t nnz = N * 8;
  int64_t* colptr = malloc(sizeof(int64_t) * (size_t)(N + 1));
  int64_t* rowval = malloc(sizeof(int64_t) * (size_t)nnz);
  qp_c128* nzval = malloc(sizeof(qp_c128) * (size_t)nnz);
  qp_c128* up = malloc(sizeof(qp_c128) * (size_t)(N * 4));   /* H[i, (i + d_k) % N] */
  uint64_t seed = 20260612;
  for (int64_t i = 0; i < N * 4; ++i) {
    const double r = rho / 8.0 * urand(&seed), ph = 6.283185307179586 * urand(&seed);
    up[i].re = r * cos(ph);
    up[i].im = r * sin(ph);
  }
  for (int64_t j = 0; j < N; ++j) {      /* column j: rows (j - d) with H[j-d, j] = up[(j-d), k], rows (j + d) conj */
    int64_t rows[8];
    qp_c128 v[8];
    for (int k = 0; k < 4; ++k) {
      const int64_t a = (j - offs[k] + N) % N, b = (j + offs[k]) % N;
      rows[2 * k] = a;
      v[2 * k] = up[a * 4 + k];
      rows[2 * k + 1] = b;
      v[2 * k + 1].re = up[j * 4 + k].re;
      v[2 * k + 1].im = -up[j * 4 + k].im;
    }
    for (int a = 1; a < 8; ++a)          /* ascending row order within the column */
      for (int b = a; b > 0 && rows[b - 1] > rows[b]; --b) {
        const int64_t t = rows[b];
        rows[b] = rows[b - 1];
        rows[b - 1] = t;
        const qp_c128 tv = v[b];
        v[b] = v[b - 1];
        v[b - 1] = tv;
      }
    colptr[j] = 8 * j + 1;
    for (int a = 0; a < 8; ++a) {
      rowval[8 * j + a] = rows[a] + 1;
      nzval[8 * j + a] = v[a];
    }
  }
  colptr[N] = nnz + 1;

  qp_ctx* ctx = NULL;
  qp_matrix* mat = NULL;
  qp_operator* op = NULL;
  CK(qp_ctx_create(0, NULL, &ctx));
  CK(qp_matrix_create(ctx, N, N, nnz, colptr, rowval, nzval, QP_VAL_C128, QP_LAYOUT_CSC, 1, QP_FMT_AUTO, &mat));
  qp_matrix* ops[1] = {mat};
  CK(qp_operator_create(ctx, ops, 1, 0, QP_FMT_AUTO, &op));
  int fmt = -1;
  int64_t nr = 0, nc = 0, nz = 0;
  CK(qp_operator_info(op, &nr, &nc, &nz, &fmt));
  printf("operator %lld x %lld, %lld entries, device format %d (3 = Hermitian-packed row blocks)\n", (long long)nr,
         (long long)nc, (long long)nz, fmt);

  qp_c128* psi0 = malloc(sizeof(qp_c128) * (size_t)N);
  double n2 = 0.0;
  for (int64_t i = 0; i < N; ++i) {
    psi0[i].re = urand(&seed) - 0.5;
    psi0[i].im = urand(&seed) - 0.5;
    n2 += psi0[i].re * psi0[i].re + psi0[i].im * psi0[i].im;
  }
  for (int64_t i = 0; i < N; ++i) {
    psi0[i].re /= sqrt(n2);
    psi0[i].im /= sqrt(n2);
  }
  qp_state *psi = NULL, *phi = NULL;
  CK(qp_state_create(ctx, N, &psi));
  CK(qp_state_create(ctx, N, &phi));
  CK(qp_state_upload(psi, psi0));
  CK(qp_state_upload(phi, psi0));

  /* cheby!: Delta = 2 rho, E_min = -rho, dt = 0.8 */
  const double Delta = 2 * rho, E_min = -rho, dt = 0.8;
  double a[512];
  int na = 0;
  CK(qp_cheby_coeffs(Delta, dt, 1e-12, a, 512, &na));
  qp_cheby* cw = NULL;
  CK(qp_cheby_create(ctx, N, &cw));
  const int nsteps = 10;
  for (int k = 0; k < nsteps; ++k) CK(qp_cheby_step(cw, op, psi, a, na, Delta, E_min, dt, dt, 1e-12, 0));
  double nrm = 0.0;
  CK(qp_norm(psi, &nrm));
  printf("cheby!: %d coefficients, |psi| after %d steps = %.15f\n", na, nsteps, nrm);
  int bad = fabs(nrm - 1.0) > 1e-12;

  /* newton! on the copy, same steps */
  qp_newton* nw = NULL;
  CK(qp_newton_create(ctx, N, 12, &nw));
  qp_newton_stats ns;
  int restarts = 0;
  for (int k = 0; k < nsteps; ++k) {
    CK(qp_newton_step(nw, op, phi, dt, QP_FUNC_EXPMI, NULL, NULL, 1e-14, 1e-12, 50, &ns));
    restarts += ns.restarts;
  }
  qp_c128 minus1 = {-1.0, 0.0};
  CK(qp_axpy(minus1, psi, phi));          /* phi <- phi - psi */
  double diff = 0.0;
  CK(qp_norm(phi, &diff));
  printf("newton!: %d restarts in %d steps, |psi_newton - psi_cheby| = %.2e\n", restarts, nsteps, diff);
  bad |= diff > 1e-10;

  /* backward: the same steps with dt < 0 return to psi0 */
  for (int k = 0; k < nsteps; ++k) CK(qp_cheby_step(cw, op, psi, a, na, Delta, E_min, -dt, dt, 1e-12, 0));
  qp_c128* back = malloc(sizeof(qp_c128) * (size_t)N);
  CK(qp_state_download(psi, back));
  double d2 = 0.0;
  for (int64_t i = 0; i < N; ++i) {
    const double x = back[i].re - psi0[i].re, y = back[i].im - psi0[i].im;
    d2 += x * x + y * y;
  }
  printf("forward + backward: |psi - psi0| = %.2e\n", sqrt(d2));
  bad |= sqrt(d2) > 1e-11;

  /* error behaviour across the boundary: the reference's assertion on dt (src/cheby.jl:157) */
  const int st = qp_cheby_step(cw, op, psi, a, na, Delta, E_min, 0.5 * dt, dt, 1e-12, 0);
  printf("dt mismatch -> %s (\"%s\")\n", qp_status_name(st), qp_last_error());
  bad |= st != QP_E_DT_MISMATCH;

  qp_stats stats;
  CK(qp_stats_get(ctx, &stats));
  printf("stats: %llu mat-vecs, %llu kernel launches\n", (unsigned long long)stats.n_matvec,
         (unsigned long long)stats.n_kernel_launches);
  CK(qp_newton_destroy(nw));
  CK(qp_cheby_destroy(cw));
  CK(qp_state_destroy(phi));
  CK(qp_state_destroy(psi));
  CK(qp_operator_destroy(op));
  CK(qp_matrix_destroy(mat));
  CK(qp_ctx_destroy(ctx));
  free(back);
  free(psi0);
  free(up);
  free(nzval);
  free(rowval);
  free(colptr);
  printf(bad ? "FAILED\n" : "C ABI demo ok\n");
  return bad;
}
