// Row-partitioned cheby! with the exchange inside the library (RCCL).
#include <dlfcn.h>

#include <rccl/rccl.h>

#include "engine.h"

// ---------------------------------------------------------------------------
// Row-partitioned cheby! with the exchange inside the library (RCCL)
// ---------------------------------------------------------------------------
namespace {

// the entry points of librccl that this file uses, resolved at run time from the library
// the caller names (so that it is the same RCCL the rest of the process uses)
struct RcclApi {
  void* handle = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;      // optional: qp_comm_info
  ncclResult_t (*CommUserRank)(const ncclComm_t, int*) = nullptr;   // optional: qp_comm_info
};

int rccl_load(const char* path, RcclApi* api) {
  if (!path || !*path) return qp::fail(QP_E_BAD_ARG, "path of librccl.so is empty");
  void* h = dlopen(path, RTLD_NOW | RTLD_GLOBAL);
  if (!h) return qp::fail(QP_E_RCCL, "dlopen(%s) failed: %s", path, dlerror());
  api->handle = h;
  api->GetUniqueId = reinterpret_cast<decltype(api->GetUniqueId)>(dlsym(h, "ncclGetUniqueId"));
  api->CommInitRank = reinterpret_cast<decltype(api->CommInitRank)>(dlsym(h, "ncclCommInitRank"));
  api->CommDestroy = reinterpret_cast<decltype(api->CommDestroy)>(dlsym(h, "ncclCommDestroy"));
  api->AllGather = reinterpret_cast<decltype(api->AllGather)>(dlsym(h, "ncclAllGather"));
  api->Send = reinterpret_cast<decltype(api->Send)>(dlsym(h, "ncclSend"));
  api->Recv = reinterpret_cast<decltype(api->Recv)>(dlsym(h, "ncclRecv"));
  api->GroupStart = reinterpret_cast<decltype(api->GroupStart)>(dlsym(h, "ncclGroupStart"));
  api->GroupEnd = reinterpret_cast<decltype(api->GroupEnd)>(dlsym(h, "ncclGroupEnd"));
  api->GetErrorString = reinterpret_cast<decltype(api->GetErrorString)>(dlsym(h, "ncclGetErrorString"));
  api->CommCount = reinterpret_cast<decltype(api->CommCount)>(dlsym(h, "ncclCommCount"));
  api->CommUserRank = reinterpret_cast<decltype(api->CommUserRank)>(dlsym(h, "ncclCommUserRank"));
  if (!api->GetUniqueId || !api->CommInitRank || !api->CommDestroy || !api->AllGather || !api->Send || !api->Recv ||
      !api->GroupStart || !api->GroupEnd || !api->GetErrorString)
    return qp::fail(QP_E_RCCL, "%s does not export the RCCL entry points", path);
  return QP_OK;
}

#define QP_RCCL(api, expr)                                                                             \
  do {                                                                                                 \
    ncclResult_t r__ = (expr);                                                                         \
    if (r__ != ncclSuccess) return qp::fail(QP_E_RCCL, "RCCL: %s failed: %s", #expr, (api).GetErrorString(r__)); \
  } while (0)

__global__ __launch_bounds__(qp::kThreads) void pack_rows_kernel(double2* __restrict__ slab,
                                                                 const double2* __restrict__ x,
                                                                 const int64_t* __restrict__ rows, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * qp::kThreads + threadIdx.x; i < n; i += (int64_t)gridDim.x * qp::kThreads)
    slab[i] = x[rows[i]];
}

}  // namespace

struct qp_comm {
  qp_ctx* ctx = nullptr;
  RcclApi api;
  ncclComm_t comm = nullptr;
  int rank = 0, world = 1;
  qp_exchange_cb cb = nullptr;   // non-null: the caller performs the exchange (no RCCL)
  void* cb_user = nullptr;
};

struct qp_sharded_cheby {
  qp_sharded_cheby_desc d;
  qp_ctx* ctx = nullptr;
  int64_t nloc = 0;
  int64_t* send_rows_dev = nullptr;   // M entries, padded with row 0
  hipStream_t side = nullptr;         // high priority: boundary blocks + collectives
  hipEvent_t ev_main = nullptr, ev_side = nullptr;
  std::vector<int> send_to, recv_from;
  bool p2p = false;
};

extern "C" {

int qp_comm_unique_id(const char* rccl_lib_path, char id_out[128]) {
  QP_TRY
  if (!id_out) return qp::fail(QP_E_BAD_ARG, "qp_comm_unique_id: NULL output");
  RcclApi api;
  QP_CHECK(rccl_load(rccl_lib_path, &api));
  ncclUniqueId id;
  static_assert(sizeof(id) == 128, "ncclUniqueId is 128 bytes");
  QP_RCCL(api, api.GetUniqueId(&id));
  std::memcpy(id_out, &id, sizeof(id));
  return QP_OK;
  QP_CATCH
}

// Communicator set-up in two phases, so that a caller can make sure EVERY rank got through the part
// that may fail locally (dlopen of librccl, symbol resolution) before any rank enters the collective
// ncclCommInitRank -- a rank that failed locally would otherwise leave the others blocked in it.
int qp_comm_prepare(qp_ctx* ctx, const char* rccl_lib_path, int rank, int world, qp_comm** out) {
  QP_TRY
  if (!ctx || !out || world < 1 || rank < 0 || rank >= world) return qp::fail(QP_E_BAD_ARG, "qp_comm_prepare: bad arguments");
  QP_CHECK(use(ctx));
  auto c = std::make_unique<qp_comm>();
  c->ctx = ctx;
  c->rank = rank;
  c->world = world;
  QP_CHECK(rccl_load(rccl_lib_path, &c->api));
  *out = c.release();
  return QP_OK;
  QP_CATCH
}

int qp_comm_connect(qp_comm* c, const char id[128]) {
  QP_TRY
  if (!c || !id) return qp::fail(QP_E_BAD_ARG, "qp_comm_connect: NULL argument");
  if (c->cb) return qp::fail(QP_E_BAD_ARG, "qp_comm_connect: a callback communicator has nothing to connect");
  if (c->comm) return qp::fail(QP_E_BAD_ARG, "qp_comm_connect: already connected");
  QP_CHECK(use(c->ctx));
  ncclUniqueId uid;
  std::memcpy(&uid, id, sizeof(uid));
  QP_RCCL(c->api, c->api.CommInitRank(&c->comm, c->world, uid, c->rank));
  return QP_OK;
  QP_CATCH
}

int qp_comm_create(qp_ctx* ctx, const char* rccl_lib_path, const char id[128], int rank, int world, qp_comm** out) {
  QP_TRY
  if (!id || !out) return qp::fail(QP_E_BAD_ARG, "qp_comm_create: bad arguments");
  qp_comm* c = nullptr;
  QP_CHECK(qp_comm_prepare(ctx, rccl_lib_path, rank, world, &c));
  const int rc = qp_comm_connect(c, id);
  if (rc != QP_OK) {
    delete c;
    return rc;
  }
  *out = c;
  return QP_OK;
  QP_CATCH
}

int qp_comm_create_callback(qp_ctx* ctx, int rank, int world, qp_exchange_cb cb, void* user, qp_comm** out) {
  QP_TRY
  if (!ctx || !out || !cb || world < 1 || rank < 0 || rank >= world) return qp::fail(QP_E_BAD_ARG, "qp_comm_create_callback: bad arguments");
  auto c = std::make_unique<qp_comm>();
  c->ctx = ctx;
  c->rank = rank;
  c->world = world;
  c->cb = cb;
  c->cb_user = user;
  *out = c.release();
  return QP_OK;
  QP_CATCH
}

// What the communicator IS, asked of RCCL itself (ncclCommCount / ncclCommUserRank of the connected communicator), so that a
// caller -- bench.py --gpus N -- can state and check "this exchange ran on an RCCL communicator of N ranks" instead of assuming it.
int qp_comm_info(const qp_comm* comm, int* world, int* rank, int* rccl_ranks, int* rccl_rank, int* is_callback) {
  QP_TRY
  if (!comm) return qp::fail(QP_E_BAD_ARG, "qp_comm_info: NULL communicator");
  if (world) *world = comm->world;
  if (rank) *rank = comm->rank;
  if (is_callback) *is_callback = comm->cb ? 1 : 0;
  int n = 0, r = -1;
  if (comm->comm) {
    if (!comm->api.CommCount || !comm->api.CommUserRank)
      return qp::fail(QP_E_RCCL, "qp_comm_info: this librccl exports neither ncclCommCount nor ncclCommUserRank");
    QP_RCCL(comm->api, comm->api.CommCount(comm->comm, &n));
    QP_RCCL(comm->api, comm->api.CommUserRank(comm->comm, &r));
  }
  if (rccl_ranks) *rccl_ranks = n;
  if (rccl_rank) *rccl_rank = r;
  return QP_OK;
  QP_CATCH
}

int qp_comm_destroy(qp_comm* comm) {
  QP_TRY
  if (!comm) return QP_OK;
  (void)hipSetDevice(comm->ctx->device);
  (void)hipDeviceSynchronize();
  if (comm->comm) (void)comm->api.CommDestroy(comm->comm);
  delete comm;
  return QP_OK;
  QP_CATCH
}

int qp_comm_allgather(qp_comm* comm, const qp_state* send, qp_state* recv, int64_t count, void* stream) {
  QP_TRY
  if (!comm || !send || !recv || count < 0 || send->n < count || recv->n < count * comm->world)
    return qp::fail(QP_E_BAD_ARG, "qp_comm_allgather: bad arguments");
  QP_CHECK(use(comm->ctx));
  hipStream_t s = stream ? (hipStream_t)stream : comm->ctx->stream;
  if (comm->cb) {
    if (comm->cb(comm->cb_user, reinterpret_cast<const qp_c128*>(send->d), count, reinterpret_cast<qp_c128*>(recv->d),
                 nullptr, -1, nullptr, 0, (void*)s) != 0)
      return qp::fail(QP_E_RCCL, "the caller's exchange callback failed");
    return QP_OK;
  }
  if (!comm->comm) return qp::fail(QP_E_BAD_ARG, "qp_comm_allgather: the communicator is not connected (qp_comm_connect)");
  QP_RCCL(comm->api, comm->api.AllGather(send->d, recv->d, (size_t)(2 * count), ncclDouble, comm->comm, s));
  return QP_OK;
  QP_CATCH
}

int qp_sharded_cheby_create(const qp_sharded_cheby_desc* desc, qp_sharded_cheby** out) {
  QP_TRY
  if (!desc || !out || !desc->op || !desc->X0 || !desc->X1 || !desc->acc) return qp::fail(QP_E_BAD_ARG, "qp_sharded_cheby_create: NULL argument");
  const qp_sharded_cheby_desc& d = *desc;
  const int64_t nloc = d.op->A.nrows, ncols = d.op->A.ncols;
  const int world = d.comm ? d.comm->world : 1;
  if (d.M < 0 || d.nsend < 0 || d.nsend > d.M || (d.nsend > 0 && !d.send_rows)) return qp::fail(QP_E_BAD_ARG, "qp_sharded_cheby_create: bad send set");
  if (d.M > 0 && !d.comm) return qp::fail(QP_E_BAD_ARG, "qp_sharded_cheby_create: an exchange needs a communicator");
  if (d.M > 0 && !d.comm->cb && !d.comm->comm) return qp::fail(QP_E_BAD_ARG, "qp_sharded_cheby_create: the communicator is not connected (qp_comm_connect)");
  if (ncols != nloc + (d.M > 0 ? (int64_t)world * d.M : 0)) return qp::fail(QP_E_BAD_ARG, "qp_sharded_cheby_create: operator has %lld columns, expected nloc + world * M = %lld", (long long)ncols, (long long)(nloc + (int64_t)world * d.M));
  if (d.X0->n != ncols || d.X1->n != ncols || d.acc->n != nloc) return qp::fail(QP_E_BAD_ARG, "qp_sharded_cheby_create: vector length mismatch");
  if (d.M > 0 && !d.direct_send && (!d.slab || d.slab->n < d.M)) return qp::fail(QP_E_BAD_ARG, "qp_sharded_cheby_create: slab too small");
  if (d.direct_send && d.M != nloc) return qp::fail(QP_E_BAD_ARG, "qp_sharded_cheby_create: direct_send needs M == nloc");
  if (d.split && (d.split->op != d.op || d.direct_send || d.M == 0)) return qp::fail(QP_E_BAD_ARG, "qp_sharded_cheby_create: split does not fit this exchange");
  for (int64_t i = 0; i < d.nsend; ++i)
    if (d.send_rows[i] < 0 || d.send_rows[i] >= nloc) return qp::fail(QP_E_BAD_ARG, "qp_sharded_cheby_create: send row out of range");
  const bool p2p = d.M > 0 && d.n_send_to >= 0;
  if (p2p) {
    if (d.n_recv_from < 0 || (d.n_send_to > 0 && !d.send_to) || (d.n_recv_from > 0 && !d.recv_from))
      return qp::fail(QP_E_BAD_ARG, "qp_sharded_cheby_create: bad neighbour lists");
    for (int i = 0; i < d.n_send_to; ++i)
      if (d.send_to[i] < 0 || d.send_to[i] >= world) return qp::fail(QP_E_BAD_ARG, "qp_sharded_cheby_create: send_to rank out of range");
    for (int i = 0; i < d.n_recv_from; ++i)
      if (d.recv_from[i] < 0 || d.recv_from[i] >= world) return qp::fail(QP_E_BAD_ARG, "qp_sharded_cheby_create: recv_from rank out of range");
  }
  qp_ctx* ctx = d.op->ctx;
  QP_CHECK(use(ctx));
  auto s = std::make_unique<qp_sharded_cheby>();
  s->d = d;
  s->d.send_rows = nullptr;   // host arrays of the caller: not kept
  s->p2p = p2p;
  if (p2p) {
    s->send_to.assign(d.send_to, d.send_to + d.n_send_to);
    s->recv_from.assign(d.recv_from, d.recv_from + d.n_recv_from);
  }
  s->d.send_to = s->d.recv_from = nullptr;
  s->ctx = ctx;
  s->nloc = nloc;
  if (d.M > 0 && !d.direct_send) {
    std::vector<int64_t> rows((size_t)d.M, 0);   // padded with row 0 (never read by anyone)
    std::copy(d.send_rows, d.send_rows + d.nsend, rows.begin());
    QP_CHECK(dev_alloc(&s->send_rows_dev, (size_t)d.M));
    QP_HIP(hipMemcpy(s->send_rows_dev, rows.data(), rows.size() * sizeof(int64_t), hipMemcpyHostToDevice));
  }
  if (d.split) {
    int least = 0, greatest = 0;
    QP_HIP(hipDeviceGetStreamPriorityRange(&least, &greatest));
    QP_HIP(hipStreamCreateWithPriority(&s->side, hipStreamNonBlocking, greatest));
    QP_HIP(hipEventCreateWithFlags(&s->ev_main, hipEventDisableTiming));
    QP_HIP(hipEventCreateWithFlags(&s->ev_side, hipEventDisableTiming));
  }
  *out = s.release();
  return QP_OK;
  QP_CATCH
}

int qp_sharded_cheby_destroy(qp_sharded_cheby* s) {
  QP_TRY
  if (!s) return QP_OK;
  (void)hipSetDevice(s->ctx->device);
  (void)hipDeviceSynchronize();
  if (s->send_rows_dev) (void)hipFree(s->send_rows_dev);
  if (s->side) (void)hipStreamDestroy(s->side);
  if (s->ev_main) (void)hipEventDestroy(s->ev_main);
  if (s->ev_side) (void)hipEventDestroy(s->ev_side);
  delete s;
  return QP_OK;
  QP_CATCH
}

int qp_sharded_cheby_step(qp_sharded_cheby* s, const double* a, int n_coeffs, double Delta, double E_min, double dt) {
  QP_TRY
  if (!s || !a) return qp::fail(QP_E_BAD_ARG, "qp_sharded_cheby_step: NULL argument");
  if (n_coeffs < 2) return qp::fail(QP_E_TOO_FEW_COEFFS, "Need at least 2 Chebychev coefficients");
  if (!(Delta > 0) || dt == 0.0) return qp::fail(QP_E_BAD_ARG, "qp_sharded_cheby_step: Delta must be positive, dt non-zero");
  const qp_sharded_cheby_desc& d = s->d;
  qp_ctx* ctx = s->ctx;
  QP_CHECK(use(ctx));
  const int64_t nloc = s->nloc;
  const int nterms = n_coeffs - 1;
  const double beta = (Delta / 2) + E_min;                               // src/cheby.jl:156
  cplx c = (dt > 0) ? cplx(0, -2.0) / Delta : cplx(0, 2.0) / Delta;      // :158-162
  const cplx phase = std::exp(cplx(0, -1) * beta * dt);                  // :211
  std::vector<qp_acc_defer> sched((size_t)nterms);
  acc_schedule(a, n_coeffs, ctx->tun.acc_defer != 0, sched.data());
  const bool exchanging = d.M > 0;
  const bool overlap = d.split != nullptr;
  hipStream_t S_c = ctx->stream;
  hipStream_t S_x = overlap ? s->side : S_c;   // the stream the collectives are ordered on
  qp_state* X[2] = {d.X0, d.X1};
  qp_state xloc[2] = {qp_state{ctx, d.X0->d, nloc, false}, qp_state{ctx, d.X1->d, nloc, false}};

  // fill the ghost slabs of X[k] with the other ranks' send rows of the same vector
  auto exchange = [&](int k, bool packed) -> int {
    if (!exchanging) return QP_OK;
    const double2* send = nullptr;
    if (d.direct_send) {
      send = X[k]->d;
    } else {
      send = d.slab->d;
      if (!packed) {
        const int grid = (int)std::min<int64_t>((d.M + qp::kThreads - 1) / qp::kThreads, 1024);
        hipLaunchKernelGGL(pack_rows_kernel, dim3(grid), dim3(qp::kThreads), 0, S_x, d.slab->d, X[k]->d, s->send_rows_dev, d.M);
        QP_HIP(hipGetLastError());
        ctx->stats.n_launch++;
      }
    }
    if (d.comm->cb) {   // the caller's transport
      const int rc = d.comm->cb(d.comm->cb_user, reinterpret_cast<const qp_c128*>(send), d.M,
                                reinterpret_cast<qp_c128*>(X[k]->d + nloc), s->p2p ? s->send_to.data() : nullptr,
                                s->p2p ? (int)s->send_to.size() : -1, s->p2p ? s->recv_from.data() : nullptr,
                                s->p2p ? (int)s->recv_from.size() : 0, (void*)S_x);
      if (rc != 0) return qp::fail(QP_E_RCCL, "the caller's exchange callback failed (%d)", rc);
      return QP_OK;
    }
    const RcclApi& api = d.comm->api;
    if (s->p2p) {   // neighbour exchange: my slab to who reads it, their slabs into their ghost slots
      QP_RCCL(api, api.GroupStart());
      // a failure inside the group must not leave the thread inside an open group (later collectives of
      // this thread -- torch's included -- would be queued and never launched): remember the first one,
      // close the group, then report
      ncclResult_t first = ncclSuccess;
      const char* what = "";
      for (int o : s->recv_from) {
        if (first != ncclSuccess) break;
        first = api.Recv(X[k]->d + nloc + (int64_t)o * d.M, (size_t)(2 * d.M), ncclDouble, o, d.comm->comm, S_x);
        what = "ncclRecv";
      }
      for (int o : s->send_to) {
        if (first != ncclSuccess) break;
        first = api.Send(send, (size_t)(2 * d.M), ncclDouble, o, d.comm->comm, S_x);
        what = "ncclSend";
      }
      const ncclResult_t end = api.GroupEnd();
      if (first != ncclSuccess) return qp::fail(QP_E_RCCL, "RCCL: %s failed: %s", what, api.GetErrorString(first));
      if (end != ncclSuccess) return qp::fail(QP_E_RCCL, "RCCL: ncclGroupEnd failed: %s", api.GetErrorString(end));
    } else {
      QP_RCCL(api, api.AllGather(send, X[k]->d + nloc, (size_t)(2 * d.M), ncclDouble, d.comm->comm, S_x));
    }
    return QP_OK;
  };

  // an in-launch wait of an EARLIER step that gave up (bounded spin) left a wrong state behind: the flag
  // sits in host-visible memory, so looking at it costs no synchronisation
  if (overlap) QP_CHECK(split_timed_out(d.split));
  if (overlap) {   // join: the side stream starts after everything queued on the main stream
    QP_HIP(hipEventRecord(s->ev_main, S_c));
    QP_HIP(hipStreamWaitEvent(S_x, s->ev_main, 0));
  }
  QP_CHECK(exchange(0, false));
  bool updated = false;
  bool result_in_acc = true;
  for (int m = 1; m <= nterms; ++m) {
    const bool last = (m == nterms);
    const int xi = (m % 2 == 1) ? 0 : 1, oi = 1 - xi;
    const qp_acc_defer& df = sched[(size_t)m - 1];
    const qp_c128 cc{c.real(), c.imag()};
    const qp_c128 ph = last ? qp_c128{phase.real(), phase.imag()} : qp_c128{1.0, 0.0};
    const qp_state* v0 = (m == 1) ? nullptr : &xloc[oi];
    qp_state* vout = last ? nullptr : &xloc[oi];
    const qp_state* acc_in = (updated && !df.skip) ? d.acc : nullptr;
    // the state buffer X0 may be written only while it is not being gathered
    qp_state* out = (m > 1 && last && xi == 1) ? &xloc[0] : d.acc;
    if (!df.skip && m > 1) result_in_acc = (out == d.acc);
    const double a_prev = updated ? 0.0 : a[0];
    if (overlap) {
      QP_CHECK(qp_cheby_term_split(d.op, d.split, (void*)S_x, m == 1 ? 1 : 0, X[xi], 0, v0, vout, acc_in,
                                   df.skip ? nullptr : out, last ? nullptr : d.slab, cc, beta, a_prev, a[m], ph, &df));
    } else {
      QP_CHECK(qp_cheby_term(d.op, X[xi], 0, v0, vout, acc_in, df.skip ? nullptr : out, cc, beta, a_prev, a[m], ph, &df));
    }
    updated = updated || !df.skip;
    if (!last) QP_CHECK(exchange(oi, overlap));
    if (m == 1) c *= 2.0;                                                 // :184
  }
  if (overlap) {   // join back: later work on the main stream sees everything the side stream did
    QP_HIP(hipEventRecord(s->ev_side, S_x));
    QP_HIP(hipStreamWaitEvent(S_c, s->ev_side, 0));
  }
  if (result_in_acc)
    QP_HIP(hipMemcpyAsync(d.X0->d, d.acc->d, (size_t)nloc * sizeof(double2), hipMemcpyDeviceToDevice, S_c));
  ctx->stats.n_cheby_steps++;
  if (overlap) QP_CHECK(split_timed_out(d.split));
  return QP_OK;
  QP_CATCH
}

}  // extern "C"

