// Data-parallel collectives of the training step behind the C ABI: an RCCL communicator owned by the library, its collectives on a
// library-owned stream ordered against the caller's compute stream with events (replaces the DDP reducer that
// accelerate wraps the reference's models with — pretrain_e4t.py:410-412, :648 / tuning_e4t.py:197-200, :328).
//
// RCCL is NOT a link dependency: a process that trains under torch.distributed already holds one copy of librccl.so (torch's own), and
// a second copy with its own bootstrap state is what must not happen.  The entry points resolve the six RCCL symbols they use from
// the copy already in the process (RTLD_NOLOAD), and only load one by name when there is none.  Without any librccl.so every
// e4t_comm_* call fails with -38 and a message; the kernels of the library do not depend on it.
#include "common.h"
#include "../../include/e4t_hip.h"
#include <dlfcn.h>
#include <stdlib.h>

namespace {

typedef struct { char internal[128]; } rccl_unique_id;      // ncclUniqueId (rccl.h: NCCL_UNIQUE_ID_BYTES 128)
typedef void* rccl_comm;
typedef int (*fn_get_unique_id)(rccl_unique_id*);
typedef int (*fn_comm_init_rank)(rccl_comm*, int, rccl_unique_id, int);
typedef int (*fn_comm_destroy)(rccl_comm);
typedef int (*fn_all_reduce)(const void*, void*, size_t, int, int, rccl_comm, hipStream_t);
typedef int (*fn_all_gather)(const void*, void*, size_t, int, rccl_comm, hipStream_t);
typedef const char* (*fn_error_string)(int);

struct Rccl {
  void* handle = nullptr;
  fn_get_unique_id get_unique_id = nullptr;
  fn_comm_init_rank comm_init_rank = nullptr;
  fn_comm_destroy comm_destroy = nullptr;
  fn_all_reduce all_reduce = nullptr;
  fn_all_gather all_gather = nullptr;
  fn_error_string error_string = nullptr;
  int state = 0;          // 0 not tried, 1 resolved, -1 unavailable
  char why[256] = "";
};
Rccl g_rccl;

int rccl_resolve() {
  if (g_rccl.state) return g_rccl.state;
  const char* names[] = {getenv("E4T_RCCL_LIB"), "librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"};
  void* h = nullptr;
  for (int pass = 0; pass < 2 && !h; ++pass)               // pass 0: a copy already mapped into the process; pass 1: load one
    for (const char* n : names) {
      if (!n || !n[0]) continue;
      h = dlopen(n, RTLD_NOW | RTLD_LOCAL | (pass == 0 ? RTLD_NOLOAD : 0));
      if (h) break;
    }
  if (!h) {
    snprintf(g_rccl.why, sizeof(g_rccl.why), "no librccl.so in the process or on the loader path (%s)", dlerror());
    return g_rccl.state = -1;
  }
  g_rccl.handle = h;
  g_rccl.get_unique_id = (fn_get_unique_id)dlsym(h, "ncclGetUniqueId");
  g_rccl.comm_init_rank = (fn_comm_init_rank)dlsym(h, "ncclCommInitRank");
  g_rccl.comm_destroy = (fn_comm_destroy)dlsym(h, "ncclCommDestroy");
  g_rccl.all_reduce = (fn_all_reduce)dlsym(h, "ncclAllReduce");
  g_rccl.all_gather = (fn_all_gather)dlsym(h, "ncclAllGather");
  g_rccl.error_string = (fn_error_string)dlsym(h, "ncclGetErrorString");
  if (!g_rccl.get_unique_id || !g_rccl.comm_init_rank || !g_rccl.comm_destroy || !g_rccl.all_reduce || !g_rccl.all_gather || !g_rccl.error_string) {
    snprintf(g_rccl.why, sizeof(g_rccl.why), "librccl.so lacks one of ncclGetUniqueId / CommInitRank / CommDestroy / AllReduce / AllGather / GetErrorString");
    return g_rccl.state = -1;
  }
  return g_rccl.state = 1;
}

// rccl.h: ncclFloat32 = 7, ncclBfloat16 = 9; ncclSum 0, ncclMax 2, ncclMin 3, ncclAvg 4
int rccl_dtype(int dtype) { return dtype == E4T_COMM_F32 ? 7 : dtype == E4T_COMM_BF16 ? 9 : -1; }
int rccl_op(int op) { return op == E4T_COMM_SUM ? 0 : op == E4T_COMM_MAX ? 2 : op == E4T_COMM_MIN ? 3 : op == E4T_COMM_AVG ? 4 : -1; }

}  // namespace

struct e4t_comm {
  rccl_comm comm = nullptr;
  hipStream_t stream = nullptr;      // the library's collective stream
  hipEvent_t ready = nullptr;        // recorded on the caller's stream: the buffer is final
  hipEvent_t done = nullptr;         // recorded on the collective stream after the last collective issued
  int rank = 0, world = 1, device = 0;
  long long issued = 0;
};

#define RCCL_OR_FAIL()                                                             \
  do {                                                                             \
    if (rccl_resolve() < 0) E4T_FAIL(-38, "e4t_comm: %s", g_rccl.why);             \
  } while (0)
#define RCCL_CALL(name, expr)                                                      \
  do {                                                                             \
    int _r = (expr);                                                               \
    if (_r != 0) E4T_FAIL(-5, "%s: %s", name, g_rccl.error_string(_r));            \
  } while (0)
#define HIP_CALL(name, expr)                                                       \
  do {                                                                             \
    hipError_t _e = (expr);                                                        \
    if (_e != hipSuccess) E4T_FAIL(-5, "%s: %s", name, hipGetErrorString(_e));     \
  } while (0)

extern "C" int e4t_comm_unique_id(void* id128) {
  E4T_REQUIRE(id128 != nullptr, "comm_unique_id: NULL buffer");
  RCCL_OR_FAIL();
  RCCL_CALL("ncclGetUniqueId", g_rccl.get_unique_id((rccl_unique_id*)id128));
  return 0;
}

extern "C" int e4t_comm_init(e4t_comm_t* out, const void* id128, int rank, int world) {
  E4T_REQUIRE(out && id128, "comm_init: NULL argument");
  E4T_REQUIRE(world >= 1 && rank >= 0 && rank < world, "comm_init: rank %d of world %d", rank, world);
  RCCL_OR_FAIL();
  e4t_comm* c = new e4t_comm();
  c->rank = rank;
  c->world = world;
  int rc = 0;
  do {
    if (hipGetDevice(&c->device) != hipSuccess) { e4t_set_error("comm_init: no HIP device"); rc = -19; break; }
    // the highest stream priority: a collective that waits behind queued compute kernels delays the optimizer's tail
    int lo = 0, hi = 0;
    (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
    if (hipStreamCreateWithPriority(&c->stream, hipStreamNonBlocking, hi) != hipSuccess ||
        hipEventCreateWithFlags(&c->ready, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->done, hipEventDisableTiming) != hipSuccess) { e4t_set_error("comm_init: stream / event creation failed"); rc = -5; break; }
    rccl_unique_id id;
    memcpy(&id, id128, sizeof(id));
    int r = g_rccl.comm_init_rank(&c->comm, world, id, rank);
    if (r != 0) {
      char b[256];
      snprintf(b, sizeof(b), "ncclCommInitRank(rank %d / %d): %s", rank, world, g_rccl.error_string(r));
      e4t_set_error(b);
      rc = -5;
      break;
    }
  } while (0);
  if (rc != 0) {
    if (c->done) (void)hipEventDestroy(c->done);
    if (c->ready) (void)hipEventDestroy(c->ready);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
    return rc;
  }
  *out = c;
  return 0;
}

// compute stream -> collective stream: everything the caller has queued so far (the backward kernels that produced the buffer)
static int order_after_caller(e4t_comm* c, hipStream_t caller) {
  HIP_CALL("hipEventRecord", hipEventRecord(c->ready, caller));
  HIP_CALL("hipStreamWaitEvent", hipStreamWaitEvent(c->stream, c->ready, 0));
  return 0;
}

extern "C" int e4t_comm_allreduce(e4t_comm_t c, void* buf, long long count, int dtype, int op, e4t_stream stream) {
  E4T_REQUIRE(c && c->comm, "comm_allreduce: communicator not initialised");
  E4T_REQUIRE(count >= 0 && (buf || count == 0), "comm_allreduce: NULL buffer");
  E4T_REQUIRE(rccl_dtype(dtype) >= 0 && rccl_op(op) >= 0, "comm_allreduce: dtype %d / op %d", dtype, op);
  if (count == 0) return 0;
  int rc = order_after_caller(c, (hipStream_t)stream);
  if (rc) return rc;
  RCCL_CALL("ncclAllReduce", g_rccl.all_reduce(buf, buf, (size_t)count, rccl_dtype(dtype), rccl_op(op), c->comm, c->stream));
  HIP_CALL("hipEventRecord", hipEventRecord(c->done, c->stream));
  c->issued++;
  return 0;
}

extern "C" int e4t_comm_allgather(e4t_comm_t c, const void* send, void* recv, long long count, int dtype, e4t_stream stream) {
  E4T_REQUIRE(c && c->comm, "comm_allgather: communicator not initialised");
  E4T_REQUIRE(count >= 0 && ((send && recv) || count == 0), "comm_allgather: NULL buffer");
  E4T_REQUIRE(rccl_dtype(dtype) >= 0, "comm_allgather: dtype %d", dtype);
  if (count == 0) return 0;
  int rc = order_after_caller(c, (hipStream_t)stream);
  if (rc) return rc;
  RCCL_CALL("ncclAllGather", g_rccl.all_gather(send, recv, (size_t)count, rccl_dtype(dtype), c->comm, c->stream));
  HIP_CALL("hipEventRecord", hipEventRecord(c->done, c->stream));
  c->issued++;
  return 0;
}

// collective stream -> compute stream: what the caller queues next sees every collective issued so far.  No host wait.
extern "C" int e4t_comm_wait(e4t_comm_t c, e4t_stream stream) {
  E4T_REQUIRE(c && c->comm, "comm_wait: communicator not initialised");
  if (c->issued == 0) return 0;
  HIP_CALL("hipStreamWaitEvent", hipStreamWaitEvent((hipStream_t)stream, c->done, 0));
  return 0;
}

extern "C" int e4t_comm_info(e4t_comm_t c, int* rank, int* world, e4t_stream* comm_stream) {
  E4T_REQUIRE(c != nullptr, "comm_info: NULL communicator");
  if (rank) *rank = c->rank;
  if (world) *world = c->world;
  if (comm_stream) *comm_stream = (e4t_stream)c->stream;
  return 0;
}

extern "C" int e4t_comm_destroy(e4t_comm_t c) {
  if (!c) return 0;
  int rc = 0;
  if (c->stream) (void)hipStreamSynchronize(c->stream);
  if (c->comm && g_rccl.comm_destroy) {
    int r = g_rccl.comm_destroy(c->comm);
    if (r != 0) { e4t_set_error(g_rccl.error_string(r)); rc = -5; }
  }
  if (c->done) (void)hipEventDestroy(c->done);
  if (c->ready) (void)hipEventDestroy(c->ready);
  if (c->stream) (void)hipStreamDestroy(c->stream);
  delete c;
  return rc;
}
