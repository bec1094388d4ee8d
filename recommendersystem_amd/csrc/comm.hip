// Communicator: RCCL bound at run time + the in-process rank group (comm.hpp).
#include <dlfcn.h>
#include <stdlib.h>
#include <string.h>

#include "comm.hpp"

namespace rsys {

// ---------------------------------------------------------------- RCCL (librccl.so.1), bound with dlopen
struct RcclApi {
  void* lib = nullptr;
  int (*GetUniqueId)(ncclUniqueId_t*) = nullptr;
  int (*CommInitRank)(ncclComm_t_*, int, ncclUniqueId_t, int) = nullptr;
  int (*CommDestroy)(ncclComm_t_) = nullptr;
  int (*AllReduce)(const void*, void*, size_t, int, int, ncclComm_t_, hipStream_t) = nullptr;
  int (*AllGather)(const void*, void*, size_t, int, ncclComm_t_, hipStream_t) = nullptr;
  int (*ReduceScatter)(const void*, void*, size_t, int, int, ncclComm_t_, hipStream_t) = nullptr;
  int (*Send)(const void*, size_t, int, int, ncclComm_t_, hipStream_t) = nullptr;
  int (*Recv)(void*, size_t, int, int, ncclComm_t_, hipStream_t) = nullptr;
  int (*GroupStart)() = nullptr;
  int (*GroupEnd)() = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
  int (*GetVersion)(int*) = nullptr;
};
static RcclApi g_rccl;
enum { NCCL_INT8 = 0, NCCL_FLOAT32 = 7, NCCL_FLOAT64 = 8 };

int load_rccl() {
  if (g_rccl.lib) return RSYS_OK;
  const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
  for (const char* n : names) { g_rccl.lib = dlopen(n, RTLD_NOW | RTLD_LOCAL); if (g_rccl.lib) break; }
  if (!g_rccl.lib) { set_error(std::string("cannot load librccl: ") + dlerror()); return RSYS_ERR_COMM; }
  void* L = g_rccl.lib;
  g_rccl.GetUniqueId = (int (*)(ncclUniqueId_t*))dlsym(L, "ncclGetUniqueId");
  g_rccl.CommInitRank = (int (*)(ncclComm_t_*, int, ncclUniqueId_t, int))dlsym(L, "ncclCommInitRank");
  g_rccl.CommDestroy = (int (*)(ncclComm_t_))dlsym(L, "ncclCommDestroy");
  g_rccl.AllReduce = (int (*)(const void*, void*, size_t, int, int, ncclComm_t_, hipStream_t))dlsym(L, "ncclAllReduce");
  g_rccl.AllGather = (int (*)(const void*, void*, size_t, int, ncclComm_t_, hipStream_t))dlsym(L, "ncclAllGather");
  g_rccl.ReduceScatter = (int (*)(const void*, void*, size_t, int, int, ncclComm_t_, hipStream_t))dlsym(L, "ncclReduceScatter");
  g_rccl.Send = (int (*)(const void*, size_t, int, int, ncclComm_t_, hipStream_t))dlsym(L, "ncclSend");
  g_rccl.Recv = (int (*)(void*, size_t, int, int, ncclComm_t_, hipStream_t))dlsym(L, "ncclRecv");
  g_rccl.GroupStart = (int (*)())dlsym(L, "ncclGroupStart");
  g_rccl.GroupEnd = (int (*)())dlsym(L, "ncclGroupEnd");
  g_rccl.GetErrorString = (const char* (*)(int))dlsym(L, "ncclGetErrorString");
  g_rccl.GetVersion = (int (*)(int*))dlsym(L, "ncclGetVersion");
  if (!g_rccl.GetUniqueId || !g_rccl.CommInitRank || !g_rccl.CommDestroy || !g_rccl.AllReduce || !g_rccl.AllGather ||
      !g_rccl.Send || !g_rccl.Recv || !g_rccl.GroupStart || !g_rccl.GroupEnd) {
    set_error("librccl is missing a required symbol"); return RSYS_ERR_COMM;
  }
  return RSYS_OK;
}
int comm_rccl_version() {
  int v = 0;
  if (g_rccl.lib && g_rccl.GetVersion && g_rccl.GetVersion(&v) == 0) return v;
  return 0;
}
#define NCCL_CHECK(expr)                                                                          \
  do {                                                                                            \
    int _r = (expr);                                                                              \
    if (_r != 0) {                                                                                \
      set_error(std::string(#expr) + ": " + (g_rccl.GetErrorString ? g_rccl.GetErrorString(_r) : "rccl error")); \
      return RSYS_ERR_COMM;                                                                       \
    }                                                                                             \
  } while (0)

int comm_unique_id(unsigned char id_buf[128]) {
  int rc = load_rccl();
  if (rc) return rc;
  ncclUniqueId_t id;
  NCCL_CHECK(g_rccl.GetUniqueId(&id));
  memcpy(id_buf, id.internal, 128);
  return RSYS_OK;
}

static int comm_common_init(rsys_comm* c) {
  HIP_CHECK(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
  HIP_CHECK(hipEventCreateWithFlags(&c->ev_ready, hipEventDisableTiming));
  HIP_CHECK(hipEventCreateWithFlags(&c->ev_done, hipEventDisableTiming));
  HIP_CHECK(hipEventCreateWithFlags(&c->ev_head, hipEventDisableTiming));
  HIP_CHECK(hipMalloc((void**)&c->scratch, 64 * sizeof(double)));
  return RSYS_OK;
}

int comm_init_rccl(const unsigned char id_buf[128], int rank, int world, int device, rsys_comm** out) {
  ARG_CHECK(world >= 1 && rank >= 0 && rank < world, "rank/world");
  int rc = load_rccl();
  if (rc) return rc;
  HIP_CHECK(hipSetDevice(device));
  rsys_comm* c = new rsys_comm();
  c->rank = rank; c->world = world; c->device = device;
  ncclUniqueId_t id;
  memcpy(id.internal, id_buf, 128);
  {
    const int r = g_rccl.CommInitRank(&c->comm, world, id, rank);
    if (r != 0) {
      set_error(std::string("ncclCommInitRank: ") + (g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "rccl error"));
      delete c;
      return RSYS_ERR_COMM;
    }
  }
  rc = comm_common_init(c);
  if (rc) { comm_destroy(c); return rc; }   // (frees whatever the failed step had created: communicator, stream, events)
  c->force = sw().force_rccl == 1;
  *out = c;
  return RSYS_OK;
}

// ---------------------------------------------------------------- in-process group
int LocalGroup::barrier() {
  std::unique_lock<std::mutex> lk(mu);
  if (broken) { set_error("in-process rank group: another rank failed inside a collective"); return RSYS_ERR_COMM; }
  const unsigned long long gen = generation;
  if (++arrived == world) { arrived = 0; ++generation; cv.notify_all(); return RSYS_OK; }
  cv.wait(lk, [&] { return generation != gen || broken; });
  if (generation == gen) { set_error("in-process rank group: another rank failed inside a collective"); return RSYS_ERR_COMM; }
  return RSYS_OK;
}
static int local_fail(LocalGroup* g, int rc) {
  std::lock_guard<std::mutex> lk(g->mu);
  g->broken = true;
  g->cv.notify_all();
  return rc;
}

int comm_init_local(LocalGroup* g, int rank, rsys_comm** out) {
  ARG_CHECK(g != nullptr && rank >= 0 && rank < g->world, "in-process group: rank");
  HIP_CHECK(hipSetDevice(g->device));
  ARG_CHECK(g->world <= 16, "in-process group: at most 16 ranks (reduce_ptrs_kernel's pointer list)");
  rsys_comm* c = new rsys_comm();
  c->rank = rank; c->world = g->world; c->device = g->device; c->lg = g;
  { std::lock_guard<std::mutex> lk(g->mu); ++g->refs; }    // (comm_destroy drops it, also on the failure paths below)
  int rc = comm_common_init(c);
  if (rc) { comm_destroy(c); return rc; }
  LocalRankSlot& sl = g->slot[rank];
  if (hipEventCreateWithFlags(&sl.ready, hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&sl.done, hipEventDisableTiming) != hipSuccess) {
    set_error("in-process group: hipEventCreate failed");
    comm_destroy(c);
    return RSYS_ERR_HIP;
  }
  *out = c;
  return RSYS_OK;
}

int comm_destroy(rsys_comm* c) {
  if (!c) return RSYS_OK;
  hipSetDevice(c->device);
  hipDeviceSynchronize();
  if (c->comm) g_rccl.CommDestroy(c->comm);
  if (c->lg) {
    LocalRankSlot& sl = c->lg->slot[c->rank];
    if (sl.ready) hipEventDestroy(sl.ready);
    if (sl.done) hipEventDestroy(sl.done);
    if (sl.tmp) hipFree(sl.tmp);
    sl = LocalRankSlot();
    { std::lock_guard<std::mutex> lk(c->lg->mu); --c->lg->refs; }   // (rsys_local_group_destroy refuses while communicators are open)
  }
  if (c->scratch) hipFree(c->scratch);
  if (c->ev_ready) hipEventDestroy(c->ev_ready);
  if (c->ev_done) hipEventDestroy(c->ev_done);
  if (c->ev_head) hipEventDestroy(c->ev_head);
  if (c->stream) hipStreamDestroy(c->stream);
  delete c;
  return RSYS_OK;
}

// tests only: occupy the communicator's stream for a while, as a collective that is late to start would (the in-process group's
// copies are otherwise too fast to expose a missing cross-stream wait)
__global__ void comm_spin_kernel(unsigned long long ticks) {   // wall_clock64: 100 MHz
  const unsigned long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(32);
}
int comm_debug_delay(rsys_comm* c, int us) {
  ARG_CHECK(c != nullptr && us >= 0 && us <= 2000000, "delay: 0 .. 2e6 microseconds");
  HIP_CHECK(hipSetDevice(c->device));
  hipLaunchKernelGGL(comm_spin_kernel, dim3(1), dim3(64), 0, c->stream, (unsigned long long)us * 100ull);
  HIP_CHECK(hipGetLastError());
  return RSYS_OK;
}

struct PtrList { const void* p[16]; int n; };
template <typename T>
__global__ void reduce_ptrs_kernel(PtrList in, T* out, size_t n, int op) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    T acc = ((const T*)in.p[0])[i];
    for (int q = 1; q < in.n; ++q) {          // rank order: every rank computes the same bits
      const T v = ((const T*)in.p[q])[i];
      acc = op == COMM_MAX ? (v > acc ? v : acc) : acc + v;
    }
    out[i] = acc;
  }
}

// publish this rank's buffers, meet the others, make stream s wait for their producers
static int local_begin(rsys_comm* c, const void* send, void* recv, const long long* send_off, hipStream_t s) {
  LocalGroup* g = c->lg;
  LocalRankSlot& me = g->slot[c->rank];
  me.send = send; me.recv = recv; me.send_off = send_off;
  HIP_CHECK(hipEventRecord(me.ready, s));
  int rc = g->barrier();
  if (rc) return rc;
  for (int q = 0; q < g->world; ++q)
    if (q != c->rank) HIP_CHECK(hipStreamWaitEvent(s, g->slot[q].ready, 0));
  return RSYS_OK;
}
// everybody has finished reading everybody's send buffer once this returns (as far as stream s is concerned)
static int local_end(rsys_comm* c, hipStream_t s) {
  LocalGroup* g = c->lg;
  HIP_CHECK(hipEventRecord(g->slot[c->rank].done, s));
  int rc = g->barrier();
  if (rc) return rc;
  for (int q = 0; q < g->world; ++q)
    if (q != c->rank) HIP_CHECK(hipStreamWaitEvent(s, g->slot[q].done, 0));
  return RSYS_OK;
}

template <typename T>
static int local_all_reduce(rsys_comm* c, T* buf, size_t n, int op, hipStream_t s) {
  LocalGroup* g = c->lg;
  LocalRankSlot& me = g->slot[c->rank];
  const size_t need = (n * sizeof(T) + 3) / 4;
  if (me.tmp_floats < need) {   // (failures return an error code: the callers route it through local_fail, which releases the peers)
    if (me.tmp) hipFree(me.tmp);
    me.tmp = nullptr; me.tmp_floats = 0;
    if (hipMalloc((void**)&me.tmp, need * 4) != hipSuccess) { me.tmp = nullptr; set_error("all-reduce: hipMalloc of the staging buffer failed"); return RSYS_ERR_HIP; }
    me.tmp_floats = need;
  }
  int rc = local_begin(c, buf, nullptr, nullptr, s);
  if (rc) return rc;
  if (g->world > 16) { set_error("in-process group: more than 16 ranks"); return RSYS_ERR_ARG; }
  PtrList pl; pl.n = g->world;
  for (int q = 0; q < g->world; ++q) pl.p[q] = g->slot[q].send;
  const int blocks = (int)std::min<size_t>((n + 255) / 256, 4096);
  hipLaunchKernelGGL((reduce_ptrs_kernel<T>), dim3(blocks), dim3(256), 0, s, pl, (T*)me.tmp, n, op);
  HIP_CHECK(hipGetLastError());
  rc = local_end(c, s);      // nobody reads a buffer any more: now it may be overwritten with the result
  if (rc) return rc;
  HIP_CHECK(hipMemcpyAsync(buf, me.tmp, n * sizeof(T), hipMemcpyDeviceToDevice, s));
  return RSYS_OK;
}

int comm_all_reduce_f32(rsys_comm* c, float* buf, size_t n, int op, hipStream_t s) {
  if (n == 0 || !comm_active(c)) return RSYS_OK;
  if (c->lg) { int rc = local_all_reduce<float>(c, buf, n, op, s); return rc ? local_fail(c->lg, rc) : RSYS_OK; }
  NCCL_CHECK(g_rccl.AllReduce(buf, buf, n, NCCL_FLOAT32, op, c->comm, s));
  return RSYS_OK;
}
// recv = sum over the ranks of send; send is left as it is (recv != send, no overlap).  The early all-reduce of the item table's
// head gradient (capi.hip, split table reduce) uses it: the local gradient keeps accumulating while the sum travels.
int comm_all_reduce_f32_to(rsys_comm* c, const float* send, float* recv, size_t n, hipStream_t s) {
  if (n == 0) return RSYS_OK;
  ARG_CHECK(send != recv, "out-of-place all-reduce: send and recv are the same buffer");
  if (!comm_active(c)) { HIP_CHECK(hipMemcpyAsync(recv, send, n * 4, hipMemcpyDeviceToDevice, s)); return RSYS_OK; }
  if (c->lg) {
    LocalGroup* g = c->lg;
    int rc = local_begin(c, send, recv, nullptr, s);
    if (rc) return local_fail(g, rc);
    if (g->world > 16) { set_error("in-process group: more than 16 ranks"); return local_fail(g, RSYS_ERR_ARG); }
    PtrList pl; pl.n = g->world;
    for (int q = 0; q < g->world; ++q) pl.p[q] = g->slot[q].send;
    hipLaunchKernelGGL((reduce_ptrs_kernel<float>), dim3((int)std::min<size_t>((n + 255) / 256, 4096)), dim3(256), 0, s, pl, recv, n, (int)COMM_SUM);
    if (hipGetLastError() != hipSuccess) { set_error("all-reduce: launch failed"); return local_fail(g, RSYS_ERR_HIP); }
    rc = local_end(c, s);
    return rc ? local_fail(g, rc) : RSYS_OK;
  }
  NCCL_CHECK(g_rccl.AllReduce(send, recv, n, NCCL_FLOAT32, COMM_SUM, c->comm, s));
  return RSYS_OK;
}
int comm_all_reduce_f64(rsys_comm* c, double* buf, size_t n, hipStream_t s) {
  if (n == 0 || !comm_active(c)) return RSYS_OK;
  if (c->lg) { int rc = local_all_reduce<double>(c, buf, n, COMM_SUM, s); return rc ? local_fail(c->lg, rc) : RSYS_OK; }
  NCCL_CHECK(g_rccl.AllReduce(buf, buf, n, NCCL_FLOAT64, COMM_SUM, c->comm, s));
  return RSYS_OK;
}

int comm_reduce_scatter_f32(rsys_comm* c, float* buf, size_t chunk, hipStream_t s) {
  if (chunk == 0 || !comm_active(c)) return RSYS_OK;
  if (c->lg) {
    LocalGroup* g = c->lg;
    LocalRankSlot& me = g->slot[c->rank];
    if (me.tmp_floats < chunk) {   // (a rank that cannot allocate breaks the group: its peers must not wait for it at the barrier)
      if (me.tmp) hipFree(me.tmp);
      me.tmp = nullptr; me.tmp_floats = 0;
      if (hipMalloc((void**)&me.tmp, chunk * 4) != hipSuccess) { me.tmp = nullptr; set_error("reduce-scatter: hipMalloc of the staging chunk failed"); return local_fail(g, RSYS_ERR_HIP); }
      me.tmp_floats = chunk;
    }
    int rc = local_begin(c, buf, nullptr, nullptr, s);
    if (rc) return local_fail(g, rc);
    if (g->world > 16) { set_error("in-process group: more than 16 ranks"); return local_fail(g, RSYS_ERR_ARG); }
    PtrList pl; pl.n = g->world;
    for (int q = 0; q < g->world; ++q) pl.p[q] = (const float*)g->slot[q].send + (size_t)c->rank * chunk;
    hipLaunchKernelGGL((reduce_ptrs_kernel<float>), dim3((int)std::min<size_t>((chunk + 255) / 256, 4096)), dim3(256), 0, s, pl, (float*)me.tmp, chunk, (int)COMM_SUM);
    if (hipGetLastError() != hipSuccess) { set_error("reduce-scatter: launch failed"); return local_fail(g, RSYS_ERR_HIP); }
    rc = local_end(c, s);      // nobody reads a buffer any more: now this rank's chunk may be overwritten with the result
    if (rc) return local_fail(g, rc);
    if (hipMemcpyAsync(buf + (size_t)c->rank * chunk, me.tmp, chunk * 4, hipMemcpyDeviceToDevice, s) != hipSuccess) { set_error("reduce-scatter: device copy failed"); return local_fail(g, RSYS_ERR_HIP); }
    return RSYS_OK;
  }
  if (!g_rccl.ReduceScatter) { set_error("librccl has no ncclReduceScatter"); return RSYS_ERR_COMM; }
  NCCL_CHECK(g_rccl.ReduceScatter(buf, buf + (size_t)c->rank * chunk, chunk, NCCL_FLOAT32, COMM_SUM, c->comm, s));
  return RSYS_OK;
}

int comm_all_gather(rsys_comm* c, const void* send, void* recv, size_t bytes, hipStream_t s) {
  if (bytes == 0) return RSYS_OK;
  if (!comm_active(c)) {
    if (send != recv) HIP_CHECK(hipMemcpyAsync(recv, send, bytes, hipMemcpyDeviceToDevice, s));
    return RSYS_OK;
  }
  if (c->lg) {
    LocalGroup* g = c->lg;
    int rc = local_begin(c, send, recv, nullptr, s);
    if (rc) return local_fail(g, rc);
    for (int q = 0; q < g->world; ++q)
      if ((char*)recv + (size_t)q * bytes != (const char*)g->slot[q].send &&   // (in place: this rank's own chunk is where it belongs)
          hipMemcpyAsync((char*)recv + (size_t)q * bytes, g->slot[q].send, bytes, hipMemcpyDeviceToDevice, s) != hipSuccess) {
        set_error("all-gather: device copy failed");
        return local_fail(g, RSYS_ERR_HIP);    // (the peers must not wait for this rank at the closing barrier)
      }
    rc = local_end(c, s);
    return rc ? local_fail(g, rc) : RSYS_OK;
  }
  NCCL_CHECK(g_rccl.AllGather(send, recv, bytes, NCCL_INT8, c->comm, s));
  return RSYS_OK;
}

int comm_exchange(rsys_comm* c, const void* send, const long long* send_off, void* recv, const long long* recv_off,
                  size_t eb, hipStream_t s) {
  const int W = c ? c->world : 1;
  if (!comm_active(c)) {
    const long long n = send_off[1] - send_off[0];
    ARG_CHECK(n == recv_off[1] - recv_off[0], "exchange: send and receive sizes differ");
    if (n) HIP_CHECK(hipMemcpyAsync((char*)recv + recv_off[0] * eb, (const char*)send + send_off[0] * eb, n * eb, hipMemcpyDeviceToDevice, s));
    return RSYS_OK;
  }
  if (c->lg) {
    LocalGroup* g = c->lg;
    int rc = local_begin(c, send, recv, send_off, s);
    if (rc) return local_fail(g, rc);
    for (int q = 0; q < W; ++q) {
      const long long* so = g->slot[q].send_off;
      const long long n = so[c->rank + 1] - so[c->rank];
      if (n != recv_off[q + 1] - recv_off[q]) { set_error("exchange: a peer's send size differs from this rank's receive size"); return local_fail(g, RSYS_ERR_COMM); }
      if (n && hipMemcpyAsync((char*)recv + recv_off[q] * eb, (const char*)g->slot[q].send + so[c->rank] * eb, n * eb, hipMemcpyDeviceToDevice, s) != hipSuccess) {
        set_error("exchange: device copy failed");
        return local_fail(g, RSYS_ERR_HIP);
      }
    }
    rc = local_end(c, s);
    return rc ? local_fail(g, rc) : RSYS_OK;
  }
  // one grouped send / receive per peer, issued in rank order on every rank.  An error inside the group must not leave it
  // open (every later RCCL call of this thread would join it): close it first, then report the first failure.
  NCCL_CHECK(g_rccl.GroupStart());
  int bad = 0; const char* what = "";
  for (int q = 0; q < W && !bad; ++q) {
    const long long ns = send_off[q + 1] - send_off[q], nr = recv_off[q + 1] - recv_off[q];
    if (ns && (bad = g_rccl.Send((const char*)send + send_off[q] * eb, (size_t)ns * eb, NCCL_INT8, q, c->comm, s)) != 0) { what = "ncclSend"; break; }
    if (nr && (bad = g_rccl.Recv((char*)recv + recv_off[q] * eb, (size_t)nr * eb, NCCL_INT8, q, c->comm, s)) != 0) { what = "ncclRecv"; break; }
  }
  const int end = g_rccl.GroupEnd();
  if (bad) { set_error(std::string(what) + " inside the row exchange: " + (g_rccl.GetErrorString ? g_rccl.GetErrorString(bad) : "rccl error")); return RSYS_ERR_COMM; }
  NCCL_CHECK(end);
  return RSYS_OK;
}

}  // namespace rsys
