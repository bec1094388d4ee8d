// Communicator of one rank (one process per GPU, transformer.py:582-587): RCCL over xGMI, bound at run time.
//
// Two transports behind one interface:
//  * RCCL (librccl.so.1): the production transport; ncclCommInitRank from an id the host distributes.
//  * an in-process group: N ranks of ONE process on ONE device, each driven by its own host thread, collectives
//    done with device copies between the ranks' buffers behind events and a host barrier.  It exists so that the
//    partition arithmetic of the multi-rank paths (row-sharded item table, vocabulary-parallel cross entropy, sparse
//    row exchange) can run on a one-GPU box with the real kernels; two RCCL ranks cannot share one GPU.
// Every collective is enqueued on the stream the caller gives; buffers are device memory.
#pragma once
#include <condition_variable>
#include <mutex>
#include <vector>

#include "common.hpp"

typedef struct { char internal[128]; } ncclUniqueId_t;
typedef void* ncclComm_t_;

namespace rsys {

struct LocalRankSlot {                 // what one rank publishes to the others for the collective in flight
  const void* send = nullptr; void* recv = nullptr;
  const long long* send_off = nullptr;   // exchange: element offsets per destination rank [world + 1] (host memory)
  hipEvent_t ready = nullptr, done = nullptr;
  float* tmp = nullptr; size_t tmp_floats = 0;
};

struct LocalGroup {
  int world = 1, device = 0;
  std::mutex mu; std::condition_variable cv;
  int arrived = 0; unsigned long long generation = 0;
  bool broken = false;                 // a rank failed inside a collective: everybody else stops waiting
  std::vector<LocalRankSlot> slot;
  int refs = 0;
  int barrier();                       // host barrier over the ranks' threads; RSYS_ERR_COMM when the group is broken
};

}  // namespace rsys

struct rsys_comm {
  int rank = 0, world = 1, device = 0;
  // RCCL
  ncclComm_t_ comm = nullptr;
  hipStream_t stream = nullptr;        // the gradient all-reduce's own stream (overlaps the backward)
  hipEvent_t ev_ready = nullptr, ev_done = nullptr;
  hipEvent_t ev_head = nullptr;        // split table reduce: recorded behind the out-of-place all-reduce that READS G[E] (capi.hip)
  double* scratch = nullptr;
  bool force = false;                  // RSYS_FORCE_RCCL=1: run the collectives even at world == 1 (exercises RCCL on one GPU)
  // in-process group (tests)
  rsys::LocalGroup* lg = nullptr;
};

namespace rsys {

enum CommOp { COMM_SUM = 0, COMM_MAX = 2 };

int load_rccl();
int comm_rccl_version();   // ncclGetVersion's code (e.g. 22703), 0 when the library is not loaded or has no such symbol
int comm_unique_id(unsigned char id_buf[128]);
int comm_init_rccl(const unsigned char id_buf[128], int rank, int world, int device, rsys_comm** out);
int comm_init_local(LocalGroup* g, int rank, rsys_comm** out);
int comm_destroy(rsys_comm* c);
// tests: a kernel that spins for `us` microseconds (<= 2 s) on the communicator's stream, i.e. a collective that starts late
int comm_debug_delay(rsys_comm* c, int us);
inline bool comm_active(const rsys_comm* c) { return c != nullptr && (c->world > 1 || c->force || c->lg != nullptr); }

// in place, float32 (op: COMM_SUM / COMM_MAX) or float64 (sum)
int comm_all_reduce_f32(rsys_comm* c, float* buf, size_t n, int op, hipStream_t s);
int comm_all_reduce_f32_to(rsys_comm* c, const float* send, float* recv, size_t n, hipStream_t s);   // recv = sum of the ranks' send
int comm_all_reduce_f64(rsys_comm* c, double* buf, size_t n, hipStream_t s);
// recv[r * bytes ..] = rank r's send (bytes per rank); in place when send == recv + rank * bytes
int comm_all_gather(rsys_comm* c, const void* send, void* recv, size_t bytes, hipStream_t s);
// buf holds world * chunk floats on every rank: afterwards this rank's chunk (buf + rank * chunk) is the sum over the ranks (in rank
// order in the in-process group); the other chunks are undefined
int comm_reduce_scatter_f32(rsys_comm* c, float* buf, size_t chunk, hipStream_t s);
// all-to-all with per-pair sizes: rank r sends elements [send_off[q], send_off[q+1]) of `send` to rank q and receives rank
// q's block for it at recv_off[q]; offsets in elements of elem_bytes, host arrays of world + 1 entries that must stay
// valid until the call returns
int comm_exchange(rsys_comm* c, const void* send, const long long* send_off, void* recv, const long long* recv_off,
                  size_t elem_bytes, hipStream_t s);

}  // namespace rsys
