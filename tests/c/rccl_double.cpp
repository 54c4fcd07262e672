// rccl_double.cpp -- a TEST DOUBLE for the handful of RCCL entry points libdctzhip.so resolves with dlopen
// (dctz_shim.hip: ncclGetUniqueId, ncclCommInitRank, ncclCommDestroy, ncclAllGather, ncclSend, ncclRecv,
// ncclGroupStart, ncclGroupEnd, ncclGetErrorString).  Test infrastructure only: the one-GPU boxes of this project cannot
// run RCCL with more than one rank (RCCL refuses two ranks on one device), so dctzhip_comm_sizes / dctzhip_comm_gather --
// the one exchange step of the multi-GPU path -- never ran with a peer.  With this library in
// DCTZHIP_RCCL_LIBRARY several PROCESSES that share one GPU talk through a POSIX shared-memory segment instead: same
// calls, same group semantics (operations queued between GroupStart and GroupEnd run at GroupEnd, sends first), real
// device buffers on both ends.  It proves the library's own logic (offsets, order, message sizes, closing the group on
// the error paths); it says nothing about xGMI.
//   RCCL_DOUBLE_FAIL=<rank>:<k>   the k-th ncclSend/ncclRecv CALL of that rank (from 0) fails with ncclInternalError
//   RCCL_DOUBLE_TIMEOUT_S=<s>     how long a receive waits for its peer (default 20)
#include <hip/hip_runtime_api.h>

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

#include <atomic>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

namespace {
constexpr int MAXW = 6;
constexpr size_t CHAN_BYTES = 32u << 20, AG_BYTES = 4096;   // a channel holds one PIECE of a message at a time
enum { OK = 0, UNHANDLED = 1, SYSTEM = 2, INTERNAL = 3, INVALID_ARG = 4, INVALID_USAGE = 5, REMOTE = 6 };

struct Chan {
  std::atomic<unsigned long long> produced, taken;   // pieces published by the sender / consumed by the receiver
  unsigned long long total, piece;                   // of the piece in the slot: length of the whole message, of this piece
};
struct Shm {
  std::atomic<unsigned> joined, left;
  std::atomic<unsigned> bar_count, bar_gen;
  unsigned char ag[MAXW][AG_BYTES];
  Chan chan[MAXW][MAXW];
  unsigned char data[MAXW][MAXW][CHAN_BYTES];        // sparse: only what is written is ever backed by memory
};
struct Comm {
  Shm* shm;
  int rank, world;
  char name[128];
};
struct Op { bool send; void* buf; size_t bytes; int peer; Comm* comm; hipStream_t stream; };
thread_local int g_depth = 0;
thread_local std::vector<Op>* g_ops = nullptr;
int g_calls = 0;

double now_s() { timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return (double)t.tv_sec + 1e-9 * (double)t.tv_nsec; }
double timeout_s() { const char* e = getenv("RCCL_DOUBLE_TIMEOUT_S"); return e ? atof(e) : 20.0; }
size_t dtype_bytes(int dt) { switch (dt) { case 0: case 1: return 1; case 2: case 3: case 7: return 4; case 4: case 5: case 8: return 8; case 6: return 2; default: return 0; } }

bool injected_failure(const Comm* c) {
  const char* e = getenv("RCCL_DOUBLE_FAIL");
  const int k = g_calls++;
  if (!e) return false;
  int r = -1, at = -1;
  return sscanf(e, "%d:%d", &r, &at) == 2 && r == c->rank && at == k;
}

int barrier(Comm* c) {
  Shm* s = c->shm;
  const unsigned gen = s->bar_gen.load(std::memory_order_acquire);
  if (s->bar_count.fetch_add(1, std::memory_order_acq_rel) + 1 == (unsigned)c->world) {
    s->bar_count.store(0, std::memory_order_relaxed);
    s->bar_gen.fetch_add(1, std::memory_order_release);
    return OK;
  }
  const double t0 = now_s();
  while (s->bar_gen.load(std::memory_order_acquire) == gen) {
    if (now_s() - t0 > timeout_s()) return REMOTE;
    usleep(50);
  }
  return OK;
}

// A message travels in pieces of at most CHAN_BYTES through the one slot of its (sender, receiver) channel: the sender
// waits until the slot is free, the receiver until it is full.  (Enough for a gather, where a rank either sends or
// receives; two ranks that first send to each other and then receive would need deeper channels.)
int wait_until(const std::atomic<unsigned long long>& a, const std::atomic<unsigned long long>& b, bool want_equal) {
  const double t0 = now_s();
  while ((a.load(std::memory_order_acquire) == b.load(std::memory_order_acquire)) != want_equal) {
    if (now_s() - t0 > timeout_s()) return REMOTE;
    usleep(20);
  }
  return OK;
}
int run_send(const Op& o) {
  Comm* c = o.comm;
  Chan& ch = c->shm->chan[c->rank][o.peer];
  unsigned char* slot = c->shm->data[c->rank][o.peer];
  if (hipStreamSynchronize(o.stream) != hipSuccess) return UNHANDLED;
  size_t done = 0;
  do {
    const size_t piece = o.bytes - done < CHAN_BYTES ? o.bytes - done : CHAN_BYTES;
    if (int rc = wait_until(ch.produced, ch.taken, true)) return rc;          // the slot is free
    if (piece && hipMemcpy(slot, (const char*)o.buf + done, piece, hipMemcpyDeviceToHost) != hipSuccess) return UNHANDLED;
    ch.total = o.bytes; ch.piece = piece;
    ch.produced.fetch_add(1, std::memory_order_release);
    done += piece;
  } while (done < o.bytes);
  return OK;
}
int run_recv(const Op& o) {
  Comm* c = o.comm;
  Chan& ch = c->shm->chan[o.peer][c->rank];
  const unsigned char* slot = c->shm->data[o.peer][c->rank];
  if (hipStreamSynchronize(o.stream) != hipSuccess) return UNHANDLED;
  size_t done = 0;
  int bad = OK;
  do {
    if (int rc = wait_until(ch.produced, ch.taken, false)) return rc;         // a piece is there
    const size_t total = ch.total, piece = ch.piece;
    if (total != o.bytes || done + piece > o.bytes) bad = INVALID_USAGE;       // a send and its receive must agree on the size
    else if (piece && hipMemcpy((char*)o.buf + done, slot, piece, hipMemcpyHostToDevice) != hipSuccess) bad = UNHANDLED;
    ch.taken.fetch_add(1, std::memory_order_release);
    if (bad) return bad;
    done += piece;
  } while (done < o.bytes);
  return OK;
}
int run_all(std::vector<Op>& ops) {
  int bad = OK;
  for (const Op& o : ops) if (o.send && !bad) bad = run_send(o);
  for (const Op& o : ops) if (!o.send && !bad) bad = run_recv(o);
  ops.clear();
  return bad;
}
int post(bool send, void* buf, size_t count, int dt, int peer, void* comm, hipStream_t s) {
  Comm* c = (Comm*)comm;
  if (!c || peer < 0 || peer >= c->world || peer == c->rank || !dtype_bytes(dt)) return INVALID_ARG;
  if (injected_failure(c)) return INTERNAL;
  Op o{send, buf, count * dtype_bytes(dt), peer, c, s};
  if (g_depth > 0) { if (!g_ops) g_ops = new std::vector<Op>(); g_ops->push_back(o); return OK; }
  return send ? run_send(o) : run_recv(o);
}
}  // namespace

extern "C" {
// (the library refuses an RCCL of another major version before its first call: RCCL_DOUBLE_VERSION sets what this one says)
int ncclGetVersion(int* v) {
  const char* e = getenv("RCCL_DOUBLE_VERSION");
  *v = e ? atoi(e) : 22707;
  return 0;
}
struct ncclUniqueId { char internal[128]; };

int ncclGetUniqueId(ncclUniqueId* id) {
  if (!id) return INVALID_ARG;
  memset(id->internal, 0, sizeof(id->internal));
  snprintf(id->internal, sizeof(id->internal), "/rccl_double_%d_%lld", (int)getpid(), (long long)(now_s() * 1e6));
  return OK;
}

int ncclCommInitRank(void** comm, int world, ncclUniqueId id, int rank) {
  if (!comm || world < 1 || world > MAXW || rank < 0 || rank >= world || id.internal[0] != '/') return INVALID_ARG;
  const int fd = shm_open(id.internal, O_CREAT | O_RDWR, 0600);
  if (fd < 0) return SYSTEM;
  if (ftruncate(fd, sizeof(Shm)) != 0) { close(fd); return SYSTEM; }      // (new pages read as zeros: the initial state)
  void* m = mmap(nullptr, sizeof(Shm), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
  close(fd);
  if (m == MAP_FAILED) return SYSTEM;
  Comm* c = new Comm();
  c->shm = (Shm*)m; c->rank = rank; c->world = world;
  memcpy(c->name, id.internal, sizeof(c->name));
  c->shm->joined.fetch_add(1, std::memory_order_acq_rel);
  const double t0 = now_s();
  while (c->shm->joined.load(std::memory_order_acquire) < (unsigned)world) {
    if (now_s() - t0 > timeout_s()) { munmap(m, sizeof(Shm)); delete c; return REMOTE; }
    usleep(100);
  }
  *comm = c;
  return OK;
}

int ncclCommDestroy(void* comm) {
  Comm* c = (Comm*)comm;
  if (!c) return INVALID_ARG;
  if (c->shm->left.fetch_add(1, std::memory_order_acq_rel) + 1 == (unsigned)c->world) shm_unlink(c->name);
  munmap(c->shm, sizeof(Shm));
  delete c;
  return OK;
}

int ncclAllGather(const void* send, void* recv, size_t count, int dt, void* comm, hipStream_t s) {
  Comm* c = (Comm*)comm;
  const size_t bytes = count * dtype_bytes(dt);
  if (!c || !bytes || bytes > AG_BYTES) return INVALID_ARG;
  if (hipStreamSynchronize(s) != hipSuccess) return UNHANDLED;
  if (hipMemcpy(c->shm->ag[c->rank], send, bytes, hipMemcpyDeviceToHost) != hipSuccess) return UNHANDLED;
  int rc = barrier(c);
  if (rc) return rc;
  for (int r = 0; r < c->world; r++)
    if (hipMemcpy((char*)recv + (size_t)r * bytes, c->shm->ag[r], bytes, hipMemcpyHostToDevice) != hipSuccess) return UNHANDLED;
  return barrier(c);                                   // nobody overwrites its piece before everybody has read it
}

int ncclSend(const void* buf, size_t count, int dt, int peer, void* comm, hipStream_t s) { return post(true, (void*)buf, count, dt, peer, comm, s); }
int ncclRecv(void* buf, size_t count, int dt, int peer, void* comm, hipStream_t s) { return post(false, buf, count, dt, peer, comm, s); }
int ncclGroupStart() { g_depth++; return OK; }
int ncclGroupEnd() {
  if (g_depth <= 0) return INVALID_USAGE;
  if (--g_depth > 0 || !g_ops) return OK;
  return run_all(*g_ops);
}
const char* ncclGetErrorString(int r) {
  switch (r) {
    case OK: return "no error";
    case UNHANDLED: return "unhandled HIP error (rccl double)";
    case SYSTEM: return "system error (rccl double)";
    case INTERNAL: return "internal error (rccl double: injected)";
    case INVALID_ARG: return "invalid argument (rccl double)";
    case INVALID_USAGE: return "invalid usage (rccl double: the sizes of a send and its receive differ)";
    case REMOTE: return "remote error (rccl double: the peer did not show up in time)";
    default: return "unknown (rccl double)";
  }
}
}
