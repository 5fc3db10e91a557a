// TEST INFRASTRUCTURE -- a test double for the five RCCL entry points csrc/agx_exchange.hip binds at run time
// (ncclGetUniqueId, ncclCommInitRank, ncclCommDestroy, ncclAllGather, ncclGetErrorString; + ncclCommCount /
// ncclCommUserRank), so that the library-side step exchange (worker thread, communication stream, device-flag
// hand-off, done events, double buffering by step parity) can run at WORLD SIZE 2 on a box with ONE GPU: RCCL itself
// refuses two ranks on one device.  The "collective" is a stream-ordered copy through POSIX shared memory:
//
//   ncclAllGather(send, recv, count, ..., stream)  (collective number s of this communicator)
//     host function  wait until every rank has consumed collective s - 2      (slot s % 2 is free)
//     copy           send -> shm slot[s % 2][rank]                             (device to host, async)
//     host function  posted[rank] = s; wait until every rank has posted s     (the rendezvous of the collective)
//     copy           shm slot[s % 2][0 .. world) -> recv                       (host to device, async)
//     host function  consumed[rank] = s
//
// i.e. like the real collective it returns at once, completes in stream order, and completes only when all ranks
// have contributed.  Waits are bounded (AGX_FAKERCCL_TIMEOUT_S, default 20 s): a rank that never arrives turns into
// ncclSystemError on the next call instead of a hang.  AGX_FAKERCCL_FAIL_RANK / AGX_FAKERCCL_FAIL_AT inject a failing
// collective (failure-path tests).  Build: hipcc -shared -fPIC (tests/fakerccl/build.py).
#include <fcntl.h>
#include <hip/hip_runtime.h>
#include <sys/mman.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>

extern "C" {
typedef enum { ncclSuccess = 0, ncclUnhandledCudaError = 1, ncclSystemError = 2, ncclInternalError = 3, ncclInvalidArgument = 4 } ncclResult_t;
typedef struct { char internal[128]; } ncclUniqueId;
typedef enum { ncclFloat32 = 7 } ncclDataType_t;
}

namespace {
constexpr int kMaxRanks = 8;
constexpr size_t kSlotBytes = 16u << 20;  // per rank and slot

struct Header {
  std::atomic<uint32_t> joined, left, failed;
  std::atomic<uint64_t> posted[kMaxRanks], consumed[kMaxRanks];
};

struct Comm {
  int rank, world;
  char name[64];
  size_t bytes;
  unsigned char *base;
  Header *h;
  bool registered;
  uint64_t issued;  // collectives enqueued by this rank
  double timeout_s;
  int fail_at;
};

struct HostOp {
  Comm *c;
  uint64_t s;
  int kind;  // 0 = wait slot free, 1 = post + rendezvous, 2 = consumed
};

double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

unsigned char *slot(Comm *c, uint64_t s, int rank) {
  return c->base + 4096 + ((s & 1) * c->world + rank) * kSlotBytes;
}

bool wait_all(Comm *c, std::atomic<uint64_t> *arr, uint64_t target) {
  const double t0 = now_s();
  for (;;) {
    bool ok = true;
    for (int r = 0; r < c->world; ++r) ok = ok && arr[r].load(std::memory_order_acquire) >= target;
    if (ok) return true;
    if (c->h->failed.load(std::memory_order_acquire)) return false;
    if (now_s() - t0 > c->timeout_s) {
      c->h->failed.store(1, std::memory_order_release);
      fprintf(stderr, "[fakerccl] rank %d: a peer did not reach collective %llu within %.0f s\n", c->rank, (unsigned long long)target,
              c->timeout_s);
      return false;
    }
    std::this_thread::sleep_for(std::chrono::microseconds(5));
  }
}

void host_fn(void *p) {
  HostOp *op = static_cast<HostOp *>(p);
  Comm *c = op->c;
  if (op->kind == 0) {
    if (op->s > 2) wait_all(c, c->h->consumed, op->s - 2);
  } else if (op->kind == 1) {
    c->h->posted[c->rank].store(op->s, std::memory_order_release);
    wait_all(c, c->h->posted, op->s);
  } else {
    c->h->consumed[c->rank].store(op->s, std::memory_order_release);
  }
  delete op;
}
}  // namespace

extern "C" {

const char *ncclGetErrorString(ncclResult_t r) {
  switch (r) {
    case ncclSuccess: return "no error";
    case ncclSystemError: return "fakerccl: a peer never arrived (timeout) or the shared segment is gone";
    case ncclInternalError: return "fakerccl: injected failure";
    case ncclInvalidArgument: return "fakerccl: invalid argument";
    default: return "fakerccl: hip error";
  }
}

ncclResult_t ncclGetUniqueId(ncclUniqueId *id) {
  memset(id->internal, 0, sizeof(id->internal));
  snprintf(id->internal, sizeof(id->internal), "/agxfakerccl_%d_%llx", (int)getpid(),
           (unsigned long long)std::chrono::steady_clock::now().time_since_epoch().count());
  return ncclSuccess;
}

ncclResult_t ncclCommInitRank(Comm **out, int world, ncclUniqueId id, int rank) {
  if (world < 1 || world > kMaxRanks || rank < 0 || rank >= world) return ncclInvalidArgument;
  Comm *c = new Comm();
  c->rank = rank;
  c->world = world;
  strncpy(c->name, id.internal, sizeof(c->name) - 1);
  c->bytes = 4096 + 2 * (size_t)world * kSlotBytes;
  const char *t = getenv("AGX_FAKERCCL_TIMEOUT_S");
  c->timeout_s = t ? atof(t) : 20.0;
  const char *fr = getenv("AGX_FAKERCCL_FAIL_RANK"), *fa = getenv("AGX_FAKERCCL_FAIL_AT");
  c->fail_at = (fr && fa && atoi(fr) == rank) ? atoi(fa) : -1;
  int fd = shm_open(c->name, O_CREAT | O_RDWR, 0600);
  if (fd < 0 || ftruncate(fd, (off_t)c->bytes) != 0) { delete c; return ncclSystemError; }
  c->base = static_cast<unsigned char *>(mmap(nullptr, c->bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0));
  close(fd);
  if (c->base == MAP_FAILED) { delete c; return ncclSystemError; }
  c->h = reinterpret_cast<Header *>(c->base);  // a fresh segment is zero-filled: all counters start at 0
  c->registered = hipHostRegister(c->base, c->bytes, hipHostRegisterDefault) == hipSuccess;
  if (!c->registered) (void)hipGetLastError();
  c->h->joined.fetch_add(1, std::memory_order_acq_rel);
  const double t0 = now_s();
  while (c->h->joined.load(std::memory_order_acquire) < (uint32_t)world) {  // ncclCommInitRank is collective
    if (now_s() - t0 > c->timeout_s) return ncclSystemError;
    std::this_thread::sleep_for(std::chrono::milliseconds(1));
  }
  *out = c;
  return ncclSuccess;
}

ncclResult_t ncclCommCount(Comm *c, int *count) { *count = c->world; return ncclSuccess; }
ncclResult_t ncclCommUserRank(Comm *c, int *rank) { *rank = c->rank; return ncclSuccess; }

ncclResult_t ncclAllGather(const void *send, void *recv, size_t count, ncclDataType_t dt, Comm *c, hipStream_t stream) {
  if (!c || !send || !recv || dt != ncclFloat32) return ncclInvalidArgument;
  const size_t bytes = count * 4;
  if (bytes > kSlotBytes) return ncclInvalidArgument;
  if (c->h->failed.load(std::memory_order_acquire)) return ncclSystemError;
  const uint64_t s = ++c->issued;
  if (c->fail_at >= 0 && (int)s == c->fail_at) {
    c->h->failed.store(1, std::memory_order_release);  // the peers' rendezvous ends with an error instead of a timeout
    return ncclInternalError;
  }
  if (hipLaunchHostFunc(stream, host_fn, new HostOp{c, s, 0}) != hipSuccess) return ncclUnhandledCudaError;
  if (hipMemcpyAsync(slot(c, s, c->rank), send, bytes, hipMemcpyDeviceToHost, stream) != hipSuccess) return ncclUnhandledCudaError;
  if (hipLaunchHostFunc(stream, host_fn, new HostOp{c, s, 1}) != hipSuccess) return ncclUnhandledCudaError;
  for (int r = 0; r < c->world; ++r)
    if (hipMemcpyAsync(static_cast<unsigned char *>(recv) + (size_t)r * bytes, slot(c, s, r), bytes, hipMemcpyHostToDevice, stream) != hipSuccess)
      return ncclUnhandledCudaError;
  if (hipLaunchHostFunc(stream, host_fn, new HostOp{c, s, 2}) != hipSuccess) return ncclUnhandledCudaError;
  return ncclSuccess;
}

ncclResult_t ncclCommDestroy(Comm *c) {
  if (!c) return ncclSuccess;
  if (c->registered) (void)hipHostUnregister(c->base);
  const uint32_t left = c->h->left.fetch_add(1, std::memory_order_acq_rel) + 1;
  munmap(c->base, c->bytes);
  if (left == (uint32_t)c->world) shm_unlink(c->name);
  delete c;
  return ncclSuccess;
}
}
