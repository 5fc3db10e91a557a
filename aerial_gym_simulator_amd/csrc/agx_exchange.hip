// The per-step observation exchange of the sharded simulator: ONE RCCL all-gather
// per env step of the [N_local, obs_dim + 3] rows the observation kernel wrote, issued from a
// worker thread of this library so that the stepping thread pays one hipEventRecord + one
// hipStreamWaitEvent per step instead of a collective enqueue (14 us through torch's process
// group, 27 us with its work-handle bookkeeping -- more than the 18 us env step itself).
//
//   stepping thread                      worker thread                     GPU
//   ---------------                      -------------                     ---
//   obs kernel(step t) -> rows[p],                                         step stream
//     last wave: signal[p] = t + 1
//   agx_exchange_step(p):
//     push job (no HIP call)          -> k_wait_signal(signal[p] >= t+1)   comm stream
//                                        ncclAllGather(rows[p] -> all[p])  comm stream (xGMI)
//     step stream waits done[1-p]        record done[p]
//   obs kernel(step t+1) -> rows[1-p]    ...                               overlaps the gather of t
//
// Why a flag in device memory and not an event for the producer side: a cross-queue
// hipStreamWaitEvent makes the HIP runtime lock and flush the OTHER queue -- measured on one MI355X
// (profiles/r01_exchange_probe.txt), env step 19.1 us; + worker with events 36-41 us; + worker with the
// flag 21 us.  Rows produced by anything but the simulator's kernels (StepGather.pack) still use the
// ready[p] event (signal == NULL).
//
// RCCL is bound at run time (dlopen of the librccl torch already loaded, path handed in by the
// host) so that the simulator library itself links against nothing but the HIP runtime.
// The reference has no distributed path at all (replicas only): there is no upstream interface
// this replaces; the host-side mirror is aerial_gym_simulator_amd/sharding.py:StepGather.
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstring>
#include <mutex>
#include <thread>

#include "agx_common.h"

namespace {

struct Rccl {
  void *handle = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  const char *(*GetErrorString)(ncclResult_t) = nullptr;
  ncclResult_t (*CommCount)(const ncclComm_t, int *) = nullptr;
  ncclResult_t (*CommUserRank)(const ncclComm_t, int *) = nullptr;
};

// one binding per process (the path of the first call wins; later calls must agree or pass NULL)
int bind_rccl(const char *path, Rccl **out) {
  static std::mutex m;
  static Rccl r;
  std::lock_guard<std::mutex> lock(m);
  if (!r.handle) {
    const char *p = (path && path[0]) ? path : "librccl.so";
    void *h = dlopen(p, RTLD_NOW | RTLD_LOCAL);
    if (!h) return agx::fail(AGX_E_ARG, "agx_exchange: cannot load RCCL from '%s': %s", p, dlerror());
#define AGX_BIND(field, sym)                                                                         \
  r.field = reinterpret_cast<decltype(r.field)>(dlsym(h, sym));                                      \
  if (!r.field) {                                                                                    \
    dlclose(h);                                                                                      \
    return agx::fail(AGX_E_ARG, "agx_exchange: '%s' has no symbol %s", p, sym);                      \
  }
    AGX_BIND(GetUniqueId, "ncclGetUniqueId")
    AGX_BIND(CommInitRank, "ncclCommInitRank")
    AGX_BIND(CommDestroy, "ncclCommDestroy")
    AGX_BIND(AllGather, "ncclAllGather")
    AGX_BIND(GetErrorString, "ncclGetErrorString")
    AGX_BIND(CommCount, "ncclCommCount")
    AGX_BIND(CommUserRank, "ncclCommUserRank")
#undef AGX_BIND
    r.handle = h;
  }
  *out = &r;
  return AGX_OK;
}

constexpr int kRing = 8;  // jobs in flight between the two threads (2 are ever used: one per parity)

struct Job {
  const float *send;
  float *recv;
  size_t count;
  int parity;
  const uint32_t *signal;  // AgxEnvBuffers.step_signal, or NULL: the ready[parity] event orders the gather
  uint32_t seq;
  uint32_t push_seq;       // peer push: sequence number of this post (slot = (push_seq - 1) % kPushSlots)
};

// 10 s of the 100 MHz wall clock: a producer that never signals (or a communication stream that ended up
// in the producer's hardware queue) becomes an error instead of a hung GPU
constexpr uint64_t kSpinLimitTicks = 1000000000ull;

// Communication-stream side of AgxEnvBuffers.step_signal (agx_step_signal.h): one lane waits until the
// row-writing kernel of step `seq` has published its rows, then the all-gather behind it may read them
// (acquire here + the cache invalidate at the start of the next kernel).
__global__ void k_set_signal(uint32_t *flag, uint32_t value) {
  if (threadIdx.x == 0) __hip_atomic_store(flag, value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__global__ void k_wait_signal(const uint32_t *flag, uint32_t seq, uint32_t *timed_out, uint64_t limit_ticks) {
  if (threadIdx.x != 0) return;
  const uint64_t t0 = wall_clock64();
  uint32_t polls = 0;
  while ((int32_t)(__hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) - seq) < 0) {
    // a 20 us dynamics step is caught within a fraction of a microsecond; behind a 3 ms ray-cast step the
    // lane backs off to one poll per ~3 us
    if (++polls < 128) __builtin_amdgcn_s_sleep(8);
    else __builtin_amdgcn_s_sleep(127);
    if (wall_clock64() - t0 > limit_ticks) {
      __hip_atomic_store(timed_out, seq ? seq : 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      return;
    }
  }
}

// ---- peer push (round 3): no collective kernel per step ---------------------------------------------------------------------
// The RCCL all-gather costs 12.5 us per launch back to back in a world of one and more over xGMI -- as long as the 13 us
// dynamics-only step itself, and the gathers of consecutive steps serialise on the communication stream (VERDICT r2 "missing" 1).
// Instead every rank PUSHES its rows straight into every peer's receive buffer (peer memory mapped through hipIpcMemHandle;
// xGMI is point to point, so the seven destinations are seven independent links) and raises a per-sender arrival flag there:
//   k_push_rows (communication stream, one launch per step): wait for step_signal (the rows of this step are written) ->
//     workgroup b copies chunk b / world of the rows to destination b % world -> system-scope release -> the last workgroup to
//     arrive stores flags[slot][rank] = seq at every destination.
//   k_wait_flags (consumer's stream): one lane per sender spins until flags[slot][sender] >= seq, acquire at system scope.
// Receive buffers have kPushSlots = 8 slots (slot = (seq - 1) % 8).  No acknowledgement travels back.  A rank reads the rows
// of step t - L after its kernels of step t and before those of step t + 1 (L = 1 behind the flag wait of this file, L = 2 when
// the observation kernels push the rows themselves); that slot is overwritten by the pushes of step t - L + S, and a push of
// step s is ordered behind the arrival, at the pushing rank, of EVERY rank's rows of step s - 2 (the stream wait of step s - 1
// / the one-wave wait at the head of the env-step kernel).  The reader's own push of step t + 1 comes after its reads, so the
// overwrite is safe when t - L + S - 2 >= t + 1, i.e. S >= L + 3: five slots would do, eight are used.
constexpr int kPushSlots = 8;
constexpr int kMaxWorld = 64;

template <typename V>
__global__ void __launch_bounds__(256) k_push_rows(const V *__restrict__ send, size_t nv, V *const *__restrict__ peer_recv,
                                                   uint32_t *const *__restrict__ peer_flags, int world, int rank, int slot,
                                                   const uint32_t *signal, uint32_t signal_seq, uint32_t flag_seq,
                                                   uint32_t *arrive, uint32_t *timed_out, uint64_t limit_ticks) {
  __shared__ int give_up;
  if (threadIdx.x == 0) {
    give_up = 0;
    if (signal) {
      const uint64_t t0 = wall_clock64();
      uint32_t polls = 0;
      while ((int32_t)(__hip_atomic_load(signal, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) - signal_seq) < 0) {
        if (++polls < 128) __builtin_amdgcn_s_sleep(8);
        else __builtin_amdgcn_s_sleep(127);
        if (wall_clock64() - t0 > limit_ticks) {
          __hip_atomic_store(timed_out, signal_seq ? signal_seq : 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
          give_up = 1;
          break;
        }
      }
    }
  }
  __syncthreads();
  if (give_up) return;  // (every workgroup times out on its own: nobody publishes)
  const int w = blockIdx.x % world, chunk = blockIdx.x / world, chunks = gridDim.x / world;
  V *dst = peer_recv[w] + ((size_t)slot * world + rank) * nv;
  const size_t per = (nv + chunks - 1) / chunks, lo = (size_t)chunk * per, hi = lo + per < nv ? lo + per : nv;
  for (size_t i = lo + threadIdx.x; i < hi; i += 256) dst[i] = send[i];
  __threadfence_system();  // this thread's stores have reached their destinations
  __syncthreads();
  if (threadIdx.x == 0) {
    const uint32_t arrived = __hip_atomic_fetch_add(arrive, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) + 1u;
    if (arrived == gridDim.x) {
      __hip_atomic_store(arrive, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // next launch (same stream)
      for (int d = 0; d < world; ++d)
        __hip_atomic_store(peer_flags[d] + (size_t)slot * world + rank, flag_seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}

// rows stored by the observation kernels themselves are announced by the first kernel of the NEXT step; a consumer that wants
// the latest step's rows (synchronous exchange, flush) announces them here, behind the kernel that stored them
__global__ void k_publish_flags(uint32_t *const *peer_flags, int world, int index, uint32_t seq) {
  const int w = threadIdx.x;
  if (w < world) __hip_atomic_store(peer_flags[w] + index, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

__global__ void k_wait_flags(const uint32_t *flags, int world, uint32_t seq, uint32_t *timed_out, uint64_t limit_ticks) {
  const int w = threadIdx.x;
  if (w >= world) return;
  const uint64_t t0 = wall_clock64();
  uint32_t polls = 0;
  while ((int32_t)(__hip_atomic_load(flags + w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) - seq) < 0) {
    if (++polls < 128) __builtin_amdgcn_s_sleep(8);
    else __builtin_amdgcn_s_sleep(127);
    if (wall_clock64() - t0 > limit_ticks) {
      __hip_atomic_store(timed_out, seq ? seq : 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      return;
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");  // once: the readers are later kernels of this stream (which start with an invalidate of their own)
}

// connection self-test (agx_exchange_push_selftest): one word per destination, stored the way the observation kernels store
// rows (plain store through the mapped address), and the comparison on the receiving side by a kernel that starts after the
// arrival flags were seen -- the path a consumer's kernels take
__global__ void k_selftest_store(float *const *peer_recv, int world, size_t offset, float token) {
  const int w = threadIdx.x;
  if (w < world) peer_recv[w][offset] = token;
}
__global__ void k_selftest_check(const float *recv, int world, size_t slot_base, size_t count, float token_base, uint32_t *bad) {
  const int w = threadIdx.x;
  if (w < world && recv[slot_base + (size_t)w * count] != token_base + (float)w)
    __hip_atomic_fetch_add(bad, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

}  // namespace

struct AgxExchange {
  Rccl *rccl = nullptr;
  ncclComm_t comm = nullptr;
  int rank = 0, world = 1, device = 0;
  // peer push (mode_push): this rank's receive buffer + arrival flags, the peers' mapped through hipIpcMemHandle
  bool mode_push = false, connected = false, flags_uncached = false;
  size_t push_count = 0;             // floats per rank and step
  float *push_recv = nullptr;        // [kPushSlots][world][push_count]
  uint32_t *push_flags = nullptr;    // [kPushSlots][world]
  uint32_t *push_arrive = nullptr;   // workgroup counter of k_push_rows
  void **peer_recv_dev = nullptr;    // device arrays [world]
  uint32_t **peer_flags_dev = nullptr;
  void *opened[2 * kMaxWorld] = {};
  int num_opened = 0;
  uint64_t push_posts = 0;           // posts so far = sequence number of the latest
  uint64_t push_seq_of_parity[2] = {0, 0};
  bool concurrent = false;  // agx_exchange_probe found the comm stream independent of the producer's queue
  uint32_t *probe_flag = nullptr;
  hipStream_t retired[8] = {};
  int num_retired = 0;
  hipStream_t comm_stream = nullptr;
  hipEvent_t ready[2] = {nullptr, nullptr};  // rows[p] written (recorded on the stepping stream)
  hipEvent_t done[2] = {nullptr, nullptr};   // gather of parity p finished (recorded on the comm stream)
  uint32_t *timed_out_host = nullptr;        // pinned, device-visible: set by k_wait_signal when it gives up
  uint32_t *timed_out_dev = nullptr;
  // single producer (stepping thread) / single consumer (worker)
  Job ring[kRing];
  std::atomic<uint64_t> pushed{0}, issued{0};
  uint64_t last_job[2] = {0, 0};  // sequence number (1-based) of the latest job of each parity
  std::atomic<int> failed{0};
  char error[256] = {0};
  std::atomic<bool> stop{false}, sleeping{false};
  std::mutex m;
  std::condition_variable cv;
  std::thread worker;

  void fail_worker(const char *what, const char *detail) {
    snprintf(error, sizeof(error), "%s: %s", what, detail);
    failed.store(1, std::memory_order_release);
  }

  void run() {
    if (hipSetDevice(device) != hipSuccess) fail_worker("hipSetDevice", "worker thread");
    auto last_work = std::chrono::steady_clock::now();
    for (;;) {
      const uint64_t done_n = issued.load(std::memory_order_relaxed);
      if (pushed.load(std::memory_order_acquire) == done_n) {
        if (stop.load(std::memory_order_acquire)) return;
        // stay hot for 2 ms after the last job (a step is tens of microseconds), then sleep
        if (std::chrono::steady_clock::now() - last_work < std::chrono::milliseconds(2)) {
          std::this_thread::yield();
          continue;
        }
        std::unique_lock<std::mutex> lock(m);
        sleeping.store(true, std::memory_order_seq_cst);
        cv.wait(lock, [&] { return pushed.load(std::memory_order_seq_cst) != done_n || stop.load(std::memory_order_seq_cst); });
        sleeping.store(false, std::memory_order_seq_cst);
        continue;
      }
      const Job j = ring[done_n % kRing];
      if (!failed.load(std::memory_order_relaxed) && mode_push) {
        hipError_t e = hipSuccess;
        if (!j.signal) {
          e = hipStreamWaitEvent(comm_stream, ready[j.parity], 0);
          if (e != hipSuccess) fail_worker("hipStreamWaitEvent", hipGetErrorString(e));
        }
        const int slot = (int)((j.push_seq - 1) % kPushSlots);
        const uint32_t *sig = j.signal ? j.signal + j.parity : nullptr;
        // 8 chunks per destination: 64 workgroups at world 8 (0.5 MB of rows per destination: 64 KB per workgroup)
        const int chunks = j.count >= (size_t)1 << 16 ? 8 : 1;
        if (j.count % 4 == 0 && ((uintptr_t)j.send % 16) == 0)
          hipLaunchKernelGGL(k_push_rows<float4>, dim3(world * chunks), dim3(256), 0, comm_stream, (const float4 *)j.send, j.count / 4,
                             (float4 *const *)peer_recv_dev, (uint32_t *const *)peer_flags_dev, world, rank, slot, sig, j.seq, j.push_seq,
                             push_arrive, timed_out_dev, kSpinLimitTicks);
        else
          hipLaunchKernelGGL(k_push_rows<float>, dim3(world * chunks), dim3(256), 0, comm_stream, j.send, j.count,
                             (float *const *)peer_recv_dev, (uint32_t *const *)peer_flags_dev, world, rank, slot, sig, j.seq, j.push_seq,
                             push_arrive, timed_out_dev, kSpinLimitTicks);
        e = hipGetLastError();
        if (e != hipSuccess) fail_worker("k_push_rows", hipGetErrorString(e));
        // the consumer orders itself behind THIS rank's push with an event, never by spinning on this rank's own flag: its
        // stream may share a hardware queue with the communication stream, and a spinning kernel in front of the kernel it
        // waits for never ends (seen as a 10 s time-out, depending on how many streams the process had created before)
        e = hipEventRecord(done[j.parity], comm_stream);
        if (e != hipSuccess) fail_worker("hipEventRecord", hipGetErrorString(e));
      } else if (!failed.load(std::memory_order_relaxed)) {
        hipError_t e;
        if (j.signal) {  // producer = a simulator kernel: spin on its flag, no cross-queue event
          hipLaunchKernelGGL(k_wait_signal, dim3(1), dim3(64), 0, comm_stream, j.signal + j.parity, j.seq, timed_out_dev,
                             kSpinLimitTicks);
          e = hipGetLastError();
          if (e != hipSuccess) fail_worker("k_wait_signal", hipGetErrorString(e));
        } else {
          e = hipStreamWaitEvent(comm_stream, ready[j.parity], 0);
          if (e != hipSuccess) fail_worker("hipStreamWaitEvent", hipGetErrorString(e));
        }
        ncclResult_t r = rccl->AllGather(j.send, j.recv, j.count, ncclFloat32, comm, comm_stream);
        if (r != ncclSuccess) fail_worker("ncclAllGather", rccl->GetErrorString(r));
        e = hipEventRecord(done[j.parity], comm_stream);
        if (e != hipSuccess) fail_worker("hipEventRecord", hipGetErrorString(e));
      }
      issued.store(done_n + 1, std::memory_order_release);
      last_work = std::chrono::steady_clock::now();
    }
  }

  // host-blocks until the worker has ISSUED (not finished) every job up to `seq`: after that
  // done[parity] refers to that job and stream waits on it are meaningful
  void wait_issued(uint64_t seq) const {
    while (issued.load(std::memory_order_acquire) < seq) std::this_thread::yield();
  }
};

extern "C" int agx_exchange_unique_id(const char *rccl_path, void *id_out, int id_bytes) {
  AGX_REQUIRE(id_out && id_bytes == NCCL_UNIQUE_ID_BYTES, "agx_exchange_unique_id: id buffer must be %d bytes", NCCL_UNIQUE_ID_BYTES);
  Rccl *r;
  if (int e = bind_rccl(rccl_path, &r)) return e;
  ncclUniqueId id;
  ncclResult_t st = r->GetUniqueId(&id);
  if (st != ncclSuccess) return agx::fail(AGX_E_LAUNCH, "ncclGetUniqueId: %s", r->GetErrorString(st));
  memcpy(id_out, id.internal, NCCL_UNIQUE_ID_BYTES);
  return AGX_OK;
}

extern "C" int agx_exchange_create(const char *rccl_path, const void *id, int id_bytes, int rank, int world, int device,
                                   AgxExchange **out) {
  AGX_REQUIRE(out && id && id_bytes == NCCL_UNIQUE_ID_BYTES, "agx_exchange_create: null argument or wrong id size");
  AGX_REQUIRE(world >= 1 && rank >= 0 && rank < world, "agx_exchange_create: rank %d of %d", rank, world);
  *out = nullptr;
  Rccl *r;
  if (int e = bind_rccl(rccl_path, &r)) return e;
  hipError_t he = hipSetDevice(device);
  if (he != hipSuccess) return agx::fail(AGX_E_LAUNCH, "hipSetDevice(%d): %s", device, hipGetErrorString(he));
  AgxExchange *x = new AgxExchange();
  x->rccl = r;
  x->rank = rank;
  x->world = world;
  x->device = device;
  ncclUniqueId uid;
  memcpy(uid.internal, id, NCCL_UNIQUE_ID_BYTES);
  ncclResult_t st = r->CommInitRank(&x->comm, world, uid, rank);  // collective over the ranks
  if (st != ncclSuccess) {
    delete x;
    return agx::fail(AGX_E_LAUNCH, "ncclCommInitRank(rank %d of %d): %s", rank, world, r->GetErrorString(st));
  }
  bool ok = hipStreamCreateWithFlags(&x->comm_stream, hipStreamNonBlocking) == hipSuccess;
  ok = ok && hipHostMalloc((void **)&x->timed_out_host, 2 * sizeof(uint32_t), hipHostMallocMapped) == hipSuccess;
  if (ok) {
    x->timed_out_host[0] = x->timed_out_host[1] = 0;
    ok = hipHostGetDevicePointer((void **)&x->timed_out_dev, x->timed_out_host, 0) == hipSuccess;
  }
  ok = ok && hipMalloc((void **)&x->probe_flag, sizeof(uint32_t)) == hipSuccess &&
       hipMemset(x->probe_flag, 0, sizeof(uint32_t)) == hipSuccess;
  for (int p = 0; p < 2 && ok; ++p)
    ok = hipEventCreateWithFlags(&x->ready[p], hipEventDisableTiming) == hipSuccess &&
         hipEventCreateWithFlags(&x->done[p], hipEventDisableTiming) == hipSuccess;
  if (!ok) {
    r->CommDestroy(x->comm);
    delete x;
    return agx::fail(AGX_E_LAUNCH, "agx_exchange_create: stream / event creation failed");
  }
  x->worker = std::thread([x] { x->run(); });
  *out = x;
  return AGX_OK;
}

// ---- peer push: construction ----------------------------------------------------------------------------------------------
extern "C" int agx_exchange_create_push(int rank, int world, int device, size_t count_per_rank, AgxExchange **out) {
  AGX_REQUIRE(out && world >= 1 && world <= kMaxWorld && rank >= 0 && rank < world && count_per_rank > 0,
              "agx_exchange_create_push: rank %d of %d, %zu floats per rank", rank, world, count_per_rank);
  *out = nullptr;
  hipError_t he = hipSetDevice(device);
  if (he != hipSuccess) return agx::fail(AGX_E_LAUNCH, "hipSetDevice(%d): %s", device, hipGetErrorString(he));
  AgxExchange *x = new AgxExchange();
  x->mode_push = true;
  x->rank = rank;
  x->world = world;
  x->device = device;
  x->push_count = count_per_rank;
  const size_t recv_bytes = (size_t)kPushSlots * world * count_per_rank * sizeof(float);
  const size_t flag_bytes = (size_t)(kPushSlots + 1) * world * sizeof(uint32_t);  // (+ one row of flags for the self-test)
  bool ok = hipMalloc((void **)&x->push_recv, recv_bytes) == hipSuccess && hipMemset(x->push_recv, 0, recv_bytes) == hipSuccess;
  // the flags are polled by a running kernel while a PEER device writes them: uncached (fine-grained) memory, like RCCL's own
  // flags; a runtime that refuses it gets ordinary device memory (the accesses are system-scope atomics either way)
  if (ok) {
    x->flags_uncached = hipExtMallocWithFlags((void **)&x->push_flags, flag_bytes, hipDeviceMallocUncached) == hipSuccess;
    if (!x->flags_uncached) {
      (void)hipGetLastError();
      ok = hipMalloc((void **)&x->push_flags, flag_bytes) == hipSuccess;
    }
  }
  ok = ok && hipMemset(x->push_flags, 0, flag_bytes) == hipSuccess;
  ok = ok && hipMalloc((void **)&x->push_arrive, sizeof(uint32_t)) == hipSuccess && hipMemset(x->push_arrive, 0, sizeof(uint32_t)) == hipSuccess;
  ok = ok && hipMalloc((void **)&x->peer_recv_dev, world * sizeof(void *)) == hipSuccess &&
       hipMalloc((void **)&x->peer_flags_dev, world * sizeof(void *)) == hipSuccess;
  ok = ok && hipStreamCreateWithFlags(&x->comm_stream, hipStreamNonBlocking) == hipSuccess;
  ok = ok && hipHostMalloc((void **)&x->timed_out_host, 2 * sizeof(uint32_t), hipHostMallocMapped) == hipSuccess;
  if (ok) {
    x->timed_out_host[0] = x->timed_out_host[1] = 0;
    ok = hipHostGetDevicePointer((void **)&x->timed_out_dev, x->timed_out_host, 0) == hipSuccess;
  }
  ok = ok && hipMalloc((void **)&x->probe_flag, sizeof(uint32_t)) == hipSuccess && hipMemset(x->probe_flag, 0, sizeof(uint32_t)) == hipSuccess;
  for (int p = 0; p < 2 && ok; ++p)
    ok = hipEventCreateWithFlags(&x->ready[p], hipEventDisableTiming) == hipSuccess &&
         hipEventCreateWithFlags(&x->done[p], hipEventDisableTiming) == hipSuccess;
  ok = ok && hipDeviceSynchronize() == hipSuccess;
  if (!ok) {
    const char *why = hipGetErrorString(hipGetLastError());
    agx_exchange_destroy(x);
    return agx::fail(AGX_E_LAUNCH, "agx_exchange_create_push: allocation failed (%s)", why);
  }
  x->worker = std::thread([x] { x->run(); });
  *out = x;
  return AGX_OK;
}

// 2 x 64 bytes: the hipIpcMemHandle of this rank's receive buffer and of its flags; they travel to the peers out of band
extern "C" int agx_exchange_push_export(AgxExchange *x, void *handles_out, int bytes) {
  AGX_REQUIRE(x && x->mode_push && handles_out && bytes == 2 * (int)sizeof(hipIpcMemHandle_t), "agx_exchange_push_export: %d bytes expected",
              2 * (int)sizeof(hipIpcMemHandle_t));
  hipIpcMemHandle_t h[2];
  hipError_t e = hipIpcGetMemHandle(&h[0], x->push_recv);
  if (e == hipSuccess) e = hipIpcGetMemHandle(&h[1], x->push_flags);
  if (e != hipSuccess) return agx::fail(AGX_E_LAUNCH, "hipIpcGetMemHandle: %s", hipGetErrorString(e));
  memcpy(handles_out, h, sizeof(h));
  return AGX_OK;
}

// all_handles: world x (2 x 64) bytes, rank-major, as gathered from agx_exchange_push_export of every rank
extern "C" int agx_exchange_push_connect(AgxExchange *x, const void *all_handles, int bytes) {
  AGX_REQUIRE(x && x->mode_push && !x->connected && all_handles && bytes == x->world * 2 * (int)sizeof(hipIpcMemHandle_t),
              "agx_exchange_push_connect: %d x %d bytes expected", x ? x->world : 0, 2 * (int)sizeof(hipIpcMemHandle_t));
  (void)hipSetDevice(x->device);
  void *recv[kMaxWorld];
  uint32_t *flags[kMaxWorld];
  const hipIpcMemHandle_t *h = (const hipIpcMemHandle_t *)all_handles;
  for (int w = 0; w < x->world; ++w) {
    if (w == x->rank) {
      recv[w] = x->push_recv;
      flags[w] = x->push_flags;
      continue;
    }
    void *pr = nullptr, *pf = nullptr;
    hipError_t e = hipIpcOpenMemHandle(&pr, h[2 * w], hipIpcMemLazyEnablePeerAccess);
    if (e == hipSuccess) {
      x->opened[x->num_opened++] = pr;
      e = hipIpcOpenMemHandle(&pf, h[2 * w + 1], hipIpcMemLazyEnablePeerAccess);
      if (e == hipSuccess) x->opened[x->num_opened++] = pf;
    }
    if (e != hipSuccess) return agx::fail(AGX_E_LAUNCH, "hipIpcOpenMemHandle (rank %d's buffers): %s", w, hipGetErrorString(e));
    recv[w] = pr;
    flags[w] = (uint32_t *)pf;
  }
  if (hipMemcpy(x->peer_recv_dev, recv, x->world * sizeof(void *), hipMemcpyHostToDevice) != hipSuccess ||
      hipMemcpy(x->peer_flags_dev, flags, x->world * sizeof(void *), hipMemcpyHostToDevice) != hipSuccess)
    return agx::fail(AGX_E_LAUNCH, "agx_exchange_push_connect: hipMemcpy failed");
  x->connected = true;
  return AGX_OK;
}

// this rank's receive buffer [slots][world][count_per_rank] (device pointer) and its slot count; the gathered rows of the
// post with sequence number s (1-based, counted per exchange) are slot (s - 1) % slots
extern "C" int agx_exchange_push_buffer(AgxExchange *x, void **recv, int *slots, int *flags_uncached) {
  AGX_REQUIRE(x && x->mode_push && recv && slots, "agx_exchange_push_buffer: null argument");
  *recv = x->push_recv;
  *slots = kPushSlots;
  if (flags_uncached) *flags_uncached = x->flags_uncached ? 1 : 0;
  return AGX_OK;
}

// the addresses, IN THIS PROCESS, of every rank's receive buffer and flag array (own included): what a kernel of this rank
// stores through when it writes its rows at all destinations itself (AgxEnvBuffers.push_delta / push_flags) -- and the
// device-visible word a bounded wait reports its time-out in
extern "C" int agx_exchange_push_peers(AgxExchange *x, void **recv_out, void **flags_out, void **timed_out) {
  AGX_REQUIRE(x && x->mode_push && x->connected && recv_out && flags_out, "agx_exchange_push_peers: connect first");
  (void)hipSetDevice(x->device);
  if (hipMemcpy(recv_out, x->peer_recv_dev, x->world * sizeof(void *), hipMemcpyDeviceToHost) != hipSuccess ||
      hipMemcpy(flags_out, x->peer_flags_dev, x->world * sizeof(void *), hipMemcpyDeviceToHost) != hipSuccess)
    return agx::fail(AGX_E_LAUNCH, "agx_exchange_push_peers: hipMemcpy failed");
  if (timed_out) *timed_out = x->timed_out_dev;
  return AGX_OK;
}

static int check_failed(AgxExchange *x);
// Connection self-test, to be called by EVERY rank after agx_exchange_push_connect and before the first post: each rank stores
// one word into every rank's receive buffer through the mapped addresses, raises a flag there, waits (bounded: timeout_ms)
// for every rank's flag and compares what has arrived.  *passed = 1 when this rank received every rank's word.  Leaves the
// buffers as it found them; the callers synchronise (a barrier of their own) before the first post.  Exists because the
// data path is plain device stores into another device's memory: if a platform maps the buffers but does not deliver such
// stores, this says so before a step depends on it.
extern "C" int agx_exchange_push_selftest(AgxExchange *x, int timeout_ms, int *passed, void *stream) {
  AGX_REQUIRE(x && x->mode_push && x->connected && passed && timeout_ms > 0, "agx_exchange_push_selftest: connect first");
  (void)hipSetDevice(x->device);
  *passed = 0;
  hipStream_t st = (hipStream_t)stream;
  const size_t slot_base = (size_t)(kPushSlots - 1) * x->world * x->push_count;
  const float token_base = 12345.0f;
  uint32_t *test_flags = x->push_flags + (size_t)kPushSlots * x->world;
  x->timed_out_host[1] = 0;
  uint32_t *bad_host = nullptr, *bad_dev = nullptr;
  if (hipHostMalloc((void **)&bad_host, sizeof(uint32_t), hipHostMallocMapped) != hipSuccess ||
      hipHostGetDevicePointer((void **)&bad_dev, bad_host, 0) != hipSuccess)
    return agx::fail(AGX_E_LAUNCH, "agx_exchange_push_selftest: hipHostMalloc failed");
  *bad_host = 0;
  hipLaunchKernelGGL(k_selftest_store, dim3(1), dim3(64), 0, st, (float *const *)x->peer_recv_dev, x->world,
                     slot_base + (size_t)x->rank * x->push_count, token_base + (float)x->rank);
  hipLaunchKernelGGL(k_publish_flags, dim3(1), dim3(64), 0, st, (uint32_t *const *)x->peer_flags_dev, x->world,
                     kPushSlots * x->world + x->rank, 1u);
  hipLaunchKernelGGL(k_wait_flags, dim3(1), dim3(64), 0, st, test_flags, x->world, 1u, x->timed_out_dev + 1,
                     (uint64_t)timeout_ms * 100000ull);
  hipLaunchKernelGGL(k_selftest_check, dim3(1), dim3(64), 0, st, x->push_recv, x->world, slot_base, x->push_count, token_base, bad_dev);
  hipError_t e = hipGetLastError();
  if (e == hipSuccess) e = hipStreamSynchronize(st);
  const uint32_t bad = *bad_host, late = x->timed_out_host[1];
  const bool ok = e == hipSuccess && late == 0 && bad == 0;
  x->timed_out_host[1] = 0;
  (void)hipHostFree(bad_host);
  if (e != hipSuccess) return agx::fail(AGX_E_LAUNCH, "agx_exchange_push_selftest: %s", hipGetErrorString(e));
  // every rank's word has arrived here (or the test failed and the exchange is about to be destroyed): clear what was written
  for (int w = 0; w < x->world && e == hipSuccess; ++w)
    e = hipMemsetAsync(x->push_recv + slot_base + (size_t)w * x->push_count, 0, sizeof(float), st);
  if (e == hipSuccess) e = hipMemsetAsync(test_flags, 0, (size_t)x->world * sizeof(uint32_t), st);
  if (e == hipSuccess) e = hipStreamSynchronize(st);
  if (e != hipSuccess) return agx::fail(AGX_E_LAUNCH, "agx_exchange_push_selftest (clean-up): %s", hipGetErrorString(e));
  if (!ok)  // (the message only: the call itself succeeded)
    (void)agx::fail(AGX_E_LAUNCH, "peer-push self-test, rank %d: %u of %d words wrong, flags %s", x->rank, bad, x->world,
                    late ? "timed out" : "arrived");
  *passed = ok ? 1 : 0;
  return AGX_OK;
}

// `stream` waits (one wave, bounded) until the rows with sequence number `seq` of EVERY rank have arrived in this rank's
// receive buffer: for rows pushed by the observation kernels themselves, where the host keeps the sequence numbers
extern "C" int agx_exchange_push_wait_seq(AgxExchange *x, uint32_t seq, void *stream) {
  AGX_REQUIRE(x && x->mode_push && x->connected && seq != 0u, "agx_exchange_push_wait_seq: bad argument");
  if (int e = check_failed(x)) return e;
  const int slot = (int)((seq - 1) % kPushSlots);
  // (announcing a step twice is harmless; announcing it HERE is right because `stream` has run the kernel that stored its rows)
  hipLaunchKernelGGL(k_publish_flags, dim3(1), dim3(64), 0, (hipStream_t)stream, (uint32_t *const *)x->peer_flags_dev, x->world,
                     slot * x->world + x->rank, seq);
  hipLaunchKernelGGL(k_wait_flags, dim3(1), dim3(64), 0, (hipStream_t)stream, x->push_flags + (size_t)slot * x->world, x->world, seq,
                     x->timed_out_dev, kSpinLimitTicks);
  return agx::check_launch("agx_exchange_push_wait_seq");
}

static int check_failed(AgxExchange *x) {
  if (x->failed.load(std::memory_order_acquire)) return agx::fail(AGX_E_LAUNCH, "agx_exchange worker: %s", x->error);
  if (*(volatile uint32_t *)x->timed_out_host)
    return agx::fail(AGX_E_LAUNCH, "agx_exchange: the rows of step %u were never signalled (10 s): gathered rows are stale",
                     *(volatile uint32_t *)x->timed_out_host);
  return AGX_OK;
}

// HIP multiplexes streams onto a few hardware queues.  If the communication stream shares the producer's
// queue, a k_wait_signal waits for a kernel queued behind itself.  Probe: a waiter on the communication
// stream (5 ms limit) and a setter on the producer's stream; on a time-out retire the stream (kept alive so
// that the runtime hands out a different queue) and try the next.  Returns 1 = flags usable, 0 = use events.
extern "C" int agx_exchange_probe(AgxExchange *x, void *producer_stream) {
  AGX_REQUIRE(x, "agx_exchange_probe: null exchange");
  x->concurrent = false;
  for (int attempt = 0; attempt < 8; ++attempt) {
    x->timed_out_host[1] = 0;
    if (hipMemsetAsync(x->probe_flag, 0, sizeof(uint32_t), (hipStream_t)producer_stream) != hipSuccess ||
        hipStreamSynchronize((hipStream_t)producer_stream) != hipSuccess)
      return agx::fail(AGX_E_LAUNCH, "agx_exchange_probe: memset failed");
    hipLaunchKernelGGL(k_wait_signal, dim3(1), dim3(64), 0, x->comm_stream, x->probe_flag, 1u, x->timed_out_dev + 1, 500000ull);
    hipLaunchKernelGGL(k_set_signal, dim3(1), dim3(64), 0, (hipStream_t)producer_stream, x->probe_flag, 1u);
    if (hipGetLastError() != hipSuccess || hipStreamSynchronize(x->comm_stream) != hipSuccess ||
        hipStreamSynchronize((hipStream_t)producer_stream) != hipSuccess)
      return agx::fail(AGX_E_LAUNCH, "agx_exchange_probe: launch failed");
    if (x->timed_out_host[1] == 0) {
      x->concurrent = true;
      return 1;
    }
    if (x->num_retired == 8) break;
    x->retired[x->num_retired++] = x->comm_stream;
    if (hipStreamCreateWithFlags(&x->comm_stream, hipStreamNonBlocking) != hipSuccess)
      return agx::fail(AGX_E_LAUNCH, "agx_exchange_probe: stream creation failed");
  }
  return 0;
}

extern "C" int agx_exchange_post(AgxExchange *x, int parity, const float *send, float *recv, size_t count_per_rank,
                                 const uint32_t *signal, uint32_t seq, void *producer_stream) {
  AGX_REQUIRE(x && send && (recv || x->mode_push) && (parity == 0 || parity == 1) && count_per_rank > 0, "agx_exchange_post: bad argument");
  AGX_REQUIRE(!signal || x->concurrent, "agx_exchange_post: step_signal mode needs a successful agx_exchange_probe first");
  AGX_REQUIRE(!x->mode_push || (x->connected && count_per_rank == x->push_count),
              "agx_exchange_post (peer push): connect first; %zu floats per rank expected", x->push_count);
  if (int e = check_failed(x)) return e;
  const uint64_t n = x->pushed.load(std::memory_order_relaxed);
  // the ring slot and the two events of this parity are free once the previous job of this parity was issued
  x->wait_issued(x->last_job[parity]);
  AGX_REQUIRE(n - x->issued.load(std::memory_order_acquire) < kRing, "agx_exchange_post: more than %d exchanges in flight", kRing);
  if (!signal) {
    hipError_t e = hipEventRecord(x->ready[parity], (hipStream_t)producer_stream);
    if (e != hipSuccess) return agx::fail(AGX_E_LAUNCH, "hipEventRecord: %s", hipGetErrorString(e));
  }
  uint32_t push_seq = 0;
  if (x->mode_push) {
    push_seq = (uint32_t)(++x->push_posts);
    x->push_seq_of_parity[parity] = x->push_posts;
  }
  x->ring[n % kRing] = Job{send, recv, count_per_rank, parity, signal, seq, push_seq};
  x->last_job[parity] = n + 1;
  x->pushed.store(n + 1, std::memory_order_seq_cst);
  if (x->sleeping.load(std::memory_order_seq_cst)) {
    std::lock_guard<std::mutex> lock(x->m);
    x->cv.notify_one();
  }
  return AGX_OK;
}

extern "C" int agx_exchange_wait(AgxExchange *x, int parity, void *consumer_stream) {
  AGX_REQUIRE(x && (parity == 0 || parity == 1), "agx_exchange_wait: bad argument");
  if (x->last_job[parity] == 0) return AGX_OK;  // nothing was ever posted for this parity
  if (x->mode_push) {
    // this rank's own push: an event behind its kernel (see the worker); the other ranks' rows: the arrival flags of that
    // post's slot, raised by kernels of other processes (spinning on those is safe: different queues)
    x->wait_issued(x->last_job[parity]);
    if (int e = check_failed(x)) return e;
    hipError_t he = hipStreamWaitEvent((hipStream_t)consumer_stream, x->done[parity], 0);
    if (he != hipSuccess) return agx::fail(AGX_E_LAUNCH, "hipStreamWaitEvent: %s", hipGetErrorString(he));
    const uint64_t seq = x->push_seq_of_parity[parity];
    const int slot = (int)((seq - 1) % kPushSlots);
    hipLaunchKernelGGL(k_wait_flags, dim3(1), dim3(64), 0, (hipStream_t)consumer_stream, x->push_flags + (size_t)slot * x->world, x->world,
                       (uint32_t)seq, x->timed_out_dev, kSpinLimitTicks);
    return agx::check_launch("agx_exchange_wait (k_wait_flags)");
  }
  x->wait_issued(x->last_job[parity]);
  if (int e = check_failed(x)) return e;
  // (measured and dropped, profiles/r01_exchange_probe.txt: eliding this wait with a host-side hipEventQuery
  //  when the gather has already finished; pinning the worker next to the stepping thread)
  hipError_t e = hipStreamWaitEvent((hipStream_t)consumer_stream, x->done[parity], 0);
  if (e != hipSuccess) return agx::fail(AGX_E_LAUNCH, "hipStreamWaitEvent: %s", hipGetErrorString(e));
  return AGX_OK;
}

extern "C" int agx_exchange_step(AgxExchange *x, int parity, const float *send, float *recv, size_t count_per_rank,
                                 const uint32_t *signal, uint32_t seq, int wait_parity, void *stream) {
  if (int e = agx_exchange_post(x, parity, send, recv, count_per_rank, signal, seq, stream)) return e;
  if (wait_parity == 0 || wait_parity == 1) return agx_exchange_wait(x, wait_parity, stream);
  return AGX_OK;
}

// anything gone wrong so far (worker thread, a device-side wait that gave up)?  Cheap: two host reads.
extern "C" int agx_exchange_check(AgxExchange *x) {
  AGX_REQUIRE(x, "agx_exchange_check: null exchange");
  return check_failed(x);
}

extern "C" int agx_exchange_info(AgxExchange *x, int *rank, int *world) {
  AGX_REQUIRE(x && rank && world, "agx_exchange_info: null argument");
  if (x->mode_push) {  // the ranks whose buffers were mapped (agx_exchange_push_connect), not a communicator
    *rank = x->rank;
    *world = x->connected ? x->world : 0;
    return AGX_OK;
  }
  ncclResult_t a = x->rccl->CommUserRank(x->comm, rank), b = x->rccl->CommCount(x->comm, world);
  if (a != ncclSuccess || b != ncclSuccess)
    return agx::fail(AGX_E_LAUNCH, "ncclCommUserRank / ncclCommCount: %s", x->rccl->GetErrorString(a != ncclSuccess ? a : b));
  return AGX_OK;
}

extern "C" int agx_exchange_destroy(AgxExchange *x) {
  if (!x) return AGX_OK;
  x->stop.store(true, std::memory_order_seq_cst);
  {
    std::lock_guard<std::mutex> lock(x->m);
    x->cv.notify_one();
  }
  if (x->worker.joinable()) x->worker.join();
  (void)hipSetDevice(x->device);
  if (x->comm_stream) (void)hipStreamSynchronize(x->comm_stream);
  if (x->comm) x->rccl->CommDestroy(x->comm);
  for (int i = 0; i < x->num_opened; ++i) (void)hipIpcCloseMemHandle(x->opened[i]);
  if (x->push_recv) (void)hipFree(x->push_recv);
  if (x->push_flags) (void)hipFree(x->push_flags);
  if (x->push_arrive) (void)hipFree(x->push_arrive);
  if (x->peer_recv_dev) (void)hipFree(x->peer_recv_dev);
  if (x->peer_flags_dev) (void)hipFree(x->peer_flags_dev);
  for (int p = 0; p < 2; ++p) {
    if (x->ready[p]) (void)hipEventDestroy(x->ready[p]);
    if (x->done[p]) (void)hipEventDestroy(x->done[p]);
  }
  if (x->comm_stream) (void)hipStreamDestroy(x->comm_stream);
  for (int i = 0; i < x->num_retired; ++i) (void)hipStreamDestroy(x->retired[i]);
  if (x->probe_flag) (void)hipFree(x->probe_flag);
  if (x->timed_out_host) (void)hipHostFree(x->timed_out_host);
  delete x;
  return AGX_OK;
}
