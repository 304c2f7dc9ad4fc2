// fake_rccl.cpp — TEST DOUBLE of the RCCL entry points libmgs resolves with dlopen (csrc/mgs_api.hip: rccl()), for the ranks of
// a multi-process job that SHARE ONE GPU.  Not part of the product: libmgs loads it only when MGS_RCCL_LIB names it (the tests
// do), and the driver's list of loaded native libraries shows it as tests/helpers/libfakerccl.so.
//
// Why: the N > 1 exchange of mgs_render_gathered (grouped in-place ncclBroadcasts, empty strips, the abort path) needs more than
// one RCCL rank, and no build session has had a second GPU ("two-GPU readiness without a node").  Real RCCL refuses
// two ranks on one device; this double gives the same call sequence something to run against: every collective is executed
// synchronously inside ncclGroupEnd (or at once outside a group) by staging through a POSIX shared-memory segment —
//   root:   hipStreamSynchronize(stream), device -> segment          | barrier |
//   others:                               segment -> device (+ sync) | barrier |
// with a sense-reversing barrier of process-shared atomics that times out (ncclSystemError) and observes ncclCommAbort
// (ncclRemoteError / ncclInProgress never returned: a peer's abort fails the collective instead of hanging it, which is what
// mgs_render_gathered relies on).  Stricter than the real thing in one way only: the data is in place when GroupEnd returns.
//
// Build: hipcc -O2 -shared -fPIC (no device code; it needs the HIP runtime for the copies) -> tests/helpers/libfakerccl.so
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

namespace {
constexpr size_t   kStageBytes = (size_t)160 << 20;  // sparse: pages exist once touched (a 4K RGBA16F frame is 66 MB)
constexpr uint32_t kMagic      = 0x4d475346u;        // "MGSF"

struct Shared
{
  std::atomic<uint32_t> magic;
  std::atomic<uint32_t> world;
  std::atomic<uint32_t> joined;    // ranks that have called ncclCommInitRank
  std::atomic<uint32_t> left;      // ranks that have destroyed / aborted
  std::atomic<uint32_t> aborted;
  std::atomic<uint32_t> arrive;    // barrier: arrivals of the current generation
  std::atomic<uint32_t> gen;       // barrier: generation
  uint32_t              pad[9];
  unsigned char         stage[1];  // kStageBytes
};

struct Comm
{
  Shared* sh    = nullptr;
  int     rank  = 0;
  int     world = 1;
  char    name[64];
};

struct Op
{
  int         kind;  // 0 broadcast, 1 all-gather
  const void* send;
  void*       recv;
  size_t      bytes;
  int         root;
  Comm*       comm;
  hipStream_t stream;
};
thread_local int             t_depth = 0;
thread_local std::vector<Op> t_ops;

double timeoutSeconds()
{
  const char* e = std::getenv("MGS_FAKE_RCCL_TIMEOUT");
  return e ? std::atof(e) : 120.0;
}

// every rank of the communicator arrives; false: timed out or a rank aborted
bool barrier(Comm* c)
{
  Shared*        s   = c->sh;
  const uint32_t g   = s->gen.load(std::memory_order_acquire);
  const uint32_t pos = s->arrive.fetch_add(1, std::memory_order_acq_rel) + 1;
  if(pos == (uint32_t)c->world)
  {
    s->arrive.store(0, std::memory_order_release);
    s->gen.fetch_add(1, std::memory_order_acq_rel);
    return s->aborted.load() == 0;
  }
  const auto t0 = std::chrono::steady_clock::now();
  while(s->gen.load(std::memory_order_acquire) == g)
  {
    if(s->aborted.load() != 0)
      return false;
    if(std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > timeoutSeconds())
      return false;
    std::this_thread::sleep_for(std::chrono::microseconds(50));
  }
  return s->aborted.load() == 0;
}

size_t dtBytes(ncclDataType_t t)
{
  switch(t)
  {
    case ncclInt8: case ncclUint8: return 1;
    case ncclFloat16: case ncclBfloat16: return 2;
    case ncclInt32: case ncclUint32: case ncclFloat32: return 4;
    case ncclInt64: case ncclUint64: case ncclFloat64: return 8;
    default: return 1;
  }
}

ncclResult_t run(std::vector<Op>& ops)
{
  if(ops.empty())
    return ncclSuccess;
  Comm* c = ops[0].comm;
  // staging offsets: the same on every rank (every rank issues the same sequence of collectives)
  std::vector<size_t> off(ops.size());
  size_t              total = 0;
  for(size_t i = 0; i < ops.size(); ++i)
  {
    off[i] = total;
    total += ops[i].kind == 0 ? ops[i].bytes : ops[i].bytes * (size_t)c->world;
    total = (total + 255) & ~(size_t)255;
  }
  if(total > kStageBytes)
    return ncclInvalidArgument;
  // phase 1: what this rank contributes, device -> segment (after everything queued before on the stream)
  for(size_t i = 0; i < ops.size(); ++i)
  {
    const Op& o = ops[i];
    if(hipStreamSynchronize(o.stream) != hipSuccess)
      return ncclUnhandledCudaError;
    if(o.kind == 0 && o.root == c->rank)
    {
      if(hipMemcpy(c->sh->stage + off[i], o.send, o.bytes, hipMemcpyDeviceToHost) != hipSuccess)
        return ncclUnhandledCudaError;
    }
    else if(o.kind == 1)
    {
      if(hipMemcpy(c->sh->stage + off[i] + (size_t)c->rank * o.bytes, o.send, o.bytes, hipMemcpyDeviceToHost) != hipSuccess)
        return ncclUnhandledCudaError;
    }
  }
  std::atomic_thread_fence(std::memory_order_seq_cst);
  if(!barrier(c))
    return c->sh->aborted.load() ? ncclRemoteError : ncclSystemError;
  // phase 2: what this rank receives, segment -> device
  for(size_t i = 0; i < ops.size(); ++i)
  {
    const Op& o = ops[i];
    if(o.kind == 0 && o.root != c->rank)
    {
      if(hipMemcpy(o.recv, c->sh->stage + off[i], o.bytes, hipMemcpyHostToDevice) != hipSuccess)
        return ncclUnhandledCudaError;
    }
    else if(o.kind == 1)
    {
      if(hipMemcpy(o.recv, c->sh->stage + off[i], o.bytes * (size_t)c->world, hipMemcpyHostToDevice) != hipSuccess)
        return ncclUnhandledCudaError;
    }
  }
  if(hipDeviceSynchronize() != hipSuccess)
    return ncclUnhandledCudaError;
  if(!barrier(c))  // nobody overwrites the segment before everybody has read it
    return c->sh->aborted.load() ? ncclRemoteError : ncclSystemError;
  return ncclSuccess;
}
}  // namespace

extern "C" {

ncclResult_t ncclGetUniqueId(ncclUniqueId* id)
{
  if(!id)
    return ncclInvalidArgument;
  static std::atomic<uint32_t> counter{0};
  std::memset(id, 0, sizeof(*id));
  const auto now = std::chrono::steady_clock::now().time_since_epoch().count();
  std::snprintf(id->internal, sizeof(id->internal), "/mgs_fakerccl_%d_%u_%llx", (int)getpid(), counter.fetch_add(1), (unsigned long long)now);
  return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t* comm, int nranks, ncclUniqueId id, int rank)
{
  if(!comm || nranks < 1 || rank < 0 || rank >= nranks || id.internal[0] != '/')
    return ncclInvalidArgument;
  Comm* c = new Comm;
  std::memset(c->name, 0, sizeof(c->name));
  std::strncpy(c->name, id.internal, sizeof(c->name) - 1);
  const size_t bytes = sizeof(Shared) + kStageBytes;
  const int    fd    = shm_open(c->name, O_CREAT | O_RDWR, 0600);
  if(fd < 0 || ftruncate(fd, (off_t)bytes) != 0)
  {
    if(fd >= 0)
      close(fd);
    delete c;
    return ncclSystemError;
  }
  void* p = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
  close(fd);
  if(p == MAP_FAILED)
  {
    delete c;
    return ncclSystemError;
  }
  c->sh    = reinterpret_cast<Shared*>(p);  // a fresh segment is zero-filled: every counter starts at 0
  c->rank  = rank;
  c->world = nranks;
  c->sh->magic.store(kMagic);
  c->sh->world.store((uint32_t)nranks);
  c->sh->joined.fetch_add(1);
  // like the real call: returns when every rank has joined
  const auto t0 = std::chrono::steady_clock::now();
  while(c->sh->joined.load() < (uint32_t)nranks)
  {
    if(std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > timeoutSeconds())
    {
      munmap(p, bytes);
      shm_unlink(c->name);
      delete c;
      return ncclSystemError;
    }
    std::this_thread::sleep_for(std::chrono::microseconds(200));
  }
  *comm = reinterpret_cast<ncclComm_t>(c);
  return ncclSuccess;
}

static ncclResult_t leave(ncclComm_t comm, bool abort)
{
  Comm* c = reinterpret_cast<Comm*>(comm);
  if(!c)
    return ncclInvalidArgument;
  if(abort)
    c->sh->aborted.store(1);
  const uint32_t gone = c->sh->left.fetch_add(1) + 1;
  const bool     last = gone >= (uint32_t)c->world;
  munmap(c->sh, sizeof(Shared) + kStageBytes);
  if(last || abort)
    shm_unlink(c->name);  // (idempotent: a second unlink fails with ENOENT)
  delete c;
  return ncclSuccess;
}
ncclResult_t ncclCommDestroy(ncclComm_t comm) { return leave(comm, false); }
ncclResult_t ncclCommAbort(ncclComm_t comm) { return leave(comm, true); }

ncclResult_t ncclGroupStart()
{
  ++t_depth;
  return ncclSuccess;
}
ncclResult_t ncclGroupEnd()
{
  if(t_depth <= 0)
    return ncclInvalidUsage;
  if(--t_depth > 0)
    return ncclSuccess;
  std::vector<Op> ops;
  ops.swap(t_ops);
  return run(ops);
}

ncclResult_t ncclBroadcast(const void* sendbuff, void* recvbuff, size_t count, ncclDataType_t datatype, int root, ncclComm_t comm,
                           hipStream_t stream)
{
  Comm* c = reinterpret_cast<Comm*>(comm);
  if(!c || root < 0 || root >= c->world)
    return ncclInvalidArgument;
  t_ops.push_back(Op{0, sendbuff, recvbuff, count * dtBytes(datatype), root, c, stream});
  if(t_depth > 0)
    return ncclSuccess;
  std::vector<Op> ops;
  ops.swap(t_ops);
  return run(ops);
}

ncclResult_t ncclAllGather(const void* sendbuff, void* recvbuff, size_t sendcount, ncclDataType_t datatype, ncclComm_t comm,
                           hipStream_t stream)
{
  Comm* c = reinterpret_cast<Comm*>(comm);
  if(!c)
    return ncclInvalidArgument;
  t_ops.push_back(Op{1, sendbuff, recvbuff, sendcount * dtBytes(datatype), 0, c, stream});
  if(t_depth > 0)
    return ncclSuccess;
  std::vector<Op> ops;
  ops.swap(t_ops);
  return run(ops);
}

const char* ncclGetErrorString(ncclResult_t r)
{
  switch(r)
  {
    case ncclSuccess: return "fake rccl: success";
    case ncclSystemError: return "fake rccl: a peer did not arrive in time";
    case ncclRemoteError: return "fake rccl: a peer aborted the communicator";
    case ncclInvalidArgument: return "fake rccl: invalid argument";
    case ncclInvalidUsage: return "fake rccl: invalid usage";
    default: return "fake rccl: error";
  }
}

// lets a test see which library answered
const char* mgs_fake_rccl_marker() { return "tests/helpers/libfakerccl.so"; }
}
