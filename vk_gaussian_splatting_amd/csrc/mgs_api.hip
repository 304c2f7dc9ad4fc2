// mgs_api.hip — implementation of include/mgs.h: handles, device scene, frame orchestration.
//
// Host orchestration replaces GaussianSplatting::processSortingOnGPU / drawSplatPrimitives
// (src/gaussian_splatting.cpp:1298-1465): everything after the parameter upload is GPU-driven —
// counts stay on the device (IndirectParams, shaders/shaderio.h:343-356) and are read back only
// for statistics, like readBackIndirectParametersIfNeeded (:1536).
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <rccl/rccl.h>  // types only: the library itself is resolved with dlopen on first use

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstddef>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <execution>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <map>
#include <vector>

#include "../../include/mgs.h"
#include "device_types.h"
#include "host_model.h"
#include "sort_plan.h"

namespace mgs {
// kernels_*.hip
void launchProject(hipStream_t stream, const FrameArgs& args, const FrameArgs* dArgs, bool full, FrameCounters* ctr, uint2* slotPairs,
                   uint32_t* slotCount, SplatRec* rec, uint32_t* rect, uint32_t* slotHist2,
                   uint32_t* top16Rec, uint32_t* top16Count, OsPlan* osPlan, const uint32_t* order);
void launchFrameInit(hipStream_t stream, uint2* ranges, uint32_t nTiles);
void launchBinning(hipStream_t stream, const uint32_t* idsX, const uint32_t* idsY, const SortPlan* planKeys,
                   const uint32_t* rect, uint32_t* blockCount, uint32_t maxBlocks, FrameCounters* ctr, uint32_t* sortedRect,
                   uint32_t* splatOffset, uint32_t* chunkStart, uint32_t* pairKey, uint32_t* pairVal, uint32_t capacity,
                   int binsX, bool gatherRects);
bool directBinningSupported(int binsX, int binsY);
void launchDirectBinning(hipStream_t stream, const uint32_t* idsX, const uint32_t* idsY, const SortPlan* planKeys,
                         const uint32_t* rect, const uint16_t* sortedCode16, uint64_t* maskBuf, uint32_t maxSplats, uint32_t* binHist, uint32_t pStride,
                         uint32_t* binTotal, uint32_t* binList, uint2* ranges, FrameCounters* ctr, uint32_t capacity,
                         int binsX, int binsY, uint32_t* binCost);
void launchTileRanges(hipStream_t stream, const uint32_t* keyX, const uint32_t* keyY, const SortPlan* planPairs,
                      uint2* ranges);
void launchComposite(hipStream_t stream, const FrameArgs& A, const uint2* ranges, const uint32_t* valX,
                     const uint32_t* valY, const SortPlan* planPairs, const SplatRec* rec, void* image, int halfOut,
                     int shFormat, FrameCounters* ctr, float* outDepth, uint32_t* outSplatId, const void* instTable, const FrameArgs* dArgs,
                     float4* outNormal, uint32_t* binCost);
void launchProjectGut(hipStream_t stream, const FrameArgs& args, const FrameArgs* dArgs, int shFormat, FrameCounters* ctr,
                      uint2* slotPairs, uint32_t* slotCount, GutRec* rec, uint32_t* rect,
                      uint32_t* slotHist2, uint32_t* top16Rec, uint32_t* top16Count, OsPlan* osPlan, const uint32_t* order);
void launchCompositeGut(hipStream_t stream, const FrameArgs& A, const FrameArgs* dArgs, const uint2* ranges, const uint32_t* valX,
                        const uint32_t* valY, const SortPlan* planPairs, const GutRec* rec, void* image, int halfOut,
                        FrameCounters* ctr, int shFormat, float* outDepth, uint32_t* outSplatId, float4* outNormal);
constexpr uint32_t kPart = 2048;  // == kPrjPart == kSortPart == kBinPart
}  // namespace mgs

using namespace mgs;

#define HIPCHK(expr)                                                                                                    \
  do                                                                                                                    \
  {                                                                                                                     \
    hipError_t e_ = (expr);                                                                                             \
    if(e_ != hipSuccess)                                                                                                \
    {                                                                                                                   \
      setError(std::string("HIP error: ") + hipGetErrorString(e_) + " at " #expr);                                      \
      return MGS_ERR_DEVICE;                                                                                            \
    }                                                                                                                   \
  } while(0)

struct MgsSplatSet_t
{
  std::shared_ptr<HostSplatSet> data;
};

namespace {

struct DeviceSet
{
  std::shared_ptr<HostSplatSet> host;
  float*   centers = nullptr;
  float*   cov6    = nullptr;
  void*    rgba    = nullptr;
  void*    sh      = nullptr;
  uint32_t count   = 0;
  int      shDegree = 0, shStride = 0;  // logical elements per splat (0/9/24/45)
  int      shPitch = 0;                  // stored elements per splat: shStride padded to a 16-byte multiple
  int      shFormat = -1, rgbaFormat = -1;
  float*   maxScale = nullptr;             // [count] max(exp(scale)), storage order
  float*   rgbaF32 = nullptr;              // [count*4] colours as the shaders read them back (== rgba when stored as fp32)
  float*   alpha = nullptr;                // [count] dequantised opacity, storage order (the projection's view of rgba)
  float*   scales = nullptr;               // [count*3] log scales, storage order   } integrated-normal side output
  float*   rotations = nullptr;            // [count*4] (w,x,y,z), storage order    }
  float*   partBox = nullptr;              // [ceil(count/2048)][8]: AABB of the centres + footprint radius bound
  std::vector<uint32_t> newToOld, oldToNew;  // storage order (Morton) <-> caller's order
};

struct Instance
{
  int   set;  // index into sets
  float M[16];
};

template <typename T>
struct DevBuf
{
  T*     p = nullptr;
  size_t n = 0;
  int    ensure(size_t count)
  {
    if(count <= n)
      return MGS_OK;
    if(p)
      (void)hipFree(p);
    p = nullptr;
    n = 0;
    if(hipMalloc((void**)&p, std::max<size_t>(count, 1) * sizeof(T)) != hipSuccess)
    {
      setError("device allocation failed (" + std::to_string(count * sizeof(T)) + " bytes)");
      return MGS_ERR_OOM;
    }
    n = count;
    return MGS_OK;
  }
  void release()
  {
    if(p)
      (void)hipFree(p);
    p = nullptr;
    n = 0;
  }
};

// a typed window into memory somebody else owns (the members of FrameState below)
template <class T>
struct DevView
{
  T* p = nullptr;
};

// One device block per handle: the per-frame device state that every frame starts from zero (counters, sort plans), followed by
// the frame's constants.  A frame begins with ONE upload that carries the zeros along with the constants — no kernel has to
// sweep them (the frame's first kernel used to).
struct FrameState
{
  FrameCounters ctr;
  FramePlans    plans;  // keys, pairs, os (sort_plan.h)
  FrameArgs     args;   // view / proj, instances, knobs: the kernels read them through this pointer, so a captured frame graph
                        // replays with nothing but the upload
};

// CPU async sorter: same protocol as SplatSorterAsync (src/splat_sorter_async.h:41-48,84-128)
struct CpuSorter
{
  enum State { READY, SORTING, SORTED, FAILURE, SHUTDOWN };
  std::thread             worker;
  std::mutex              mtx;
  std::condition_variable cv;
  State                   state = READY;
  bool                    started = false;
  // job
  struct Job
  {
    float dir[3], cop[3];
    bool  frontToBack;
    struct Inst
    {
      std::shared_ptr<HostSplatSet> set;
      float                         M[16];
      uint32_t                      offset, count;
    };
    std::vector<Inst> inst;
    uint32_t          total;
  } job;
  float                 lastDir[3] = {0, 0, 0}, lastCop[3] = {0, 0, 0};  // viewpoint of the last sort that was started
  bool                  haveLast   = false;
  std::vector<float>    distances;
  std::vector<uint32_t> indices;
  double                distMs = 0, sortMs = 0;

  void run()
  {
    std::unique_lock<std::mutex> lk(mtx);
    for(;;)
    {
      cv.wait(lk, [&] { return state == SORTING || state == SHUTDOWN; });
      if(state == SHUTDOWN)
        return;
      lk.unlock();
      innerSort();
      lk.lock();
      if(state != SHUTDOWN)
        state = SORTED;
      cv.notify_all();
    }
  }
  // SplatSorterAsync::innerSort (src/splat_sorter_async.cpp:92-141): plane distance keys, then
  // std::sort(par_unseq) of the index array with a comparator on the distances.
  void innerSort()
  {
    const auto  t0 = std::chrono::high_resolution_clock::now();
    const float plane[4] = {job.dir[0], job.dir[1], job.dir[2],
                            -job.dir[0] * job.cop[0] - job.dir[1] * job.cop[1] - job.dir[2] * job.cop[2]};
    const float divider = 1.0f / std::sqrt(plane[0] * plane[0] + plane[1] * plane[1] + plane[2] * plane[2]);
    distances.resize(job.total);
    indices.resize(job.total);
    for(const auto& I : job.inst)
    {
      const float* pos = I.set->positions.data();
      parallelBatches(I.count, [&](size_t s) {
        const float v[4] = {pos[3 * s], pos[3 * s + 1], pos[3 * s + 2], 1.0f};
        float       p[4];
        mat4MulVec4(I.M, v, p);
        distances[I.offset + s] = std::fabs(plane[0] * p[0] + plane[1] * p[1] + plane[2] * p[2] + plane[3]) * divider;
        indices[I.offset + s]   = I.offset + (uint32_t)s;
      });
    }
    const auto   t1 = std::chrono::high_resolution_clock::now();
    const float* d  = distances.data();
    if(job.frontToBack)
      std::sort(std::execution::par_unseq, indices.begin(), indices.end(), [d](uint32_t i, uint32_t j) { return d[i] < d[j]; });
    else
      std::sort(std::execution::par_unseq, indices.begin(), indices.end(), [d](uint32_t i, uint32_t j) { return d[i] > d[j]; });
    const auto t2 = std::chrono::high_resolution_clock::now();
    distMs        = std::chrono::duration<double, std::milli>(t1 - t0).count();
    sortMs        = std::chrono::duration<double, std::milli>(t2 - t1).count();
  }
  void ensureStarted()
  {
    if(!started)
    {
      started = true;
      worker  = std::thread([this] { run(); });
    }
  }
  void shutdown()
  {
    if(!started)
      return;
    {
      std::lock_guard<std::mutex> lk(mtx);
      state = SHUTDOWN;
    }
    cv.notify_all();
    worker.join();
    started = false;
  }
};

}  // namespace

// What a commit produces: the resident splat buffers and the instance list.  ONE copy per scene, shared read-only by the scene
// handle and every frame context created from it — the reference likewise keeps one copy of the splat buffers however many
// frames are in flight, which is why processUpdateRequests waits for the device before it touches them
// (src/gaussian_splatting.cpp:1092-1111); mgs_scene_commit does the same over all contexts' streams.
struct SceneData
{
  int                    device = 0;
  std::vector<DeviceSet> sets;
  std::vector<Instance>  instances;
  bool                   committed = false;
  int                    shFormat = 0, rgbaFormat = 0;
  uint32_t               totalSplats = 0, totalParts = 0;
  DevBuf<CompositeArgs::Inst> compInst;  // SH table of all instances for the compositor (scenes with > 16 instances)
  uint64_t               epoch = 0;      // bumped by every commit: a context re-sizes its working set when it lags
  std::mutex             mtx;            // guards `handles`
  std::vector<MgsScene_t*> handles;      // the owning scene and its live frame contexts
  ~SceneData();
};

struct MgsScene_t
{
  int         device = 0;
  hipStream_t stream = nullptr, ownStream = nullptr;

  std::shared_ptr<SceneData> d;   // shared with the frame contexts (mgs_frame_context_create)
  uint64_t    listCapacityOverride = 0;  // mgs_scene_set_list_capacity: entries of the per-bin lists (0 = 32 per splat)
  bool        isContext = false;  // a frame context: own stream, working buffers and graphs; the scene data is the parent's
  uint64_t    wsEpoch   = ~0ull;  // d->epoch the working set below was sized for

  // frame buffers
  DevBuf<uint2>         pairA, pairB;  // ping-pong of the key sort; the project kernels leave their (key, id) pairs in B, one
                                       // slot of 2048 entries per partition
  DevBuf<uint32_t>      slotHist2, top16Rec, top16Count, osStatus;  // what the key sort needs besides (k_osort.hip)
  DevBuf<uint32_t>      prjOrder;                    // [parts] dispatch order of the project kernels' partitions (identity until a key sort has run)
  DevBuf<uint32_t>      slotCount, chunkSum, runTab;  // pairs per slot; the tables of the sort's virtual pass 0 (k_os_prepare)
  DevBuf<uint16_t>      sortedCode16;                // the bin rectangles' codes in sorted order (they ride through the key sort)
  DevBuf<uint32_t>      binCost;                     // [256] per bin: its slowest region in the last frame -> the next frame's bin order
  DevBuf<uint32_t>      keysA, idsA;  // the sorted ids (and, for the sort-only hook, the sorted keys)
  DevBuf<uint32_t>      rect, partHist, blockCount;
  DevBuf<uint32_t>      sortedRect, splatOffset, chunkStart;
  DevBuf<uint64_t>      dbinMasks;
  DevBuf<FrameState>    fstate;      // counters + plans + this frame's constants: one block, one upload per frame (FrameState)
  DevView<FrameArgs>    dArgs;       // = &fstate->args
  // the upload's source: zeros for counters and plans, then the constants.  Pinned, a ring of four guarded by events: the copy
  // is asynchronous for real (nothing is staged behind the caller's back), and a buffer is rewritten only after the copy
  // that read it has completed — with several frames in flight a late read would hand one frame the constants of another
  static constexpr int kUpRing = 4;
  uint8_t*              upBuf[kUpRing] = {};
  hipEvent_t            upEv[kUpRing]  = {};
  bool                  upBusy[kUpRing] = {};
  uint32_t              upNext = 0;
  struct GraphKey
  {
    int32_t v[16];
    const void* p[2];
    bool operator<(const GraphKey& o) const { return std::memcmp(this, &o, sizeof(*this)) < 0; }
  };
  std::map<GraphKey, hipGraphExec_t> graphs;  // captured frames, one per (resolution, strip, mode); cleared at commit
  bool                  graphOk = true;
  DevBuf<float>         surfDepth;   // FTB side outputs of the last frame rendered with surface_outputs
  DevBuf<uint32_t>      surfId;
  DevBuf<float4>        surfNormal;
  DevBuf<float4>        accum;       // temporal accumulation (post.comp.slang): running mean of the frame samples, fp32
  bool                  haveSurface = false;
  DevBuf<SplatRec>      rec;
  DevBuf<GutRec>        recGut;  // 3DGUT records (96 B per splat), allocated by the first 3DGUT frame
  DevBuf<uint32_t>      pairKey0, pairVal0, pairKey1, pairVal1;
  DevBuf<uint2>         ranges;
  DevBuf<uint8_t>       image;
  DevView<FrameCounters> ctr;    // = &fstate->ctr
  DevView<FramePlans>    plans;  // = &fstate->plans
  uint32_t              pairCapacity = 0, pStride = 0;

  // pinned readback
  FrameCounters* hCtr   = nullptr;
  FramePlans*    hPlans = nullptr;

  static constexpr int kRing = 128;
  hipEvent_t ev[8] = {};              // [0..2] sort-only hook, [6..7] raw radix sort
  hipEvent_t evRing[kRing][7] = {};   // per-frame stage brackets ([6]: end of the partition cull), so timed frames need no host sync
  uint64_t   frameIndex = 0;          // frames rendered with collect_timings
  bool       evReady = false;

  // last frame
  MgsFrameParams lastParams{};
  bool           haveFrame = false, lastTimed = false, lastWasSortOnly = false;
  int            lastBinShift[2] = {4, 3};  // the last frame's bin size (the adaptive policy may pick another one for the next frame)
  int            lastRide[5] = {0, 0, 0, 0, 0};  // the last frame's rideShift, code bits, binsX, binsY, 1 if the GPU key sort ran (mgs_frame_download_projected)
  bool           lastListsPartial = false;  // the last frame came from mgs_render_gathered: its bin lists cover this rank's rows only
  size_t         imageBytes = 0, imageRowBytes = 0;
  MgsSortOut     lastSort{};

  DevBuf<uint32_t>      rsKeys, rsVals, rsHist, rsCount;  // mgs_radix_sort_u32 scratch
  DevBuf<uint2>         rsPairA, rsPairB;
  DevBuf<uint32_t>      rsStatus;
  uint32_t              rsStatusParts = 0;  // partition bound the look-back words of the stand-alone sort are laid out for
  DevBuf<OsPlan>        rsOsPlan;
  DevBuf<SortPlan>      rsPlan;
  // multi-GPU strips (RCCL)
  ncclComm_t            comm = nullptr;
  int                   commRank = 0, commWorld = 1;
  std::vector<int32_t>  stripBounds;  // [world + 1] tile rows; empty = equal strips

  // Adaptive bin size (round 6).  A region walks its bin's list until its pixels saturate; where regions never
  // saturate (sparse or translucent scenes) every region re-scans the WHOLE list of a 256x128-px bin, and 128x128-px bins halve
  // that — measured in round 5: train-sized composite 137 -> 124 us for +6 us of binning, fog -2 %, the benchmark scene +4.6 %
  // (its regions stop after a few hundred entries whatever the list's length).  The signal is the fraction of the list entries
  // the regions look at, scanned / (D x regions per bin): 0.008 on the benchmark scene, 0.034 fog, 0.056 sparse, 0.077 train-sized.
  // Sampled every 32nd frame of a context (one 4 KB device-to-host copy of the statistics lines behind the frame), applied eight
  // frames later — by frame count, not by when the copy lands, so that a sequence renders the same way every time.  Scheduling
  // only: frames do not depend on the bin size (test_binning_paths_bit_identical).  MGS_BIN_ADAPT=0 turns it off; MGS_BIN_SHIFT wins.
  struct BinPolicy
  {
    int        fine = 0;           // 1: 128x128-px bins (where the frame allows: <= 256 bins)
    uint32_t   frames = 0;         // eligible frames rendered by this context
    bool       pending = false;    // a sample is in flight / waiting to be applied
    int        sampledFine = 0, sampledRegions = 64;
    uint32_t   sampledRegionCount = 1;  // 32x16-px regions of the sampled frame (its strip)
    hipEvent_t ev = nullptr;
    uint32_t*  host = nullptr;     // pinned, kFrameStatSlots x 32 words
    float      lastRatio = 0.0f;
  } binPolicy;

  CpuSorter             cpu;
  std::vector<float>    cpuDistances;  // distances of the consumed sort (swapped out under the sorter's lock, like the indices)
  std::vector<uint32_t> cpuIndices;  // consumed result (caller's id space)
  std::vector<uint32_t> cpuStorageIds;
  bool                  cpuHaveIndices = false;
  DevBuf<float>         cpuDistDev;
};

// No exception crosses the C ABI (include/mgs.h): every entry point that can allocate host memory or parse untrusted
// input runs inside this guard.
template <typename Fn>
static int guarded(const char* what, Fn&& fn) noexcept
{
  try
  {
    return fn();
  }
  catch(const std::bad_alloc&)
  {
    setError(std::string(what) + ": out of host memory");
    return MGS_ERR_OOM;
  }
  catch(const std::length_error&)
  {
    setError(std::string(what) + ": size exceeds what the host can allocate");
    return MGS_ERR_OOM;
  }
  catch(const std::exception& e)
  {
    setError(std::string(what) + ": " + e.what());
    return MGS_ERR_FORMAT;
  }
  catch(...)
  {
    setError(std::string(what) + ": unknown failure");
    return MGS_ERR_FORMAT;
  }
}

// ------------------------------------------------------------------------------------------------
extern "C" {

const char* mgs_last_error(void) { return lastError(); }
const char* mgs_version(void) { return "mgs 0.4 (gfx950, ABI 4)"; }

static int mgs_splatset_load_impl(const char* path, MgsSplatSet* out);
int mgs_splatset_load(const char* path, MgsSplatSet* out)
{
  return guarded("mgs_splatset_load", [&] { return mgs_splatset_load_impl(path, out); });
}
static int mgs_splatset_load_impl(const char* path, MgsSplatSet* out)
{
  if(!path || !out)
  {
    setError("mgs_splatset_load: null argument");
    return MGS_ERR_INVALID_ARG;
  }
  std::string p(path), ext;
  const auto  dot = p.find_last_of('.');
  if(dot != std::string::npos)
    ext = p.substr(dot);
  for(auto& c : ext)
    c = (char)std::tolower((unsigned char)c);  // hasExtension lower-cases, src/utilities.h:65-70
  auto set = std::make_shared<HostSplatSet>();
  int  rc;
  if(ext == ".splat")
    rc = loadSplat(p, *set);
  else if(ext == ".spz")
    rc = loadSpz(p, *set);
  else
    rc = loadPly(p, *set);  // the reference hands every other extension to miniply
  if(rc != MGS_OK)
    return rc;
  *out = new MgsSplatSet_t{set};
  return MGS_OK;
}

static int mgs_splatset_from_arrays_impl(const MgsSplatSetView* v, MgsSplatSet* out);
int mgs_splatset_from_arrays(const MgsSplatSetView* v, MgsSplatSet* out)
{
  return guarded("mgs_splatset_from_arrays", [&] { return mgs_splatset_from_arrays_impl(v, out); });
}
static int mgs_splatset_from_arrays_impl(const MgsSplatSetView* v, MgsSplatSet* out)
{
  if(!v || !out || !v->positions || !v->f_dc || !v->opacity || !v->scale || !v->rotation)
  {
    setError("mgs_splatset_from_arrays: null argument");
    return MGS_ERR_INVALID_ARG;
  }
  if(v->splat_count == 0 || v->splat_count > 0xFFFFFFFFull)
  {
    setError("mgs_splatset_from_arrays: splat_count must be in [1, 2^32)");
    return MGS_ERR_INVALID_ARG;
  }
  if(v->f_rest_per_splat != 0 && v->f_rest == nullptr)
  {
    setError("mgs_splatset_from_arrays: f_rest is null but f_rest_per_splat != 0");
    return MGS_ERR_INVALID_ARG;
  }
  if(v->f_rest_per_splat % 3 != 0 || v->f_rest_per_splat > 45)
  {
    setError("mgs_splatset_from_arrays: f_rest_per_splat must be a multiple of 3, at most 45");
    return MGS_ERR_INVALID_ARG;
  }
  const size_t n   = (size_t)v->splat_count;
  auto         set = std::make_shared<HostSplatSet>();
  set->positions.assign(v->positions, v->positions + 3 * n);
  set->f_dc.assign(v->f_dc, v->f_dc + 3 * n);
  if(v->f_rest_per_splat)
    set->f_rest.assign(v->f_rest, v->f_rest + (size_t)v->f_rest_per_splat * n);
  set->opacity.assign(v->opacity, v->opacity + n);
  set->scale.assign(v->scale, v->scale + 3 * n);
  set->rotation.assign(v->rotation, v->rotation + 4 * n);
  *out = new MgsSplatSet_t{set};
  return MGS_OK;
}

int mgs_splatset_view(MgsSplatSet set, MgsSplatSetView* out)
{
  if(!set || !out)
  {
    setError("mgs_splatset_view: null argument");
    return MGS_ERR_INVALID_ARG;
  }
  const HostSplatSet& s = *set->data;
  out->positions        = s.positions.data();
  out->f_dc             = s.f_dc.data();
  out->f_rest           = s.f_rest.empty() ? nullptr : s.f_rest.data();
  out->opacity          = s.opacity.data();
  out->scale            = s.scale.data();
  out->rotation         = s.rotation.data();
  out->splat_count      = s.size();
  out->f_rest_per_splat = s.fRestPerSplat();
  out->sh_degree        = s.maxShDegree();
  return MGS_OK;
}

void mgs_splatset_destroy(MgsSplatSet set) { delete set; }

// ---- asynchronous loader + request queue (PlyLoaderAsync + sceneLoadQueue) -----------------------------------------
}  // extern "C"
#include <deque>
struct MgsLoader_t
{
  std::thread             worker;
  std::mutex              mtx;
  std::condition_variable cv;
  std::deque<std::string> queue;     // waiting requests; front() is the head while LOADING / LOADED / FAILURE
  int                     state = MGS_LOADER_READY;
  bool                    shutdown = false;
  MgsSplatSet             result = nullptr;
  int                     resultCode = MGS_OK;
  std::string             resultError;

  void run()
  {
    std::unique_lock<std::mutex> lk(mtx);
    for(;;)
    {
      cv.wait(lk, [&] { return shutdown || (state == MGS_LOADER_READY && !queue.empty()); });
      if(shutdown)
        return;
      state                  = MGS_LOADER_LOADING;
      const std::string path = queue.front();
      lk.unlock();
      MgsSplatSet set = nullptr;
      const int   rc  = mgs_splatset_load(path.c_str(), &set);  // the synchronous loader (thread-local error string)
      const std::string err = rc == MGS_OK ? std::string() : std::string(mgs_last_error());
      lk.lock();
      result      = set;
      resultCode  = rc;
      resultError = err;
      state       = rc == MGS_OK ? MGS_LOADER_LOADED : MGS_LOADER_FAILURE;
      cv.notify_all();
    }
  }
};
extern "C" {

int mgs_loader_create(MgsLoader* out)
{
  if(!out)
  {
    setError("mgs_loader_create: null argument");
    return MGS_ERR_INVALID_ARG;
  }
  return guarded("mgs_loader_create", [&] {
    auto* L   = new MgsLoader_t();
    L->worker = std::thread([L] { L->run(); });
    *out      = L;
    return (int)MGS_OK;
  });
}

void mgs_loader_destroy(MgsLoader L)
{
  if(!L)
    return;
  {
    std::lock_guard<std::mutex> lk(L->mtx);
    L->shutdown = true;
  }
  L->cv.notify_all();
  L->worker.join();
  if(L->result)
    mgs_splatset_destroy(L->result);
  delete L;
}

int mgs_loader_push(MgsLoader L, const char* path)
{
  if(!L || !path)
  {
    setError("mgs_loader_push: null argument");
    return MGS_ERR_INVALID_ARG;
  }
  return guarded("mgs_loader_push", [&] {
    {
      std::lock_guard<std::mutex> lk(L->mtx);
      L->queue.emplace_back(path);
    }
    L->cv.notify_all();
    return (int)MGS_OK;
  });
}

int mgs_loader_status(MgsLoader L, int* state, uint32_t* queued, char* pathOut, size_t cap)
{
  if(!L || !state)
  {
    setError("mgs_loader_status: null argument");
    return MGS_ERR_INVALID_ARG;
  }
  std::lock_guard<std::mutex> lk(L->mtx);
  // a request that has been queued but not picked up yet already counts as LOADING for the poller
  const bool busy = !L->queue.empty();
  *state          = (L->state == MGS_LOADER_READY && busy) ? MGS_LOADER_LOADING : L->state;
  if(queued)
    *queued = busy ? (uint32_t)L->queue.size() - 1u : 0u;
  if(pathOut && cap)
  {
    const std::string& p = busy ? L->queue.front() : std::string();
    std::snprintf(pathOut, cap, "%s", busy ? p.c_str() : "");
  }
  return MGS_OK;
}

int mgs_loader_take(MgsLoader L, MgsSplatSet* out)
{
  if(!L || !out)
  {
    setError("mgs_loader_take: null argument");
    return MGS_ERR_INVALID_ARG;
  }
  int rc;
  {
    std::lock_guard<std::mutex> lk(L->mtx);
    if(L->state != MGS_LOADER_LOADED && L->state != MGS_LOADER_FAILURE)
    {
      setError("mgs_loader_take: nothing loaded (poll mgs_loader_status)");
      return MGS_ERR_STATE;
    }
    rc = L->resultCode;
    if(L->state == MGS_LOADER_LOADED)
      *out = L->result;
    else
      setError(L->resultError);
    L->result = nullptr;
    L->queue.pop_front();
    L->state = MGS_LOADER_READY;  // reset(): the next queued file may start
  }
  L->cv.notify_all();
  return rc;
}

// ------------------------------------------------------------------------------------------------
static int mgs_scene_create_impl(int device, MgsScene* out);
int mgs_scene_create(int device, MgsScene* out)
{
  return guarded("mgs_scene_create", [&] { return mgs_scene_create_impl(device, out); });
}
static int mgs_scene_create_impl(int device, MgsScene* out)
{
  if(!out)
  {
    setError("mgs_scene_create: null argument");
    return MGS_ERR_INVALID_ARG;
  }
  int count = 0;
  if(hipGetDeviceCount(&count) != hipSuccess || count == 0)
  {
    setError("mgs_scene_create: no HIP device available — the MI355X path has no CPU fallback");
    return MGS_ERR_DEVICE;
  }
  if(device < 0 || device >= count)
  {
    setError("mgs_scene_create: device ordinal out of range");
    return MGS_ERR_INVALID_ARG;
  }
  HIPCHK(hipSetDevice(device));
  auto* s   = new MgsScene_t();
  s->device = device;
  s->d      = std::make_shared<SceneData>();
  s->d->device = device;
  s->d->handles.push_back(s);
  if(hipStreamCreateWithFlags(&s->ownStream, hipStreamNonBlocking) != hipSuccess)
  {
    s->ownStream = nullptr;
    delete s;
    setError("mgs_scene_create: hipStreamCreate failed");
    return MGS_ERR_DEVICE;
  }
  s->stream = s->ownStream;
  if(hipHostMalloc((void**)&s->hCtr, sizeof(FrameCounters)) != hipSuccess
     || hipHostMalloc((void**)&s->hPlans, sizeof(FramePlans)) != hipSuccess)
  {
    if(s->hCtr) (void)hipHostFree(s->hCtr);
    (void)hipStreamDestroy(s->ownStream);
    delete s;
    setError("mgs_scene_create: pinned allocation failed");
    return MGS_ERR_OOM;
  }
  std::memset(s->hCtr, 0, sizeof(FrameCounters));
  std::memset(s->hPlans, 0, sizeof(FramePlans));
  bool evOk = true;
  for(auto& e : s->ev)
    evOk = evOk && hipEventCreate(&e) == hipSuccess;
  for(auto& set : s->evRing)
    for(auto& e : set)
      evOk = evOk && hipEventCreate(&e) == hipSuccess;
  if(!evOk)
  {
    for(auto& e : s->ev)
      if(e) (void)hipEventDestroy(e);
    for(auto& set : s->evRing)
      for(auto& e : set)
        if(e) (void)hipEventDestroy(e);
    mgs_scene_destroy(s);  // evReady is still false: the events are not touched again
    setError("mgs_scene_create: hipEventCreate failed");
    return MGS_ERR_DEVICE;
  }
  s->evReady = true;
  *out       = s;
  return MGS_OK;
}

static void freeSet(DeviceSet& d)
{
  if(d.centers) (void)hipFree(d.centers);
  if(d.cov6) (void)hipFree(d.cov6);
  if(d.rgba) (void)hipFree(d.rgba);
  if(d.sh) (void)hipFree(d.sh);
  if(d.partBox) (void)hipFree(d.partBox);
  if(d.maxScale) (void)hipFree(d.maxScale);
  if(d.alpha) (void)hipFree(d.alpha);
  if(d.rgbaF32 && (void*)d.rgbaF32 != d.rgba) (void)hipFree(d.rgbaF32);
  d.rgbaF32 = nullptr;
  if(d.scales) (void)hipFree(d.scales);
  if(d.rotations) (void)hipFree(d.rotations);
  d.centers = d.cov6 = d.partBox = d.maxScale = d.scales = d.rotations = d.alpha = nullptr;
  d.rgba = d.sh = nullptr;
  d.shFormat = d.rgbaFormat = -1;  // a freed (or half-built) set never matches a requested format: commit rebuilds it
}

SceneData::~SceneData()
{  // the last handle (scene or context) is gone: every stream that read these buffers was synchronised by its handle's destroy
  (void)hipSetDevice(device);
  for(auto& d : sets)
    freeSet(d);
  compInst.release();
}

void mgs_scene_destroy(MgsScene s)
{
  if(!s)
    return;
  s->cpu.shutdown();
  if(s->comm)
    (void)mgs_scene_comm_destroy(s);
  (void)hipSetDevice(s->device);
  (void)hipStreamSynchronize(s->stream);  // like vkDeviceWaitIdle before destruction (gaussian_splatting.cpp:1096)
  if(s->d)
  {
    std::lock_guard<std::mutex> lk(s->d->mtx);
    auto& h = s->d->handles;
    h.erase(std::remove(h.begin(), h.end(), s), h.end());
  }
  s->pairA.release(); s->prjOrder.release(); s->pairB.release(); s->slotCount.release(); s->chunkSum.release(); s->runTab.release(); s->sortedCode16.release(); s->binCost.release(); s->slotHist2.release(); s->top16Rec.release(); s->top16Count.release();
  s->osStatus.release(); s->keysA.release(); s->idsA.release(); s->rect.release(); s->partHist.release(); s->blockCount.release();
  s->rec.release(); s->recGut.release(); s->pairKey0.release(); s->pairVal0.release(); s->pairKey1.release(); s->pairVal1.release();
  s->sortedRect.release(); s->splatOffset.release(); s->chunkStart.release();
  s->surfDepth.release(); s->surfId.release(); s->surfNormal.release(); s->accum.release(); s->fstate.release(); s->dbinMasks.release();
  for(auto& g : s->graphs) (void)hipGraphExecDestroy(g.second);
  s->graphs.clear();
  s->ranges.release(); s->image.release(); s->cpuDistDev.release();
  s->rsKeys.release(); s->rsVals.release(); s->rsHist.release(); s->rsCount.release(); s->rsPlan.release();
  s->rsPairA.release(); s->rsPairB.release(); s->rsStatus.release(); s->rsOsPlan.release();
  if(s->hCtr) (void)hipHostFree(s->hCtr);
  if(s->hPlans) (void)hipHostFree(s->hPlans);
  if(s->binPolicy.host) (void)hipHostFree(s->binPolicy.host);
  if(s->binPolicy.ev) (void)hipEventDestroy(s->binPolicy.ev);
  for(int k = 0; k < MgsScene_t::kUpRing; ++k)
    if(s->upBuf[k])
    {
      (void)hipHostFree(s->upBuf[k]);
      (void)hipEventDestroy(s->upEv[k]);
    }
  if(s->evReady)
  {
    for(auto& e : s->ev)
      (void)hipEventDestroy(e);
    for(auto& set : s->evRing)
      for(auto& e : set)
        (void)hipEventDestroy(e);
  }
  if(s->ownStream)
    (void)hipStreamDestroy(s->ownStream);
  delete s;
}

int mgs_scene_set_stream(MgsScene s, void* stream)
{
  if(!s)
  {
    setError("mgs_scene_set_stream: null scene");
    return MGS_ERR_INVALID_ARG;
  }
  s->stream = stream ? (hipStream_t)stream : s->ownStream;
  return MGS_OK;
}

// ---- frame contexts -----------------------------------------------------------------------------------------------------
// The reference keeps ONE copy of the splat buffers however many frames its application loop has in flight
// (src/gaussian_splatting.cpp:1092-1111: updates wait for the device before they touch the buffers all frames share).  A frame
// context is that: its own HIP stream, working buffers, counters and captured frame graphs over the scene's committed data.
static int mgs_frame_context_create_impl(MgsScene parent, MgsScene* out)
{
  if(!parent || !out)
  {
    setError("mgs_frame_context_create: null argument");
    return MGS_ERR_INVALID_ARG;
  }
  MgsScene c  = nullptr;
  const int rc = mgs_scene_create(parent->device, &c);
  if(rc != MGS_OK)
    return rc;
  {
    std::lock_guard<std::mutex> lk(parent->d->mtx);
    c->d         = parent->d;  // drops the context's own (empty) SceneData
    c->isContext = true;
    c->d->handles.push_back(c);
  }
  *out = c;
  return MGS_OK;
}
int mgs_frame_context_create(MgsScene parent, MgsScene* out)
{
  return guarded("mgs_frame_context_create", [&] { return mgs_frame_context_create_impl(parent, out); });
}
void mgs_frame_context_destroy(MgsScene ctx) { mgs_scene_destroy(ctx); }

int mgs_scene_set_list_capacity(MgsScene s, uint64_t entries)
{
  if(!s)
  {
    setError("mgs_scene_set_list_capacity: null scene");
    return MGS_ERR_INVALID_ARG;
  }
  s->listCapacityOverride = entries;
  s->wsEpoch              = ~0ull;  // the working set is re-sized before the next frame (or by the next commit)
  return MGS_OK;
}

static uint64_t setBytes(const DeviceSet& d)
{
  if(!d.centers)
    return 0;
  const uint64_t n   = d.count;
  const uint64_t fmt[3] = {4, 2, 1};
  uint64_t       b   = n * (12 + 24 + 4 + 4 + 12 + 16) + ((n + kPart - 1) / kPart) * 32;  // centres, cov6, alpha, maxScale, scales, rotations, partBox
  b += n * 4 * fmt[d.rgbaFormat] + (d.rgbaFormat != MGS_FORMAT_FLOAT32 ? n * 16 : 0);
  b += d.sh ? n * (uint64_t)d.shPitch * fmt[d.shFormat] : 0;
  return b;
}
int mgs_scene_memory_usage(MgsScene s, uint64_t* sceneBytes, uint64_t* workingBytes)
{
  if(!s)
  {
    setError("mgs_scene_memory_usage: null scene");
    return MGS_ERR_INVALID_ARG;
  }
  if(sceneBytes)
  {
    uint64_t b = s->d->compInst.n * sizeof(CompositeArgs::Inst);
    for(const auto& d : s->d->sets)
      b += setBytes(d);
    *sceneBytes = b;
  }
  if(workingBytes)
  {
    uint64_t b = 0;
    auto add = [&](auto& buf) { b += (uint64_t)buf.n * sizeof(*buf.p); };
    add(s->pairA); add(s->pairB); add(s->prjOrder); add(s->slotCount); add(s->chunkSum); add(s->runTab); add(s->sortedCode16); add(s->binCost); add(s->slotHist2); add(s->top16Rec); add(s->top16Count); add(s->osStatus); add(s->keysA); add(s->idsA); add(s->rect);
    add(s->partHist); add(s->blockCount); add(s->sortedRect); add(s->splatOffset); add(s->chunkStart);
    add(s->dbinMasks); add(s->fstate); add(s->surfDepth); add(s->surfId); add(s->surfNormal); add(s->accum); add(s->rec); add(s->recGut);
    add(s->pairKey0); add(s->pairVal0); add(s->pairKey1); add(s->pairVal1); add(s->ranges); add(s->image);
    add(s->rsKeys); add(s->rsVals); add(s->rsHist); add(s->rsCount); add(s->rsPairA); add(s->rsPairB); add(s->rsStatus); add(s->rsOsPlan); add(s->rsPlan); add(s->cpuDistDev);
    *workingBytes = b;
  }
  return MGS_OK;
}

static int mgs_instance_add_impl(MgsScene s, MgsSplatSet set, const float m[16], int* id);
int mgs_instance_add(MgsScene s, MgsSplatSet set, const float m[16], int* id)
{
  return guarded("mgs_instance_add", [&] { return mgs_instance_add_impl(s, set, m, id); });
}
static int mgs_instance_add_impl(MgsScene s, MgsSplatSet set, const float m[16], int* id)
{
  if(!s || !set || !m)
  {
    setError("mgs_instance_add: null argument");
    return MGS_ERR_INVALID_ARG;
  }
  if(s->isContext)
  {
    setError("mgs_instance_add: a frame context is read-only; edit the scene it was created from");
    return MGS_ERR_STATE;
  }
  if((int)s->d->instances.size() >= kMaxInstances)
  {
    setError("mgs_instance_add: at most " + std::to_string(kMaxInstances) + " instances per scene in this build");
    return MGS_ERR_UNSUPPORTED;
  }
  int idx = -1;
  for(size_t i = 0; i < s->d->sets.size(); ++i)
    if(s->d->sets[i].host == set->data)
      idx = (int)i;
  if(idx < 0)
  {
    DeviceSet d;
    d.host = set->data;
    s->d->sets.push_back(d);
    idx = (int)s->d->sets.size() - 1;
  }
  uint64_t total = 0;
  for(const auto& I : s->d->instances)
    total += s->d->sets[I.set].host->size();
  total += set->data->size();
  if(total >= kOsMaxPairs)
  {  // the reference's ids are u32; this build's key sort carries 30-bit digit prefixes in its look-back words (k_osort.hip)
    // (ADVICE r4 asked for the generic sort as a fall-back here instead.  Not built: a scene that large cannot be resident — 2^30
    //  splats are >= 107 GB of splat data in the smallest storage formats plus >= 200 B per splat of working set per frame in
    //  flight, beyond the 288 GB of one MI355X — so the limit that binds first is memory, and this one is never the reason.)
    setError("mgs_instance_add: 2^30 or more global splats (the key sort's look-back words hold 30-bit prefixes; such a scene would not fit 288 GB either)");
    return MGS_ERR_UNSUPPORTED;
  }
  Instance I;
  I.set = idx;
  std::memcpy(I.M, m, sizeof(I.M));
  s->d->instances.push_back(I);
  s->d->committed = false;
  if(id)
    *id = (int)s->d->instances.size() - 1;
  return MGS_OK;
}

int mgs_instance_set_transform(MgsScene s, int id, const float m[16])
{
  if(!s || !m || id < 0 || id >= (int)s->d->instances.size())
  {
    setError("mgs_instance_set_transform: bad argument");
    return MGS_ERR_INVALID_ARG;
  }
  if(s->isContext)
  {
    setError("mgs_instance_set_transform: a frame context is read-only; edit the scene it was created from");
    return MGS_ERR_STATE;
  }
  std::memcpy(s->d->instances[id].M, m, sizeof(float) * 16);
  return MGS_OK;  // transforms travel with every frame (of every context); no re-commit needed
}

uint64_t mgs_scene_splat_count(MgsScene s)
{
  if(!s)
    return 0;
  uint64_t total = 0;
  for(const auto& I : s->d->instances)
    total += s->d->sets[I.set].host->size();
  return total;
}

static int uploadFormatted(const std::vector<float>& src, int format, bool isSh, void** dev)
{
  const size_t n = src.size();
  if(n == 0)
  {
    *dev = nullptr;
    return MGS_OK;
  }
  const size_t esz = format == MGS_FORMAT_FLOAT32 ? 4 : format == MGS_FORMAT_FLOAT16 ? 2 : 1;
  if(hipMalloc(dev, n * esz) != hipSuccess)
  {
    setError("commit: device allocation failed");
    return MGS_ERR_OOM;
  }
  if(format == MGS_FORMAT_FLOAT32)
  {
    HIPCHK(hipMemcpy(*dev, src.data(), n * 4, hipMemcpyHostToDevice));
  }
  else if(format == MGS_FORMAT_FLOAT16)
  {
    std::vector<uint16_t> tmp(n);
    parallelBatches(n, [&](size_t i) { tmp[i] = floatToHalf(src[i]); });
    HIPCHK(hipMemcpy(*dev, tmp.data(), n * 2, hipMemcpyHostToDevice));
  }
  else
  {
    std::vector<uint8_t> tmp(n);
    const float          lo = isSh ? -1.f : 0.f;
    parallelBatches(n, [&](size_t i) { tmp[i] = toUint8(src[i], lo, 1.f); });
    HIPCHK(hipMemcpy(*dev, tmp.data(), n, hipMemcpyHostToDevice));
  }
  return MGS_OK;
}

// the handle's FrameState block (allocated once; its address is baked into captured frame graphs)
static int ensureFrameState(MgsScene s)
{
  if(s->fstate.p)
    return MGS_OK;
  int rc = s->fstate.ensure(1);
  if(rc)
    return rc;
  HIPCHK(hipMemset(s->fstate.p, 0, sizeof(FrameState)));
  HIPCHK(hipDeviceSynchronize());
  s->ctr.p   = &s->fstate.p->ctr;
  s->plans.p = &s->fstate.p->plans;
  s->dArgs.p = &s->fstate.p->args;
  return MGS_OK;
}

// A frame's first operation on its stream: counters and plans to zero, this frame's constants behind them, in one upload
// from the handle's ring of pinned buffers (a buffer is reused four uploads later, after its copy's event).
static int uploadFrameState(MgsScene s, const FrameArgs& A, hipStream_t st)
{
  int rc = ensureFrameState(s);
  if(rc)
    return rc;
  const size_t head = offsetof(FrameState, args), used = offsetof(FrameArgs, inst) + (size_t)A.f.nInstances * sizeof(InstanceConst);
  const int    k    = (int)(s->upNext++ % MgsScene_t::kUpRing);
  if(!s->upBuf[k])
  {
    if(hipHostMalloc((void**)&s->upBuf[k], head + sizeof(FrameArgs)) != hipSuccess || hipEventCreateWithFlags(&s->upEv[k], hipEventDisableTiming) != hipSuccess)
    {
      if(s->upBuf[k]) (void)hipHostFree(s->upBuf[k]);
      s->upBuf[k] = nullptr;
      setError("frame: pinned allocation for the per-frame upload failed");
      return MGS_ERR_OOM;
    }
    std::memset(s->upBuf[k], 0, head);  // the head stays zero for good
  }
  if(s->upBusy[k])
    HIPCHK(hipEventSynchronize(s->upEv[k]));
  std::memcpy(s->upBuf[k] + head, &A, used);
  HIPCHK(hipMemcpyAsync(s->fstate.p, s->upBuf[k], head + used, hipMemcpyHostToDevice, st));
  HIPCHK(hipEventRecord(s->upEv[k], st));
  s->upBusy[k] = true;
  return MGS_OK;
}

static int sizeWorkingSet(MgsScene s);
// a frame context that has not seen the scene's latest commit re-sizes its working set before it renders
static int ensureWorkingSet(MgsScene s)
{
  return s->wsEpoch == s->d->epoch ? MGS_OK : sizeWorkingSet(s);
}
static int mgs_scene_commit_impl(MgsScene s, int shFormat, int rgbaFormat);
int mgs_scene_commit(MgsScene s, int shFormat, int rgbaFormat)
{
  return guarded("mgs_scene_commit", [&] { return mgs_scene_commit_impl(s, shFormat, rgbaFormat); });
}
static int mgs_scene_commit_impl(MgsScene s, int shFormat, int rgbaFormat)
{
  if(!s)
  {
    setError("mgs_scene_commit: null scene");
    return MGS_ERR_INVALID_ARG;
  }
  if(shFormat < 0 || shFormat > 2 || rgbaFormat < 0 || rgbaFormat > 2)
  {
    setError("mgs_scene_commit: formats must be MGS_FORMAT_FLOAT32/FLOAT16/UINT8");
    return MGS_ERR_INVALID_ARG;
  }
  if(s->isContext)
  {
    setError("mgs_scene_commit: a frame context shares its scene's data; commit the scene it was created from");
    return MGS_ERR_STATE;
  }
  if(s->d->instances.empty())
  {
    setError("mgs_scene_commit: scene has no instances");
    return MGS_ERR_STATE;
  }
  HIPCHK(hipSetDevice(s->device));
  {  // every frame in flight on any context reads the buffers this call may free: wait for all of them, like the reference's
     // vkDeviceWaitIdle before it touches the splat buffers (gaussian_splatting.cpp:1092-1111)
    std::lock_guard<std::mutex> lk(s->d->mtx);
    for(MgsScene_t* h : s->d->handles)
      HIPCHK(hipStreamSynchronize(h->stream));
  }
  for(auto& g : s->graphs)  // captured frames hold the old buffers and grid sizes (contexts drop theirs when they see the new epoch)
    (void)hipGraphExecDestroy(g.second);
  s->graphs.clear();
  for(auto& d : s->d->sets)
  {
    if(d.centers && d.shFormat == shFormat && d.rgbaFormat == rgbaFormat)
      continue;  // idempotent
    freeSet(d);
    const HostSplatSet& h = *d.host;
    const size_t        n = h.size();
    d.count               = (uint32_t)n;
    d.shDegree            = std::max(0, h.maxShDegree());
    d.shStride            = shStride(h.fRestPerSplat());
    // storage order: Morton order of the centres (MGS_REORDER=0 keeps the caller's order).  Global ids
    // used inside the pipeline are STORAGE ids; the API maps them back (mgs_sort_download,
    // mgs_scene_download_set, mgs_scene_storage_order).  Ties between equal depth keys resolve in
    // storage order — the reference's tie order is nondeterministic (dist.comp.slang:137-139).
    static const bool kReorder = [] { const char* e = std::getenv("MGS_REORDER"); return e ? std::atoi(e) != 0 : true; }();
    std::vector<float> cov;
    buildCov6(h, cov);
    if(kReorder)
    {
      std::vector<float> radius(n);
      parallelBatches(n, [&](size_t i) { radius[i] = std::sqrt(8.0f * (cov[6 * i] + cov[6 * i + 3] + cov[6 * i + 5])); });
      mortonOrder(h, &radius, d.newToOld);
    }
    else
    {
      d.newToOld.resize(n);
      for(size_t i = 0; i < n; ++i)
        d.newToOld[i] = (uint32_t)i;
    }
    d.oldToNew.resize(n);
    parallelBatches(n, [&](size_t i) { d.oldToNew[d.newToOld[i]] = (uint32_t)i; });
    const uint32_t* perm = d.newToOld.data();
    // a4: SplatSetVk::initDataBuffers (src/splat_set_vk.cpp:188-480), host loops like the reference
    {
      std::vector<float> pos(n * 3);
      parallelBatches(n, [&](size_t i) { std::memcpy(&pos[3 * i], &h.positions[3 * (size_t)perm[i]], 12); });
      HIPCHK(hipMalloc((void**)&d.centers, n * 3 * sizeof(float)));
      HIPCHK(hipMemcpy(d.centers, pos.data(), n * 3 * sizeof(float), hipMemcpyHostToDevice));
      // per-partition bounds (model space): AABB of the centres and rmax = sqrt(8 * max trace(Sigma3D)),
      // an upper bound of the sqrt(8)-sigma footprint radius of any splat of the partition
      const size_t       np = (n + kPart - 1) / kPart;
      std::vector<float> box(np * 8);
      parallelBatches(np, [&](size_t pIdx) {
        float mn[3] = {3.4e38f, 3.4e38f, 3.4e38f}, mx[3] = {-3.4e38f, -3.4e38f, -3.4e38f}, tr = 0.f;
        bool  bad = false;
        for(size_t i = pIdx * kPart; i < std::min(n, (pIdx + 1) * kPart); ++i)
        {
          for(int a = 0; a < 3; ++a)
          {
            const float v = pos[3 * i + a];
            if(!std::isfinite(v)) bad = true;
            mn[a] = std::min(mn[a], v);
            mx[a] = std::max(mx[a], v);
          }
          const float* c = &cov[6 * (size_t)perm[i]];
          const float  t = c[0] + c[3] + c[5];
          if(!std::isfinite(t)) bad = true;
          tr = std::max(tr, t);
        }
        float* b = &box[8 * pIdx];
        for(int a = 0; a < 3; ++a)
        {
          b[a]     = mn[a];
          b[3 + a] = mx[a];
        }
        b[6] = std::sqrt(8.0f * tr);
        b[7] = bad ? 1.0f : 0.0f;  // non-finite data: never cull this partition
      });
      HIPCHK(hipMalloc((void**)&d.partBox, np * 8 * sizeof(float)));
      HIPCHK(hipMemcpy(d.partBox, box.data(), np * 8 * sizeof(float), hipMemcpyHostToDevice));
    }
    {  // size culling (dist.comp.slang:93-134) needs max(exp(scale)): 4 B/splat, exp() on the host like the oracle
      std::vector<float> ms(n);
      parallelBatches(n, [&](size_t i) {
        const float* sc = &h.scale[3 * (size_t)perm[i]];
        ms[i]           = std::max(std::exp(sc[0]), std::max(std::exp(sc[1]), std::exp(sc[2])));
      });
      HIPCHK(hipMalloc((void**)&d.maxScale, n * sizeof(float)));
      HIPCHK(hipMemcpy(d.maxScale, ms.data(), n * sizeof(float), hipMemcpyHostToDevice));
    }
    {  // raw scales / rotations (splat_set_vk.cpp:221-251): 28 B/splat, read by the surface normal only
      std::vector<float> sc(n * 3), rq(n * 4);
      parallelBatches(n, [&](size_t i) {
        std::memcpy(&sc[3 * i], &h.scale[3 * (size_t)perm[i]], 12);
        std::memcpy(&rq[4 * i], &h.rotation[4 * (size_t)perm[i]], 16);
      });
      HIPCHK(hipMalloc((void**)&d.scales, n * 3 * sizeof(float)));
      HIPCHK(hipMemcpy(d.scales, sc.data(), n * 3 * sizeof(float), hipMemcpyHostToDevice));
      HIPCHK(hipMalloc((void**)&d.rotations, n * 4 * sizeof(float)));
      HIPCHK(hipMemcpy(d.rotations, rq.data(), n * 4 * sizeof(float), hipMemcpyHostToDevice));
    }
    {
      std::vector<float> planar(n * 6);
      // HBM layout: float4 covA[n] = (S00,S01,S02,S11) followed by float2 covB[n] = (S12,S22)
      parallelBatches(n, [&](size_t i) {
        std::memcpy(&planar[4 * i], &cov[6 * (size_t)perm[i]], 16);
        std::memcpy(&planar[4 * n + 2 * i], &cov[6 * (size_t)perm[i] + 4], 8);
      });
      HIPCHK(hipMalloc((void**)&d.cov6, n * 6 * sizeof(float)));
      HIPCHK(hipMemcpy(d.cov6, planar.data(), n * 6 * sizeof(float), hipMemcpyHostToDevice));
    }
    {
      std::vector<float> rgba, re(n * 4);
      buildRgba(h, rgba);
      parallelBatches(n, [&](size_t i) { std::memcpy(&re[4 * i], &rgba[4 * (size_t)perm[i]], 16); });
      int rc = uploadFormatted(re, rgbaFormat, false, &d.rgba);
      if(rc != MGS_OK)
        return rc;
      // the colours exactly as fetchColor reads them back from the formatted buffer (threedgs_particle_buffers.h.slang:
      // 72-90): fp32 for the compositor's shading phase, and the opacity as a plane of its own — all the projection needs
      std::vector<float> al(n);
      if(rgbaFormat != MGS_FORMAT_FLOAT32)
        parallelBatches(n * 4, [&](size_t i) {
          re[i] = rgbaFormat == MGS_FORMAT_FLOAT16 ? halfToFloat(floatToHalf(re[i])) : (float)toUint8(re[i], 0.f, 1.f) / 255.0f;
        });
      parallelBatches(n, [&](size_t i) { al[i] = re[4 * i + 3]; });
      if(rgbaFormat == MGS_FORMAT_FLOAT32)
        d.rgbaF32 = static_cast<float*>(d.rgba);
      else
      {
        HIPCHK(hipMalloc((void**)&d.rgbaF32, n * 4 * sizeof(float)));
        HIPCHK(hipMemcpy(d.rgbaF32, re.data(), n * 4 * sizeof(float), hipMemcpyHostToDevice));
      }
      HIPCHK(hipMalloc((void**)&d.alpha, n * sizeof(float)));
      HIPCHK(hipMemcpy(d.alpha, al.data(), n * sizeof(float), hipMemcpyHostToDevice));
    }
    if(d.shStride)
    {
      std::vector<float> sh, padded;
      buildShInterleaved(h, sh);
      // one 48-element record per (storage) splat, [coef][rgb] like the reference, zero padded: the compositor
      // fetches whole records of the splats it stages (192 B fp32 = three 64-byte sectors)
      d.shPitch = 48;
      padded.assign(n * (size_t)d.shPitch, 0.f);
      const int stride = d.shStride;
      parallelBatches(n, [&](size_t i) {
        std::memcpy(&padded[i * (size_t)d.shPitch], &sh[(size_t)perm[i] * (size_t)stride], sizeof(float) * stride);
      });
      int rc = uploadFormatted(padded, shFormat, true, &d.sh);
      if(rc != MGS_OK)
        return rc;
    }
    d.shFormat   = shFormat;
    d.rgbaFormat = rgbaFormat;
  }
  s->d->shFormat   = shFormat;
  s->d->rgbaFormat = rgbaFormat;

  // global id space + partitions (a5: rebuildGlobalIndexTables, splat_set_manager_vk.cpp:2304-2360,
  // here a closed-form prefix instead of an 8-byte-per-splat table)
  uint64_t total = 0, parts = 0;
  for(const auto& I : s->d->instances)
  {
    const uint32_t c = s->d->sets[I.set].count;
    total += c;
    parts += (c + kPart - 1) / kPart;
  }
  s->d->totalSplats = (uint32_t)total;
  s->d->totalParts  = (uint32_t)parts;
  int rc = MGS_OK;
  {  // the compositor's SH table of all instances (it carries the first 16 by value)
    std::vector<CompositeArgs::Inst> tab(s->d->instances.size());
    uint32_t                         off = 0;
    for(size_t k = 0; k < s->d->instances.size(); ++k)
    {
      const DeviceSet& d  = s->d->sets[s->d->instances[k].set];
      tab[k].sh           = d.sh;
      tab[k].rgba         = reinterpret_cast<const float4*>(d.rgbaF32);
      tab[k].centers      = d.centers;
      tab[k].globalOffset = off;
      tab[k].shDegree     = d.shDegree;
      off += d.count;
    }
    if((rc = s->d->compInst.ensure(tab.size()))) return rc;
    HIPCHK(hipMemcpy(s->d->compInst.p, tab.data(), tab.size() * sizeof(tab[0]), hipMemcpyHostToDevice));
  }
  s->d->committed = true;
  s->d->epoch += 1;
  return sizeWorkingSet(s);
}

// The per-handle working set (slots, sort ping-pong, records, lists, counters): sized for the committed scene.  A scene sizes
// its own at commit; a frame context when it first renders after a commit (wsEpoch lags d->epoch).
static int sizeWorkingSet(MgsScene s)
{
  HIPCHK(hipSetDevice(s->device));
  HIPCHK(hipStreamSynchronize(s->stream));
  for(auto& g : s->graphs)
    (void)hipGraphExecDestroy(g.second);
  s->graphs.clear();
  const uint64_t total = s->d->totalSplats, parts = s->d->totalParts;
  int rc = MGS_OK;
  if((rc = s->slotHist2.ensure((size_t)parts * kSlotHistWords))) return rc;
  if((rc = s->top16Rec.ensure((size_t)parts * 128u))) return rc;
  // occurrences of key >> 16: the sort's prepare kernel consumes and clears it every frame; zeroed here as well, so that a
  // frame that died between the two kernels cannot leak counts into the next scene
  if((rc = s->top16Count.ensure(65536u))) return rc;
  HIPCHK(hipMemset(s->top16Count.p, 0, 65536u * 4u));
  {  // look-back words of the key sort's passes: zero once, the passes clear them for their successors (k_osort.hip)
    // (the three sets' offsets depend on the element bound: a re-sized working set starts from zeroed words)
    const size_t words = 3u * osSortStatusWords(osSortMaxParts((uint32_t)total));
    if((rc = s->osStatus.ensure(words))) return rc;
    HIPCHK(hipMemset(s->osStatus.p, 0, words * 4u));
  }
  if((rc = s->pairA.ensure(total))) return rc;
  if((rc = s->pairB.ensure(std::max<uint64_t>(total, parts * (uint64_t)kOsSlot)))) return rc;  // whole slots
  if((rc = s->slotCount.ensure(parts))) return rc;
  {  // the partitions' dispatch order starts as the identity; every key sort leaves the next frame's (always a permutation)
    if((rc = s->prjOrder.ensure(std::max<uint64_t>(parts, 1)))) return rc;
    std::vector<uint32_t> iota(parts);
    for(uint32_t i = 0; i < (uint32_t)parts; ++i)
      iota[i] = i;
    if(parts)
      HIPCHK(hipMemcpy(s->prjOrder.p, iota.data(), (size_t)parts * 4, hipMemcpyHostToDevice));
  }
  if((rc = s->sortedCode16.ensure(total))) return rc;
  if((rc = s->binCost.ensure(256))) return rc;
  HIPCHK(hipMemset(s->binCost.p, 0, 256 * 4));
  if((rc = s->chunkSum.ensure((size_t)osSortChunks(parts) * 256u))) return rc;
  if((rc = s->runTab.ensure((size_t)osSortChunks(parts) * kOsChunk * 256u))) return rc;
  if((rc = s->idsA.ensure(total))) return rc;
  if((rc = s->rect.ensure(total))) return rc;
  if((rc = s->rec.ensure(total))) return rc;
  if((rc = s->sortedRect.ensure(total))) return rc;
  if((rc = ensureFrameState(s))) return rc;

  uint64_t cap = std::max<uint64_t>(32ull * total, 64ull << 20);  // 16 B per pair: 3 GB for a garden-sized scene
  if(const char* e = std::getenv("MGS_PAIR_CAPACITY"))
    cap = std::strtoull(e, nullptr, 10);
  if(s->listCapacityOverride)
    cap = s->listCapacityOverride;
  cap = std::min<uint64_t>(std::max<uint64_t>(cap, kPart), 0xFFFFF000ull);
  s->pairCapacity = (uint32_t)cap;
  if((rc = s->pairVal1.ensure(cap))) return rc;  // the per-bin lists
  // bit masks handed from k_dbin_count to k_dbin_emit: <= 64 x 8 B per 64 sorted splats
  if((rc = s->dbinMasks.ensure((size_t)total + 8192))) return rc;  // 64 masks per 64 sorted splats at most, chunk size independent
  // the record path's buffers (16 B per record + per-splat offsets) are allocated when a frame first needs them
  const uint64_t maxParts = std::max<uint64_t>((cap + kPart - 1) / kPart, std::max<uint64_t>(parts, (total + 1023) / 1024 + 1));  // direct binning scans rows of 1024-splat chunks
  s->pStride              = (uint32_t)maxParts;
  if((rc = s->partHist.ensure(256ull * maxParts))) return rc;
  if((rc = s->blockCount.ensure(std::max<uint64_t>((total + kPart - 1) / kPart, 1)))) return rc;
  HIPCHK(hipMemset(s->fstate.p, 0, offsetof(FrameState, args)));
  // hipMemset on device memory is asynchronous on the NULL stream, and the render stream is non-blocking:
  // without this the memsets above can land in the middle of the first frame (caught by the test suite)
  HIPCHK(hipDeviceSynchronize());
  s->wsEpoch = s->d->epoch;
  {  // a changed scene invalidates the CPU sorter's last result (lazy sorting must not reuse it)
    std::lock_guard<std::mutex> lk(s->cpu.mtx);
    s->cpu.haveLast   = false;
    s->cpuHaveIndices = false;
  }
  s->haveFrame = false;
  return MGS_OK;
}

int mgs_scene_storage_order(MgsScene s, int instance, uint32_t* newToOld, size_t count)
{
  if(!s || !newToOld || instance < 0 || instance >= (int)s->d->instances.size())
  {
    setError("mgs_scene_storage_order: bad argument");
    return MGS_ERR_INVALID_ARG;
  }
  if(!s->d->committed)
  {
    setError("mgs_scene_storage_order: scene not committed");
    return MGS_ERR_STATE;
  }
  const DeviceSet& d = s->d->sets[s->d->instances[instance].set];
  if(count < d.count)
  {
    setError("mgs_scene_storage_order: destination too small");
    return MGS_ERR_INVALID_ARG;
  }
  std::memcpy(newToOld, d.newToOld.data(), (size_t)d.count * 4);
  return MGS_OK;
}

int mgs_scene_download_set(MgsScene s, int instance, int which, float* dst, size_t count)
{
  if(!s || !dst || instance < 0 || instance >= (int)s->d->instances.size())
  {
    setError("mgs_scene_download_set: bad argument");
    return MGS_ERR_INVALID_ARG;
  }
  if(!s->d->committed)
  {
    setError("mgs_scene_download_set: scene not committed");
    return MGS_ERR_STATE;
  }
  HIPCHK(hipSetDevice(s->device));
  const DeviceSet& d = s->d->sets[s->d->instances[instance].set];
  const size_t     n = d.count;
  size_t           need;
  const void*      src;
  int              fmt = MGS_FORMAT_FLOAT32;
  bool             isSh = false, isCov = false;
  switch(which)
  {
    case 0: need = 3 * n; src = d.centers; break;
    case 1: need = 6 * n; src = d.cov6; isCov = true; break;
    case 2: need = 4 * n; src = d.rgba; fmt = d.rgbaFormat; break;
    case 3: need = (size_t)d.shStride * n; src = d.sh; fmt = d.shFormat; isSh = true; break;
    default: setError("mgs_scene_download_set: unknown buffer"); return MGS_ERR_INVALID_ARG;
  }
  if(count < need)
  {
    setError("mgs_scene_download_set: destination too small");
    return MGS_ERR_INVALID_ARG;
  }
  if(need == 0)
    return MGS_OK;
  const uint32_t* n2o = d.newToOld.data();  // storage index -> caller's index
  if(isCov)
  {  // planar (float4 covA[n], float2 covB[n]) in storage order -> the reference's 6 floats per splat, caller's order
    std::vector<float> raw(need);
    HIPCHK(hipMemcpy(raw.data(), src, need * 4, hipMemcpyDeviceToHost));
    for(size_t i = 0; i < n; ++i)
    {
      std::memcpy(dst + 6 * (size_t)n2o[i], &raw[4 * i], 16);
      std::memcpy(dst + 6 * (size_t)n2o[i] + 4, &raw[4 * n + 2 * i], 8);
    }
    return MGS_OK;
  }
  if(isSh)
  {  // padded records in storage order: fetch, dequantise, restore [splat][coef][rgb] in the caller's order
    const size_t         tot = (size_t)d.shPitch * n;
    const size_t         esz = fmt == MGS_FORMAT_FLOAT32 ? 4 : fmt == MGS_FORMAT_FLOAT16 ? 2 : 1;
    std::vector<uint8_t> raw(tot * esz);
    HIPCHK(hipMemcpy(raw.data(), src, raw.size(), hipMemcpyDeviceToHost));
    for(size_t i = 0; i < n; ++i)
      for(int k = 0; k < d.shStride; ++k)
      {
        const size_t j = i * (size_t)d.shPitch + (size_t)k;
        float        v;
        if(fmt == MGS_FORMAT_FLOAT32)
          std::memcpy(&v, raw.data() + 4 * j, 4);
        else if(fmt == MGS_FORMAT_FLOAT16)
        {
          uint16_t hv;
          std::memcpy(&hv, raw.data() + 2 * j, 2);
          v = halfToFloat(hv);
        }
        else
          v = (float)raw[j] / 255.0f * 2.0f - 1.0f;
        dst[(size_t)n2o[i] * (size_t)d.shStride + k] = v;
      }
    return MGS_OK;
  }
  {  // centres (3 floats) and rgba (4 elements): fetch in storage order, dequantise, scatter to the caller's order
    const size_t       w = (which == 0) ? 3 : 4;
    std::vector<float> tmp(need);
    if(fmt == MGS_FORMAT_FLOAT32)
      HIPCHK(hipMemcpy(tmp.data(), src, need * 4, hipMemcpyDeviceToHost));
    else if(fmt == MGS_FORMAT_FLOAT16)
    {
      std::vector<uint16_t> raw(need);
      HIPCHK(hipMemcpy(raw.data(), src, need * 2, hipMemcpyDeviceToHost));
      for(size_t i = 0; i < need; ++i)
        tmp[i] = halfToFloat(raw[i]);
    }
    else
    {
      std::vector<uint8_t> raw(need);
      HIPCHK(hipMemcpy(raw.data(), src, need, hipMemcpyDeviceToHost));
      for(size_t i = 0; i < need; ++i)
        tmp[i] = (float)raw[i] / 255.0f;
    }
    for(size_t i = 0; i < n; ++i)
      std::memcpy(dst + w * (size_t)n2o[i], &tmp[w * i], w * sizeof(float));
    return MGS_OK;
  }
}

void mgs_frame_params_default(MgsFrameParams* p)
{
  if(!p)
    return;
  std::memset(p, 0, sizeof(*p));
  const float id[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
  std::memcpy(p->view, id, sizeof(id));
  std::memcpy(p->proj, id, sizeof(id));
  p->width                = 1920;
  p->height               = 1080;
  p->splat_scale          = 1.0f;
  p->frustum_dilation     = 0.2f;
  p->alpha_cull_threshold = 1.0f / 255.0f;
  p->sh_degree            = 3;
  p->sort_mode            = MGS_SORT_GPU_RADIX;
  p->frustum_culling      = MGS_CULL_AT_DIST;
  p->target_format        = MGS_TARGET_RGBA16F;
  p->alpha_mode           = MGS_ALPHA_COVERAGE;
  p->size_culling_min_pixels = 1.0f;
  p->cpu_lazy_sort           = 1;     // parameters.h:183
  p->thin_particle_threshold = 1e-6f; // parameters.h:163
  p->quantize_normals        = 1;     // parameters.h:195
  p->pipeline                = MGS_PIPELINE_3DGS;
  p->camera_model            = MGS_CAMERA_PINHOLE;
  p->extent_method           = MGS_EXTENT_CONIC;  // parameters.h:190 (read by the 3DGUT pipeline only)
  p->fov_rad                 = 0.0f;
  p->alpha_clamp             = 0.99f;    // shaderio.h:271
  p->kernel_min_response     = 0.0113f;  // parameters.h:216
  p->surface_outputs         = 0;
  p->depth_iso_threshold     = 0.7f;  // parameters.h:200
  p->dof_mode                = MGS_DOF_DISABLED;
  p->focus_dist              = 1.3f;    // shaderio.h:278
  p->aperture                = 0.001f;  // shaderio.h:279
  p->frame_sample_id         = 0;
  p->temporal_sampling       = 0;
  p->kernel_degree           = 2;       // parameters.h:215
  p->normal_method           = MGS_NORMAL_MAX_DENSITY_PLANE;  // parameters.h:162
}

// storage global id <-> caller global id.  Instances are concatenated in creation order in both spaces;
// only the index inside a splat set is permuted.
static void mapIdsToCaller(MgsScene s, uint32_t* ids, size_t n)
{
  std::vector<uint32_t> offs;
  uint32_t              o = 0;
  for(const auto& I : s->d->instances)
  {
    offs.push_back(o);
    o += s->d->sets[I.set].count;
  }
  offs.push_back(o);
  parallelBatches(n, [&](size_t i) {
    const uint32_t g = ids[i];
    size_t         k = 0;
    while(k + 1 < s->d->instances.size() && g >= offs[k + 1])
      ++k;
    ids[i] = offs[k] + s->d->sets[s->d->instances[k].set].newToOld[g - offs[k]];
  });
}
static void mapIdsToStorage(MgsScene s, uint32_t* ids, size_t n)
{
  std::vector<uint32_t> offs;
  uint32_t              o = 0;
  for(const auto& I : s->d->instances)
  {
    offs.push_back(o);
    o += s->d->sets[I.set].count;
  }
  offs.push_back(o);
  parallelBatches(n, [&](size_t i) {
    const uint32_t g = ids[i];
    size_t         k = 0;
    while(k + 1 < s->d->instances.size() && g >= offs[k + 1])
      ++k;
    ids[i] = offs[k] + s->d->sets[s->d->instances[k].set].oldToNew[g - offs[k]];
  });
}

// ------------------------------------------------------------------------------------------------
// coarse bins (the default): stable multi-split straight into the per-bin lists, no records, no pair sort; MGS_DIRECT_BIN=0
// forces the record + pair-sort path of frames with more than 256 bins
static const bool kDirectBin = [] { const char* e = std::getenv("MGS_DIRECT_BIN"); return e ? std::atoi(e) != 0 : true; }();

// The bin rectangles ride through the key sort as codes in the id word's spare bits (kernels_common.h: rideEncode) when the
// frame bins directly and the GPU sorts: as many of the shapes 1x1, 2x1, 1x2, 2x2 as fit the spare bits (at most 16).
// MGS_RECT_RIDE=0: k_dbin_count gathers every rectangle by id instead (A/B).
static const bool kRectRide = [] { const char* e = std::getenv("MGS_RECT_RIDE"); return e ? std::atoi(e) != 0 : true; }();
static void chooseRide(MgsScene s, FrameConst& F, bool cpuMode)
{
  F.rideShift = F.rideShapes = F.rideSplit = 0;
  F.rideEscape = 0;
  if(!kRectRide || cpuMode || !kDirectBin || !directBinningSupported(F.binsX, F.binsY) || s->d->totalSplats == 0)
    return;
  int idBits = 1;
  while(idBits < 32 && (1ull << idBits) < (uint64_t)s->d->totalSplats)
    ++idBits;
  const int bx = F.binsX, by = F.binsY, nb = bx * by;
  const int codes[4] = {nb, nb + (bx - 1) * by, nb + (bx - 1) * by + bx * (by - 1), nb + (bx - 1) * by + bx * (by - 1) + (bx - 1) * (by - 1)};
  // Where the id word's spare bits hold all four shapes the code rides there whole.  Otherwise (round 5; 8 garden instances
  // need 26 id bits) it is split: its low 8 bits replace the KEY's low byte in the slot — the slot is grouped by that byte, so
  // the sort never looks at it again (slot_emit.h) — and only the rest rides above the id: 8 + spare bits.  MGS_RIDE_SPLIT=0:
  // as before (fewer shapes, or the gather by id); =2: always split (tests: small scenes take the split path too).
  static const int kSplitMode = [] { const char* e = std::getenv("MGS_RIDE_SPLIT"); return e ? std::atoi(e) : 1; }();
  const bool kSplit = kSplitMode != 0;
  const int  spare  = 32 - idBits;
  for(int pass = kSplitMode == 2 ? 1 : 0; pass < 2; ++pass)
  {
    const bool split    = pass == 1;
    const int  codeBits = std::min(split ? spare + 8 : spare, 16);
    if(split && (!kSplit || spare < 1))
      break;
    for(int shapes = 4; shapes >= (split ? 1 : 4); --shapes)
      if(codeBits >= 1 && codes[shapes - 1] + 1 <= (1 << codeBits))
      {
        F.rideShift  = idBits;
        F.rideShapes = shapes;
        F.rideEscape = (1u << codeBits) - 1u;
        F.rideSplit  = split ? 1 : 0;
        return;
      }
  }
  if(kSplit)
    return;
  const int codeBits = std::min(spare, 16);
  for(int shapes = 3; shapes >= 1; --shapes)
    if(codeBits >= 1 && codes[shapes - 1] + 1 <= (1 << codeBits))
    {
      F.rideShift  = idBits;
      F.rideShapes = shapes;
      F.rideEscape = (1u << codeBits) - 1u;
      return;
    }
}


// the frames the adaptive bin size applies to: the default compositing mode of the 3DGS pipeline with the GPU sort
static const bool kBinAdapt = [] { const char* e = std::getenv("MGS_BIN_ADAPT"); return (e ? std::atoi(e) != 0 : true) && std::getenv("MGS_BIN_SHIFT") == nullptr; }();
static bool binPolicyEligible(const MgsFrameParams* p)
{
  return kBinAdapt && p->alpha_mode != MGS_ALPHA_SUM && p->pipeline == MGS_PIPELINE_3DGS && p->sort_mode == MGS_SORT_GPU_RADIX && p->surface_outputs == 0;
}

static int buildFrameArgs(MgsScene s, const MgsFrameParams* p, FrameArgs& A)
{
  if(p->width <= 0 || p->height <= 0 || p->width > 8192 || p->height > 8192)
  {
    setError("frame: width/height must be in [1,8192] (8-bit bin coordinates, bins of at least 32 pixels)");
    return MGS_ERR_INVALID_ARG;
  }
  std::memset(&A, 0, sizeof(A));
  FrameConst& F = A.f;
  std::memcpy(F.view, p->view, sizeof(F.view));
  std::memcpy(F.proj, p->proj, sizeof(F.proj));
  {  // affine view + perspective projection, bit patterns and magnitudes checked: the project kernels may drop the products with
     // the exact zeros (kernels_common.h: mulMat4ExactAffineW1 / mulPerspExactW1*).  MGS_EXACT_SHORTCUTS=0: never (A/B).
    static const bool kShort = [] { const char* e = std::getenv("MGS_EXACT_SHORTCUTS"); return e ? std::atoi(e) != 0 : true; }();
    const float vrow[4] = {p->view[3], p->view[7], p->view[11], p->view[15]}, want[4] = {0.0f, 0.0f, 0.0f, 1.0f};
    bool        ok      = kShort && std::memcmp(vrow, want, sizeof(want)) == 0;
    static const int kZero[9] = {1, 2, 3, 4, 6, 7, 12, 13, 15};
    const float      zero     = 0.0f;
    for(int q = 0; q < 9 && ok; ++q)
      ok = std::memcmp(&p->proj[kZero[q]], &zero, 4) == 0;  // +0 bitwise
    ok = ok && p->proj[14] != 0.0f;
    for(int q = 0; q < 16 && ok; ++q)
      ok = std::fabs(p->view[q]) < 16777216.0f && std::fabs(p->proj[q]) < 16777216.0f;  // 2^24 (with |coordinate| < 2^40 and |M| < 2^24 no product of the chain P*V*M*p reaches 2^127); false for NaN
    F.perspAffine = ok ? 1 : 0;
  }
  F.focal[0] = p->proj[0] * 0.5f * (float)p->width;   // gaussian_splatting.cpp:1248-1250
  F.focal[1] = p->proj[5] * 0.5f * (float)p->height;
  F.width    = p->width;
  F.height   = p->height;
  F.tilesX   = (p->width + kTilePx - 1) / kTilePx;
  F.tilesY   = (p->height + kTilePx - 1) / kTilePx;
  int r0 = p->strip_row_begin, r1 = p->strip_row_end;
  if(r0 == 0 && r1 == 0)
    r1 = F.tilesY;
  if(r0 < 0 || r1 > F.tilesY || r0 >= r1)
  {
    setError("frame: strip rows out of range");
    return MGS_ERR_INVALID_ARG;
  }
  F.stripRow0       = r0;
  F.stripRow1       = r1;
  // bins: lists are built per (16<<shift)-pixel bin; the compositor culls per 16x16 tile on chip.
  // Default 256x128 px (MGS_BIN_SHIFT="x,y" overrides, x 1..5, y 0..5; measured sweep in DESIGN.md §3.3).
  int bsx = 4, bsy = 3;  // 256x128 px
  // MGS_ALPHA_SUM has no early termination: every region walks its bin's WHOLE list, and the walk (the cull of stage A) is what
  // that mode pays for — 64 regions re-scanning a 256x128-px bin's list.  Smaller bins there, as small as the direct binning
  // allows (<= 256 bins): 128x64 px at 1080p = 255 bins; composite 6.6 -> 3.8 ms, binning +0.05 ms (profiles/r4_a_alpha_sum_sweep.log)
  if(p->alpha_mode == MGS_ALPHA_SUM && p->sort_mode != MGS_SORT_STOCHASTIC)
  {
    bsx = 3;
    bsy = 2;
  }
  else if(binPolicyEligible(p) && s->binPolicy.fine)
    bsx = bsy = 3;  // 128x128 px: this context's regions scan most of their lists (BinPolicy)
  if(const char* e = std::getenv("MGS_BIN_SHIFT"))
    std::sscanf(e, "%d,%d", &bsx, &bsy);
  else
  {  // keep the frame at <= 256 bins so that the direct (record-free) binning applies: 4K -> 256x128 px bins
    // (8K UHD: 512x512 px bins)
    while(((F.tilesX + (1 << bsx) - 1) >> bsx) * ((F.tilesY + (1 << bsy) - 1) >> bsy) > 256 && (bsx < 5 || bsy < 5))
    {
      if(bsx <= bsy && bsx < 5)
        ++bsx;
      else if(bsy < 5)
        ++bsy;
      else
        ++bsx;
    }
  }
  bsx         = std::min(std::max(bsx, 1), 5);  // the compositor's 32x16-px regions must not straddle bins
  bsy         = std::min(std::max(bsy, 0), 5);
  if(((F.tilesX + (1 << bsx) - 1) >> bsx) > 256 || ((F.tilesY + (1 << bsy) - 1) >> bsy) > 256)
  {
    setError("frame: more than 256 bins along an axis (raise MGS_BIN_SHIFT)");
    return MGS_ERR_INVALID_ARG;
  }
  F.binShiftX = bsx;
  F.binShiftY = bsy;
  F.binsX     = (F.tilesX + (1 << bsx) - 1) >> bsx;
  F.binsY     = (F.tilesY + (1 << bsy) - 1) >> bsy;
  F.splatScale      = p->splat_scale;
  F.frustumDilation = p->frustum_dilation;
  F.alphaCull       = p->alpha_cull_threshold;
  F.shDegree        = p->sh_degree;
  F.frontToBack     = 0;  // keys of -depth: ascending = far to near, the reference's default order
  F.cullMode        = p->frustum_culling;
  if(p->sort_mode == MGS_SORT_CPU_ASYNC && F.cullMode == MGS_CULL_AT_DIST)
    F.cullMode = MGS_CULL_AT_RASTER;  // gaussian_splatting_ui.cpp:1469-1479
  F.msAA            = p->ms_antialiasing;
  F.alphaMode       = p->alpha_mode;
  F.debugFlags      = p->debug_flags;
  if(std::getenv("MGS_LOOSE_MASK")) F.debugFlags |= 256;
  F.sizeCulling     = p->size_culling;
  F.sizeCullingMinPixels = p->size_culling_min_pixels;
  F.surfaceOutputs  = p->surface_outputs ? 1 : 0;
  F.depthIsoThreshold = p->depth_iso_threshold;
  F.thinParticleThreshold = p->thin_particle_threshold;
  F.quantizeNormals       = p->quantize_normals ? 1 : 0;
  // 3DGUT
  F.pipeline          = p->pipeline;
  F.cameraModel       = p->camera_model;
  F.extentMethod      = p->extent_method;
  F.fovRad            = p->fov_rad > 0.0f ? p->fov_rad : 2.0f * std::atan(1.0f / std::fabs(p->proj[5]));
  F.alphaClamp        = p->alpha_clamp;
  F.kernelMinResponse = p->kernel_min_response;
  F.stochastic        = p->sort_mode == MGS_SORT_STOCHASTIC ? 1 : 0;
  if(F.stochastic)
    F.alphaMode = MGS_ALPHA_COVERAGE;  // opaque writes: the pixel's alpha is "a fragment was accepted"
  F.dofMode          = p->pipeline == MGS_PIPELINE_3DGUT ? p->dof_mode : 0;
  F.focusDist        = p->focus_dist;
  F.aperture         = p->aperture;
  F.frameSampleId    = p->frame_sample_id;
  F.temporalSampling = p->temporal_sampling ? 1 : 0;
  F.kernelDegree     = p->kernel_degree;
  F.normalMethod     = p->normal_method;
  chooseRide(s, F, p->sort_mode == MGS_SORT_CPU_ASYNC);
  if(p->camera_model == MGS_CAMERA_FISHEYE && p->pipeline == MGS_PIPELINE_3DGUT)
  {  // gaussian_splatting.cpp:1239-1244: frameInfo.focal is the fisheye focal only for the 3DGUT pipelines; a fisheye camera on
     // the 3DGS pipelines keeps the pinhole focal (and dist.comp's fisheye cull then runs on that)
    F.gutFocal[0] = (float)p->width / F.fovRad;
    F.gutFocal[1] = -(float)p->height / F.fovRad;
  }
  else
  {
    F.gutFocal[0] = F.focal[0];
    F.gutFocal[1] = F.focal[1];
  }
  {  // computeMaxAngle, threedgut_camera_models.h.slang:87-118 (principal point at the viewport centre)
    const float mdx = (float)p->width * 0.5f, mdy = (float)p->height * 0.5f;
    const float maxR = std::sqrt(mdx * mdx + mdy * mdy);
    F.gutMaxAngle = std::max(2.0f * maxR / F.gutFocal[0], 2.0f * maxR / F.gutFocal[1]) / 2.0f;
  }
  mat4Inverse(p->view, F.viewInv);
  mat4Inverse(p->proj, F.projInv);
  F.maxFocal        = std::max(std::fabs(F.gutFocal[0]), std::fabs(F.gutFocal[1]));  // frameInfo.focal (dist.comp.slang:125)
  F.targetFormat    = p->target_format;
  F.nInstances      = (int)s->d->instances.size();
  F.totalSplats     = s->d->totalSplats;
  F.totalPartitions = s->d->totalParts;
  static const bool kPartCull = [] { const char* e = std::getenv("MGS_PARTITION_CULL"); return e ? std::atoi(e) != 0 : true; }();
  F.partitionCull   = (kPartCull && F.cullMode == MGS_CULL_AT_DIST) ? 1 : 0;
  uint32_t offset = 0, block = 0;
  for(int k = 0; k < F.nInstances; ++k)
  {
    const Instance&  I = s->d->instances[k];
    const DeviceSet& d = s->d->sets[I.set];
    InstanceConst&   C = A.inst[k];
    C.centers = d.centers;
    C.cov6    = d.cov6;
    C.rgba    = d.rgba;
    C.sh      = d.sh;
    C.partBox = d.partBox;
    C.maxScale = d.maxScale;
    C.scales    = d.scales;
    C.alpha     = d.alpha;
    C.rgbaF32   = d.rgbaF32;
    C.rotations = d.rotations;
    {
      auto len3 = [&](int c) { return std::sqrt((I.M[4 * c] * I.M[4 * c] + I.M[4 * c + 1] * I.M[4 * c + 1]) + I.M[4 * c + 2] * I.M[4 * c + 2]); };
      C.modelAxisMax = std::max(len3(0), std::max(len3(1), len3(2)));
    }
    std::memcpy(C.model, I.M, sizeof(C.model));
    mat4Mul(p->view, I.M, C.modelView);  // mul(desc.transform, viewMatrix), mesh.slang:175
    float inv[16], cam[4] = {p->camera_pos[0], p->camera_pos[1], p->camera_pos[2], 1.0f}, cm[4];
    mat4Inverse(I.M, inv);
    std::memcpy(C.modelInv, inv, sizeof(inv));
    mat4MulVec4(inv, cam, cm);  // mesh.slang:240-241
    C.camModel[0]  = cm[0];
    C.camModel[1]  = cm[1];
    C.camModel[2]  = cm[2];
    {  // largest singular value of the 3x3: power iteration on A^T A (tiny, per frame, per instance)
      float AtA[9];
      for(int a = 0; a < 3; ++a)
        for(int b = 0; b < 3; ++b)
          AtA[3 * a + b] = I.M[4 * a] * I.M[4 * b] + I.M[4 * a + 1] * I.M[4 * b + 1] + I.M[4 * a + 2] * I.M[4 * b + 2];
      float v[3] = {0.577f, 0.577f, 0.577f}, lam = 0.f;
      for(int it = 0; it < 32; ++it)
      {
        float w[3];
        for(int a = 0; a < 3; ++a)
          w[a] = AtA[3 * a] * v[0] + AtA[3 * a + 1] * v[1] + AtA[3 * a + 2] * v[2];
        lam = std::sqrt(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]);
        if(!(lam > 0.f))
          break;
        for(int a = 0; a < 3; ++a)
          v[a] = w[a] / lam;
      }
      const float fro = std::sqrt(AtA[0] + AtA[4] + AtA[8]);
      C.modelScale    = std::min(fro, std::sqrt(lam) * 1.02f + 1e-6f);  // never above the Frobenius bound
      if(!(C.modelScale > 0.f))
        C.modelScale = fro;
    }
    C.count        = d.count;
    {
      static const float kId[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
      C.modelIsIdentity = std::memcmp(I.M, kId, sizeof(kId)) == 0 ? 1u : 0u;  // bitwise (+0 only)
      {  // last row bitwise (+0, +0, +0, 1), moderate entries: w stays exactly 1 through M (kernels_common.h: mulMat4ExactAffineW1)
        const float    row[4] = {0.0f, 0.0f, 0.0f, 1.0f}, got[4] = {I.M[3], I.M[7], I.M[11], I.M[15]};
        bool           ok     = std::memcmp(row, got, sizeof(row)) == 0;
        for(int q = 0; q < 16 && ok; ++q)
          ok = std::fabs(I.M[q]) < 16777216.0f;  // 2^24; also false for NaN
        C.modelIsAffine = ok ? 1u : 0u;
      }
    }
    C.globalOffset = offset;
    C.blockBegin   = block;
    C.shDegree     = d.shDegree;
    C.shStride     = d.shPitch;  // kernels index by the stored pitch
    offset += d.count;
    block += (d.count + kPart - 1) / kPart;
  }
  return MGS_OK;
}

static int pairSortBits(int nTiles)
{
  int bits = 1;
  while((1 << bits) < nTiles)
    ++bits;
  return ((bits + 7) / 8) * 8;
}


// the compositor's bin order from the previous frame's region times (k_dbin_emit); MGS_BIN_HISTORY=0: longest list first (A/B)
static const bool kBinHistory = [] { const char* e = std::getenv("MGS_BIN_HISTORY"); return e ? std::atoi(e) != 0 : true; }();

// pass elision of the key sort (sort_plan.h): on by default, MGS_SORT_REMAP=0 keeps the four plain passes
static const bool kRemap = [] { const char* e = std::getenv("MGS_SORT_REMAP"); return e ? std::atoi(e) != 0 : true; }();

// the project kernels' partitions are dispatched fullest slot of the previous frame first (k_os_prepare writes the order,
// k_project reads it; scheduling only); MGS_PRJ_ORDER=0: storage order (A/B)
static const bool kPrjOrder = [] { const char* e = std::getenv("MGS_PRJ_ORDER"); return e ? std::atoi(e) != 0 : true; }();

// the frame's key sort (k_osort.hip): slots of the project kernel -> sorted ids in idsA (keys in keysA when wanted)
static void keySort(MgsScene s, hipStream_t st, bool wantKeys, bool allowRemap, const FrameConst* ride = nullptr)
{
  OsLaunch L{};
  L.pairs0       = s->pairB.p;
  L.prjParts     = s->d->totalParts;
  L.slotCount    = s->slotCount.p;
  L.chunkSum     = s->chunkSum.p;
  L.runTab       = s->runTab.p;
  L.nOut         = &s->ctr.p->sortedCount;
  L.prjOrderOut  = kPrjOrder ? s->prjOrder.p : nullptr;
  L.slotHist     = s->slotHist2.p;
  L.top16Rec     = s->top16Rec.p;
  L.top16Count   = s->top16Count.p;
  L.nPtr         = &s->ctr.p->sortedCount;
  L.maxElems     = s->d->totalSplats;
  L.pairA        = s->pairA.p;
  L.pairB        = s->pairB.p;
  L.outVals      = s->idsA.p;
  L.outKeys      = wantKeys ? s->keysA.p : nullptr;
  L.plan         = &s->plans.p->os;
  L.planOut      = &s->plans.p->keys;
  L.status       = s->osStatus.p;
  L.ctr          = s->ctr.p;
  L.allowRemap   = allowRemap;
  if(ride != nullptr && ride->rideShift != 0 && !wantKeys)
  {
    int codeBits = 0;
    while((1u << codeBits) - 1u < ride->rideEscape)
      ++codeBits;
    L.rideShift = (uint32_t)ride->rideShift;
    L.rideSplit = (uint32_t)ride->rideSplit;
    L.rideInfo  = (uint32_t)ride->rideShapes | ((uint32_t)codeBits << 8);
    L.outCode16 = s->sortedCode16.p;
  }
  launchOsSort(st, L);
}

// CPU_ASYNC path: tryConsumeAndUploadCpuSortingResult (src/splat_set_manager_vk.cpp:3334-3416)
static int cpuSortStep(MgsScene s, const MgsFrameParams* p, bool blocking)
{
  CpuSorter& c = s->cpu;
  c.ensureStarted();
  std::unique_lock<std::mutex> lk(c.mtx);
  auto submit = [&]() {
    // view direction = -Z axis of the camera in world space; centre of projection = camera position
    // (SplatSetManagerVk passes cameraManip's eye/centre; with matrices only, the third row of the
    //  view matrix is the same direction)
    c.job.dir[0] = -p->view[2];
    c.job.dir[1] = -p->view[6];
    c.job.dir[2] = -p->view[10];
    std::memcpy(c.job.cop, p->camera_pos, sizeof(float) * 3);
    c.job.frontToBack = false;
    c.job.inst.clear();
    uint32_t offset = 0;
    for(const auto& I : s->d->instances)
    {
      CpuSorter::Job::Inst ji;
      ji.set = s->d->sets[I.set].host;
      std::memcpy(ji.M, I.M, sizeof(ji.M));
      ji.offset = offset;
      ji.count  = s->d->sets[I.set].count;
      offset += ji.count;
      c.job.inst.push_back(ji);
    }
    c.job.total = offset;
    c.state     = CpuSorter::SORTING;
    c.cv.notify_all();
  };
  if(c.state == CpuSorter::SORTED)
  {
    s->cpuIndices.swap(c.indices);  // consume()
    s->cpuDistances.swap(c.distances);  // the worker resizes and rewrites its own copy on the next job
    s->cpuHaveIndices = true;
    c.state           = CpuSorter::READY;
  }
  // lazy (parameters.h:183, splat_sorter_async.h:81-97): a new sort starts only if the viewpoint changed since the
  // last one that was started (direction, centre of projection, order — instance transforms are not compared there either)
  const float dirNow[3] = {-p->view[2], -p->view[6], -p->view[10]};
  const bool  sameView  = c.haveLast && std::memcmp(dirNow, c.lastDir, sizeof(dirNow)) == 0
                        && std::memcmp(p->camera_pos, c.lastCop, sizeof(float) * 3) == 0;
  const bool  lazySkip  = p->cpu_lazy_sort != 0 && sameView && s->cpuHaveIndices;
  if(c.state == CpuSorter::READY && !lazySkip)
  {
    submit();
    std::memcpy(c.lastDir, dirNow, sizeof(dirNow));
    std::memcpy(c.lastCop, p->camera_pos, sizeof(float) * 3);
    c.haveLast = true;
  }
  if(blocking && c.state == CpuSorter::SORTING)
  {
    c.cv.wait(lk, [&] { return c.state == CpuSorter::SORTED; });
    s->cpuIndices.swap(c.indices);
    s->cpuDistances.swap(c.distances);
    s->cpuHaveIndices = true;
    c.state           = CpuSorter::READY;
  }
  s->lastSort.key_ms  = (float)c.distMs;
  s->lastSort.sort_ms = (float)c.sortMs;
  return MGS_OK;
}

__global__ void k_set_plan_n(SortPlan* plan, FrameCounters* ctr, uint32_t n)
{
  plan->n         = n;
  plan->finalSel  = 0;
  plan->passesRun = 0;
  ctr->sortedCount = n;
}
__global__ void k_fill_u32(uint32_t* p, uint32_t v, uint32_t n)
{
  for(uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
    p[i] = v;
}
// post.comp.slang:29-43: main = lerp(main, aux1, 1 / (frameSampleId + 1)) — aux1 is the frame just rendered, main the
// accumulated image.  Here the accumulator is a separate fp32 image and the result is written back over the frame in the
// target's format, so the caller reads the running mean where it reads every frame.
__global__ void k_post_accumulate(const FrameArgs* __restrict__ Ap, float4* __restrict__ acc, void* __restrict__ image, int halfOut)
{
  const FrameConst& F  = Ap->f;
  const int         y0 = F.stripRow0 * kTilePx, y1 = min(F.stripRow1 * kTilePx, F.height);
  const size_t      n  = (size_t)(y1 - y0) * (size_t)F.width;
  const float       a  = 1.0f / (float)(F.frameSampleId + 1);
  for(size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
  {
    const size_t o = (size_t)y0 * F.width + i;
    float4       c;
    if(halfOut == 1)
    {
      const uint2   pk = reinterpret_cast<const uint2*>(image)[o];
      const float2  lo = __half22float2(*reinterpret_cast<const __half2*>(&pk.x)), hi = __half22float2(*reinterpret_cast<const __half2*>(&pk.y));
      c = make_float4(lo.x, lo.y, hi.x, hi.y);
    }
    else if(halfOut == 2)
    {
      const uint32_t pk = reinterpret_cast<const uint32_t*>(image)[o];
      c = make_float4((float)(pk & 255u) / 255.0f, (float)((pk >> 8) & 255u) / 255.0f, (float)((pk >> 16) & 255u) / 255.0f, (float)(pk >> 24) / 255.0f);
    }
    else
      c = reinterpret_cast<const float4*>(image)[o];
    float4 m = acc[o];
    if(F.frameSampleId <= 0)
      m = c;  // lerp(main, aux1, 1) without touching what the accumulator held (it may be uninitialised)
    else
    {  // lerp(x, y, s) = x + s * (y - x)
      m.x = m.x + a * (c.x - m.x);
      m.y = m.y + a * (c.y - m.y);
      m.z = m.z + a * (c.z - m.z);
      m.w = m.w + a * (c.w - m.w);
    }
    acc[o] = m;
    if(halfOut == 1)
    {
      const __half2 lo = __floats2half2_rn(m.x, m.y), hi = __floats2half2_rn(m.z, m.w);
      uint2         pk;
      pk.x = *reinterpret_cast<const uint32_t*>(&lo);
      pk.y = *reinterpret_cast<const uint32_t*>(&hi);
      reinterpret_cast<uint2*>(image)[o] = pk;
    }
    else if(halfOut == 2)
    {
      auto q8 = [](float v) { return (uint32_t)(fminf(fmaxf(v, 0.0f), 1.0f) * 255.0f + 0.5f); };
      reinterpret_cast<uint32_t*>(image)[o] = q8(m.x) | (q8(m.y) << 8) | (q8(m.z) << 16) | (q8(m.w) << 24);
    }
    else
      reinterpret_cast<float4*>(image)[o] = m;
  }
}
__global__ void k_iota_u32(uint32_t* p, uint32_t n)
{
  for(uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
    p[i] = i;
}

static int mgs_render_impl(MgsScene s, const MgsFrameParams* p, MgsFrameOut* out);
int mgs_render(MgsScene s, const MgsFrameParams* p, MgsFrameOut* out)
{
  return guarded("mgs_render", [&] { return mgs_render_impl(s, p, out); });
}
static int mgs_render_impl(MgsScene s, const MgsFrameParams* p, MgsFrameOut* out)
{
  if(!s || !p)
  {
    setError("mgs_render: null argument");
    return MGS_ERR_INVALID_ARG;
  }
  if(!s->d->committed)
  {
    setError("mgs_render: call mgs_scene_commit first");
    return MGS_ERR_STATE;
  }
  if(int wrc = ensureWorkingSet(s))
    return wrc;
  HIPCHK(hipSetDevice(s->device));
  // adaptive bin size: a sample taken 8 eligible frames ago is applied now (BinPolicy)
  const bool policyFrame = binPolicyEligible(p);
  if(policyFrame && s->binPolicy.pending && (s->binPolicy.frames & 31u) == 24u)
  {
    MgsScene_t::BinPolicy& B = s->binPolicy;
    HIPCHK(hipEventSynchronize(B.ev));  // (eight frames of this context's stream ago: long done)
    uint64_t scanned = 0;
    for(uint32_t i = 0; i < kFrameStatSlots; ++i)
      scanned += B.host[32u * i + 1u];
    const uint64_t D = B.host[4];
    B.pending        = false;
    if(D > 0)
    {
      B.lastRatio = (float)((double)scanned / ((double)D * (double)B.sampledRegions));
      // ... and how far a region walks in absolute terms: one stage-A round is 512 entries, and a region that saturates inside its
      // first round or two (686 entries on the benchmark scene; 1 140-2 570 on the sparse / train-sized / fog scenes) has nothing
      // to gain from a shorter list, however short the list is (a small scene's lists are a few rounds long: its ratio is high for
      // that reason alone)
      const float perRegion = (float)((double)scanned / (double)B.sampledRegionCount);
      // (under the finer bins a saturating region's share of its — shorter — list is about twice what it is under the coarse ones)
      B.fine = B.sampledFine ? ((B.lastRatio > 0.030f && perRegion > 450.0f) ? 1 : 0) : ((B.lastRatio > 0.025f && perRegion > 900.0f) ? 1 : 0);
    }
  }
  FrameArgs A;
  int       rc = buildFrameArgs(s, p, A);
  if(rc != MGS_OK)
    return rc;
  const FrameConst& F      = A.f;
  const uint32_t    nTiles = (uint32_t)(F.binsX * F.binsY);  // lists (and ranges) exist per bin
  if(p->target_format < MGS_TARGET_RGBA16F || p->target_format > MGS_TARGET_RGBA8)
  {
    setError("frame: target_format must be MGS_TARGET_RGBA16F / RGBA32F / RGBA8");
    return MGS_ERR_INVALID_ARG;
  }
  // compositor output mode: 1 = RGBA16F, 0 = RGBA32F, 2 = RGBA8 (linear UNORM, rounded once at the end)
  const int         half   = (p->target_format == MGS_TARGET_RGBA16F) ? 1 : (p->target_format == MGS_TARGET_RGBA8 ? 2 : 0);
  const size_t      pixB   = half == 1 ? 8 : (half == 2 ? 4 : 16);
  s->imageRowBytes         = (size_t)F.width * pixB;
  s->imageBytes            = s->imageRowBytes * (size_t)F.height;
  const bool gut = p->pipeline == MGS_PIPELINE_3DGUT;
  if(p->pipeline != MGS_PIPELINE_3DGS && !gut)
  {
    setError("frame: pipeline must be MGS_PIPELINE_3DGS or MGS_PIPELINE_3DGUT");
    return MGS_ERR_INVALID_ARG;
  }
  if(p->camera_model < MGS_CAMERA_PINHOLE || p->camera_model > MGS_CAMERA_FISHEYE)
  {  // CAMERA_TYPE also selects the dist-stage cull of the 3DGS pipelines (dist.comp.slang:64-91)
    setError("frame: camera_model out of range");
    return MGS_ERR_INVALID_ARG;
  }
  if(gut)
  {
    if(p->extent_method < MGS_EXTENT_EIGEN || p->extent_method > MGS_EXTENT_CONIC)
    {
      setError("frame: extent_method out of range");
      return MGS_ERR_INVALID_ARG;
    }
    // the float knobs of the particle response and the lens: values outside these ranges make every ray NaN (negative or
    // non-finite aperture, focus distance <= 0) or reject / accept every hit differently from the reference (alphaClamp <=
    // 1/255 rejects everything in threedgrt.h.slang:181-184, which the per-record cutoff of the packed compositor would not)
    if(!(p->alpha_clamp > 1.0f / 255.0f && p->alpha_clamp <= 1.0f) || !(p->kernel_min_response >= 0.0f && p->kernel_min_response < 1.0f))
    {
      setError("frame: alpha_clamp must be in (1/255, 1] and kernel_min_response in [0, 1)");
      return MGS_ERR_INVALID_ARG;
    }
    if(p->dof_mode != MGS_DOF_DISABLED && (!(p->aperture >= 0.0f) || !std::isfinite(p->aperture) || !(p->focus_dist > 0.0f) || !std::isfinite(p->focus_dist)))
    {
      setError("frame: depth of field needs a finite aperture >= 0 and a finite focus_dist > 0");
      return MGS_ERR_INVALID_ARG;
    }
    if(p->kernel_degree != 8 && (p->kernel_degree < 0 || p->kernel_degree > 5))
    {
      setError("frame: kernel_degree must be one of 0, 1, 2, 3, 4, 5, 8 (shaderio.h:112-119)");
      return MGS_ERR_INVALID_ARG;
    }
    if(p->normal_method != MGS_NORMAL_MAX_DENSITY_PLANE && p->normal_method != MGS_NORMAL_ISO_SURFACE)
    {
      setError("frame: normal_method must be MGS_NORMAL_MAX_DENSITY_PLANE or MGS_NORMAL_ISO_SURFACE (shaderio.h:126-128)");
      return MGS_ERR_INVALID_ARG;
    }
    if(s->recGut.n < s->d->totalSplats)
    {  // first 3DGUT frame of this scene: its record buffer (captured frames do not reference it yet)
      if((rc = s->recGut.ensure(s->d->totalSplats))) return rc;
    }
  }
  else if(p->dof_mode != MGS_DOF_DISABLED)
  {
    setError("frame: depth of field is a feature of the 3DGUT pipeline (per-pixel rays); the 3DGS pipeline has none");
    return MGS_ERR_UNSUPPORTED;
  }
  if(p->sort_mode != MGS_SORT_GPU_RADIX && p->sort_mode != MGS_SORT_CPU_ASYNC && p->sort_mode != MGS_SORT_STOCHASTIC)
  {
    setError("frame: sort_mode must be MGS_SORT_GPU_RADIX, MGS_SORT_CPU_ASYNC or MGS_SORT_STOCHASTIC");
    return MGS_ERR_INVALID_ARG;
  }
  if(p->dof_mode < MGS_DOF_DISABLED || p->dof_mode > MGS_DOF_FIXED_FOCUS || p->frame_sample_id < 0)
  {
    setError("frame: dof_mode / frame_sample_id out of range");
    return MGS_ERR_INVALID_ARG;
  }
  const void* before[6] = {s->ranges.p, s->image.p, s->surfDepth.p, s->surfId.p, s->surfNormal.p, s->accum.p};
  if((rc = s->ranges.ensure(std::max<uint32_t>(nTiles, 256u)))) return rc;
  if(s->image.n < s->imageBytes)
  {
    if((rc = s->image.ensure(s->imageBytes))) return rc;
    HIPCHK(hipMemsetAsync(s->image.p, 0, s->imageBytes, s->stream));
  }
  if(F.surfaceOutputs)
  {
    if((rc = s->surfDepth.ensure((size_t)F.width * F.height))) return rc;
    if((rc = s->surfId.ensure((size_t)F.width * F.height))) return rc;
    if((rc = s->surfNormal.ensure((size_t)F.width * F.height))) return rc;
  }
  if(F.temporalSampling)
    if((rc = s->accum.ensure((size_t)F.width * F.height))) return rc;
  {
    if(!(kDirectBin && directBinningSupported(F.binsX, F.binsY)) && s->pairKey0.n < s->pairCapacity)
    {  // first frame on the record + pair-sort path (> 256 bins, or forced): its buffers
      const uint64_t cap = s->pairCapacity;
      if((rc = s->pairKey0.ensure(cap))) return rc;
      if((rc = s->pairVal0.ensure(cap))) return rc;
      if((rc = s->pairKey1.ensure(cap))) return rc;
      if((rc = s->chunkStart.ensure(cap / kPart + 4))) return rc;
      if((rc = s->splatOffset.ensure(s->d->totalSplats))) return rc;
    }
  }
  s->haveSurface    = F.surfaceOutputs != 0;
  {
    const void* after[6] = {s->ranges.p, s->image.p, s->surfDepth.p, s->surfId.p, s->surfNormal.p, s->accum.p};
    if(std::memcmp(before, after, sizeof(before)) != 0 && !s->graphs.empty())
    {  // a buffer moved: captured frames point at the old one
      HIPCHK(hipStreamSynchronize(s->stream));
      for(auto& g : s->graphs)
        (void)hipGraphExecDestroy(g.second);
      s->graphs.clear();
    }
  }
  hipStream_t st    = s->stream;
  const bool  timed = p->collect_timings != 0;
  FrameCounters* ctr = s->ctr.p;
  SortPlan*      planK = &s->plans.p->keys;
  SortPlan*      planP = &s->plans.p->pairs;

  hipEvent_t* fev = s->evRing[s->frameIndex % MgsScene_t::kRing];
  const bool  cpuModeOuter = (p->sort_mode == MGS_SORT_CPU_ASYNC);
  // this frame's constants, and its counters and plans zeroed: one 18 KB upload
  if((rc = uploadFrameState(s, A, st))) return rc;
  auto issue = [&](bool withEvents) -> int {
    if(withEvents) HIPCHK(hipEventRecord(fev[0], st));
    // (counters and sort plans arrive zeroed with the upload)
    if(!(kDirectBin && directBinningSupported(F.binsX, F.binsY)))
      launchFrameInit(st, s->ranges.p, nTiles);  // record path: tiles without entries keep an empty range
    if(withEvents) HIPCHK(hipEventRecord(fev[6], st));  // MGS_STAGE_CULL ends here; it is part of MGS_STAGE_PROJECT too
    const bool cpuMode = (p->sort_mode == MGS_SORT_CPU_ASYNC);
    if(cpuMode)  // rejected splats must look empty to the binning stage: rect with x0 > x1
      hipLaunchKernelGGL(k_fill_u32, dim3(1024), dim3(256), 0, st, s->rect.p, 1u, s->d->totalSplats);
    if(gut)
      launchProjectGut(st, A, s->dArgs.p, s->d->shFormat, ctr, s->pairB.p, s->slotCount.p, s->recGut.p, s->rect.p,
                       s->slotHist2.p, s->top16Rec.p, cpuMode ? nullptr : s->top16Count.p, &s->plans.p->os, kPrjOrder ? s->prjOrder.p : nullptr);
    else
      launchProject(st, A, s->dArgs.p, true, ctr, s->pairB.p, s->slotCount.p, s->rec.p, s->rect.p,
                    s->slotHist2.p, s->top16Rec.p, cpuMode ? nullptr : s->top16Count.p, &s->plans.p->os, kPrjOrder ? s->prjOrder.p : nullptr);
    if(withEvents) HIPCHK(hipEventRecord(fev[1], st));
    if(!cpuMode)
      keySort(s, st, false, kRemap, &F);
    else
    {
      rc = cpuSortStep(s, p, p->cpu_sort_blocking != 0);
      if(rc != MGS_OK)
        return rc;
      if(s->cpuHaveIndices && s->cpuIndices.size() == s->d->totalSplats)
      {
        s->cpuStorageIds = s->cpuIndices;  // the sorter works in the caller's id space
        mapIdsToStorage(s, s->cpuStorageIds.data(), s->cpuStorageIds.size());
        HIPCHK(hipMemcpyAsync(s->idsA.p, s->cpuStorageIds.data(), (size_t)s->d->totalSplats * 4, hipMemcpyHostToDevice, st));
      }
      else  // no result yet: the reference draws with whatever the index buffer holds; we use identity order
        hipLaunchKernelGGL(k_iota_u32, dim3(1024), dim3(256), 0, st, s->idsA.p, s->d->totalSplats);
      hipLaunchKernelGGL(k_set_plan_n, dim3(1), dim3(1), 0, st, planK, ctr, s->d->totalSplats);
    }
    if(withEvents) HIPCHK(hipEventRecord(fev[2], st));
    // coarse bins (the default): stable multi-split straight into the per-bin lists, no records, no pair sort
    const bool direct = kDirectBin && directBinningSupported(F.binsX, F.binsY);
    if(direct)
    {
      launchDirectBinning(st, s->idsA.p, s->idsA.p, planK, s->rect.p, s->sortedCode16.p, s->dbinMasks.p, s->d->totalSplats,
                          s->partHist.p, s->pStride, &planP->ghist[0][0], s->pairVal1.p, s->ranges.p, ctr, s->pairCapacity,
                          F.binsX, F.binsY, kBinHistory ? s->binCost.p : nullptr);
      if(withEvents) HIPCHK(hipEventRecord(fev[3], st));
    }
    else
    {
      launchBinning(st, s->idsA.p, s->idsA.p, planK, s->rect.p, s->blockCount.p, (s->d->totalSplats + kPart - 1) / kPart, ctr,
                    s->sortedRect.p, s->splatOffset.p, s->chunkStart.p, s->pairKey0.p, s->pairVal0.p, s->pairCapacity, F.binsX, true);
      if(withEvents) HIPCHK(hipEventRecord(fev[3], st));
      {
        SortLaunch L{};
        L.keys0 = s->pairKey0.p;
        L.vals0 = s->pairVal0.p;
        L.keysX = s->pairKey1.p;
        L.valsX = s->pairVal1.p;
        L.keysY = s->pairKey0.p;
        L.valsY = s->pairVal0.p;
        L.nPtr      = &ctr->pairCount;
        L.plan      = planP;
        L.partHist  = s->partHist.p;
        L.pStride   = s->pStride;
        L.maxElems  = s->pairCapacity;
        L.beginBit  = 0;
        L.endBit    = pairSortBits((int)nTiles);
        const bool onePass = L.endBit <= 8;  // <= 256 bins: the digit histogram IS the range table
        L.ranges    = onePass ? s->ranges.p : nullptr;
        launchRadixSort(st, L);
        if(!onePass)
          launchTileRanges(st, s->pairKey1.p, s->pairKey0.p, planP, s->ranges.p);
      }
    }
    if(withEvents) HIPCHK(hipEventRecord(fev[4], st));
    if(gut)
      launchCompositeGut(st, A, s->dArgs.p, s->ranges.p, s->pairVal1.p, s->pairVal0.p, planP, s->recGut.p, s->image.p, half, ctr, s->d->shFormat,
                         F.surfaceOutputs ? s->surfDepth.p : nullptr, F.surfaceOutputs ? s->surfId.p : nullptr,
                         F.surfaceOutputs ? s->surfNormal.p : nullptr);
    else
      launchComposite(st, A, s->ranges.p, s->pairVal1.p, s->pairVal0.p, planP, s->rec.p, s->image.p, half, s->d->shFormat, ctr,
                      F.surfaceOutputs ? s->surfDepth.p : nullptr, F.surfaceOutputs ? s->surfId.p : nullptr, s->d->compInst.p,
                      s->dArgs.p, F.surfaceOutputs ? s->surfNormal.p : nullptr, kBinHistory ? s->binCost.p : nullptr);
    if(F.temporalSampling)
      hipLaunchKernelGGL(k_post_accumulate, dim3(2048), dim3(256), 0, st, s->dArgs.p, s->accum.p, s->image.p, half);
    if(withEvents) HIPCHK(hipEventRecord(fev[5], st));
    return MGS_OK;
  };
  // The frame is 17 dependent launches (18 with temporal accumulation); on the submitting thread that is ~0.15 ms of API calls.  A frame whose
  // launch sequence does not depend on host-side data (GPU sort, no per-stage events) is captured ONCE per
  // (resolution, strip, mode) into a hipGraph and replayed: the kernels read everything that changes from frame to
  // frame through dArgs.  MGS_GRAPH=0 forces plain launches.
  static const bool kUseGraph = [] { const char* e = std::getenv("MGS_GRAPH"); return e ? std::atoi(e) != 0 : true; }();
  bool launched = false;
  if(kUseGraph && s->graphOk && !timed && !cpuModeOuter)
  {
    MgsScene_t::GraphKey key;
    std::memset(&key, 0, sizeof(key));
    int32_t isoBits;
    std::memcpy(&isoBits, &F.depthIsoThreshold, 4);
    // everything the compositor receives by value (CompositeArgs) must be part of the key
    const int32_t kv[16] = {F.width, F.height, F.stripRow0, F.stripRow1, F.binShiftX, F.binShiftY, F.partitionCull, F.alphaMode,
                            F.debugFlags & (2 | 4 | 256), F.surfaceOutputs, half, F.nInstances, F.shDegree, isoBits,
                            F.pipeline, F.stochastic | (F.dofMode << 1) | (F.temporalSampling << 2) | ((F.pipeline == 1 && F.kernelDegree != 2) ? 8 : 0) |
                                ((F.pipeline == 1 && F.normalMethod == 1) ? 16 : 0) |
                                (F.rideShift << 8) | (F.rideShapes << 16) | (F.rideSplit << 24)};  // ... and everything that selects a kernel variant or a launch argument
    std::memcpy(key.v, kv, sizeof(kv));
    key.p[0] = s->image.p;
    key.p[1] = s->surfDepth.p;
    auto it = s->graphs.find(key);
    if(it == s->graphs.end())
    {
      hipGraph_t     graph = nullptr;
      hipGraphExec_t exec  = nullptr;
      bool           ok    = hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal) == hipSuccess;
      if(ok)
      {
        const int rcI = issue(false);
        ok            = (hipStreamEndCapture(st, &graph) == hipSuccess) && rcI == MGS_OK && graph != nullptr;
      }
      if(ok)
        ok = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0) == hipSuccess;
      if(graph)
        (void)hipGraphDestroy(graph);
      if(ok)
        it = s->graphs.emplace(key, exec).first;
      else
      {
        (void)hipGetLastError();
        s->graphOk = false;  // this runtime cannot capture the frame: stay with plain launches
      }
    }
    if(it != s->graphs.end())
    {
      HIPCHK(hipGraphLaunch(it->second, st));
      launched = true;
    }
  }
  if(!launched)
  {
    rc = issue(timed);
    if(rc != MGS_OK)
      return rc;
  }
  if(policyFrame && kDirectBin && directBinningSupported(F.binsX, F.binsY))
  {  // adaptive bin size: every 32nd eligible frame leaves a copy of its statistics lines (4 KB) for the frame eight later
    MgsScene_t::BinPolicy& B = s->binPolicy;
    if((B.frames & 31u) == 16u && !B.pending)
    {
      if(!B.host)
      {
        if(hipHostMalloc((void**)&B.host, kFrameStatSlots * 32u * sizeof(uint32_t)) != hipSuccess || hipEventCreateWithFlags(&B.ev, hipEventDisableTiming) != hipSuccess)
        {
          (void)hipGetLastError();
          if(B.host) (void)hipHostFree(B.host);
          B.host = nullptr;
        }
      }
      if(B.host)
      {
        HIPCHK(hipMemcpyAsync(B.host, &s->plans.p->keys.ghist[0][0], kFrameStatSlots * 32u * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
        HIPCHK(hipEventRecord(B.ev, st));
        B.pending        = true;
        B.sampledFine    = (F.binShiftX == 3 && F.binShiftY == 3) ? 1 : 0;
        B.sampledRegions = (1 << (F.binShiftX - 1)) * (1 << F.binShiftY);
        B.sampledRegionCount = (uint32_t)std::max(1, ((F.tilesX + 1) / 2) * (F.stripRow1 - F.stripRow0));
      }
    }
    ++B.frames;
  }
  // the counters stay on the device; mgs_frame_stats fetches them when somebody asks (two API calls per frame less
  // on the submitting thread, which spends ~6 us per call)
  HIPCHK(hipGetLastError());
  s->lastParams      = *p;
  s->lastParams.strip_row_begin = F.stripRow0;
  s->lastParams.strip_row_end   = F.stripRow1;
  s->haveFrame       = true;
  s->lastTimed       = timed;
  s->lastWasSortOnly = false;
  s->lastListsPartial = false;
  {
    int codeBits = 0;
    while(F.rideShift != 0 && (1u << codeBits) - 1u < F.rideEscape)
      ++codeBits;
    const int lr[5] = {F.rideShift, codeBits, F.binsX, F.binsY, cpuModeOuter ? 0 : 1};
    std::memcpy(s->lastRide, lr, sizeof(lr));
    s->lastBinShift[0] = F.binShiftX;
    s->lastBinShift[1] = F.binShiftY;
  }
  if(timed)
    ++s->frameIndex;
  if(out)
  {
    std::memset(out, 0, sizeof(*out));
    out->rgba_device = s->image.p;
    out->rgba_bytes  = s->imageBytes;
    if(p->collect_timings == 1)
      return mgs_frame_stats(s, out);
  }
  return MGS_OK;
}

int mgs_timings_query(MgsScene s, uint32_t framesBack, float* stageMs)
{
  if(!s || !stageMs)
  {
    setError("mgs_timings_query: null argument");
    return MGS_ERR_INVALID_ARG;
  }
  if(framesBack >= (uint32_t)MgsScene_t::kRing || framesBack >= s->frameIndex)
  {
    setError("mgs_timings_query: no timed frame that far back (ring of 128)");
    return MGS_ERR_INVALID_ARG;
  }
  HIPCHK(hipSetDevice(s->device));
  hipEvent_t* fev = s->evRing[(s->frameIndex - 1 - framesBack) % MgsScene_t::kRing];
  HIPCHK(hipEventSynchronize(fev[5]));
  std::memset(stageMs, 0, sizeof(float) * MGS_STAGE_COUNT);
  float ms = 0;
  for(int i = 0; i < 5; ++i)
  {
    HIPCHK(hipEventElapsedTime(&ms, fev[i], fev[i + 1]));
    stageMs[i] = ms;
  }
  HIPCHK(hipEventElapsedTime(&ms, fev[0], fev[5]));
  stageMs[MGS_STAGE_TOTAL] = ms;
  HIPCHK(hipEventElapsedTime(&ms, fev[0], fev[6]));
  stageMs[MGS_STAGE_CULL] = ms;
  return MGS_OK;
}

int mgs_frame_stats(MgsScene s, MgsFrameOut* out)
{
  if(!s || !out)
  {
    setError("mgs_frame_stats: null argument");
    return MGS_ERR_INVALID_ARG;
  }
  if(!s->haveFrame)
  {
    setError("mgs_frame_stats: no frame rendered yet");
    return MGS_ERR_STATE;
  }
  HIPCHK(hipSetDevice(s->device));
  if(!s->lastWasSortOnly)
  {
    HIPCHK(hipMemcpyAsync(s->hCtr, s->ctr.p, sizeof(FrameCounters), hipMemcpyDeviceToHost, s->stream));
    HIPCHK(hipMemcpyAsync(s->hPlans, s->plans.p, sizeof(FramePlans), hipMemcpyDeviceToHost, s->stream));
  }
  HIPCHK(hipStreamSynchronize(s->stream));
  std::memset(out->stage_ms, 0, sizeof(out->stage_ms));
  out->rgba_device   = s->image.p;
  out->rgba_bytes    = s->imageBytes;
  out->frustum_count = s->hCtr->frustumCount;
  out->sorted_count  = s->hCtr->sortedCount;
  out->tile_pairs    = s->hCtr->pairCount;
  out->error_flags   = s->hCtr->errorFlags;
  out->shaded_count  = 0;
  out->scanned_entries = 0;
  out->escape_count  = 0;
  for(uint32_t i = 0; i < kFrameStatSlots; ++i)
    {  // (sort_plan.h: frameStatSlot — the kernels' counts live in the keys plan's histogram rows, one 128-byte line per slot; after
       //  mgs_sort_keys the plans were fetched by that call)
      out->frustum_count += (&s->hPlans->keys.ghist[0][0])[32u * i + 2u];
      out->shaded_count += (&s->hPlans->keys.ghist[0][0])[32u * i];
      out->scanned_entries += (&s->hPlans->keys.ghist[0][0])[32u * i + 1u];
      out->escape_count += (&s->hPlans->keys.ghist[0][0])[32u * i + 3u];
    }
  if(!s->lastWasSortOnly && s->lastRide[0] == 0)
    out->escape_count = out->sorted_count;  // nothing rode through the sort: every sorted splat's rectangle was stored and gathered by id
  if(s->lastTimed && !s->lastWasSortOnly)
  {
    int rc = mgs_timings_query(s, 0, out->stage_ms);
    if(rc != MGS_OK)
      return rc;
  }
  if(out->error_flags & kErrPairOverflow)
  {
    setError("frame: tile-pair capacity exceeded (raise MGS_PAIR_CAPACITY); frame is incomplete");
    return MGS_ERR_OVERFLOW;
  }
  if(out->error_flags & kErrSpinTimeout)
  {  // a look-back wait of the key sort ran into its bound (k_osort.hip): the sorted order, hence the frame, is not to be trusted
    setError("frame: a look-back wait of the key sort gave up (kErrSpinTimeout); the frame is invalid");
    return MGS_ERR_DEVICE;
  }
  return MGS_OK;
}

int mgs_frame_download_surface(MgsScene s, int which, void* dst, size_t bytes)
{
  if(!s || !dst || which < 0 || which > 2)
  {
    setError("mgs_frame_download_surface: bad argument");
    return MGS_ERR_INVALID_ARG;
  }
  if(!s->haveFrame || !s->haveSurface || s->lastWasSortOnly)
  {
    setError("mgs_frame_download_surface: the last frame was not rendered with surface_outputs = 1");
    return MGS_ERR_STATE;
  }
  const size_t n = (size_t)s->lastParams.width * (size_t)s->lastParams.height;
  if(bytes < n * (which == 2 ? 16 : 4))
  {
    setError("mgs_frame_download_surface: destination too small");
    return MGS_ERR_INVALID_ARG;
  }
  HIPCHK(hipSetDevice(s->device));
  HIPCHK(hipStreamSynchronize(s->stream));
  if(which == 2)
  {
    HIPCHK(hipMemcpy(dst, s->surfNormal.p, n * 16, hipMemcpyDeviceToHost));
    return MGS_OK;
  }
  HIPCHK(hipMemcpy(dst, which == 0 ? (const void*)s->surfDepth.p : (const void*)s->surfId.p, n * 4, hipMemcpyDeviceToHost));
  if(which == 1)
  {  // the pipeline works on storage ids: hand out the caller's
    uint32_t*             ids = static_cast<uint32_t*>(dst);
    std::vector<uint32_t> hit;
    std::vector<size_t>   where;
    for(size_t i = 0; i < n; ++i)
      if(ids[i] != 0xFFFFFFFFu)
      {
        hit.push_back(ids[i]);
        where.push_back(i);
      }
    mapIdsToCaller(s, hit.data(), hit.size());
    for(size_t i = 0; i < hit.size(); ++i)
      ids[where[i]] = hit[i];
  }
  return MGS_OK;
}

int mgs_frame_download(MgsScene s, void* dst, size_t bytes)
{
  if(!s || !dst)
  {
    setError("mgs_frame_download: null argument");
    return MGS_ERR_INVALID_ARG;
  }
  if(!s->haveFrame || s->lastWasSortOnly)
  {
    setError("mgs_frame_download: no frame rendered yet");
    return MGS_ERR_STATE;
  }
  if(bytes < s->imageBytes)
  {
    setError("mgs_frame_download: destination too small");
    return MGS_ERR_INVALID_ARG;
  }
  HIPCHK(hipSetDevice(s->device));
  HIPCHK(hipMemcpyAsync(dst, s->image.p, s->imageBytes, hipMemcpyDeviceToHost, s->stream));
  HIPCHK(hipStreamSynchronize(s->stream));
  return MGS_OK;
}

int mgs_frame_copy_strip(MgsScene s, void* dst, size_t bytes)
{
  if(!s || !dst)
  {
    setError("mgs_frame_copy_strip: null argument");
    return MGS_ERR_INVALID_ARG;
  }
  if(!s->haveFrame || s->lastWasSortOnly)
  {
    setError("mgs_frame_copy_strip: no frame rendered yet");
    return MGS_ERR_STATE;
  }
  const int    y0 = s->lastParams.strip_row_begin * kTilePx;
  const int    y1 = std::min(s->lastParams.strip_row_end * kTilePx, s->lastParams.height);
  const size_t n  = (size_t)(y1 - y0) * s->imageRowBytes;
  if(bytes < n)
  {
    setError("mgs_frame_copy_strip: destination too small");
    return MGS_ERR_INVALID_ARG;
  }
  HIPCHK(hipSetDevice(s->device));
  HIPCHK(hipMemcpyAsync(dst, s->image.p + (size_t)y0 * s->imageRowBytes, n, hipMemcpyDeviceToDevice, s->stream));
  return MGS_OK;
}

// ---- RCCL, resolved at first use ------------------------------------------------------------------------------------
namespace {
struct Rccl
{
  void* lib = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*)                                                             = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int)                                      = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t)                                                                = nullptr;
  ncclResult_t (*CommAbort)(ncclComm_t)                                                                  = nullptr;
  ncclResult_t (*GroupStart)()                                                                           = nullptr;
  ncclResult_t (*GroupEnd)()                                                                             = nullptr;
  ncclResult_t (*Broadcast)(const void*, void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t)    = nullptr;
  ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t)         = nullptr;
  const char* (*GetErrorString)(ncclResult_t)                                                            = nullptr;
  bool ok = false;
};
Rccl& rccl()
{
  static Rccl R = [] {
    Rccl r;
    // MGS_RCCL_LIB=path: load THIS library instead (and nothing else if it fails) — the seam of the test double that lets several
    // ranks share one GPU (tests/helpers/fake_rccl.cpp); never set in production, never a fallback
    if(const char* forced = std::getenv("MGS_RCCL_LIB"))
      r.lib = dlopen(forced, RTLD_NOW | RTLD_LOCAL);
    else
      for(const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"})
        if((r.lib = dlopen(name, RTLD_NOW | RTLD_LOCAL)))
          break;
    if(!r.lib)
      return r;
    auto sym = [&](const char* n) { return dlsym(r.lib, n); };
    r.GetUniqueId    = (decltype(r.GetUniqueId))sym("ncclGetUniqueId");
    r.CommInitRank   = (decltype(r.CommInitRank))sym("ncclCommInitRank");
    r.CommDestroy    = (decltype(r.CommDestroy))sym("ncclCommDestroy");
    r.CommAbort      = (decltype(r.CommAbort))sym("ncclCommAbort");
    r.GroupStart     = (decltype(r.GroupStart))sym("ncclGroupStart");
    r.GroupEnd       = (decltype(r.GroupEnd))sym("ncclGroupEnd");
    r.Broadcast      = (decltype(r.Broadcast))sym("ncclBroadcast");
    r.AllGather      = (decltype(r.AllGather))sym("ncclAllGather");
    r.GetErrorString = (decltype(r.GetErrorString))sym("ncclGetErrorString");
    r.ok = r.GetUniqueId && r.CommInitRank && r.CommDestroy && r.GroupStart && r.GroupEnd && r.Broadcast && r.AllGather;
    return r;
  }();
  return R;
}
int rcclFail(const char* what, ncclResult_t e)
{
  setError(std::string(what) + ": RCCL error " + std::to_string((int)e) + (rccl().GetErrorString ? std::string(" (") + rccl().GetErrorString(e) + ")" : ""));
  return MGS_ERR_DEVICE;
}
}  // namespace

int mgs_comm_unique_id(void* idOut)
{
  static_assert(sizeof(ncclUniqueId) == MGS_COMM_ID_BYTES, "ncclUniqueId is 128 bytes");
  if(!idOut)
  {
    setError("mgs_comm_unique_id: null argument");
    return MGS_ERR_INVALID_ARG;
  }
  if(!rccl().ok)
  {
    setError("mgs_comm_unique_id: librccl could not be loaded");
    return MGS_ERR_UNSUPPORTED;
  }
  ncclUniqueId id;
  const ncclResult_t e = rccl().GetUniqueId(&id);
  if(e != ncclSuccess)
    return rcclFail("ncclGetUniqueId", e);
  std::memcpy(idOut, &id, sizeof(id));
  return MGS_OK;
}

int mgs_scene_comm_destroy(MgsScene s)
{
  if(!s)
  {
    setError("mgs_scene_comm_destroy: null scene");
    return MGS_ERR_INVALID_ARG;
  }
  if(s->comm)
  {
    (void)hipSetDevice(s->device);
    (void)hipStreamSynchronize(s->stream);
    (void)rccl().CommDestroy(s->comm);
    s->comm = nullptr;
  }
  s->commRank  = 0;
  s->commWorld = 1;
  return MGS_OK;
}

int mgs_scene_comm_init(MgsScene s, int rank, int world, const void* id)
{
  if(!s || !id || world < 1 || rank < 0 || rank >= world)
  {
    setError("mgs_scene_comm_init: bad argument");
    return MGS_ERR_INVALID_ARG;
  }
  if(!rccl().ok)
  {
    setError("mgs_scene_comm_init: librccl could not be loaded");
    return MGS_ERR_UNSUPPORTED;
  }
  (void)mgs_scene_comm_destroy(s);
  HIPCHK(hipSetDevice(s->device));
  ncclUniqueId uid;
  std::memcpy(&uid, id, sizeof(uid));
  const ncclResult_t e = rccl().CommInitRank(&s->comm, world, uid, rank);
  if(e != ncclSuccess)
  {
    s->comm = nullptr;
    return rcclFail("ncclCommInitRank", e);
  }
  s->commRank  = rank;
  s->commWorld = world;
  s->stripBounds.clear();
  return MGS_OK;
}

int mgs_scene_set_strip_rows(MgsScene s, const int32_t* bounds, int count)
{
  if(!s)
  {
    setError("mgs_scene_set_strip_rows: null scene");
    return MGS_ERR_INVALID_ARG;
  }
  if(!bounds)
  {
    s->stripBounds.clear();
    return MGS_OK;
  }
  if(count != s->commWorld + 1 || bounds[0] != 0)
  {
    setError("mgs_scene_set_strip_rows: need world_size + 1 ascending tile-row bounds starting at 0");
    return MGS_ERR_INVALID_ARG;
  }
  for(int i = 0; i < count - 1; ++i)
    if(bounds[i + 1] < bounds[i])
    {
      setError("mgs_scene_set_strip_rows: bounds must be ascending");
      return MGS_ERR_INVALID_ARG;
    }
  s->stripBounds.assign(bounds, bounds + count);
  return MGS_OK;
}

static void stripOfRank(MgsScene s, int tilesY, int r, int& b, int& e)
{
  if(!s->stripBounds.empty())
  {
    b = std::min(s->stripBounds[r], tilesY);
    e = std::min(s->stripBounds[r + 1], tilesY);
    if(r == s->commWorld - 1)
      e = tilesY;  // the last strip takes whatever the table left over
    return;
  }
  const int per = (tilesY + s->commWorld - 1) / s->commWorld;
  b             = std::min(r * per, tilesY);
  e             = std::min(b + per, tilesY);
}

// the frame buffer of a rank that does not render this frame (empty strip) or whose render failed: the exchange still
// receives everybody else's rows into it
static int ensureFrameBufferFor(MgsScene s, const MgsFrameParams* p)
{
  if(p->width <= 0 || p->height <= 0 || p->target_format < MGS_TARGET_RGBA16F || p->target_format > MGS_TARGET_RGBA8)
  {
    setError("frame: bad size / target_format");
    return MGS_ERR_INVALID_ARG;
  }
  const size_t pixB = p->target_format == MGS_TARGET_RGBA16F ? 8 : (p->target_format == MGS_TARGET_RGBA8 ? 4 : 16);
  s->imageRowBytes  = (size_t)p->width * pixB;
  s->imageBytes     = s->imageRowBytes * (size_t)p->height;
  if(s->image.n < s->imageBytes)
  {
    HIPCHK(hipStreamSynchronize(s->stream));
    for(auto& g : s->graphs)  // captured frames hold the old image pointer
      (void)hipGraphExecDestroy(g.second);
    s->graphs.clear();
    int rc = s->image.ensure(s->imageBytes);
    if(rc != MGS_OK)
      return rc;
    HIPCHK(hipMemsetAsync(s->image.p, 0, s->imageBytes, s->stream));
  }
  return MGS_OK;
}

static int mgs_render_gathered_impl(MgsScene s, const MgsFrameParams* p, MgsFrameOut* out)
{
  if(!s || !p)
  {
    setError("mgs_render_gathered: null argument");
    return MGS_ERR_INVALID_ARG;
  }
  if(!s->comm)
  {
    setError("mgs_render_gathered: call mgs_scene_comm_init first");
    return MGS_ERR_STATE;
  }
  const int tilesY = (p->height + kTilePx - 1) / kTilePx;
  MgsFrameParams q = *p;
  int b, e;
  stripOfRank(s, tilesY, s->commRank, b, e);
  // A rank must reach the exchange whatever happens to its own strip: the peers block in the collective otherwise.  A failed
  // (or absent) local render still joins with a frame buffer of the right size — its rows are stale, the error is returned
  // after the exchange.  Only if no buffer can be had at all is the communicator aborted, which fails the peers' collective
  // instead of hanging it.
  int         localRc = MGS_OK;
  std::string localErr;
  if(e > b)
  {
    q.strip_row_begin = b;
    q.strip_row_end   = e;
    localRc           = mgs_render(s, &q, out);
  }
  else if(out)
    std::memset(out, 0, sizeof(*out));  // more ranks than tile rows (or an empty strip in the table): nothing to render
  if(localRc != MGS_OK)
    localErr = lastError();
  HIPCHK(hipSetDevice(s->device));
  const int bufRc = ensureFrameBufferFor(s, p);
  if(bufRc != MGS_OK)
  {
    if(rccl().CommAbort)
      (void)rccl().CommAbort(s->comm);
    else
      (void)rccl().CommDestroy(s->comm);
    s->comm      = nullptr;
    s->commWorld = 1;
    s->commRank  = 0;
    return localRc != MGS_OK ? localRc : bufRc;
  }
  // exchange in place: rank r's rows are broadcast from r into the same rows of everybody's frame buffer.  One group
  // = one fused launch on the render stream; it overlaps with the next frame's key/sort when frames are in flight.
  ncclResult_t ne = rccl().GroupStart();
  if(ne != ncclSuccess)
    return rcclFail("ncclGroupStart", ne);
  for(int r = 0; r < s->commWorld; ++r)
  {
    int rb, re;
    stripOfRank(s, tilesY, r, rb, re);
    const int    y0 = rb * kTilePx, y1 = std::min(re * kTilePx, p->height);
    if(y1 <= y0)
      continue;
    uint8_t*     ptr = s->image.p + (size_t)y0 * s->imageRowBytes;
    const size_t n   = (size_t)(y1 - y0) * s->imageRowBytes;
    ne               = rccl().Broadcast(ptr, ptr, n, ncclUint8, r, s->comm, s->stream);
    if(ne != ncclSuccess)
    {
      (void)rccl().GroupEnd();
      return rcclFail("ncclBroadcast", ne);
    }
  }
  ne = rccl().GroupEnd();
  if(ne != ncclSuccess)
    return rcclFail("ncclGroupEnd", ne);
  if(localRc != MGS_OK)
  {
    setError("mgs_render_gathered: this rank's strip failed (" + localErr + "); the exchange was still joined");
    return localRc;
  }
  // the frame is whole again: downloads address all rows.  The bin lists of the last frame cover this rank's rows only.
  s->lastParams                 = q;
  s->lastParams.strip_row_begin = 0;
  s->lastParams.strip_row_end   = tilesY;
  s->haveFrame                  = true;
  s->lastWasSortOnly            = false;
  s->lastListsPartial           = s->commWorld > 1;
  return MGS_OK;
}
int mgs_render_gathered(MgsScene s, const MgsFrameParams* p, MgsFrameOut* out)
{
  return guarded("mgs_render_gathered", [&] { return mgs_render_gathered_impl(s, p, out); });
}

int mgs_frame_row_costs(MgsScene s, uint32_t* cost, size_t rows)
{
  if(!s || !cost)
  {
    setError("mgs_frame_row_costs: null argument");
    return MGS_ERR_INVALID_ARG;
  }
  if(!s->haveFrame || s->lastWasSortOnly)
  {
    setError("mgs_frame_row_costs: no frame rendered yet");
    return MGS_ERR_STATE;
  }
  if(s->lastListsPartial)
  {  // every rank would see its own rows only and derive a different table: the exchange sizes would no longer agree
    setError("mgs_frame_row_costs: the last frame came from mgs_render_gathered (bin lists of this rank's strip only); "
             "calibrate on a full-frame mgs_render");
    return MGS_ERR_STATE;
  }
  return guarded("mgs_frame_row_costs", [&]() -> int {
    FrameArgs A;
    int       rc = buildFrameArgs(s, &s->lastParams, A);
    if(rc != MGS_OK)
      return rc;
    FrameConst& F = A.f;
    // the lists are those of the LAST frame: its bin grid, not the one the adaptive policy would pick for the next frame
    F.binShiftX = s->lastBinShift[0];
    F.binShiftY = s->lastBinShift[1];
    F.binsX     = (F.tilesX + (1 << F.binShiftX) - 1) >> F.binShiftX;
    F.binsY     = (F.tilesY + (1 << F.binShiftY) - 1) >> F.binShiftY;
    if(rows < (size_t)F.tilesY)
    {
      setError("mgs_frame_row_costs: need one entry per 16-pixel tile row");
      return MGS_ERR_INVALID_ARG;
    }
    HIPCHK(hipSetDevice(s->device));
    HIPCHK(hipStreamSynchronize(s->stream));
    const int          bins = F.binsX * F.binsY;
    std::vector<uint2> rg((size_t)bins);
    HIPCHK(hipMemcpy(rg.data(), s->ranges.p, (size_t)bins * sizeof(uint2), hipMemcpyDeviceToHost));
    std::vector<double> acc((size_t)F.tilesY, 0.0);
    const int           rowsPerBin = 1 << F.binShiftY;
    for(int by = 0; by < F.binsY; ++by)
    {
      double len = 0;
      for(int bx = 0; bx < F.binsX; ++bx)
        len += (double)(rg[(size_t)by * F.binsX + bx].y - rg[(size_t)by * F.binsX + bx].x);
      const int r0 = by * rowsPerBin, r1 = std::min(r0 + rowsPerBin, F.tilesY);
      for(int r = r0; r < r1; ++r)
        acc[(size_t)r] += len / (double)(r1 - r0);
    }
    for(int r = 0; r < F.tilesY; ++r)
      cost[r] = (uint32_t)std::min(acc[(size_t)r], 4.0e9);
    for(size_t r = (size_t)F.tilesY; r < rows; ++r)
      cost[r] = 0u;
    return MGS_OK;
  });
}

// test/debug hook: the projected records of the last full frame for the given global ids (caller's id space)
static int mgs_frame_download_projected_impl(MgsScene s, const uint32_t* ids, size_t count, float* out10, uint32_t* rectOut)
{
  if(!s || !ids || !out10)
  {
    setError("mgs_frame_download_projected: null argument");
    return MGS_ERR_INVALID_ARG;
  }
  if(!s->haveFrame || s->lastWasSortOnly || s->lastParams.pipeline != MGS_PIPELINE_3DGS)
  {
    setError("mgs_frame_download_projected: no 3DGS frame rendered yet");
    return MGS_ERR_STATE;
  }
  for(size_t i = 0; i < count; ++i)
    if(ids[i] >= s->d->totalSplats)
    {
      setError("mgs_frame_download_projected: id out of range");
      return MGS_ERR_INVALID_ARG;
    }
  HIPCHK(hipSetDevice(s->device));
  HIPCHK(hipStreamSynchronize(s->stream));
  std::vector<uint32_t> sid(ids, ids + count);
  mapIdsToStorage(s, sid.data(), count);
  std::vector<SplatRec> rec(s->d->totalSplats);
  std::vector<uint32_t> rect(s->d->totalSplats);
  HIPCHK(hipMemcpy(rec.data(), s->rec.p, (size_t)s->d->totalSplats * sizeof(SplatRec), hipMemcpyDeviceToHost));
  HIPCHK(hipMemcpy(rect.data(), s->rect.p, (size_t)s->d->totalSplats * 4, hipMemcpyDeviceToHost));
  if(rectOut && s->lastRide[0] != 0 && s->lastRide[4] != 0)
  {  // the rectangles rode through the key sort as codes (kernels_common.h: rideEncode): rect[id] holds the escapes only, the
     // others are rebuilt from the code at the id's sorted position (they used to come back stale)
    FrameCounters c;
    HIPCHK(hipMemcpy(&c, s->ctr.p, sizeof(c), hipMemcpyDeviceToHost));
    const uint32_t        n = std::min<uint32_t>(c.sortedCount, (uint32_t)s->d->totalSplats);
    std::vector<uint32_t> sortedIds(n);
    std::vector<uint16_t> codes(n);
    if(n)
    {
      HIPCHK(hipMemcpy(sortedIds.data(), s->idsA.p, (size_t)n * 4, hipMemcpyDeviceToHost));
      HIPCHK(hipMemcpy(codes.data(), s->sortedCode16.p, (size_t)n * 2, hipMemcpyDeviceToHost));
    }
    const int      bx = s->lastRide[2], by = s->lastRide[3], nb = bx * by;
    const uint32_t escape = (1u << s->lastRide[1]) - 1u;
    const uint32_t base[4] = {0u, (uint32_t)nb, (uint32_t)(nb + (bx - 1) * by), (uint32_t)(nb + (bx - 1) * by + bx * (by - 1))};
    for(uint32_t e = 0; e < n; ++e)
    {
      const uint32_t code = codes[e];
      if(code == escape || sortedIds[e] >= rect.size())
        continue;
      const uint32_t shape = code >= base[3] ? 3u : (code >= base[2] ? 2u : (code >= base[1] ? 1u : 0u));
      const uint32_t r = code - base[shape], dx = shape & 1u, dy = shape >> 1, wS = (uint32_t)bx - dx;
      const uint32_t y0 = r / wS, x0 = r - y0 * wS;
      rect[sortedIds[e]] = x0 | (y0 << 8) | ((x0 + dx) << 16) | ((y0 + dy) << 24);
    }
  }
  for(size_t i = 0; i < count; ++i)
  {
    const SplatRec& r = rec[sid[i]];
    float*          o = out10 + 10 * i;
    // p = 2 b / |b|^2  =>  b = 2 p / |p|^2
    const double n1 = (double)r.p1x * r.p1x + (double)r.p1y * r.p1y, n2 = (double)r.p2x * r.p2x + (double)r.p2y * r.p2y;
    o[0] = r.cx;
    o[1] = r.cy;
    o[2] = (float)(2.0 * r.p1x / n1);
    o[3] = (float)(2.0 * r.p1y / n1);
    o[4] = (float)(2.0 * r.p2x / n2);
    o[5] = (float)(2.0 * r.p2y / n2);
    o[6] = r.a;
    uint16_t h[2];
    std::memcpy(h, &r.exey, 4);
    o[7] = halfToFloat(h[0]);
    o[8] = halfToFloat(h[1]);
    o[9] = 0.0f;
    if(rectOut)
      rectOut[i] = rect[sid[i]];
  }
  return MGS_OK;
}
int mgs_frame_download_projected(MgsScene s, const uint32_t* ids, size_t count, float* out10, uint32_t* rectOut)
{
  return guarded("mgs_frame_download_projected", [&] { return mgs_frame_download_projected_impl(s, ids, count, out10, rectOut); });
}

int mgs_sync(MgsScene s)
{
  if(!s)
  {
    setError("mgs_sync: null scene");
    return MGS_ERR_INVALID_ARG;
  }
  HIPCHK(hipSetDevice(s->device));
  HIPCHK(hipStreamSynchronize(s->stream));
  return MGS_OK;
}

// ------------------------------------------------------------------------------------------------
int mgs_sort_keys(MgsScene s, const MgsFrameParams* p, MgsSortOut* out)
{
  if(!s || !p || !out)
  {
    setError("mgs_sort_keys: null argument");
    return MGS_ERR_INVALID_ARG;
  }
  if(!s->d->committed)
  {
    setError("mgs_sort_keys: call mgs_scene_commit first");
    return MGS_ERR_STATE;
  }
  if(int wrc = ensureWorkingSet(s))
    return wrc;
  HIPCHK(hipSetDevice(s->device));
  std::memset(out, 0, sizeof(*out));
  if(p->sort_mode == MGS_SORT_CPU_ASYNC)
  {
    int rc = cpuSortStep(s, p, true);
    if(rc != MGS_OK)
      return rc;
    out->count   = (uint32_t)s->cpuIndices.size();
    out->key_ms  = s->lastSort.key_ms;
    out->sort_ms = s->lastSort.sort_ms;
    s->lastSort  = *out;
    s->lastWasSortOnly = true;
    s->haveFrame       = true;
    s->lastParams      = *p;
    return MGS_OK;
  }
  FrameArgs A;
  int       rc = buildFrameArgs(s, p, A);
  if(rc != MGS_OK)
    return rc;
  hipStream_t st = s->stream;
  // the metric hook returns dist.comp.slang's stream for the whole frame: a strip set on the scene culls footprints, which is
  // the raster stage's business, so it does not apply here
  A.f.stripRow0 = 0;
  A.f.stripRow1 = A.f.tilesY;
  if((rc = s->ranges.ensure(1))) return rc;
  if((rc = uploadFrameState(s, A, st))) return rc;
  HIPCHK(hipEventRecord(s->ev[0], st));
  launchProject(st, A, s->dArgs.p, false, s->ctr.p, s->pairB.p, s->slotCount.p, s->rec.p, s->rect.p,
                s->slotHist2.p, s->top16Rec.p, s->top16Count.p, &s->plans.p->os, kPrjOrder ? s->prjOrder.p : nullptr);
  HIPCHK(hipEventRecord(s->ev[1], st));
  if((rc = s->keysA.ensure(s->d->totalSplats))) return rc;  // the hook returns the sorted keys too
  keySort(s, st, true, kRemap);
  HIPCHK(hipEventRecord(s->ev[2], st));
  HIPCHK(hipMemcpyAsync(s->hCtr, s->ctr.p, sizeof(FrameCounters), hipMemcpyDeviceToHost, st));
  HIPCHK(hipMemcpyAsync(s->hPlans, s->plans.p, sizeof(FramePlans), hipMemcpyDeviceToHost, st));
  HIPCHK(hipStreamSynchronize(st));
  HIPCHK(hipGetLastError());
  float ms = 0;
  HIPCHK(hipEventElapsedTime(&ms, s->ev[0], s->ev[1]));
  out->key_ms = ms;
  HIPCHK(hipEventElapsedTime(&ms, s->ev[1], s->ev[2]));
  out->sort_ms = ms;
  if(s->hCtr->errorFlags & kErrSpinTimeout)
  {
    setError("mgs_sort_keys: a look-back wait of the key sort gave up (kErrSpinTimeout); the order is invalid");
    return MGS_ERR_DEVICE;
  }
  out->count   = s->hCtr->sortedCount;
  out->passes  = s->hPlans->keys.passesRun;
  out->reserved[0] = s->hPlans->os.remapOn;     // pass 2 sorted on the rank of key >> 16
  out->reserved[1] = s->hPlans->os.remapCount;  // occurring values of key >> 16
  s->lastSort  = *out;
  s->lastWasSortOnly = true;
  s->haveFrame       = true;
  s->lastParams      = *p;
  return MGS_OK;
}

int mgs_sort_download(MgsScene s, uint32_t* keys, uint32_t* ids, uint32_t capacity)
{
  if(!s || !ids)
  {
    setError("mgs_sort_download: null argument");
    return MGS_ERR_INVALID_ARG;
  }
  if(!s->haveFrame)
  {
    setError("mgs_sort_download: nothing sorted yet");
    return MGS_ERR_STATE;
  }
  HIPCHK(hipSetDevice(s->device));
  if(s->lastParams.sort_mode == MGS_SORT_CPU_ASYNC)
  {
    const size_t n = s->cpuIndices.size();
    if(capacity < n)
    {
      setError("mgs_sort_download: capacity too small");
      return MGS_ERR_INVALID_ARG;
    }
    std::memcpy(ids, s->cpuIndices.data(), n * 4);
    if(keys)  // the snapshot taken together with the indices (the worker may already be rewriting its own array)
      for(size_t i = 0; i < n; ++i)
      {
        const uint32_t g = s->cpuIndices[i];
        const float    d = g < s->cpuDistances.size() ? s->cpuDistances[g] : 0.0f;
        std::memcpy(&keys[i], &d, 4);
      }
    return MGS_OK;
  }
  if(!s->lastWasSortOnly)
  {  // after a full frame the counters are still on the device
    HIPCHK(hipMemcpyAsync(s->hCtr, s->ctr.p, sizeof(FrameCounters), hipMemcpyDeviceToHost, s->stream));
    HIPCHK(hipMemcpyAsync(s->hPlans, s->plans.p, sizeof(FramePlans), hipMemcpyDeviceToHost, s->stream));
  }
  HIPCHK(hipStreamSynchronize(s->stream));
  const uint32_t n = s->hCtr->sortedCount;
  if(capacity < n)
  {
    setError("mgs_sort_download: capacity too small");
    return MGS_ERR_INVALID_ARG;
  }
  if(n)
  {
    if(keys)
    {
      if(!s->lastWasSortOnly || s->keysA.n < n)
      {
        setError("mgs_sort_download: the sorted keys exist after mgs_sort_keys only (a frame's last sort pass writes the ids alone)");
        return MGS_ERR_STATE;
      }
      HIPCHK(hipMemcpy(keys, s->keysA.p, (size_t)n * 4, hipMemcpyDeviceToHost));
    }
    HIPCHK(hipMemcpy(ids, s->idsA.p, (size_t)n * 4, hipMemcpyDeviceToHost));
    mapIdsToCaller(s, ids, n);  // the pipeline works on storage ids
  }
  return MGS_OK;
}

int mgs_radix_sort_u32(MgsScene s, void* keysDev, void* valsDev, uint32_t count, int beginBit, int endBit, float* ms)
{
  if(!s || !keysDev || !valsDev || beginBit < 0 || endBit > 32 || beginBit >= endBit)
  {
    setError("mgs_radix_sort_u32: bad argument");
    return MGS_ERR_INVALID_ARG;
  }
  HIPCHK(hipSetDevice(s->device));
  if(count == 0)
    return MGS_OK;
  // scratch of the stand-alone sort: owned by the scene (its device, its stream), released in mgs_scene_destroy
  DevBuf<uint32_t>&kX = s->rsKeys, &vX = s->rsVals, &hist = s->rsHist, &nDev = s->rsCount;
  DevBuf<SortPlan>& plan = s->rsPlan;
  int rc;
  const uint32_t parts = (count + kPart - 1) / kPart;
  if((rc = kX.ensure(count))) return rc;
  if((rc = vX.ensure(count))) return rc;
  if((rc = hist.ensure(256ull * parts))) return rc;
  if((rc = nDev.ensure(1))) return rc;
  if((rc = plan.ensure(1))) return rc;
  // a full-width sort runs on the frame key sort's kernels (k_osort.hip, uniform input, four plain passes): the battery of
  // the stand-alone sort tests exercises exactly what the frame uses; partial bit ranges take the generic sort (k_sort.hip).
  // MGS_RAW_SORT=generic forces the generic one for every range (A/B).
  static const bool kRawGeneric = [] { const char* e = std::getenv("MGS_RAW_SORT"); return e && std::strcmp(e, "generic") == 0; }();
  // (2^30 pairs or more: the look-back words of k_os_pass hold 30-bit prefixes — the generic sort has no such limit)
  const bool os = !kRawGeneric && beginBit == 0 && endBit == 32 && (uint64_t)count < kOsMaxPairs;
  if(os)
  {
    if((rc = s->rsPairA.ensure(count))) return rc;
    if((rc = s->rsPairB.ensure(count))) return rc;
    if((rc = s->rsOsPlan.ensure(1))) return rc;
    // the passes keep the three sets of look-back words zeroed for each other (k_osort.hip); the sets' offsets depend on
    // the count, so a sort of another size starts from freshly zeroed words
    const size_t words = 3u * osSortStatusWords(osSortMaxParts(count));
    if(s->rsStatus.n < words || s->rsStatusParts != osSortMaxParts(count))
    {
      if((rc = s->rsStatus.ensure(words))) return rc;
      HIPCHK(hipMemset(s->rsStatus.p, 0, s->rsStatus.n * 4u));
      HIPCHK(hipDeviceSynchronize());
      s->rsStatusParts = osSortMaxParts(count);
    }
  }
  hipStream_t st = s->stream;
  HIPCHK(hipMemcpyAsync(nDev.p, &count, 4, hipMemcpyHostToDevice, st));
  HIPCHK(hipStreamSynchronize(st));
  HIPCHK(hipEventRecord(s->ev[6], st));
  launchSortClearPlan(st, plan.p);
  if(os)
  {
    launchOsSortClearPlan(st, s->rsOsPlan.p);
    OsLaunch O{};
    O.keys0    = (const uint32_t*)keysDev;
    O.vals0    = (const uint32_t*)valsDev;
    O.nPtr     = nDev.p;
    O.maxElems = count;
    O.pairA    = s->rsPairA.p;
    O.pairB    = s->rsPairB.p;
    O.outKeys  = kX.p;
    O.outVals  = vX.p;
    O.plan     = s->rsOsPlan.p;
    O.planOut  = plan.p;  // finalSel = 0: the result is in X
    if((rc = ensureFrameState(s))) return rc;
    O.ctr      = s->ctr.p;
    O.status   = s->rsStatus.p;
    O.allowRemap = false;
    HIPCHK(hipMemsetAsync(&s->ctr.p->errorFlags, 0, sizeof(uint32_t), st));  // whatever an earlier frame left there is not this sort's
    launchOsSort(st, O);
  }
  else
  {
    SortLaunch L{};
    L.keys0 = (uint32_t*)keysDev;
    L.vals0 = (uint32_t*)valsDev;
    L.keysX = kX.p;
    L.valsX = vX.p;
    L.keysY = (uint32_t*)keysDev;
    L.valsY = (uint32_t*)valsDev;
    L.nPtr     = nDev.p;
    L.plan     = plan.p;
    L.partHist = hist.p;
    L.pStride  = parts;
    L.maxElems = count;
    L.beginBit = beginBit;
    L.endBit   = endBit;
    launchRadixSort(st, L);
  }
  HIPCHK(hipEventRecord(s->ev[7], st));
  SortPlan hp;
  uint32_t sortFlags = 0;
  HIPCHK(hipMemcpyAsync(&hp, plan.p, sizeof(SortPlan), hipMemcpyDeviceToHost, st));
  if(os)
    HIPCHK(hipMemcpyAsync(&sortFlags, &s->ctr.p->errorFlags, sizeof(uint32_t), hipMemcpyDeviceToHost, st));
  HIPCHK(hipStreamSynchronize(st));
  if(sortFlags & kErrSpinTimeout)
  {
    setError("mgs_radix_sort_u32: a look-back wait gave up (kErrSpinTimeout); the caller's arrays are untouched");
    return MGS_ERR_DEVICE;
  }
  if(ms)
    HIPCHK(hipEventElapsedTime(ms, s->ev[6], s->ev[7]));
  if(hp.finalSel == 0)
  {  // result is in X: bring it home (outside the timed region)
    HIPCHK(hipMemcpyAsync(keysDev, kX.p, (size_t)count * 4, hipMemcpyDeviceToDevice, st));
    HIPCHK(hipMemcpyAsync(valsDev, vX.p, (size_t)count * 4, hipMemcpyDeviceToDevice, st));
    HIPCHK(hipStreamSynchronize(st));
  }
  HIPCHK(hipGetLastError());
  return MGS_OK;
}

int mgs_radix_sort_host(MgsScene s, uint32_t* keys, uint32_t* vals, uint32_t count, int beginBit, int endBit, float* ms)
{
  if(!s || !keys || !vals)
  {
    setError("mgs_radix_sort_host: null argument");
    return MGS_ERR_INVALID_ARG;
  }
  if(count == 0)
    return MGS_OK;
  HIPCHK(hipSetDevice(s->device));
  DevBuf<uint32_t> dk, dv;  // released on every exit path
  int              rc = dk.ensure(count);
  if(rc == MGS_OK)
    rc = dv.ensure(count);
  auto copy = [&](void* dst, const void* src, hipMemcpyKind kind) {
    if(rc == MGS_OK && hipMemcpy(dst, src, (size_t)count * 4, kind) != hipSuccess)
    {
      setError("mgs_radix_sort_host: hipMemcpy failed");
      rc = MGS_ERR_DEVICE;
    }
  };
  copy(dk.p, keys, hipMemcpyHostToDevice);
  copy(dv.p, vals, hipMemcpyHostToDevice);
  if(rc == MGS_OK)
    rc = mgs_radix_sort_u32(s, dk.p, dv.p, count, beginBit, endBit, ms);
  copy(keys, dk.p, hipMemcpyDeviceToHost);
  copy(vals, dv.p, hipMemcpyDeviceToHost);
  dk.release();
  dv.release();
  return rc;
}

// ------------------------------------------------------------------------------------------------
// build-defined camera helper (nvutils::CameraManipulator is absent)
void mgs_camera_lookat_perspective(const float eye[3], const float center[3], const float up[3], float fovDeg, float zn,
                                   float zf, int width, int height, int flipY, float view[16], float proj[16])
{
  auto norm3 = [](float v[3]) {
    const float l = std::sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
    v[0] /= l; v[1] /= l; v[2] /= l;
  };
  float f[3] = {center[0] - eye[0], center[1] - eye[1], center[2] - eye[2]};
  norm3(f);
  float sx[3] = {f[1] * up[2] - f[2] * up[1], f[2] * up[0] - f[0] * up[2], f[0] * up[1] - f[1] * up[0]};
  norm3(sx);
  const float u[3] = {sx[1] * f[2] - sx[2] * f[1], sx[2] * f[0] - sx[0] * f[2], sx[0] * f[1] - sx[1] * f[0]};
  // right-handed lookAt, column-major
  view[0] = sx[0]; view[4] = sx[1]; view[8]  = sx[2]; view[12] = -(sx[0] * eye[0] + sx[1] * eye[1] + sx[2] * eye[2]);
  view[1] = u[0];  view[5] = u[1];  view[9]  = u[2];  view[13] = -(u[0] * eye[0] + u[1] * eye[1] + u[2] * eye[2]);
  view[2] = -f[0]; view[6] = -f[1]; view[10] = -f[2]; view[14] = (f[0] * eye[0] + f[1] * eye[1] + f[2] * eye[2]);
  view[3] = 0; view[7] = 0; view[11] = 0; view[15] = 1;
  // right-handed perspective, clip z in [0,1]
  const float aspect = (float)width / (float)height;
  const float t      = std::tan(fovDeg * 0.017453292519943295f * 0.5f);
  std::memset(proj, 0, sizeof(float) * 16);
  proj[0]  = 1.0f / (aspect * t);
  proj[5]  = (flipY ? -1.0f : 1.0f) / t;
  proj[10] = zf / (zn - zf);
  proj[11] = -1.0f;
  proj[14] = -(zf * zn) / (zf - zn);
}

// T*R*S with R from Euler angles (degrees) through a quaternion, computeTransform (src/utilities.h:170-199)
void mgs_compute_transform(const float scale[3], const float rotDeg[3], const float tr[3], float M[16], float Minv[16])
{
  const float d2r = 0.017453292519943295f;
  const float hx = rotDeg[0] * d2r * 0.5f, hy = rotDeg[1] * d2r * 0.5f, hz = rotDeg[2] * d2r * 0.5f;
  const float cx = std::cos(hx), sx = std::sin(hx), cy = std::cos(hy), sy = std::sin(hy), cz = std::cos(hz),
              sz = std::sin(hz);
  // [glm] quat(eulerAngles): w = cx*cy*cz + sx*sy*sz, x = sx*cy*cz - cx*sy*sz, y = cx*sy*cz + sx*cy*sz, z = cx*cy*sz - sx*sy*cz
  const float w = cx * cy * cz + sx * sy * sz, x = sx * cy * cz - cx * sy * sz, y = cx * sy * cz + sx * cy * sz,
              z = cx * cy * sz - sx * sy * cz;
  const float R[9] = {1 - 2 * (y * y + z * z), 2 * (x * y + w * z),     2 * (x * z - w * y),
                      2 * (x * y - w * z),     1 - 2 * (x * x + z * z), 2 * (y * z + w * x),
                      2 * (x * z + w * y),     2 * (y * z - w * x),     1 - 2 * (x * x + y * y)};  // columns
  for(int c = 0; c < 3; ++c)
  {
    for(int r = 0; r < 3; ++r)
      M[c * 4 + r] = R[c * 3 + r] * scale[c];
    M[c * 4 + 3] = 0.f;
  }
  M[12] = tr[0]; M[13] = tr[1]; M[14] = tr[2]; M[15] = 1.f;
  if(Minv)
    mat4Inverse(M, Minv);
}

}  // extern "C"
