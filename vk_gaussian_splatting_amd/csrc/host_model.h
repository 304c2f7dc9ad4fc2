// host_model.h — RAM-side splat model and helpers shared by the C ABI implementation.
// Mirrors struct SplatSet of the reference (src/splat_set.h:33-48): INRIA SoA arrays in RUB.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

namespace mgs {

struct HostSplatSet
{
  std::vector<float> positions;  // 3n
  std::vector<float> f_dc;       // 3n
  std::vector<float> f_rest;     // (3*coeffsPerChannel) n, channel-major per splat
  std::vector<float> opacity;    // n (logit)
  std::vector<float> scale;      // 3n (log)
  std::vector<float> rotation;   // 4n (w,x,y,z)
  std::string        path;

  size_t   size() const { return positions.size() / 3; }
  uint32_t fRestPerSplat() const { return size() ? uint32_t(f_rest.size() / size()) : 0u; }
  int      maxShDegree() const;                 // src/splat_set.h:52-74
  void     convertRdfToRub();                   // src/splat_set.h:78-114 with (RDF -> RUB)
};

// thread-local error message used by the C ABI
void        setError(const std::string& msg);
const char* lastError();

// loaders: return 0 or a negative MgsStatus
int loadPly(const std::string& path, HostSplatSet& out);    // src/ply_loader_async.cpp:357-441
int loadSpz(const std::string& path, HostSplatSet& out);    // src/ply_loader_async.cpp:304-353 + 3rdparty/spz
int loadSplat(const std::string& path, HostSplatSet& out);  // src/ply_loader_async.cpp:43-183

// 4x4 helpers, glm column-major memory (m[col*4+row]); plain unfused fp32 like glm on the host
void mat4Mul(const float a[16], const float b[16], float out[16]);
void mat4Inverse(const float m[16], float out[16]);
void mat4MulVec4(const float m[16], const float v[4], float out[4]);

// upload transform of SplatSetVk::initDataBuffers (src/splat_set_vk.cpp:263-435), host side, threaded
void buildCov6(const HostSplatSet& s, std::vector<float>& cov6);
void buildRgba(const HostSplatSet& s, std::vector<float>& rgba);
int  shStride(uint32_t fRestPerSplat);
void buildShInterleaved(const HostSplatSet& s, std::vector<float>& sh);
uint16_t floatToHalf(float f);
float    halfToFloat(uint16_t h);
uint8_t  toUint8(float v, float lo, float hi);  // src/splat_set_vk.cpp:85-89

// spatially coherent storage order: permutation newToOld sorted by the 63-bit Morton code of the centre
// `radius` (optional, one per splat): splats are first split into 16 size classes (octaves of the radius
// around the median), then Morton-sorted inside each class, so that a partition's footprint bound is not
// set by a few outliers.
void mortonOrder(const HostSplatSet& s, const std::vector<float>* radius, std::vector<uint32_t>& newToOld);

// parallel-for over [0,n) in batches of 8192 (START_PAR_LOOP, src/utilities.h:52-59)
template <typename F>
void parallelBatches(size_t n, F&& fn);

}  // namespace mgs

#include <algorithm>
#include <thread>
namespace mgs {
template <typename F>
void parallelBatches(size_t n, F&& fn)
{
  const size_t batch   = 8192;
  const size_t nBatch  = (n + batch - 1) / batch;
  unsigned     threads = std::max(1u, std::thread::hardware_concurrency());
  threads              = (unsigned)std::min<size_t>(threads, std::max<size_t>(1, nBatch));
  if(threads > 32)
    threads = 32;
  auto worker = [&](unsigned tid) {
    for(size_t b = tid; b < nBatch; b += threads)
    {
      const size_t lo = b * batch, hi = std::min(n, lo + batch);
      for(size_t i = lo; i < hi; ++i)
        fn(i);
    }
  };
  std::vector<std::thread> pool;
  for(unsigned t = 1; t < threads; ++t)
    pool.emplace_back(worker, t);
  worker(0);
  for(auto& th : pool)
    th.join();
}
}  // namespace mgs
