// device_types.h — structs shared between the host orchestration and the gfx950 kernels.
#pragma once
#include <cstdint>

namespace mgs {

constexpr int kMaxInstances       = 256;  // instances per scene: FrameArgs lives in device memory, only the used part is uploaded
constexpr int kMaxInlineInstances = 16;   // SH-table entries the compositor carries by value; larger scenes read a device table
constexpr int kTilePx             = 16;  // compositing tile edge in pixels (one workgroup; 8x8 pixels per wave)

// error bits reported through MgsFrameOut.error_flags
enum : uint32_t {
  kErrPairOverflow  = 1u << 0,  // tile-pair capacity exceeded
  kErrSpinTimeout   = 1u << 1,  // a bounded look-back spin gave up (never hangs the GPU)
  kErrSortOverflow  = 1u << 2,
};

// per-instance constants (SplatSetDesc, shaders/shaderio.h:439-479, reduced to what the path reads)
struct InstanceConst
{
  const float* centers;  // [count*3]
  const float* cov6;     // planar: float4 covA[count] = (S00,S01,S02,S11) then float2 covB[count] = (S12,S22)
  const void*  rgba;     // [count*4] fp32 | fp16 | u8 as stored
  const float* rgbaF32;  // [count*4] the same, dequantised the way fetchColor does (== rgba for fp32 storage): what the
                         // compositor's shading phase reads for the records it stages
  const float* alpha;    // [count] opacity as the shaders read it back from `rgba` (dequantised), planar: all the
                         // projection needs of the colour
  const float* maxScale; // [count] max(exp(scale)) per splat, host-computed (only read by size culling)
  const float* partBox;  // per 2048-splat partition: min xyz, max xyz (model space), rmax (sqrt(8*trace(Sigma))), pad
  const void*  sh;       // [count] records of 48 elements, [coef][rgb] + padding (192 B fp32 / 96 B fp16 / 48 B uint8)
  const float* scales;    // [count*3] log-space, as stored (scalesAddress)     } read only by the integrated-normal
  const float* rotations; // [count*4] (w,x,y,z), as stored (rotationsAddress)  } side output (mesh.slang:209-235)
  float        model[16];      // M   (glm column-major)
  float        modelInv[16];   // M^-1 (3DGUT: model-space ray of a fragment, frag.slang:113-118)
  float        modelView[16];  // V*M (host-computed with the same unfused fp32 products the shader does per thread)
  float        camModel[3];    // M^-1 * cameraPosition
  float        modelAxisMax;   // max length of the model matrix columns (size culling, dist.comp.slang:110-115)
  float        modelScale;     // largest singular value of the model 3x3 (upper bound of any length stretch)
  uint32_t     count;
  uint32_t     modelIsIdentity; // M is bitwise the identity: M*p == p + 0.0f for finite p (k_project skips the product)
  uint32_t     modelIsAffine;   // M's last row is bitwise (+0, +0, +0, 1) and its entries are < 2^24 in magnitude (kernels_common.h: mulMat4ExactAffineW1)
  uint32_t     globalOffset;   // first global splat id of this instance
  uint32_t     blockBegin;     // first project-kernel partition of this instance
  int32_t      shDegree;       // of the splat set
  int32_t      shStride;       // stored elements per splat (logical 0/9/24/45 padded to a 16-byte multiple)
};

// frame constants (shaderio::FrameInfo, shaders/shaderio.h:238-317, reduced)
struct FrameConst
{
  float    view[16];
  float    proj[16];
  float    focal[2];      // (P00*W/2, P11*H/2), src/gaussian_splatting.cpp:1248-1250
  int32_t  width, height;
  int32_t  tilesX, tilesY;
  int32_t  binShiftX, binShiftY;  // a bin = (1<<shift) x (1<<shift) tiles; lists are built per bin
  int32_t  binsX, binsY;
  int32_t  stripRow0, stripRow1;  // tile rows rendered by this device
  float    splatScale, frustumDilation, alphaCull;
  int32_t  shDegree;
  int32_t  frontToBack;   // key sign
  int32_t  cullMode;
  int32_t  msAA;
  int32_t  alphaMode;
  int32_t  targetFormat;
  int32_t  nInstances;
  uint32_t totalSplats;
  uint32_t totalPartitions;  // project-kernel partitions
  int32_t  partitionCull;    // 1: the project kernels test their partition as a whole first (partition_cull.h)
  int32_t  debugFlags;       // MGS_DEBUG_* bits
  int32_t  sizeCulling;      // dist.comp.slang:93-134
  float    sizeCullingMinPixels;
  float    maxFocal;         // max(|focal.x|, |focal.y|)
  int32_t  surfaceOutputs;   // picked depth + splat id side outputs (frag.slang:320-349)
  float    depthIsoThreshold;
  float    thinParticleThreshold;  // shaderio.h:316: scale below which a particle axis counts as degenerate
  int32_t  quantizeNormals;        // QUANTIZE_NORMALS (parameters.h:195): octahedral 2x16-bit round trip of the splat normal
  // 3DGUT pipeline
  int32_t  pipeline;               // 0 3DGS, 1 3DGUT
  int32_t  cameraModel;            // 0 pinhole, 1 fisheye
  int32_t  extentMethod;           // 0 eigen, 1 conic
  float    fovRad;
  float    alphaClamp, kernelMinResponse;
  float    gutFocal[2];            // pinhole: == focal; fisheye: (1,-1) * viewport / fovRad (gaussian_splatting.cpp:1243)
  float    gutMaxAngle;            // computeMaxAngle (threedgut_camera_models.h.slang:87-118)
  float    viewInv[16], projInv[16];  // glm::inverse (gaussian_splatting.cpp:1166,1200)
  // stochastic splats (SORTING_STOCHASTIC_SPLAT, frag.slang:265-290) / depth of field (3DGUT, frag.slang:104-109) /
  // temporal accumulation (post.comp.slang)
  int32_t  stochastic;             // 1: binary accept/reject per fragment, the nearest accepted fragment is the pixel
  int32_t  dofMode;                // 0 DOF_DISABLED, 1 DOF_FIXED_FOCUS (shaderio.h:136-138)
  int32_t  frameSampleId;          // frameInfo.frameSampleId: seeds the per-pixel random numbers
  int32_t  temporalSampling;       // 1: the frame is folded into the running mean of the samples 0..frameSampleId
  float    focusDist, aperture;    // shaderio.h:278-279
  int32_t  kernelDegree;           // KERNEL_DEGREE (3DGUT particle response), 2 = quadratic
  int32_t  normalMethod;           // NORMAL_METHOD (shaderio.h:126-128): 0 max-density plane, 1 iso-surface (3DGUT fragment normal)
  // the bin rectangles ride through the key sort in the id word's spare bits (kernels_common.h: rideEncode)
  int32_t  rideShift;              // bits the ids need; 0 = no ride
  int32_t  rideShapes;             // how many of the shapes 1x1, 2x1, 1x2, 2x2 have codes
  uint32_t rideEscape;             // the code of every other rectangle: (1 << code bits) - 1
  int32_t  perspAffine;            // 1: view's last row is bitwise (+0,+0,+0,1), proj has the perspective zero pattern with P[14] != 0, all
                                   // entries < 2^24 in magnitude: the project kernels may take the exact shortcuts of kernels_common.h
  int32_t  rideSplit;              // 1: the id word's spare bits do not hold the whole code (> 8 M splats): its low 8 bits travel in the
                                   // key's low byte — dead weight once the slot is grouped by it (slot_emit.h) —, the rest above the id
};

struct FrameArgs
{
  FrameConst    f;
  InstanceConst inst[kMaxInstances];
};

// What the compositor needs besides the lists: screen geometry, mode knobs and the SH table of the instances.  All of
// it is constant for a captured frame graph (it is part of the graph's key, or scene state that a commit
// invalidates), so it travels BY VALUE: kernel arguments are preloaded into SGPRs, while reading the same fields through
// the per-frame FrameArgs pointer cost the compositor 10 us (an extra dependent scalar load at every workgroup start).
struct CompositeArgs
{
  int32_t width, height;
  int32_t tilesX;
  int32_t binShiftX, binShiftY;
  int32_t binsX, binsY;
  int32_t stripRow0, stripRow1;
  int32_t nInstances;
  int32_t shDegree;
  int32_t looseMask;          // A/B knob (MGS_LOOSE_MASK)
#ifdef MGS_CMP_TRACE
  uint64_t* trace;            // debug build only (tools/cmp_trace.py): per-workgroup time stamps and counts
#endif
  float   depthIsoThreshold;
  int32_t shOnly;             // SHOW_SH_ONLY (mesh.slang:205-207): base colour 0.5
  uint32_t* binCost;          // [256] per bin: the longest region of this frame (100 MHz ticks) -> the NEXT frame's bin order
  struct Inst
  {
    const void*   sh;
    const float4* rgba;     // colours as fetchColor returns them (dequantised fp32; the fp32 storage buffer itself when
                            // the set is stored as fp32)
    const float*  centers;
    uint32_t     globalOffset;
    int32_t      shDegree;
  } inst[kMaxInlineInstances];
  const Inst* instTable;      // all instances (device memory, rebuilt at commit); used when nInstances > kMaxInlineInstances
};

// projected splat record consumed by the compositor: 32 B = half a 64-byte sector, 16-B aligned.  It holds only what
// the per-fragment arithmetic and the region cull need; base colour, view direction and fragCoord.z are rebuilt by the
// compositor for the records it stages (a quarter of them), from the splat's own buffers.
struct alignas(16) SplatRec
{
  float    cx, cy;    // centre in pixels
  float    p1x, p1y;  // 2*b1/|b1|^2 : (d.p1)^2 + (d.p2)^2 == A/2 of threedgs_raster.frag.slang:236
  float    p2x, p2y;
  float    a;         // opacity (after MS_ANTIALIASING)
  uint32_t exey;      // half2: tight half extents of the visible footprint in pixels, rounded UP (cull tests only)
};

// 3DGUT projected record: 96 B, indexed by global id.  The per-fragment evaluator needs the particle itself, not a 2D
// conic: with A = S^-1 R^T (canonical frame), N = 3x3 of M^-1, o = camera origin, the canonical ray of a fragment with
// world direction d is  origin ro = A (M^-1 o - p) (per splat)  and  direction ~ B d, B = A N (per splat), so
// dist^2 = |B d x ro|^2 / |B d|^2 (threedgrt.h.slang:57-81) costs 9 FMAs + a cross product per fragment.
struct alignas(16) GutRec
{
  float cx, cy;        // UT mean in pixels (quad centre)
  float q1x, q1y;      // half1 / |half1|^2 : |d.q1| <= 1 and |d.q2| <= 1  <=>  the pixel centre is inside the quad
  float q2x, q2y;
  float bex, bey;      // half extents of the quad's bounding box in pixels (culling only)
  float B[9];
  float ro[3];
  float r, g, b, a;    // colour incl. SH; opacity after MS antialiasing
};
static_assert(sizeof(GutRec) == 96, "six 16-byte vectors");

// device-resident counters of one frame
struct FrameCounters
{
  uint32_t frustumCount;   // (zero since round 5: the dist stage's survivors are counted in the statistics lines, sort_plan.h: frameStatSlot)
  uint32_t sortedCount;    // V: elements handed to the radix sort
  uint32_t pairCount;      // D: (tile, splat) records
  uint32_t errorFlags;
  // (the compositors' statistics — staged records, scanned entries — and the project kernels' frustum survivors are counted on 32
  //  lines of their own since round 5: sort_plan.h, frameStatSlot)
  uint32_t pad[27];
};

}  // namespace mgs
