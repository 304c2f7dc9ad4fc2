/*
 * mgs_oracle.cpp — TEST INFRASTRUCTURE ONLY (see mgs_oracle.h for the rules and pinning status).
 *
 * Plain fp32 CPU restatement of the reference's VK3DGSR path.  Compiled with
 * -ffp-contract=off so every a*b+c rounds twice, i.e. "unfused fp32" is the oracle's
 * arithmetic; GPU results are compared under the tolerances stated in tests/.
 *
 * Matrix convention: the Slang shaders are compiled row-major and use
 * mul(v, M) over glm column-major memory (src/gaussian_splatting.cpp:125), which is the
 * ordinary column-vector product  out = M * v  with  M(r,c) = mem[c*4 + r].
 *
 * Third-party arithmetic that is NOT under /root/reference (nvpro_core2 is an empty
 * submodule, .gitmodules:1-3): glm (quat normalise, mat3_cast, inverse, packHalf1x16).
 * Their published algorithms are restated below and marked [glm].
 */
#include "mgs_oracle.h"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstring>
#include <execution>
#include <thread>
#include <vector>

namespace {

inline float m_at(const float* m, int r, int c) { return m[c * 4 + r]; }

// out = M * v (column vector), summation order x*col0 + y*col1 + z*col2 + w*col3
inline void mat4_mul_vec4(const float* m, const float v[4], float out[4])
{
  for(int r = 0; r < 4; ++r)
    out[r] = ((v[0] * m_at(m, r, 0) + v[1] * m_at(m, r, 1)) + v[2] * m_at(m, r, 2)) + v[3] * m_at(m, r, 3);
}

const float kSqrt8 = std::sqrt(8.0f);  // threedgs_particle_storage.h.slang:48

}  // namespace

extern "C" {

// ---------------------------------------------------------------------------------------
// src/splat_set.h:52-74
int orc_max_sh_degree(size_t f_rest_len, size_t splat_count)
{
  if(splat_count == 0)
    return -1;
  const size_t total      = (uint32_t)f_rest_len / splat_count;
  const size_t perChannel = total / 3;
  int          degree     = 0;
  if(perChannel >= 3)
    degree = 1;
  if(perChannel >= 8)
    degree = 2;
  if(perChannel == 15)
    degree = 3;
  return degree;
}

// 3rdparty/spz/src/cc/splat-types.h:55-83 evaluated for (from=RDF, to=RUB): x matches, y and z do not.
void orc_flip_sh_rdf_to_rub(float out[15])
{
  const float x = 1.f, y = -1.f, z = -1.f;
  const float f[15] = {y, z, x, x * y, y * z, 1.0f, x * z, 1.0f, y, x * y * z, y, z, x, z, x};
  std::memcpy(out, f, sizeof(f));
}

// src/splat_set.h:78-114 with coordinateConverter(RDF, RUB): flipP=(1,-1,-1), flipQ=(y*z,x*z,x*y)=(1,-1,-1)
void orc_convert_rdf_to_rub(float* positions, float* rotation, float* f_rest, size_t n, size_t coeffs_per_channel)
{
  float flipSh[15];
  orc_flip_sh_rdf_to_rub(flipSh);
  const float flipP[3] = {1.f, -1.f, -1.f};
  const float flipQ[3] = {1.f, -1.f, -1.f};
  for(size_t i = 0; i < n; ++i)
  {
    positions[3 * i + 0] *= flipP[0];
    positions[3 * i + 1] *= flipP[1];
    positions[3 * i + 2] *= flipP[2];
    // scalar component (index 0) untouched
    rotation[4 * i + 1] *= flipQ[0];
    rotation[4 * i + 2] *= flipQ[1];
    rotation[4 * i + 3] *= flipQ[2];
  }
  if(f_rest && coeffs_per_channel)
  {
    size_t idx = 0;
    for(size_t i = 0; i < n; ++i)
    {
      for(size_t j = 0; j < coeffs_per_channel; ++j)
      {
        const float flip = flipSh[j];
        f_rest[idx + j] *= flip;
        f_rest[idx + coeffs_per_channel + j] *= flip;
        f_rest[idx + 2 * coeffs_per_channel + j] *= flip;
      }
      idx += 3 * coeffs_per_channel;
    }
  }
}

// ---------------------------------------------------------------------------------------
// src/splat_set_vk.cpp:263-288.  [glm] normalize(quat): len=sqrt(dot); len<=0 -> (1,0,0,0); q*(1/len).
// [glm] mat3_cast: the standard quaternion->matrix (column-major Result[col][row]).
void orc_cov3d(const float* scale, const float* rotation, size_t n, float* cov6)
{
  for(size_t i = 0; i < n; ++i)
  {
    const float s[3] = {std::exp(scale[3 * i + 0]), std::exp(scale[3 * i + 1]), std::exp(scale[3 * i + 2])};
    float       w = rotation[4 * i + 0], x = rotation[4 * i + 1], y = rotation[4 * i + 2], z = rotation[4 * i + 3];
    const float len = std::sqrt(((w * w + x * x) + y * y) + z * z);
    if(len <= 0.f)
    {
      w = 1.f;
      x = y = z = 0.f;
    }
    else
    {
      const float inv = 1.f / len;
      w *= inv;
      x *= inv;
      y *= inv;
      z *= inv;
    }
    const float qxx = x * x, qyy = y * y, qzz = z * z, qxz = x * z, qxy = x * y, qyz = y * z, qwx = w * x,
                qwy = w * y, qwz = w * z;
    // R[col][row]
    float R[3][3];
    R[0][0] = 1.f - 2.f * (qyy + qzz);
    R[0][1] = 2.f * (qxy + qwz);
    R[0][2] = 2.f * (qxz - qwy);
    R[1][0] = 2.f * (qxy - qwz);
    R[1][1] = 1.f - 2.f * (qxx + qzz);
    R[1][2] = 2.f * (qyz + qwx);
    R[2][0] = 2.f * (qxz + qwy);
    R[2][1] = 2.f * (qyz - qwx);
    R[2][2] = 1.f - 2.f * (qxx + qyy);
    // M = R * diag(s): column j scaled by s[j]
    float M[3][3];
    for(int c = 0; c < 3; ++c)
      for(int r = 0; r < 3; ++r)
        M[c][r] = R[c][r] * s[c];
    // Sigma = M * M^T : Sigma(r,c) = sum_k M(r,k) M(c,k)
    auto sig = [&](int r, int c) { return (M[0][r] * M[0][c] + M[1][r] * M[1][c]) + M[2][r] * M[2][c]; };
    cov6[6 * i + 0] = sig(0, 0);
    cov6[6 * i + 1] = sig(0, 1);
    cov6[6 * i + 2] = sig(0, 2);
    cov6[6 * i + 3] = sig(1, 1);
    cov6[6 * i + 4] = sig(1, 2);
    cov6[6 * i + 5] = sig(2, 2);
  }
}

// src/splat_set_vk.cpp:313-345
void orc_rgba(const float* f_dc, const float* opacity, size_t n, float* rgba)
{
  const float SH_C0 = 0.28209479177387814f;
  auto        clamp01 = [](float v) { return std::min(std::max(v, 0.0f), 1.0f); };
  for(size_t i = 0; i < n; ++i)
  {
    rgba[4 * i + 0] = clamp01(0.5f + SH_C0 * f_dc[3 * i + 0]);
    rgba[4 * i + 1] = clamp01(0.5f + SH_C0 * f_dc[3 * i + 1]);
    rgba[4 * i + 2] = clamp01(0.5f + SH_C0 * f_dc[3 * i + 2]);
    rgba[4 * i + 3] = clamp01(1.0f / (1.0f + std::exp(-opacity[i])));
  }
}

int orc_sh_stride(int cpc)
{
  int stride = 0;
  if(cpc >= 3)
    stride += 9;
  if(cpc >= 8)
    stride += 15;
  if(cpc == 15)
    stride += 21;
  return stride;
}

// src/splat_set_vk.cpp:356-435: channel-major f_rest -> [coef][rgb] interleave
void orc_sh_interleave(const float* f_rest, size_t n, int cpc, float* out)
{
  const int stride = orc_sh_stride(cpc);
  const int ncoef  = stride / 3;
  const int srcStride = 3 * cpc;
  for(size_t i = 0; i < n; ++i)
    for(int k = 0; k < ncoef; ++k)
      for(int c = 0; c < 3; ++c)
        out[i * stride + 3 * k + c] = f_rest[i * srcStride + cpc * c + k];
}

// [glm] packHalf1x16 / unpackHalf1x16: IEEE binary16, round-to-nearest-even
uint16_t orc_float_to_half(float f)
{
  uint32_t x;
  std::memcpy(&x, &f, 4);
  const uint32_t sign = (x >> 16) & 0x8000u;
  const uint32_t absx = x & 0x7fffffffu;
  if(absx >= 0x7f800000u)  // inf / nan
    return (uint16_t)(sign | 0x7c00u | ((absx > 0x7f800000u) ? 0x200u : 0u));
  if(absx >= 0x477ff000u)  // rounds to >= 65520 -> inf
    return (uint16_t)(sign | 0x7c00u);
  if(absx < 0x33000001u)  // < 2^-25 (or exactly 2^-25 -> ties to even 0)
    return (uint16_t)sign;
  int32_t  exp  = (int32_t)(absx >> 23) - 127;
  uint32_t mant = (absx & 0x7fffffu) | 0x800000u;
  uint32_t shift;
  uint32_t hexp;
  if(exp < -14)
  {  // subnormal half
    shift = (uint32_t)(13 + (-14 - exp));
    hexp  = 0;
  }
  else
  {
    shift = 13;
    hexp  = (uint32_t)(exp + 15);
  }
  uint32_t hm        = mant >> shift;
  uint32_t rem       = mant & ((1u << shift) - 1u);
  uint32_t halfway   = 1u << (shift - 1);
  if(rem > halfway || (rem == halfway && (hm & 1u)))
    hm += 1;
  uint32_t h;
  if(hexp == 0)
    h = hm;  // may carry into exponent 1: correct by construction
  else
    h = ((hexp << 10) + (hm - 0x400u));  // hm has implicit bit at 0x400; carry propagates
  return (uint16_t)(sign | h);
}

float orc_half_to_float(uint16_t h)
{
  const uint32_t sign = (uint32_t)(h & 0x8000u) << 16;
  const uint32_t exp  = (h >> 10) & 0x1fu;
  const uint32_t mant = h & 0x3ffu;
  uint32_t       x;
  if(exp == 0)
  {
    if(mant == 0)
      x = sign;
    else
    {
      float v = std::ldexp((float)mant, -24);
      std::memcpy(&x, &v, 4);
      x |= sign;
    }
  }
  else if(exp == 31)
    x = sign | 0x7f800000u | (mant << 13);
  else
    x = sign | ((exp + 112u) << 23) | (mant << 13);
  float f;
  std::memcpy(&f, &x, 4);
  return f;
}

// src/splat_set_vk.cpp:85-112 (toUint8/storeSh) then the shader-side dequantisation
// shaders/threedgs_particle_buffers.h.slang:72-90 (colour v/255), :112-131 (SH v/255*2-1)
void orc_quantize_roundtrip(float* data, size_t count, int format, int is_sh)
{
  if(format == ORC_FORMAT_FLOAT32)
    return;
  if(format == ORC_FORMAT_FLOAT16)
  {
    for(size_t i = 0; i < count; ++i)
      data[i] = orc_half_to_float(orc_float_to_half(data[i]));
    return;
  }
  const float lo = is_sh ? -1.f : 0.f, hi = 1.f;
  for(size_t i = 0; i < count; ++i)
  {
    const float   normalized = (data[i] - lo) / (hi - lo);
    const uint8_t q          = (uint8_t)std::min(std::max(std::round(normalized * 255.0f), 0.0f), 255.0f);
    if(is_sh)
      data[i] = float(q) / 255.0f * 2.0f - 1.0f;
    else
      data[i] = float(q) / 255.0f;
  }
}

// ---------------------------------------------------------------------------------------
// [glm] inverse(mat4): cofactor expansion, as in glm/detail/func_matrix.inl compute_inverse<4,4>
void orc_mat4_inverse(const float* m, float* out)
{
  auto  M      = [&](int c, int r) { return m[c * 4 + r]; };
  float Coef00 = M(2, 2) * M(3, 3) - M(3, 2) * M(2, 3);
  float Coef02 = M(1, 2) * M(3, 3) - M(3, 2) * M(1, 3);
  float Coef03 = M(1, 2) * M(2, 3) - M(2, 2) * M(1, 3);
  float Coef04 = M(2, 1) * M(3, 3) - M(3, 1) * M(2, 3);
  float Coef06 = M(1, 1) * M(3, 3) - M(3, 1) * M(1, 3);
  float Coef07 = M(1, 1) * M(2, 3) - M(2, 1) * M(1, 3);
  float Coef08 = M(2, 1) * M(3, 2) - M(3, 1) * M(2, 2);
  float Coef10 = M(1, 1) * M(3, 2) - M(3, 1) * M(1, 2);
  float Coef11 = M(1, 1) * M(2, 2) - M(2, 1) * M(1, 2);
  float Coef12 = M(2, 0) * M(3, 3) - M(3, 0) * M(2, 3);
  float Coef14 = M(1, 0) * M(3, 3) - M(3, 0) * M(1, 3);
  float Coef15 = M(1, 0) * M(2, 3) - M(2, 0) * M(1, 3);
  float Coef16 = M(2, 0) * M(3, 2) - M(3, 0) * M(2, 2);
  float Coef18 = M(1, 0) * M(3, 2) - M(3, 0) * M(1, 2);
  float Coef19 = M(1, 0) * M(2, 2) - M(2, 0) * M(1, 2);
  float Coef20 = M(2, 0) * M(3, 1) - M(3, 0) * M(2, 1);
  float Coef22 = M(1, 0) * M(3, 1) - M(3, 0) * M(1, 1);
  float Coef23 = M(1, 0) * M(2, 1) - M(2, 0) * M(1, 1);

  const float Fac0[4] = {Coef00, Coef00, Coef02, Coef03};
  const float Fac1[4] = {Coef04, Coef04, Coef06, Coef07};
  const float Fac2[4] = {Coef08, Coef08, Coef10, Coef11};
  const float Fac3[4] = {Coef12, Coef12, Coef14, Coef15};
  const float Fac4[4] = {Coef16, Coef16, Coef18, Coef19};
  const float Fac5[4] = {Coef20, Coef20, Coef22, Coef23};
  const float Vec0[4] = {M(1, 0), M(0, 0), M(0, 0), M(0, 0)};
  const float Vec1[4] = {M(1, 1), M(0, 1), M(0, 1), M(0, 1)};
  const float Vec2[4] = {M(1, 2), M(0, 2), M(0, 2), M(0, 2)};
  const float Vec3[4] = {M(1, 3), M(0, 3), M(0, 3), M(0, 3)};
  float       Inv[4][4];
  const float SignA[4] = {+1, -1, +1, -1};
  const float SignB[4] = {-1, +1, -1, +1};
  for(int i = 0; i < 4; ++i)
  {
    Inv[0][i] = (Vec1[i] * Fac0[i] - Vec2[i] * Fac1[i] + Vec3[i] * Fac2[i]) * SignA[i];
    Inv[1][i] = (Vec0[i] * Fac0[i] - Vec2[i] * Fac3[i] + Vec3[i] * Fac4[i]) * SignB[i];
    Inv[2][i] = (Vec0[i] * Fac1[i] - Vec1[i] * Fac3[i] + Vec3[i] * Fac5[i]) * SignA[i];
    Inv[3][i] = (Vec0[i] * Fac2[i] - Vec1[i] * Fac4[i] + Vec2[i] * Fac5[i]) * SignB[i];
  }
  const float Dot1 = (M(0, 0) * Inv[0][0] + M(0, 1) * Inv[1][0]) + (M(0, 2) * Inv[2][0] + M(0, 3) * Inv[3][0]);
  const float OneOverDet = 1.0f / Dot1;
  for(int c = 0; c < 4; ++c)
    for(int r = 0; r < 4; ++r)
      out[c * 4 + r] = Inv[c][r] * OneOverDet;
}

// out = a * b (both glm column-major); [glm] operator*(mat4,mat4)
void orc_mat4_mul(const float* a, const float* b, float* out)
{
  float tmp[16];
  for(int c = 0; c < 4; ++c)
    for(int r = 0; r < 4; ++r)
      tmp[c * 4 + r] = ((a[0 * 4 + r] * b[c * 4 + 0] + a[1 * 4 + r] * b[c * 4 + 1]) + a[2 * 4 + r] * b[c * 4 + 2])
                       + a[3 * 4 + r] * b[c * 4 + 3];
  std::memcpy(out, tmp, sizeof(tmp));
}

// shaders/dist.comp.slang:33-38
uint32_t orc_encode_key(float v)
{
  uint32_t bits;
  std::memcpy(&bits, &v, 4);
  bits ^= (uint32_t)((int32_t)bits >> 31) | 0x80000000u;
  return bits;
}

// ---- CAMERA_TYPE == CAMERA_FISHEYE branch of the dist stage (dist.comp.slang:75-90) ----
// atan2 is implementation-defined in the reference (SPIR-V Atan2).  The cull decision selects the sorted set, which the
// tests compare bit for bit, so oracle and kernels share ONE definition: the single-precision arctangent of the Cephes
// library (range reduction at tan(pi/8) and tan(3pi/8), degree-9 odd polynomial, about 2 ulp), every operation rounded
// separately (this file is compiled with -ffp-contract=off).  y > 0 (rho >= 1e-7).
float orc_atan2_det(float y, float x)
{
  const float ax = std::fabs(x);
  const float t  = y / ax;
  float base = 0.0f, u = t;
  if(t > 2.414213562373095f)
  {
    base = 1.5707963267948966f;
    u    = -1.0f / t;
  }
  else if(t > 0.4142135623730950f)
  {
    base = 0.7853981633974483f;
    u    = (t - 1.0f) / (t + 1.0f);
  }
  const float z = u * u;
  const float p = ((((8.05374449538e-2f * z - 1.38776856032e-1f) * z + 1.99777106478e-1f) * z - 3.33329491539e-1f) * z) * u + u;
  const float a = base + p;
  return (x < 0.0f) ? (3.14159274101257f - a) : a;
}

namespace {
// frameInfo.focal as the dist stage reads it (gaussian_splatting.cpp:1239-1251)
void distStageFocal(const OrcFrame* f, float focal[2])
{
  if(f->camera_model == 1 && f->pipeline_3dgut)
  {
    focal[0] = (float)f->width / f->fov_rad;
    focal[1] = -(float)f->height / f->fov_rad;
  }
  else
  {
    focal[0] = f->proj[0] * 0.5f * (float)f->width;
    focal[1] = f->proj[5] * 0.5f * (float)f->height;
  }
}
}  // namespace

// dist.comp.slang:78-86: initPerfectFisheyeCamera(viewport, focal) (threedgut_camera_models.h.slang:120-136: principal point at
// the viewport centre, zero radial coefficients, maxAngle = computeMaxAngle :87-118) and projectPointFisheye
// (threedgut_camera_projections.h.slang:149-171) on float3(1,1,-1) * viewPos with GUT_IN_IMAGE_MARGIN_FACTOR = 0.1.
// min / max are written as selects so that a NaN anywhere fails the validity test (the reference's min/max on NaN are
// undefined in SPIR-V; the kernels use the same selects).
int orc_fisheye_cull_valid(const OrcFrame* f, const float view_pos[3])
{
  float focal[2];
  distStageFocal(f, focal);
  const float resx = (float)f->width, resy = (float)f->height;
  const float mdx = resx - resx / 2.0f, mdy = resy - resy / 2.0f;  // computeMaxDistanceToBorder, principal point at the centre
  const float maxR     = std::sqrt(mdx * mdx + mdy * mdy);
  const float maxAngle = std::max(2.0f * maxR / focal[0], 2.0f * maxR / focal[1]) / 2.0f;
  const float px = view_pos[0], py = view_pos[1], pz = -view_pos[2];
  const float ax = std::fabs(px), ay = std::fabs(py);  // stableNorm2 :32-44
  const float mn = (ax < ay) ? ax : ay, mx = (ax < ay) ? ay : ax;
  float       nrm = 0.0f;
  if(mx > 0.0f)
  {
    const float r = mn / mx;
    nrm           = mx * std::sqrt(1.0f + r * r);
  }
  const float rho       = (nrm > 1e-7f) ? nrm : 1e-7f;
  const float thetaFull = orc_atan2_det(rho, pz);
  const float theta     = (thetaFull < maxAngle) ? thetaFull : maxAngle;
  const float theta2    = theta * theta;
  const float delta     = (theta * (0.0f * theta2 + 1.0f)) / rho;
  const float ox = (focal[0] * px) * delta + resx / 2.0f;
  const float oy = (focal[1] * py) * delta + resy / 2.0f;
  const float tx = resx * 0.1f, ty = resy * 0.1f;  // withinResolution :78-83
  return (theta < maxAngle) && (ox > -tx) && (oy > -ty) && (ox < resx + tx) && (oy < resy + ty);
}

// shaders/dist.comp.slang:40-171 (both CAMERA_TYPE branches of the dist-stage cull, size culling)
// Global ids are the concatenation of instances in creation order
// (src/splat_set_manager_vk.cpp:2319-2357).  Survivor order: ascending global id — a
// deterministic refinement of the reference's atomic append order (dist.comp.slang:137-139).
uint32_t orc_key_cull(const OrcFrame* f, const OrcInstance* inst, int n_inst, uint32_t* keys, uint32_t* ids)
{
  uint32_t v      = 0;
  uint32_t offset = 0;
  for(int k = 0; k < n_inst; ++k)
  {
    const OrcInstance& I = inst[k];
    for(uint32_t i = 0; i < I.count; ++i)
    {
      const float p[4] = {I.centers[3 * i + 0], I.centers[3 * i + 1], I.centers[3 * i + 2], 1.0f};
      float       world[4], view[4], clip[4];
      mat4_mul_vec4(I.transform, p, world);  // mul(splatPos, desc.transform)          :58
      mat4_mul_vec4(f->view, world, view);   // mul(..., frameInfo.viewMatrix)         :58
      mat4_mul_vec4(f->proj, view, clip);    // mul(viewPos, projectionMatrix)         :60
      const float ndc[3] = {clip[0] / clip[3], clip[1] / clip[3], clip[2] / clip[3]};  // :61
      const float depth  = ndc[2];
      if(f->frustum_culling == 1 && f->camera_model != 1)
      {
        const float c = 1.0f + f->frustum_dilation;  // :71-73
        if(std::fabs(ndc[0]) > c || std::fabs(ndc[1]) > c || ndc[2] < 0.f - f->frustum_dilation || ndc[2] > 1.0f)
          continue;
        // NaN compares false in every test above, exactly as in the shader: a NaN splat survives.
      }
      else if(f->frustum_culling == 1)
      {  // :75-90, CAMERA_FISHEYE
        if(!orc_fisheye_cull_valid(f, view))
          continue;
        if(ndc[2] < 0.f - f->frustum_dilation || ndc[2] > 1.0f)
          continue;
      }
      if(f->size_culling && I.scales)
      {  // dist.comp.slang:93-134
        const float sx = std::exp(I.scales[3 * i + 0]) * f->splat_scale, sy = std::exp(I.scales[3 * i + 1]) * f->splat_scale,
                    sz = std::exp(I.scales[3 * i + 2]) * f->splat_scale;
        const float radius = std::max(sx, std::max(sy, sz));
        const float sqrt8  = 2.8284271247f;
        float       extent = radius * sqrt8 * 2.0f;
        auto        len3   = [&](int c) {
          const float* m = I.transform + 4 * c;  // shader row c of the row-major view == glm column c
          return std::sqrt((m[0] * m[0] + m[1] * m[1]) + m[2] * m[2]);
        };
        extent *= std::max(len3(0), std::max(len3(1), len3(2)));
        const float viewDist = std::fabs(view[2]);
        if(viewDist > 0.0001f)
        {
          float focal[2];
          distStageFocal(f, focal);  // frameInfo.focal :125
          const float maxFocal = std::max(std::fabs(focal[0]), std::fabs(focal[1]));
          const float projectedPixels = (extent * maxFocal) / viewDist;
          if(projectedPixels < f->size_culling_min_pixels)
            continue;
        }
      }
      ids[v]  = offset + i;
      keys[v] = f->front_to_back ? orc_encode_key(depth) : orc_encode_key(-depth);  // :163-167
      ++v;
    }
    offset += I.count;
  }
  return v;
}

// 3rdparty/vrdx/src/vk_radix_sort.cc:262-416: stable ascending LSD radix sort, 8-bit digits, 4 passes
void orc_sort_stable(uint32_t* keys, uint32_t* ids, uint32_t n)
{
  std::vector<uint32_t> k2(n), v2(n);
  uint32_t *            ks = keys, *vs = ids, *kd = k2.data(), *vd = v2.data();
  for(int pass = 0; pass < 4; ++pass)
  {
    size_t hist[257] = {0};
    const int shift  = 8 * pass;
    for(uint32_t i = 0; i < n; ++i)
      hist[((ks[i] >> shift) & 0xffu) + 1]++;
    for(int d = 0; d < 256; ++d)
      hist[d + 1] += hist[d];
    for(uint32_t i = 0; i < n; ++i)
    {
      const size_t dst = hist[(ks[i] >> shift) & 0xffu]++;
      kd[dst]          = ks[i];
      vd[dst]          = vs[i];
    }
    std::swap(ks, kd);
    std::swap(vs, vd);
  }
  // even number of passes: result is back in keys/ids
}

// ---------------------------------------------------------------------------------------
// shaders/threedgs_particle_storage.h.slang:48-52,103-159
static void sh_radiance(const OrcInstance& I, uint32_t idx, int requested, const float d[3], float rgb[3])
{
  rgb[0] = rgb[1] = rgb[2] = 0.f;
  const int degree = std::min(I.sh_degree, requested);
  if(degree < 1 || !I.sh)
    return;
  const float  SH_C1    = 0.4886025119029199f;
  const float  SH_C2[5] = {1.0925484f, -1.0925484f, 0.3153916f, -1.0925484f, 0.5462742f};
  const float  SH_C3[7] = {-0.5900435899266435f, 2.890611442640554f, -0.4570457994644658f, 0.3731763325901154f,
                           -0.4570457994644658f, 1.445305721320277f, -0.5900435899266435f};
  const float* s        = I.sh + (size_t)idx * I.sh_stride;
  const float  x = d[0], y = d[1], z = d[2];
  for(int c = 0; c < 3; ++c)
  {
    auto  S = [&](int k) { return s[3 * k + c]; };
    float r = SH_C1 * (-S(0) * y + S(1) * z - S(2) * x);
    if(degree >= 2)
    {
      const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
      r += (SH_C2[0] * xy) * S(3) + (SH_C2[1] * yz) * S(4) + (SH_C2[2] * (2.0f * zz - xx - yy)) * S(5)
           + (SH_C2[3] * xz) * S(6) + (SH_C2[4] * (xx - yy)) * S(7);
      if(degree >= 3)
      {
        r += SH_C3[0] * S(8) * (3.0f * x * x - y * y) * y + SH_C3[1] * S(9) * x * y * z
             + SH_C3[2] * S(10) * (4.0f * z * z - x * x - y * y) * y
             + SH_C3[3] * S(11) * z * (2.0f * z * z - 3.0f * x * x - 3.0f * y * y)
             + SH_C3[4] * S(12) * x * (4.0f * z * z - x * x - y * y) + SH_C3[5] * S(13) * (x * x - y * y) * z
             + SH_C3[6] * S(14) * x * (x * x - 3.0f * y * y);
      }
    }
    rgb[c] = r;
  }
}

// shaders/threedgs_raster.mesh.slang:111-291 + shaders/threedgs.h.slang:26-121
void orc_project(const OrcFrame* f, const OrcInstance* Ip, uint32_t i, OrcProjected* out)
{
  const OrcInstance& I = *Ip;
  std::memset(out, 0, sizeof(*out));
  float rgba[4] = {I.rgba[4 * i + 0], I.rgba[4 * i + 1], I.rgba[4 * i + 2], I.rgba[4 * i + 3]};
  if(rgba[3] < f->alpha_cull_threshold)  // mesh.slang:164-170
    return;
  const float p[4] = {I.centers[3 * i + 0], I.centers[3 * i + 1], I.centers[3 * i + 2], 1.0f};
  float       MV[16];
  orc_mat4_mul(f->view, I.transform, MV);  // mul(desc.transform, viewMatrix) == V*M      mesh.slang:175
  float viewC[4], clip[4];
  mat4_mul_vec4(MV, p, viewC);         // :178
  mat4_mul_vec4(f->proj, viewC, clip);  // :179 (projectionMatrixJittered == projectionMatrix w/o DLSS)
  if(f->frustum_culling == 2)
  {  // mesh.slang:181-190
    const float c = (1.0f + f->frustum_dilation) * clip[3];
    if(std::fabs(clip[0]) > c || std::fabs(clip[1]) > c || clip[2] < (0.0f - f->frustum_dilation) * clip[3]
       || clip[2] > clip[3])
      return;
  }
  // SH: direction in model space                                                           :240-243
  const float cam[4] = {f->camera_pos[0], f->camera_pos[1], f->camera_pos[2], 1.0f};
  float       camM[4];
  mat4_mul_vec4(I.transform_inv, cam, camM);
  float       dir[3] = {p[0] - camM[0], p[1] - camM[1], p[2] - camM[2]};
  const float dl     = std::sqrt((dir[0] * dir[0] + dir[1] * dir[1]) + dir[2] * dir[2]);
  dir[0] /= dl;
  dir[1] /= dl;
  dir[2] /= dl;
  if(f->debug_flags & 2)  // SHOW_SH_ONLY, mesh.slang:205-207
    rgba[0] = rgba[1] = rgba[2] = 0.5f;
  float sh[3];
  sh_radiance(I, i, f->sh_degree, dir, sh);
  rgba[0] += sh[0];
  rgba[1] += sh[1];
  rgba[2] += sh[2];  // no clamp afterwards

  // covariance projection                                                            threedgs.h.slang:26-56
  const float* c6       = I.cov6 + (size_t)6 * i;
  const float  S[3][3]  = {{c6[0], c6[1], c6[2]}, {c6[1], c6[3], c6[4]}, {c6[2], c6[4], c6[5]}};
  const float  focal[2] = {f->proj[0] * 0.5f * (float)f->width, f->proj[5] * 0.5f * (float)f->height};  // cpp:1248-1250
  const float  tz = viewC[2];
  const float  s  = 1.0f / (tz * tz);
  const float  J[3][3] = {{focal[0] / tz, 0.f, -(focal[0] * viewC[0]) * s},
                          {0.f, focal[1] / tz, -(focal[1] * viewC[1]) * s},
                          {0.f, 0.f, 0.f}};
  float        W[3][3];  // W(r,c) = MV(r,c)
  for(int r = 0; r < 3; ++r)
    for(int c = 0; c < 3; ++c)
      W[r][c] = m_at(MV, r, c);
  float T[3][3], TS[3][3];
  for(int r = 0; r < 3; ++r)
    for(int c = 0; c < 3; ++c)
      T[r][c] = (J[r][0] * W[0][c] + J[r][1] * W[1][c]) + J[r][2] * W[2][c];
  for(int r = 0; r < 3; ++r)
    for(int c = 0; c < 3; ++c)
      TS[r][c] = (T[r][0] * S[0][c] + T[r][1] * S[1][c]) + T[r][2] * S[2][c];
  auto  c2  = [&](int r, int c) { return (TS[r][0] * T[c][0] + TS[r][1] * T[c][1]) + TS[r][2] * T[c][2]; };
  float a = c2(0, 0), b = c2(0, 1), d = c2(1, 1);

  // extent basis                                                                     threedgs.h.slang:60-121
  float detOrig = 0.f;
  if(f->ms_antialiasing)
    detOrig = a * d - b * b;
  a += 0.3f;
  d += 0.3f;
  if(f->ms_antialiasing)
  {
    const float detBlur = a * d - b * b;
    rgba[3] *= std::sqrt(std::max(detOrig / detBlur, 0.0f));
  }
  const float D          = a * d - b * b;
  const float traceOver2 = 0.5f * (a + d);
  const float term2      = std::sqrt(std::max(0.1f, traceOver2 * traceOver2 - D));
  float ev1 = traceOver2 + term2, ev2 = traceOver2 - term2;
  if(ev2 <= 0.0f)
    return;
  if(f->debug_flags & 1)  // POINT_CLOUD_MODE, threedgs.h.slang:108-110
    ev1 = ev2 = 0.2f;
  float       e1[2] = {(std::fabs(b) < 0.001f) ? 1.0f : b, ev1 - a};
  const float el    = std::sqrt(e1[0] * e1[0] + e1[1] * e1[1]);
  e1[0] /= el;
  e1[1] /= el;
  const float e2[2] = {e1[1], -e1[0]};
  const float l1    = std::min(kSqrt8 * std::sqrt(ev1), 2048.0f);
  const float l2    = std::min(kSqrt8 * std::sqrt(ev2), 2048.0f);
  out->basis1[0]    = e1[0] * f->splat_scale * l1;
  out->basis1[1]    = e1[1] * f->splat_scale * l1;
  out->basis2[0]    = e2[0] * f->splat_scale * l2;
  out->basis2[1]    = e2[1] * f->splat_scale * l2;

  const float ndc[3] = {clip[0] / clip[3], clip[1] / clip[3], clip[2] / clip[3]};  // :268
  // Fixed-function clipping of the emitted quad (all four vertices carry z = ndc.z, w = 1):
  // Vulkan's clip volume is 0 <= z <= w with depth clamp off (the pipeline never enables it,
  // src/gaussian_splatting.cpp:2053-2181), which is also what emitDegeneratedQuad relies on
  // when it parks rejected quads at z = 2 (mesh.slang:102-109).
  if(!(ndc[2] >= 0.0f && ndc[2] <= 1.0f))
    return;
  out->ndc_z        = ndc[2];
  out->center_px[0] = (ndc[0] + 1.0f) * 0.5f * (float)f->width;  // viewport transform, origin (0,0)
  out->center_px[1] = (ndc[1] + 1.0f) * 0.5f * (float)f->height;
  std::memcpy(out->rgba, rgba, sizeof(rgba));
  out->opacity_disabled = (f->debug_flags & 4) ? 1 : 0;
  out->valid            = 1;
}

// fragments + blending: shaders/threedgs_raster.frag.slang:223-309, src/gaussian_splatting.cpp:2066-2087
// `win` (optional) = {x0, y0, x1, y1}: only pixels inside this inclusive window are evaluated, and `img` is then the
// window's own [y1-y0+1][x1-x0+1][4] buffer.  The per-pixel arithmetic is the same either way; pixels are independent
// given the draw order, so a windowed render equals the crop of the full one bit for bit.
// ---- nvshaders/random.h.slang (nvpro_core2, absent from the reference tree): restated from the published file ----------
uint32_t orc_xxhash32(uint32_t x, uint32_t y, uint32_t z)
{  // xxhash32 over a uint3 (Jarzynski & Olano, "Hash Functions for GPU Rendering"; shadertoy XlGcRh)
  const uint32_t p0 = 2246822519u, p1 = 3266489917u, p2 = 668265263u, p3 = 374761393u;
  uint32_t       h  = z + p3 + x * p1;
  h                 = p2 * ((h << 17) | (h >> (32 - 17)));
  h += y * p1;
  h = p2 * ((h << 17) | (h >> (32 - 17)));
  h = p0 * (h ^ (h >> 15));
  h = p1 * (h ^ (h >> 13));
  return h ^ (h >> 16);
}
static uint32_t pcg(uint32_t* state)
{  // pcg-random.org, RXS-M-XS 32
  const uint32_t prev = *state * 747796405u + 2891336453u;
  const uint32_t word = ((prev >> ((prev >> 28u) + 4u)) ^ prev) * 277803737u;
  *state              = prev;
  return (word >> 22u) ^ word;
}
float orc_rand(uint32_t* seed)
{  // a float in [0,1): the top 23 bits of pcg as the mantissa of a number in [1,2), minus one
  const uint32_t r = pcg(seed);
  const uint32_t b = 0x3f800000u | (r >> 9);
  float          v;
  std::memcpy(&v, &b, 4);
  return v - 1.0f;
}

void orc_post_accumulate(float* main_image, const float* aux1, size_t n, int frame_sample_id)
{  // post.comp.slang:37-43
  const float a = 1.0f / (float)(frame_sample_id + 1);
  for(size_t i = 0; i < n; ++i)
    main_image[i] = main_image[i] + a * (aux1[i] - main_image[i]);  // lerp(main, aux1, a)
}

// `depth` (stochastic splats only): the depth attachment, cleared to 1; `gid` = the splat's global id
static uint64_t raster_one(const OrcFrame* f, const OrcProjected& P, float* img, const int* win = nullptr, float* depth = nullptr,
                           uint32_t gid = 0)
{
  const int   W = f->width, H = f->height;
  const float ex = std::fabs(P.basis1[0]) + std::fabs(P.basis2[0]);
  const float ey = std::fabs(P.basis1[1]) + std::fabs(P.basis2[1]);
  // pixel centres (x+0.5) inside [cx-ex, cx+ex]
  const float fx0 = P.center_px[0] - ex - 0.5f, fx1 = P.center_px[0] + ex - 0.5f;
  const float fy0 = P.center_px[1] - ey - 0.5f, fy1 = P.center_px[1] + ey - 0.5f;
  if(!(fx1 >= 0.f && fy1 >= 0.f && fx0 <= (float)(W - 1) && fy0 <= (float)(H - 1)))
    return 0;
  const int   x0 = (int)std::max(0.0f, std::floor(fx0)), x1 = (int)std::min((float)(W - 1), std::ceil(fx1));
  int         y0 = (int)std::max(0.0f, std::floor(fy0)), y1 = (int)std::min((float)(H - 1), std::ceil(fy1));
  int         ox = 0, oy = 0, pitch = W;
  int         xa = x0, xb = x1;
  if(win)
  {
    xa = std::max(xa, win[0]);
    xb = std::min(xb, win[2]);
    y0 = std::max(y0, win[1]);
    y1 = std::min(y1, win[3]);
    ox = win[0];
    oy = win[1];
    pitch = win[2] - win[0] + 1;
  }
  const float n1 = P.basis1[0] * P.basis1[0] + P.basis1[1] * P.basis1[1];
  const float n2 = P.basis2[0] * P.basis2[0] + P.basis2[1] * P.basis2[1];
  uint64_t    frags = 0;
  for(int y = y0; y <= y1; ++y)
  {
    for(int x = xa; x <= xb; ++x)
    {
      const float dx = ((float)x + 0.5f) - P.center_px[0];
      const float dy = ((float)y + 0.5f) - P.center_px[1];
      // interpolated fragPos = sqrt8 * (u, v) with pixel = centre + u*b1 + v*b2 (b1 ⟂ b2)
      const float u  = (dx * P.basis1[0] + dy * P.basis1[1]) / n1;
      const float v  = (dx * P.basis2[0] + dy * P.basis2[1]) / n2;
      const float px = u * kSqrt8, py = v * kSqrt8;
      const float A  = px * px + py * py;  // frag.slang:236
      if(A > 8.0f)                         // :242-245
        continue;
      const float opacity = P.opacity_disabled ? 1.0f : std::exp(-0.5f * A) * P.rgba[3];  // :248-254
      if(opacity <= 1.0f / 255.0f)                            // :258-262
        continue;
      float* dst = img + ((size_t)(y - oy) * pitch + (x - ox)) * 4;
      if(f->stochastic && depth)
      {  // frag.slang:265-290: accept with probability `opacity`, write opaque; depth test LESS_OR_EQUAL + depth write.
        // primitiveID: the quad (-1,-1),(1,-1),(1,1),(-1,1) is the triangles (0,2,1) and (2,0,3) (mesh.slang:158-159,193):
        // triangle 0 holds u > v.  The index of the splat inside its mesh workgroup comes from an atomically compacted,
        // unsorted list in the reference (not reproducible); restated as global id mod RASTER_MESH_WORKGROUP_SIZE (32).
        uint32_t seed = orc_xxhash32((uint32_t)x, (uint32_t)y, (uint32_t)f->frame_sample_id);
        seed          = orc_xxhash32(seed, gid, 2u * (gid & 31u) + (u > v ? 0u : 1u));
        if(!(orc_rand(&seed) < opacity))
          continue;  // discard
        float& dz = depth[(size_t)(y - oy) * pitch + (x - ox)];
        if(!(P.ndc_z <= dz))
          continue;
        dz     = P.ndc_z;
        dst[0] = P.rgba[0];
        dst[1] = P.rgba[1];
        dst[2] = P.rgba[2];
        dst[3] = 1.0f;
        ++frags;
        continue;
      }
      if(f->front_to_back)
      {  // src rgb premultiplied (:303); C = Cs*(1-Ad) + Cd ; A = As*(1-Ad) + Ad
        const float oma = 1.0f - dst[3];
        dst[0]          = (P.rgba[0] * opacity) * oma + dst[0];
        dst[1]          = (P.rgba[1] * opacity) * oma + dst[1];
        dst[2]          = (P.rgba[2] * opacity) * oma + dst[2];
        dst[3]          = opacity * oma + dst[3];
      }
      else
      {  // C = Cs*As + Cd*(1-As) ; A = As + Ad   (:308, cpp:2081-2086)
        const float oma = 1.0f - opacity;
        dst[0]          = P.rgba[0] * opacity + dst[0] * oma;
        dst[1]          = P.rgba[1] * opacity + dst[1] * oma;
        dst[2]          = P.rgba[2] * opacity + dst[2] * oma;
        dst[3]          = opacity + dst[3];
      }
      if(f->target_fp16)  // RGBA16F colour target (gaussian_splatting.h:338,340)
        for(int c = 0; c < 4; ++c)
          dst[c] = orc_half_to_float(orc_float_to_half(dst[c]));
      ++frags;
    }
  }
  return frags;
}

uint64_t orc_render_order(const OrcFrame* f, const OrcInstance* inst, int n_inst, const uint32_t* ids, uint32_t v,
                          float* rgba_out, uint64_t* stats)
{
  std::memset(rgba_out, 0, (size_t)f->width * f->height * 4 * sizeof(float));  // clear (0,0,0,0)
  std::vector<uint32_t> offsets(n_inst + 1, 0);
  for(int k = 0; k < n_inst; ++k)
    offsets[k + 1] = offsets[k] + inst[k].count;
  uint64_t frags = 0, quads = 0;
  std::vector<float> depth;
  if(f->stochastic)
    depth.assign((size_t)f->width * f->height, 1.0f);
  for(uint32_t s = 0; s < v; ++s)
  {
    const uint32_t g = ids[s];
    int            k = 0;
    while(k + 1 < n_inst && g >= offsets[k + 1])
      ++k;
    OrcProjected P;
    orc_project(f, &inst[k], g - offsets[k], &P);
    if(!P.valid)
      continue;
    ++quads;
    frags += raster_one(f, P, rgba_out, nullptr, f->stochastic ? depth.data() : nullptr, g);
  }
  if(stats)
  {
    stats[0] = v;
    stats[1] = quads;
  }
  return frags;
}

// =======================================================================================================
// 3DGUT: unscented-transform projection + per-pixel particle response
// =======================================================================================================
namespace {
const float GUT_DELTA = 1.73205080757f, GUT_LAMBDA = 0.0f, GUT_ALPHA = 1.0f, GUT_BETA = 2.0f;  // threedgut_definitions.h.slang:45-50
const int   GUT_D = 3;
const float GUT_IN_IMAGE_MARGIN_FACTOR = 0.1f, GUT_COVARIANCE_DILATION = 0.3f, GUT_ALPHA_THRESHOLD = 0.01f;  // :51-57

struct GutSensor
{
  int   fisheye;
  float focal[2], principal[2], resolution[2], maxAngle;
};

// threedgut_camera_projections.h.slang:32-44
float stableNorm2(float x, float y)
{
  const float ax = std::fabs(x), ay = std::fabs(y);
  const float mn = std::min(ax, ay), mx = std::max(ax, ay);
  if(mx <= 0.0f)
    return 0.0f;
  const float r = mn / mx;
  return mx * std::sqrt(1.0f + r * r);
}
// :78-83
bool withinResolution(const float res[2], float tol, const float p[2])
{
  const float mx = res[0] * tol, my = res[1] * tol;
  return (p[0] > -mx) && (p[1] > -my) && (p[0] < res[0] + mx) && (p[1] < res[1] + my);
}
// projectPointPinhole with every distortion coefficient zero (initPerfectPinholeCamera, camera_models.h.slang:65-83):
// icD = 1, delta = 0 -> uvND = uv; :85-137
bool projectPinhole(const GutSensor& S, const float pos[3], float out[2])
{
  if(pos[2] <= 0.0f)
  {
    out[0] = out[1] = 0.0f;
    return false;
  }
  const float u = pos[0] / pos[2], v = pos[1] / pos[2];
  out[0] = u * S.focal[0] + S.principal[0];
  out[1] = v * S.focal[1] + S.principal[1];
  return withinResolution(S.resolution, GUT_IN_IMAGE_MARGIN_FACTOR, out);
}
// projectPointFisheye with radialCoeffs = 0 (initPerfectFisheyeCamera, camera_models.h.slang:120-136): delta = theta / rho; :151-176
bool projectFisheye(const GutSensor& S, const float pos[3], float out[2])
{
  const float eps = 1e-7f;
  const float rho = std::max(stableNorm2(pos[0], pos[1]), eps);
  const float thetaFull = std::atan2(rho, pos[2]);
  const float theta = std::min(thetaFull, S.maxAngle);
  const float theta2 = theta * theta;
  const float delta = (theta * (0.0f * theta2 + 1.0f)) / rho;
  out[0] = S.focal[0] * pos[0] * delta + S.principal[0];
  out[1] = S.focal[1] * pos[1] * delta + S.principal[1];
  return (theta < S.maxAngle) && withinResolution(S.resolution, GUT_IN_IMAGE_MARGIN_FACTOR, out);
}

GutSensor makeSensor(const OrcFrame* f)
{
  GutSensor S;
  S.fisheye       = f->camera_model == 1;
  S.resolution[0] = (float)f->width;
  S.resolution[1] = (float)f->height;
  S.principal[0]  = S.resolution[0] / 2.0f;
  S.principal[1]  = S.resolution[1] / 2.0f;
  if(S.fisheye)
  {  // gaussian_splatting.cpp:1243: focal = (1,-1) * viewport / fovRad
    S.focal[0] = S.resolution[0] / f->fov_rad;
    S.focal[1] = -S.resolution[1] / f->fov_rad;
  }
  else
  {  // :1248-1250
    S.focal[0] = f->proj[0] * 0.5f * S.resolution[0];
    S.focal[1] = f->proj[5] * 0.5f * S.resolution[1];
  }
  // computeMaxAngle, camera_models.h.slang:87-118 (principal point at the centre: max distance = the half size)
  const float mdx = S.resolution[0] - S.principal[0], mdy = S.resolution[1] - S.principal[1];
  const float maxR = std::sqrt(mdx * mdx + mdy * mdy);
  S.maxAngle = std::max(2.0f * maxR / S.focal[0], 2.0f * maxR / S.focal[1]) / 2.0f;
  return S;
}

// projectPointWithShutter, global shutter (:186-201): the RUB -> RUF flip of position, translation and quaternion is
// F (R p + t) with F = diag(1,1,-1), i.e. the view-space point with z negated
bool projectWorldPoint(const OrcFrame* f, const GutSensor& S, const float world[3], float out[2])
{
  const float p[4] = {world[0], world[1], world[2], 1.0f};
  float       v[4];
  mat4_mul_vec4(f->view, p, v);
  const float cam[3] = {v[0], v[1], -v[2]};
  return S.fisheye ? projectFisheye(S, cam, out) : projectPinhole(S, cam, out);
}
}  // namespace

void orc_project_gut(const OrcFrame* f, const OrcInstance* Ip, uint32_t i, OrcGutProjected* out)
{
  const OrcInstance& I = *Ip;
  std::memset(out, 0, sizeof(*out));
  // threedgut_raster.mesh.slang:116-122 — colour, centre, exp(scale), rotation matrix of the normalised quaternion
  float rgba[4] = {I.rgba[4 * i + 0], I.rgba[4 * i + 1], I.rgba[4 * i + 2], I.rgba[4 * i + 3]};
  const float p[3] = {I.centers[3 * i + 0], I.centers[3 * i + 1], I.centers[3 * i + 2]};
  const float sc[3] = {std::exp(I.scales[3 * i + 0]), std::exp(I.scales[3 * i + 1]), std::exp(I.scales[3 * i + 2])};
  float qw = I.rotations[4 * i + 0], qx = I.rotations[4 * i + 1], qy = I.rotations[4 * i + 2], qz = I.rotations[4 * i + 3];
  {
    const float ql = std::sqrt(((qw * qw + qx * qx) + qy * qy) + qz * qz);
    qw /= ql; qx /= ql; qy /= ql; qz /= ql;
  }
  // quatToMat3, quaternions.h.slang:39-58 — rows of the Slang matrix (used as mul(v, M)); row i = i-th principal axis
  const float xx = qx * qx, yy = qy * qy, zz = qz * qz, xy = qx * qy, xz = qx * qz, yz = qy * qz, wx = qw * qx, wy = qw * qy, wz = qw * qz;
  const float R[3][3] = {{1.0f - 2.0f * (yy + zz), 2.0f * (xy + wz), 2.0f * (xz - wy)},
                         {2.0f * (xy - wz), 1.0f - 2.0f * (xx + zz), 2.0f * (yz + wx)},
                         {2.0f * (xz + wy), 2.0f * (yz - wx), 1.0f - 2.0f * (xx + yy)}};
  // SH in model coordinates (:142-148), then the alpha cull (:150-155)
  const float cam[4] = {f->camera_pos[0], f->camera_pos[1], f->camera_pos[2], 1.0f};
  float       camM[4];
  mat4_mul_vec4(I.transform_inv, cam, camM);
  float       dir[3] = {p[0] - camM[0], p[1] - camM[1], p[2] - camM[2]};
  const float dl     = std::sqrt((dir[0] * dir[0] + dir[1] * dir[1]) + dir[2] * dir[2]);
  dir[0] /= dl; dir[1] /= dl; dir[2] /= dl;
  if(f->debug_flags & 2)
    rgba[0] = rgba[1] = rgba[2] = 0.5f;
  float sh[3];
  sh_radiance(I, i, f->sh_degree, dir, sh);
  rgba[0] += sh[0]; rgba[1] += sh[1]; rgba[2] += sh[2];
  if(rgba[3] < f->alpha_cull_threshold)
    return;
  // threedgutParticleProjection, threedgut.h.slang:26-110
  const GutSensor S = makeSensor(f);
  float           sp[2 * GUT_D + 1][2];
  int             numValid = 0;
  auto projectModel = [&](const float m[3], float o[2]) {
    const float mp[4] = {m[0], m[1], m[2], 1.0f};
    float       w[4];
    mat4_mul_vec4(I.transform, mp, w);
    return projectWorldPoint(f, S, w, o);
  };
  if(projectModel(p, sp[0]))
    ++numValid;
  float       c[2]    = {sp[0][0] * (GUT_LAMBDA / (GUT_D + GUT_LAMBDA)), sp[0][1] * (GUT_LAMBDA / (GUT_D + GUT_LAMBDA))};
  const float weightI = 1.0f / (2.0f * (GUT_D + GUT_LAMBDA));
  for(int a = 0; a < GUT_D; ++a)
  {
    const float d[3] = {GUT_DELTA * sc[a] * R[a][0], GUT_DELTA * sc[a] * R[a][1], GUT_DELTA * sc[a] * R[a][2]};
    const float pp[3] = {p[0] + d[0], p[1] + d[1], p[2] + d[2]}, pm[3] = {p[0] - d[0], p[1] - d[1], p[2] - d[2]};
    if(projectModel(pp, sp[a + 1]))
      ++numValid;
    c[0] += weightI * sp[a + 1][0];
    c[1] += weightI * sp[a + 1][1];
    if(projectModel(pm, sp[a + 1 + GUT_D]))
      ++numValid;
    c[0] += weightI * sp[a + 1 + GUT_D][0];
    c[1] += weightI * sp[a + 1 + GUT_D][1];
  }
  if(numValid == 0)  // GUT_REQUIRE_ALL_SIGMA_POINTS_VALID false
    return;
  float cov[3];
  {
    const float cx = sp[0][0] - c[0], cy = sp[0][1] - c[1];
    const float w0 = GUT_LAMBDA / (GUT_D + GUT_LAMBDA) + (1.0f - GUT_ALPHA * GUT_ALPHA + GUT_BETA);
    cov[0] = w0 * (cx * cx);
    cov[1] = w0 * (cx * cy);
    cov[2] = w0 * (cy * cy);
  }
  for(int a = 0; a < 2 * GUT_D; ++a)
  {
    const float cx = sp[a + 1][0] - c[0], cy = sp[a + 1][1] - c[1];
    cov[0] += weightI * (cx * cx);
    cov[1] += weightI * (cx * cy);
    cov[2] += weightI * (cy * cy);
  }
  if(f->extent_method == 1)
  {  // threedgutProjectedExtentConicOpacity, threedgut.h.slang:113-160 (TIGHT_OPACITY_BOUNDING, RECT_BOUNDING)
    const float dx = cov[0] + GUT_COVARIANCE_DILATION, dy = cov[1], dz = cov[2] + GUT_COVARIANCE_DILATION;
    const float det = dx * dz - dy * dy;
    if(det == 0.0f)
      return;
    float w = rgba[3];
    if(f->ms_antialiasing)
    {
      const float covDet = cov[0] * cov[2] - cov[1] * cov[1];
      w = rgba[3] * std::sqrt(std::max(0.000025f, covDet / det));
    }
    if(w < GUT_ALPHA_THRESHOLD)
      return;
    const float maxPower = std::log(w / GUT_ALPHA_THRESHOLD);
    const float factor   = std::min(3.33f, std::sqrt(2.0f * maxPower));
    const float mid      = 0.5f * (dx + dz);
    const float lambda   = mid + std::sqrt(std::max(0.01f, mid * mid - det));
    const float radius   = factor * std::sqrt(lambda);
    const float ex = std::min(factor * std::sqrt(dx), radius), ey = std::min(factor * std::sqrt(dz), radius);
    if(!(radius > 0.0f))
      return;
    if(f->ms_antialiasing)
      rgba[3] = w;  // mesh.slang:193-195
    out->half1[0] = ex; out->half1[1] = 0.f;
    out->half2[0] = 0.f; out->half2[1] = ey;
  }
  else
  {  // threedgsProjectedExtentBasis(cov, 3.33, splatScale, a, ...), threedgs.h.slang:60-121
    float a = cov[0], b = cov[1], d = cov[2], detOrig = 0.f;
    if(f->ms_antialiasing)
      detOrig = a * d - b * b;
    a += 0.3f;
    d += 0.3f;
    if(f->ms_antialiasing)
      rgba[3] *= std::sqrt(std::max(detOrig / (a * d - b * b), 0.0f));
    const float D = a * d - b * b, half = 0.5f * (a + d);
    const float term2 = std::sqrt(std::max(0.1f, half * half - D));
    float ev1 = half + term2, ev2 = half - term2;
    if(ev2 <= 0.0f)
      return;
    if(f->debug_flags & 1)
      ev1 = ev2 = 0.2f;
    float       e1[2] = {(std::fabs(b) < 0.001f) ? 1.0f : b, ev1 - a};
    const float el    = std::sqrt(e1[0] * e1[0] + e1[1] * e1[1]);
    e1[0] /= el; e1[1] /= el;
    const float l1 = std::min(3.33f * std::sqrt(ev1), 2048.0f), l2 = std::min(3.33f * std::sqrt(ev2), 2048.0f);
    out->half1[0] = e1[0] * f->splat_scale * l1; out->half1[1] = e1[1] * f->splat_scale * l1;
    out->half2[0] = e1[1] * f->splat_scale * l2; out->half2[1] = -e1[0] * f->splat_scale * l2;
  }
  // quad centre: the UT mean in pixels; depth from the (pinhole) projection matrix, "a coarse approx" for fisheye (:205-214)
  float MV[16], viewC[4], clip[4];
  orc_mat4_mul(f->view, I.transform, MV);
  const float p4[4] = {p[0], p[1], p[2], 1.0f};
  mat4_mul_vec4(MV, p4, viewC);
  mat4_mul_vec4(f->proj, viewC, clip);
  const float ndcz = clip[2] / clip[3];
  if(!(ndcz >= 0.0f && ndcz <= 1.0f))  // fixed-function clip of the quad emitted at z = ndc.z, w = 1 (see orc_project)
    return;
  out->ndc_z        = ndcz;
  out->center_px[0] = c[0];
  out->center_px[1] = c[1];
  std::memcpy(out->rgba, rgba, sizeof(rgba));
  std::memcpy(out->position, p, sizeof(p));
  std::memcpy(out->scale, sc, sizeof(sc));
  for(int r = 0; r < 3; ++r)  // splatInvRotation = transpose(splatRotation)
    for(int k = 0; k < 3; ++k)
      out->inv_rot[3 * r + k] = R[k][r];
  out->valid = 1;
}

// threedgut_raster.frag.slang:87-127 for pixel (px, py): ray, model-space ray, particleProcessHitGut
int orc_gut_fragment(const OrcFrame* f, const OrcInstance* I, const OrcGutProjected* P, int px, int py, float* opacity)
{
  return orc_gut_fragment_iso(f, I, P, px, py, opacity, 0.0f, nullptr);
}

int orc_gut_fragment_iso(const OrcFrame* f, const OrcInstance* I, const OrcGutProjected* P, int px, int py, float* opacity,
                         float thinThreshold, float* normalWorld)
{
  float viewInv[16], projInv[16];
  orc_mat4_inverse(f->view, viewInv);   // gaussian_splatting.cpp:1166
  orc_mat4_inverse(f->proj, projInv);   // :1200
  float       ro[3], rd[3];
  const float posx = (float)px + 0.5f, posy = (float)py + 0.5f;  // SV_Position of the fragment
  const float o4[4] = {0.f, 0.f, 0.f, 1.f};
  float       t4[4];
  mat4_mul_vec4(viewInv, o4, t4);
  ro[0] = t4[0]; ro[1] = t4[1]; ro[2] = t4[2];
  if(f->camera_model == 1)
  {  // generateFisheyeRay(input.position.xy, viewport, fovRad, principalPoint = 0, viewInverse), cameras.h.slang:46-82
    const float u = (posx / ((float)f->width - 1.0f)) * 2.0f - 1.0f, v = (posy / ((float)f->height - 1.0f)) * 2.0f - 1.0f;
    const float r = std::sqrt(u * u + v * v);
    if(r > 1.0f)
      return 0;  // out of fov: discard
    float phiCos = std::fabs(r) > 1e-9f ? u / r : 0.0f;
    phiCos       = std::min(std::max(phiCos, -1.0f), 1.0f);
    float phi    = std::acos(phiCos);
    phi          = v < 0.0f ? -phi : phi;
    const float theta = r * f->fov_rad * 0.5f;
    const float d4[4] = {std::cos(phi) * std::sin(theta), -std::sin(phi) * std::sin(theta), -std::cos(theta), 0.f};
    mat4_mul_vec4(viewInv, d4, t4);
  }
  else
  {  // generatePinholeRay(input.position.xy, float2(0.5), ...), cameras.h.slang:27-43 — the 0.5 is added to SV_Position literally
    const float inu = (posx + 0.5f) / (float)f->width, inv = (posy + 0.5f) / (float)f->height;
    const float d4[4] = {inu * 2.0f - 1.0f, inv * 2.0f - 1.0f, 1.f, 1.f};
    float       target[4];
    mat4_mul_vec4(projInv, d4, target);
    const float tg[4] = {target[0], target[1], target[2], 0.f};
    mat4_mul_vec4(viewInv, tg, t4);
  }
  {
    const float l = std::sqrt((t4[0] * t4[0] + t4[1] * t4[1]) + t4[2] * t4[2]);
    rd[0] = t4[0] / l; rd[1] = t4[1] / l; rd[2] = t4[2] / l;
  }
  if(f->dof_mode != 0)
  {  // frag.slang:104-109 + depthOfField, cameras.h.slang:85-108
    uint32_t    seed = orc_xxhash32((uint32_t)(int)posx, (uint32_t)(int)posy, (uint32_t)f->frame_sample_id);
    const float fp[3] = {rd[0] * f->focus_dist, rd[1] * f->focus_dist, rd[2] * f->focus_dist};
    const float r1 = orc_rand(&seed) * 6.28318530717958647692f;  // M_TWO_PI
    const float r2 = orc_rand(&seed) * f->aperture;
    const float e1[4] = {1.f, 0.f, 0.f, 0.f}, e2[4] = {0.f, 1.f, 0.f, 0.f};
    float       right[4], up[4];
    mat4_mul_vec4(viewInv, e1, right);
    mat4_mul_vec4(viewInv, e2, up);
    const float c = std::cos(r1), sn = std::sin(r1), sq = std::sqrt(r2);
    const float ap[3] = {(c * right[0] + sn * up[0]) * sq, (c * right[1] + sn * up[1]) * sq, (c * right[2] + sn * up[2]) * sq};
    float       nd[3] = {fp[0] - ap[0], fp[1] - ap[1], fp[2] - ap[2]};
    const float l = std::sqrt((nd[0] * nd[0] + nd[1] * nd[1]) + nd[2] * nd[2]);
    for(int c3 = 0; c3 < 3; ++c3)
    {
      ro[c3] += ap[c3];
      rd[c3] = nd[c3] / l;
    }
  }
  // model-space ray (:113-118)
  const float ro4[4] = {ro[0], ro[1], ro[2], 1.f};
  float       mo[4];
  mat4_mul_vec4(I->transform_inv, ro4, mo);
  float md[3];
  for(int r = 0; r < 3; ++r)
    md[r] = (rd[0] * m_at(I->transform_inv, r, 0) + rd[1] * m_at(I->transform_inv, r, 1)) + rd[2] * m_at(I->transform_inv, r, 2);
  {
    const float l = std::sqrt((md[0] * md[0] + md[1] * md[1]) + md[2] * md[2]);
    md[0] /= l; md[1] /= l; md[2] /= l;
  }
  // particleProcessHitGut, threedgrt.h.slang:238-278
  const float density = P->rgba[3];
  if(density <= f->alpha_cull_threshold)
    return 0;
  // particleCannonicalRay, :57-75 — mul(v, invRotation) = sum_r v[r] * row r
  const float g[3] = {mo[0] - P->position[0], mo[1] - P->position[1], mo[2] - P->position[2]};
  float       gr[3], dr[3];
  for(int k = 0; k < 3; ++k)
  {
    gr[k] = (g[0] * P->inv_rot[0 + k] + g[1] * P->inv_rot[3 + k]) + g[2] * P->inv_rot[6 + k];
    dr[k] = (md[0] * P->inv_rot[0 + k] + md[1] * P->inv_rot[3 + k]) + md[2] * P->inv_rot[6 + k];
  }
  const float pro[3] = {gr[0] / P->scale[0], gr[1] / P->scale[1], gr[2] / P->scale[2]};
  float       prd[3] = {dr[0] / P->scale[0], dr[1] / P->scale[1], dr[2] / P->scale[2]};
  {
    const float l = std::sqrt((prd[0] * prd[0] + prd[1] * prd[1]) + prd[2] * prd[2]);
    prd[0] /= l; prd[1] /= l; prd[2] /= l;
  }
  // particleRayMinSquaredDistance, :77-81
  const float cr[3] = {prd[1] * pro[2] - prd[2] * pro[1], prd[2] * pro[0] - prd[0] * pro[2], prd[0] * pro[1] - prd[1] * pro[0]};
  const float dist2 = (cr[0] * cr[0] + cr[1] * cr[1]) + cr[2] * cr[2];
  // particleRayMaxKernelResponse<KERNEL_DEGREE>(rayDist), threedgrt.h.slang:83-127 (rayDist is the squared distance)
  float maxResponse;
  switch(f->kernel_degree)
  {
    case 8: { const float sq = dist2 * dist2; maxResponse = std::exp(-0.000685871056241f * sq * sq); break; }
    case 5: maxResponse = std::exp(-0.0185185185185f * dist2 * dist2 * std::sqrt(dist2)); break;
    case 4: maxResponse = std::exp(-0.0555555555556f * dist2 * dist2); break;
    case 3: maxResponse = std::exp(-0.166666666667f * dist2 * std::sqrt(dist2)); break;
    case 1: maxResponse = std::exp(-1.5f * std::sqrt(dist2)); break;
    case 0: maxResponse = std::max(1.0f + -0.329630334487f * std::sqrt(dist2), 0.0f); break;
    default: maxResponse = std::exp(-0.5f * dist2); break;  // quadratic
  }
  const float alpha       = std::min(f->alpha_clamp, maxResponse * density);
  const bool  accept      = ((double)alpha > (double)(1.0f / 255.0f)) && (maxResponse > f->kernel_min_response);
  if(!accept)
    return 0;
  *opacity = (f->debug_flags & 4) ? 1.0f : alpha;
  if(normalWorld)
  {  // computeEllipsoidNormal, threedgrt.h.slang:423-497
    const float maxScale  = std::max(std::max(P->scale[0], P->scale[1]), P->scale[2]);
    const float flatness  = std::max(0.02f * maxScale, thinThreshold);
    const int   small[3]  = {P->scale[0] < flatness, P->scale[1] < flatness, P->scale[2] < flatness};
    const int   smallCount = small[0] + small[1] + small[2];
    const float local[3]  = {g[0], g[1], g[2]};  // modelRayOrigin - particle.position
    float       nm[3];
    bool        haveNormal = false;
    if(smallCount == 0)
    {  // raySphereIntersection(particleRayOrigin, particleRayDirection, 3.0, 0, INF), :502-540
      const float a = (prd[0] * prd[0] + prd[1] * prd[1]) + prd[2] * prd[2];
      const float b = 2.0f * ((pro[0] * prd[0] + pro[1] * prd[1]) + pro[2] * prd[2]);
      const float c = ((pro[0] * pro[0] + pro[1] * pro[1]) + pro[2] * pro[2]) - 3.0f * 3.0f;
      const float disc = b * b - 4.0f * a * c;
      if(disc >= 0.0f)
      {
        const float sq = std::sqrt(disc), invA = 1.0f / (2.0f * a);
        const float t1 = (-b - sq) * invA, t2 = (-b + sq) * invA;
        float       t  = -1.0f;
        if(t1 >= 0.0f)
          t = t1;
        else if(t2 >= 0.0f)
          t = t2;
        if(t >= 0.0f)
        {
          float h[3] = {pro[0] + t * prd[0], pro[1] + t * prd[1], pro[2] + t * prd[2]};
          const float hl = std::sqrt((h[0] * h[0] + h[1] * h[1]) + h[2] * h[2]);
          const float ns[3] = {h[0] / hl / P->scale[0], h[1] / hl / P->scale[1], h[2] / hl / P->scale[2]};
          // mul(normalScaled, rotMat), rotMat = transpose(invRotation)
          for(int k = 0; k < 3; ++k)
            nm[k] = (ns[0] * P->inv_rot[3 * k] + ns[1] * P->inv_rot[3 * k + 1]) + ns[2] * P->inv_rot[3 * k + 2];
          const float l = std::sqrt((nm[0] * nm[0] + nm[1] * nm[1]) + nm[2] * nm[2]);
          nm[0] /= l; nm[1] /= l; nm[2] /= l;
          haveNormal = true;
        }
      }
    }
    else if(smallCount == 1)
    {
      const int a = small[0] ? 0 : (small[1] ? 1 : 2);
      for(int k = 0; k < 3; ++k)
        nm[k] = P->inv_rot[3 * k + a];  // mul(axisLocal, rotMat): row a of transpose(invRotation)
      if((nm[0] * local[0] + nm[1] * local[1]) + nm[2] * local[2] < 0.0f)
        for(int k = 0; k < 3; ++k)
          nm[k] = -nm[k];
      haveNormal = true;
    }
    if(!haveNormal)
      for(int k = 0; k < 3; ++k)
        nm[k] = -md[k];  // -modelRayDirection
    // normalize(mul(normalModel, modelToWorldRS)), :337-345
    const float* M = I->transform;
    float        nw[3];
    for(int r = 0; r < 3; ++r)
      nw[r] = M[r] * nm[0] + M[4 + r] * nm[1] + M[8 + r] * nm[2];
    const float l = std::sqrt((nw[0] * nw[0] + nw[1] * nw[1]) + nw[2] * nw[2]);
    for(int r = 0; r < 3; ++r)
      normalWorld[r] = nw[r] / l;
  }
  return 1;
}

uint64_t orc_render_gut_order(const OrcFrame* f, const OrcInstance* inst, int n_inst, const uint32_t* ids, uint32_t v,
                              float* rgba_out, uint64_t* stats)
{
  const int W = f->width, H = f->height;
  std::memset(rgba_out, 0, (size_t)W * H * 4 * sizeof(float));
  std::vector<uint32_t> offsets(n_inst + 1, 0);
  for(int k = 0; k < n_inst; ++k)
    offsets[k + 1] = offsets[k] + inst[k].count;
  uint64_t frags = 0, quads = 0;
  std::vector<float> gutDepth;
  if(f->stochastic)
    gutDepth.assign((size_t)W * H, 1.0f);
  for(uint32_t s = 0; s < v; ++s)
  {
    const uint32_t g = ids[s];
    int            k = 0;
    while(k + 1 < n_inst && g >= offsets[k + 1])
      ++k;
    OrcGutProjected P;
    orc_project_gut(f, &inst[k], g - offsets[k], &P);
    if(!P.valid)
      continue;
    ++quads;
    // pixel centres covered by the quad centre +- half1 +- half2
    const float ex = std::fabs(P.half1[0]) + std::fabs(P.half2[0]), ey = std::fabs(P.half1[1]) + std::fabs(P.half2[1]);
    const float fx0 = P.center_px[0] - ex - 0.5f, fx1 = P.center_px[0] + ex - 0.5f;
    const float fy0 = P.center_px[1] - ey - 0.5f, fy1 = P.center_px[1] + ey - 0.5f;
    if(!(fx1 >= 0.f && fy1 >= 0.f && fx0 <= (float)(W - 1) && fy0 <= (float)(H - 1)))
      continue;
    const int   x0 = (int)std::max(0.0f, std::floor(fx0)), x1 = (int)std::min((float)(W - 1), std::ceil(fx1));
    const int   y0 = (int)std::max(0.0f, std::floor(fy0)), y1 = (int)std::min((float)(H - 1), std::ceil(fy1));
    const float n1 = P.half1[0] * P.half1[0] + P.half1[1] * P.half1[1], n2 = P.half2[0] * P.half2[0] + P.half2[1] * P.half2[1];
    for(int y = y0; y <= y1; ++y)
      for(int x = x0; x <= x1; ++x)
      {
        const float dx = ((float)x + 0.5f) - P.center_px[0], dy = ((float)y + 0.5f) - P.center_px[1];
        const float u = (dx * P.half1[0] + dy * P.half1[1]) / n1, w = (dx * P.half2[0] + dy * P.half2[1]) / n2;
        if(std::fabs(u) > 1.0f || std::fabs(w) > 1.0f)
          continue;  // outside the quad: no fragment
        float opacity;
        if(!orc_gut_fragment(f, &inst[k], &P, x, y, &opacity))
          continue;
        float* dst = rgba_out + ((size_t)y * W + x) * 4;
        if(f->stochastic)
        {  // threedgut_raster.frag.slang:150-172 (same seed chain and depth state as the 3DGS fragment shader)
          uint32_t seed = orc_xxhash32((uint32_t)x, (uint32_t)y, (uint32_t)f->frame_sample_id);
          seed          = orc_xxhash32(seed, g, 2u * (g & 31u) + (u > w ? 0u : 1u));
          if(!(orc_rand(&seed) < opacity))
            continue;
          float& dz = gutDepth[(size_t)y * W + x];
          if(!(P.ndc_z <= dz))
            continue;
          dz     = P.ndc_z;
          dst[0] = P.rgba[0];
          dst[1] = P.rgba[1];
          dst[2] = P.rgba[2];
          dst[3] = 1.0f;
          ++frags;
          continue;
        }
        if(f->front_to_back)
        {
          const float oma = 1.0f - dst[3];
          dst[0] = (P.rgba[0] * opacity) * oma + dst[0];
          dst[1] = (P.rgba[1] * opacity) * oma + dst[1];
          dst[2] = (P.rgba[2] * opacity) * oma + dst[2];
          dst[3] = opacity * oma + dst[3];
        }
        else
        {
          const float oma = 1.0f - opacity;
          dst[0] = P.rgba[0] * opacity + dst[0] * oma;
          dst[1] = P.rgba[1] * opacity + dst[1] * oma;
          dst[2] = P.rgba[2] * opacity + dst[2] * oma;
          dst[3] = opacity + dst[3];
        }
        if(f->target_fp16)
          for(int c = 0; c < 4; ++c)
            dst[c] = orc_half_to_float(orc_float_to_half(dst[c]));
        ++frags;
      }
  }
  if(stats)
  {
    stats[0] = v;
    stats[1] = quads;
  }
  return frags;
}

// orc_render_order restricted to the inclusive pixel window {x0,y0,x1,y1}; rgba_out is the window's own buffer.
// Used to pin full-size frames with a few crops (a whole 5.8 M-splat frame is ~7 G fragments on one core).
uint64_t orc_render_window(const OrcFrame* f, const OrcInstance* inst, int n_inst, const uint32_t* ids, uint32_t v,
                           const int win[4], float* rgba_out)
{
  const int ww = win[2] - win[0] + 1, wh = win[3] - win[1] + 1;
  std::memset(rgba_out, 0, (size_t)ww * wh * 4 * sizeof(float));
  std::vector<uint32_t> offsets(n_inst + 1, 0);
  for(int k = 0; k < n_inst; ++k)
    offsets[k + 1] = offsets[k] + inst[k].count;
  uint64_t frags = 0;
  for(uint32_t s = 0; s < v; ++s)
  {
    const uint32_t g = ids[s];
    int            k = 0;
    while(k + 1 < n_inst && g >= offsets[k + 1])
      ++k;
    OrcProjected P;
    orc_project(f, &inst[k], g - offsets[k], &P);
    if(P.valid)
      frags += raster_one(f, P, rgba_out, win);
  }
  return frags;
}

// NEED_SURFACE_INFO + FRONT_TO_BACK side output of the fragment shader (threedgs_raster.frag.slang:320-349):
// depthTransmittanceBuffer starts at (depth 0, transmittance 1); every fragment that passes the discards does
//   transmittance *= (1 - opacity);  if(depth == 0 && transmittance < depthIsoThreshold) depth = fragCoord.z;
// `ids` is the FRONT-TO-BACK draw order.  id_out receives the global id of the splat that set the depth
// (0xFFFFFFFF where none did) — the build's reading of "splat id" for a picked depth.
// shaders/octahedral_normal.h.slang:27-87
static void octWrap(const float v[2], float out[2])
{
  out[0] = (1.0f - std::fabs(v[1])) * (v[0] >= 0.0f ? 1.0f : -1.0f);
  out[1] = (1.0f - std::fabs(v[0])) * (v[1] >= 0.0f ? 1.0f : -1.0f);
}

uint32_t orc_oct_encode(const float n[3])
{
  const float inv  = 1.0f / (std::fabs(n[0]) + std::fabs(n[1]) + std::fabs(n[2]));
  float       p[2] = {n[0] * inv, n[1] * inv};
  if(n[2] < 0.0f)
  {
    float w[2];
    octWrap(p, w);
    p[0] = w[0];
    p[1] = w[1];
  }
  const uint32_t x = (uint32_t)std::min(std::max((p[0] * 0.5f + 0.5f) * 65535.0f, 0.0f), 65535.0f);
  const uint32_t y = (uint32_t)std::min(std::max((p[1] * 0.5f + 0.5f) * 65535.0f, 0.0f), 65535.0f);
  return (y << 16) | x;
}

void orc_oct_decode(uint32_t packed, float out[3])
{
  const float fx = (float)(packed & 0xFFFFu) / 65535.0f * 2.0f - 1.0f;
  const float fy = (float)(packed >> 16) / 65535.0f * 2.0f - 1.0f;
  float       n[3] = {fx, fy, 1.0f - std::fabs(fx) - std::fabs(fy)};
  if(n[2] < 0.0f)
  {
    float w[2];
    octWrap(n, w);
    n[0] = w[0];
    n[1] = w[1];
  }
  const float l = std::sqrt(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]);
  out[0] = n[0] / l;
  out[1] = n[1] / l;
  out[2] = n[2] / l;
}

// threedgs_raster.mesh.slang:209-235 ; threedgrt.h.slang:42-48 (particle), :358-419 (max density plane)
void orc_splat_normal(const OrcFrame* f, const OrcInstance* I, uint32_t i, float thinThreshold, int quantize, float out[3])
{
  const float* p  = &I->centers[3 * (size_t)i];
  const float* ls = &I->scales[3 * (size_t)i];
  const float* rq = &I->rotations[4 * (size_t)i];
  const float  scale[3] = {std::exp(ls[0]), std::exp(ls[1]), std::exp(ls[2])};
  // vec4toQuat(normalize(fetchRotation)): stored scalar first
  const float ql = std::sqrt(rq[0] * rq[0] + rq[1] * rq[1] + rq[2] * rq[2] + rq[3] * rq[3]);
  const float w = rq[0] / ql, x = rq[1] / ql, y = rq[2] / ql, z = rq[3] / ql;
  // quatToMat3Transpose (quaternions.h.slang:56-73), rows as written; mul(v, M) is row-vector times matrix
  const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, xz = x * z, yz = y * z, wx = w * x, wy = w * y, wz = w * z;
  const float invRot[9] = {1.0f - 2.0f * (yy + zz), 2.0f * (xy - wz), 2.0f * (xz + wy),
                           2.0f * (xy + wz), 1.0f - 2.0f * (xx + zz), 2.0f * (yz - wx),
                           2.0f * (xz - wy), 2.0f * (yz + wx), 1.0f - 2.0f * (xx + yy)};
  // modelCameraPos = mul(float4(cameraPosition,1), transformInverse).xyz  (glm memory: column-major)
  const float* Mi = I->transform_inv;
  float        cam[3];
  for(int r = 0; r < 3; ++r)
    cam[r] = Mi[r] * f->camera_pos[0] + Mi[4 + r] * f->camera_pos[1] + Mi[8 + r] * f->camera_pos[2] + Mi[12 + r];
  const float local[3] = {cam[0] - p[0], cam[1] - p[1], cam[2] - p[2]};  // modelRayOrigin - particle.position
  const int   small[3] = {scale[0] < thinThreshold, scale[1] < thinThreshold, scale[2] < thinThreshold};
  const int   smallCount = small[0] + small[1] + small[2];
  float       nm[3];
  if(smallCount == 0)
  {
    float canon[3], sv[3];
    for(int c = 0; c < 3; ++c)
      canon[c] = local[0] * invRot[c] + local[1] * invRot[3 + c] + local[2] * invRot[6 + c];
    for(int c = 0; c < 3; ++c)
      sv[c] = canon[c] * (1.0f / (scale[c] * scale[c]));
    float g[3];  // mul(scaledVector, transpose(invRotation))
    for(int c = 0; c < 3; ++c)
      g[c] = sv[0] * invRot[3 * c] + sv[1] * invRot[3 * c + 1] + sv[2] * invRot[3 * c + 2];
    const float rl = 1.0f / std::sqrt(g[0] * g[0] + g[1] * g[1] + g[2] * g[2]);
    for(int c = 0; c < 3; ++c)
      nm[c] = g[c] * rl;
    if(nm[0] * local[0] + nm[1] * local[1] + nm[2] * local[2] < 0.0f)
      for(int c = 0; c < 3; ++c)
        nm[c] = -nm[c];
  }
  else if(smallCount == 1)
  {
    const int a = small[0] ? 0 : (small[1] ? 1 : 2);
    for(int c = 0; c < 3; ++c)
      nm[c] = invRot[3 * c + a];  // mul(axisLocal, rotMat): row a of transpose(invRotation)
    if(nm[0] * local[0] + nm[1] * local[1] + nm[2] * local[2] < 0.0f)
      for(int c = 0; c < 3; ++c)
        nm[c] = -nm[c];
  }
  else
  {  // -modelRayDir, modelRayDir = normalize(splatCenter - modelCameraPos)
    const float l = std::sqrt(local[0] * local[0] + local[1] * local[1] + local[2] * local[2]);
    for(int c = 0; c < 3; ++c)
      nm[c] = local[c] / l;
  }
  // normalize(mul(float4(normalModel,0), transform).xyz)
  const float* M = I->transform;
  float        nw[3];
  for(int r = 0; r < 3; ++r)
    nw[r] = M[r] * nm[0] + M[4 + r] * nm[1] + M[8 + r] * nm[2];
  const float l = std::sqrt(nw[0] * nw[0] + nw[1] * nw[1] + nw[2] * nw[2]);
  for(int r = 0; r < 3; ++r)
    nw[r] /= l;
  if(quantize)
    orc_oct_decode(orc_oct_encode(nw), out);
  else
    for(int r = 0; r < 3; ++r)
      out[r] = nw[r];
}

void orc_render_surface(const OrcFrame* f, const OrcInstance* inst, int n_inst, const uint32_t* ids, uint32_t v,
                        float depth_iso_threshold, float thin_particle_threshold, int quantize_normals, float* depth_out,
                        uint32_t* id_out, float* normal_out)
{
  const int          W = f->width, H = f->height;
  const size_t       np = (size_t)W * H;
  std::vector<float> trans(np, 1.0f);
  std::fill(depth_out, depth_out + np, 0.0f);
  std::fill(id_out, id_out + np, 0xFFFFFFFFu);
  if(normal_out)
    std::fill(normal_out, normal_out + 4 * np, 0.0f);
  std::vector<uint32_t> offsets(n_inst + 1, 0);
  for(int k = 0; k < n_inst; ++k)
    offsets[k + 1] = offsets[k] + inst[k].count;
  for(uint32_t s = 0; s < v; ++s)
  {
    const uint32_t g = ids[s];
    int            k = 0;
    while(k + 1 < n_inst && g >= offsets[k + 1])
      ++k;
    OrcProjected P;
    orc_project(f, &inst[k], g - offsets[k], &P);
    if(!P.valid)
      continue;
    const float ex = std::fabs(P.basis1[0]) + std::fabs(P.basis2[0]);
    const float ey = std::fabs(P.basis1[1]) + std::fabs(P.basis2[1]);
    const float fx0 = P.center_px[0] - ex - 0.5f, fx1 = P.center_px[0] + ex - 0.5f;
    const float fy0 = P.center_px[1] - ey - 0.5f, fy1 = P.center_px[1] + ey - 0.5f;
    if(!(fx1 >= 0.f && fy1 >= 0.f && fx0 <= (float)(W - 1) && fy0 <= (float)(H - 1)))
      continue;
    const int   x0 = (int)std::max(0.0f, std::floor(fx0)), x1 = (int)std::min((float)(W - 1), std::ceil(fx1));
    const int   y0 = (int)std::max(0.0f, std::floor(fy0)), y1 = (int)std::min((float)(H - 1), std::ceil(fy1));
    const float n1 = P.basis1[0] * P.basis1[0] + P.basis1[1] * P.basis1[1];
    const float n2 = P.basis2[0] * P.basis2[0] + P.basis2[1] * P.basis2[1];
    float       nrm[3] = {0.f, 0.f, 0.f};
    if(normal_out)
      orc_splat_normal(f, &inst[k], g - offsets[k], thin_particle_threshold, quantize_normals, nrm);
    for(int y = y0; y <= y1; ++y)
      for(int x = x0; x <= x1; ++x)
      {
        const float dx = ((float)x + 0.5f) - P.center_px[0];
        const float dy = ((float)y + 0.5f) - P.center_px[1];
        const float u  = (dx * P.basis1[0] + dy * P.basis1[1]) / n1;
        const float vv = (dx * P.basis2[0] + dy * P.basis2[1]) / n2;
        const float px = u * kSqrt8, py = vv * kSqrt8;
        const float A  = px * px + py * py;
        if(A > 8.0f)
          continue;
        const float opacity = P.opacity_disabled ? 1.0f : std::exp(-0.5f * A) * P.rgba[3];
        if(opacity <= 1.0f / 255.0f)
          continue;
        const size_t i = (size_t)y * W + x;
        if(normal_out)
        {  // src = (normal*opacity, opacity); dst = src * (1 - dst.a) + dst   (gaussian_splatting.cpp:2095-2107)
          float*      d  = &normal_out[4 * i];
          const float om = 1.0f - d[3];
          d[0] += nrm[0] * opacity * om;
          d[1] += nrm[1] * opacity * om;
          d[2] += nrm[2] * opacity * om;
          d[3] += opacity * om;
        }
        trans[i] *= (1.0f - opacity);
        if(depth_out[i] == 0.0f && trans[i] < depth_iso_threshold)
        {
          depth_out[i] = P.ndc_z;
          id_out[i]    = g;
        }
      }
  }
}

// 3DGUT with NEED_SURFACE_INFO, FTB (threedgut_raster.frag.slang:127-131,195-228; particleProcessHitGutWithNormal,
// threedgrt.h.slang:281-345): ids nearest first.  The max-density-plane normal depends on the ray ORIGIN only (per splat
// without depth of field) except for particles with two degenerate axes, whose normal is minus the pixel's ray; it is
// not quantised (computed in the fragment shader).
void orc_render_surface_gut(const OrcFrame* f, const OrcInstance* inst, int n_inst, const uint32_t* ids, uint32_t v,
                            float depth_iso_threshold, float thin_particle_threshold, float* depth_out, uint32_t* id_out,
                            float* normal_out)
{
  const int          W = f->width, H = f->height;
  const size_t       np = (size_t)W * H;
  std::vector<float> trans(np, 1.0f);
  std::fill(depth_out, depth_out + np, 0.0f);
  std::fill(id_out, id_out + np, 0xFFFFFFFFu);
  if(normal_out)
    std::fill(normal_out, normal_out + 4 * np, 0.0f);
  std::vector<uint32_t> offsets(n_inst + 1, 0);
  for(int k = 0; k < n_inst; ++k)
    offsets[k + 1] = offsets[k] + inst[k].count;
  for(uint32_t s = 0; s < v; ++s)
  {
    const uint32_t g = ids[s];
    int            k = 0;
    while(k + 1 < n_inst && g >= offsets[k + 1])
      ++k;
    OrcGutProjected P;
    orc_project_gut(f, &inst[k], g - offsets[k], &P);
    if(!P.valid)
      continue;
    const float ex = std::fabs(P.half1[0]) + std::fabs(P.half2[0]), ey = std::fabs(P.half1[1]) + std::fabs(P.half2[1]);
    const float fx0 = P.center_px[0] - ex - 0.5f, fx1 = P.center_px[0] + ex - 0.5f;
    const float fy0 = P.center_px[1] - ey - 0.5f, fy1 = P.center_px[1] + ey - 0.5f;
    if(!(fx1 >= 0.f && fy1 >= 0.f && fx0 <= (float)(W - 1) && fy0 <= (float)(H - 1)))
      continue;
    const int   x0 = (int)std::max(0.0f, std::floor(fx0)), x1 = (int)std::min((float)(W - 1), std::ceil(fx1));
    const int   y0 = (int)std::max(0.0f, std::floor(fy0)), y1 = (int)std::min((float)(H - 1), std::ceil(fy1));
    const float n1 = P.half1[0] * P.half1[0] + P.half1[1] * P.half1[1], n2 = P.half2[0] * P.half2[0] + P.half2[1] * P.half2[1];
    float       nrm[3] = {0.f, 0.f, 0.f};
    if(normal_out && f->normal_method != 1)
      orc_splat_normal(f, &inst[k], g - offsets[k], thin_particle_threshold, 0, nrm);
    for(int y = y0; y <= y1; ++y)
      for(int x = x0; x <= x1; ++x)
      {
        const float dx = ((float)x + 0.5f) - P.center_px[0], dy = ((float)y + 0.5f) - P.center_px[1];
        const float u = (dx * P.half1[0] + dy * P.half1[1]) / n1, w = (dx * P.half2[0] + dy * P.half2[1]) / n2;
        if(std::fabs(u) > 1.0f || std::fabs(w) > 1.0f)
          continue;
        float opacity;
        const bool iso = normal_out && f->normal_method == 1;
        if(!orc_gut_fragment_iso(f, &inst[k], &P, x, y, &opacity, thin_particle_threshold, iso ? nrm : nullptr))
          continue;
        const size_t i = (size_t)y * W + x;
        if(normal_out)
        {
          float*      d  = &normal_out[4 * i];
          const float om = 1.0f - d[3];
          d[0] += nrm[0] * opacity * om;
          d[1] += nrm[1] * opacity * om;
          d[2] += nrm[2] * opacity * om;
          d[3] += opacity * om;
        }
        trans[i] *= (1.0f - opacity);
        if(depth_out[i] == 0.0f && trans[i] < depth_iso_threshold)
        {
          depth_out[i] = P.ndc_z;
          id_out[i]    = g;
        }
      }
  }
}

uint64_t orc_render(const OrcFrame* f, const OrcInstance* inst, int n_inst, float* rgba_out, uint64_t* stats)
{
  size_t total = 0;
  for(int k = 0; k < n_inst; ++k)
    total += inst[k].count;
  std::vector<uint32_t> keys(total), ids(total);
  const uint32_t        v = orc_key_cull(f, inst, n_inst, keys.data(), ids.data());
  orc_sort_stable(keys.data(), ids.data(), v);
  return orc_render_order(f, inst, n_inst, ids.data(), v, rgba_out, stats);
}

// shaders/image_compare_metric.comp.slang:116-130 ; src/image_compare.cpp:869-893
double orc_psnr_rgb(const float* a, const float* b, int width, int height)
{
  double       se = 0.0;
  const size_t n  = (size_t)width * height;
  for(size_t i = 0; i < n; ++i)
    for(int c = 0; c < 3; ++c)
    {
      const double d = (double)a[4 * i + c] - (double)b[4 * i + c];
      se += d * d;
    }
  const double mse = se / ((double)n * 3.0);
  if(mse <= 0.0)
    return 99.99;
  return std::min(99.99, 10.0 * std::log10(1.0 / mse));
}

// ---------------------------------------------------------------------------------------
// src/splat_sorter_async.cpp:92-141 ; parallel loop = nvutils::parallel_batches_pooled<8192>
// (src/utilities.h:52-59, nvpro_core2 absent: restated as fixed 8192-element batches handed
// to a pool of `threads` workers).
int orc_cpu_sort(const float dir[3], const float cop[3], const OrcSortInstance* inst, int n_inst, uint32_t total,
                 int front_to_back, int threads, float* distances, uint32_t* indices, double* dist_ms, double* sort_ms)
{
  if(n_inst <= 0)
    return -1;
  const auto  t0       = std::chrono::high_resolution_clock::now();
  const float plane[4] = {dir[0], dir[1], dir[2], -dir[0] * cop[0] - dir[1] * cop[1] - dir[2] * cop[2]};
  const float divider  = 1.0f / std::sqrt(plane[0] * plane[0] + plane[1] * plane[1] + plane[2] * plane[2]);
  if(threads <= 0)
    threads = (int)std::max(1u, std::thread::hardware_concurrency());
  for(int k = 0; k < n_inst; ++k)
  {
    const OrcSortInstance& I = inst[k];
    if(!I.positions)
      continue;
    const uint32_t batch   = 8192;
    const uint32_t nBatch  = (I.count + batch - 1) / batch;
    auto           worker  = [&](int tid) {
      for(uint32_t bi = (uint32_t)tid; bi < nBatch; bi += (uint32_t)threads)
      {
        const uint32_t b = bi * batch, e = std::min(I.count, b + batch);
        for(uint32_t s = b; s < e; ++s)
        {
          const float v[4] = {I.positions[3 * s], I.positions[3 * s + 1], I.positions[3 * s + 2], 1.0f};
          float       p[4];
          mat4_mul_vec4(I.transform, v, p);
          const float dist = std::fabs(plane[0] * p[0] + plane[1] * p[1] + plane[2] * p[2] + plane[3]) * divider;
          distances[I.global_offset + s] = dist;
          indices[I.global_offset + s]   = I.global_offset + s;
        }
      }
    };
    std::vector<std::thread> pool;
    for(int t = 1; t < threads; ++t)
      pool.emplace_back(worker, t);
    worker(0);
    for(auto& th : pool)
      th.join();
  }
  const auto t1 = std::chrono::high_resolution_clock::now();
  if(front_to_back)
    std::sort(std::execution::par_unseq, indices, indices + total,
              [&](uint32_t i, uint32_t j) { return distances[i] < distances[j]; });
  else
    std::sort(std::execution::par_unseq, indices, indices + total,
              [&](uint32_t i, uint32_t j) { return distances[i] > distances[j]; });
  const auto t2 = std::chrono::high_resolution_clock::now();
  if(dist_ms)
    *dist_ms = std::chrono::duration<double, std::milli>(t1 - t0).count();
  if(sort_ms)
    *sort_ms = std::chrono::duration<double, std::milli>(t2 - t1).count();
  return 0;
}

}  // extern "C"
