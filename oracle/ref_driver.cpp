/*
 * ref_driver.cpp — TEST INFRASTRUCTURE ONLY.  Thin extern "C" glue (ours) around the parts of the
 * reference that compile from their own sources with plain g++:
 *   /root/reference/3rdparty/miniply/miniply.cpp, /root/reference/3rdparty/spz/src/cc/*.cc (-lz),
 *   /root/reference/src/splat_set.h.
 * The reference sources are compiled WHERE THEY LIE (see oracle/Makefile); nothing is copied.
 * Output goes to oracle/_ref/ only.  The glue repeats the property extraction of
 * src/ply_loader_async.cpp:383-441 and the SPZ re-layout of :304-353 (that file itself needs
 * Vulkan headers through splat_set_vk.h and is unbuildable here).
 *
 * NOT built: src/splat_sorter_async.cpp (needs glm, fmt, nvutils, nvvk = nvpro_core2, absent) —
 * unbuildable here without stand-in headers, so the CPU-sorter oracle stays a restatement.
 */
#include <cstring>
#include <string>
#include <vector>

#include "miniply.h"
#include "load-spz.h"
#include "splat_set.h"

using vk_gaussian_splatting::SplatSet;

static std::vector<float>* field(SplatSet* s, int which)
{
  switch(which)
  {
    case 0: return &s->positions;
    case 1: return &s->f_dc;
    case 2: return &s->f_rest;
    case 3: return &s->opacity;
    case 4: return &s->scale;
    case 5: return &s->rotation;
  }
  return nullptr;
}

extern "C" {

void* ref_ply_load(const char* path)
{
  miniply::PLYReader reader(path);
  if(!reader.valid())
    return nullptr;
  auto*    out = new SplatSet();
  uint32_t indices[45];
  bool     gsFound = false;
  while(reader.has_element() && !gsFound)
  {
    if(reader.element_is(miniply::kPLYVertexElement) && reader.load_element())
    {
      const uint32_t numVerts = reader.num_rows();
      if(numVerts == 0)
        continue;
      if(reader.find_properties(indices, 45, "f_rest_0", "f_rest_1", "f_rest_2", "f_rest_3", "f_rest_4", "f_rest_5",
                                "f_rest_6", "f_rest_7", "f_rest_8", "f_rest_9", "f_rest_10", "f_rest_11", "f_rest_12",
                                "f_rest_13", "f_rest_14", "f_rest_15", "f_rest_16", "f_rest_17", "f_rest_18",
                                "f_rest_19", "f_rest_20", "f_rest_21", "f_rest_22", "f_rest_23", "f_rest_24",
                                "f_rest_25", "f_rest_26", "f_rest_27", "f_rest_28", "f_rest_29", "f_rest_30",
                                "f_rest_31", "f_rest_32", "f_rest_33", "f_rest_34", "f_rest_35", "f_rest_36",
                                "f_rest_37", "f_rest_38", "f_rest_39", "f_rest_40", "f_rest_41", "f_rest_42",
                                "f_rest_43", "f_rest_44"))
      {
        out->f_rest.resize(size_t(numVerts) * 45);
        reader.extract_properties(indices, 45, miniply::PLYPropertyType::Float, out->f_rest.data());
      }
      if(reader.find_properties(indices, 3, "x", "y", "z"))
      {
        out->positions.resize(size_t(numVerts) * 3);
        reader.extract_properties(indices, 3, miniply::PLYPropertyType::Float, out->positions.data());
      }
      if(reader.find_properties(indices, 1, "opacity"))
      {
        out->opacity.resize(numVerts);
        reader.extract_properties(indices, 1, miniply::PLYPropertyType::Float, out->opacity.data());
      }
      if(reader.find_properties(indices, 3, "scale_0", "scale_1", "scale_2"))
      {
        out->scale.resize(size_t(numVerts) * 3);
        reader.extract_properties(indices, 3, miniply::PLYPropertyType::Float, out->scale.data());
      }
      if(reader.find_properties(indices, 4, "rot_0", "rot_1", "rot_2", "rot_3"))
      {
        out->rotation.resize(size_t(numVerts) * 4);
        reader.extract_properties(indices, 4, miniply::PLYPropertyType::Float, out->rotation.data());
      }
      if(reader.find_properties(indices, 3, "f_dc_0", "f_dc_1", "f_dc_2"))
      {
        out->f_dc.resize(size_t(numVerts) * 3);
        reader.extract_properties(indices, 3, miniply::PLYPropertyType::Float, out->f_dc.data());
      }
      gsFound = true;
    }
    reader.next_element();
  }
  if(!gsFound)
  {
    delete out;
    return nullptr;
  }
  out->convertCoordinates(spz::CoordinateSystem::RDF, spz::CoordinateSystem::RUB);
  return out;
}

void* ref_spz_load(const char* path)
{
  spz::UnpackOptions options;
  options.to               = spz::CoordinateSystem::RUB;
  spz::GaussianCloud cloud = spz::loadSpz(std::string(path), options);
  if(cloud.numPoints == 0)
    return nullptr;
  auto* out = new SplatSet();
  out->positions.swap(cloud.positions);
  out->rotation.resize(cloud.rotations.size());
  const uint32_t n = uint32_t(out->positions.size() / 3);
  for(uint32_t i = 0; i < n; i++)
  {
    out->rotation[4 * i + 0] = cloud.rotations[4 * i + 3];
    out->rotation[4 * i + 1] = cloud.rotations[4 * i + 0];
    out->rotation[4 * i + 2] = cloud.rotations[4 * i + 1];
    out->rotation[4 * i + 3] = cloud.rotations[4 * i + 2];
  }
  out->scale.swap(cloud.scales);
  out->opacity.swap(cloud.alphas);
  out->f_dc               = cloud.colors;
  const size_t shCoefs    = cloud.sh.size() / n / 3;
  out->f_rest.resize(cloud.sh.size());
  for(size_t i = 0; i < n; i++)
  {
    const size_t offset = i * shCoefs * 3;
    for(size_t c = 0; c < 3; ++c)
      for(size_t j = 0; j < shCoefs; j++)
        out->f_rest[offset + shCoefs * c + j] = cloud.sh[(i * shCoefs + j) * 3 + c];
  }
  return out;
}

// writes an .spz with the reference's own packer (input in spz conventions: rot xyzw, sh [coef][rgb])
int ref_spz_save(const char* path, int n, int shDegree, const float* pos, const float* scales, const float* rot_xyzw,
                 const float* alphas, const float* colors, const float* sh, int from_coord)
{
  spz::GaussianCloud g;
  g.numPoints = n;
  g.shDegree  = shDegree;
  const int shDim = shDegree == 0 ? 0 : shDegree == 1 ? 3 : shDegree == 2 ? 8 : 15;
  g.positions.assign(pos, pos + 3 * size_t(n));
  g.scales.assign(scales, scales + 3 * size_t(n));
  g.rotations.assign(rot_xyzw, rot_xyzw + 4 * size_t(n));
  g.alphas.assign(alphas, alphas + size_t(n));
  g.colors.assign(colors, colors + 3 * size_t(n));
  if(shDim)
    g.sh.assign(sh, sh + size_t(n) * shDim * 3);
  spz::PackOptions po;
  po.from = (spz::CoordinateSystem)from_coord;
  return spz::saveSpz(g, po, std::string(path)) ? 0 : -1;
}

size_t ref_set_size(void* h, int which) { return field((SplatSet*)h, which)->size(); }
void   ref_set_copy(void* h, int which, float* out)
{
  auto* v = field((SplatSet*)h, which);
  std::memcpy(out, v->data(), v->size() * sizeof(float));
}
int  ref_set_max_sh_degree(void* h) { return ((SplatSet*)h)->maxShDegree(); }
void ref_set_free(void* h) { delete(SplatSet*)h; }

// SplatSet::maxShDegree on a synthetic set with the given sizes (src/splat_set.h:52-74)
int ref_max_sh_degree(size_t f_rest_len, size_t splat_count)
{
  SplatSet s;
  s.positions.resize(splat_count * 3);
  s.f_rest.resize(f_rest_len);
  return s.maxShDegree();
}

void ref_flip_sh(int from, int to, float* out15, float* flipP3, float* flipQ3)
{
  spz::CoordinateConverter c = spz::coordinateConverter((spz::CoordinateSystem)from, (spz::CoordinateSystem)to);
  for(int i = 0; i < 15; ++i)
    out15[i] = c.flipSh[i];
  for(int i = 0; i < 3; ++i)
  {
    flipP3[i] = c.flipP[i];
    flipQ3[i] = c.flipQ[i];
  }
}
}
