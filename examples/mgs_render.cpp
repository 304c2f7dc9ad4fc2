// mgs_render — the C-ABI caller a C++ host would write (the reference is a C++ application; INTEGRATION.md shows
// the same calls wrapped as its MgsBackend).  Plain C++17, links only against libmgs.so:
//   make -C examples        (g++ -std=c++17 -I include examples/mgs_render.cpp -L vk_gaussian_splatting_amd/csrc -lmgs)
//   examples/mgs_render scene.ply|scene.spz|scene.splat out.ppm [W H] [eye x y z] [instances n]
// Loads the file (PlyLoaderAsync semantics), adds n instances on a row (createInstance + computeTransform), commits with
// the reference's default storage (uint8 SH + RGBA, parameters.h:88-89), renders one frame with the default camera
// (camera_set.h:48-53) and writes a binary PPM (linear RGB clamped to [0,1], no tonemap: the screenshot path).
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <cmath>
#include <vector>

#include "mgs.h"

static float halfToFloat(uint16_t h)
{
  const uint32_t s = (h >> 15) & 1u, e = (h >> 10) & 31u, m = h & 1023u;
  uint32_t       bits;
  if(e == 0)
  {
    if(m == 0)
      bits = s << 31;
    else
    {  // subnormal
      int      ee = -1;
      uint32_t mm = m;
      do
      {
        ++ee;
        mm <<= 1;
      } while((mm & 1024u) == 0);
      bits = (s << 31) | ((uint32_t)(127 - 15 - ee) << 23) | ((mm & 1023u) << 13);
    }
  }
  else if(e == 31)
    bits = (s << 31) | 0x7F800000u | (m << 13);
  else
    bits = (s << 31) | ((e + 127 - 15) << 23) | (m << 13);
  float f;
  std::memcpy(&f, &bits, 4);
  return f;
}

#define CHECK(call)                                                                                    \
  do                                                                                                   \
  {                                                                                                    \
    const int rc_ = (call);                                                                            \
    if(rc_ != MGS_OK)                                                                                  \
    {                                                                                                  \
      std::fprintf(stderr, "%s failed (%d): %s\n", #call, rc_, mgs_last_error());                      \
      return 1;                                                                                        \
    }                                                                                                  \
  } while(0)

int main(int argc, char** argv)
{
  if(argc < 3)
  {
    std::fprintf(stderr,
                 "usage: %s scene.ply|.spz|.splat out.ppm [W H] [ex ey ez] [instances] [mode]\n"
                 "  mode: 3dgs (default) | 3dgut | fisheye | stochastic[:samples] | dof[:samples]  (the last two accumulate `samples` frames)\n%s\n",
                 argv[0], mgs_version());
    return 2;
  }
  const int   W = argc > 4 ? std::atoi(argv[3]) : 1920, H = argc > 4 ? std::atoi(argv[4]) : 1080;
  const float eye[3]    = {argc > 7 ? (float)std::atof(argv[5]) : 1.7f, argc > 7 ? (float)std::atof(argv[6]) : 1.5f,
                           argc > 7 ? (float)std::atof(argv[7]) : 1.7f};
  const int   instances = argc > 8 ? std::max(1, std::atoi(argv[8])) : 1;
  const std::string mode = argc > 9 ? argv[9] : "3dgs";
  const size_t      colon = mode.find(':');
  const std::string kind  = mode.substr(0, colon);
  const int         samples = colon == std::string::npos ? 1 : std::max(1, std::atoi(mode.c_str() + colon + 1));

  MgsSplatSet set = nullptr;
  CHECK(mgs_splatset_load(argv[1], &set));
  MgsScene scene = nullptr;
  CHECK(mgs_scene_create(0, &scene));
  for(int i = 0; i < instances; ++i)
  {
    const float scale[3] = {1, 1, 1}, rot[3] = {0, 0, 0}, pos[3] = {3.0f * (float)i, 0, 0};
    float       M[16], Minv[16];
    mgs_compute_transform(scale, rot, pos, M, Minv);
    int id = -1;
    CHECK(mgs_instance_add(scene, set, M, &id));
  }
  mgs_splatset_destroy(set);  // the scene keeps its own reference
  CHECK(mgs_scene_commit(scene, MGS_FORMAT_UINT8, MGS_FORMAT_UINT8));

  MgsFrameParams p;
  mgs_frame_params_default(&p);
  const float ctr[3] = {0, 0, 0}, up[3] = {0, 1, 0};
  mgs_camera_lookat_perspective(eye, ctr, up, 60.0f, 0.1f, 2000.0f, W, H, 0, p.view, p.proj);
  std::memcpy(p.camera_pos, eye, sizeof(eye));
  p.width           = W;
  p.height          = H;
  p.collect_timings = 1;
  // the other pipelines and modes of the C ABI: the 3DGUT raster pipeline (pinhole / fisheye), stochastic splats and depth of
  // field, the latter two averaged over `samples` frames by the library (temporal_sampling = post.comp.slang)
  if(kind == "3dgut" || kind == "fisheye" || kind == "dof")
    p.pipeline = MGS_PIPELINE_3DGUT;
  if(kind == "fisheye")
  {
    p.camera_model = MGS_CAMERA_FISHEYE;
    p.fov_rad      = 2.6f;
  }
  if(kind == "stochastic")
    p.sort_mode = MGS_SORT_STOCHASTIC;
  if(kind == "dof")
  {
    p.dof_mode   = MGS_DOF_FIXED_FOCUS;
    p.focus_dist = std::sqrt(eye[0] * eye[0] + eye[1] * eye[1] + eye[2] * eye[2]);
    p.aperture   = 0.02f;
  }
  p.temporal_sampling = samples > 1 ? 1 : 0;
  MgsFrameOut out;
  for(int k = 0; k < samples; ++k)
  {
    p.frame_sample_id = k;
    CHECK(mgs_render(scene, &p, &out));
  }
  std::vector<uint16_t> img((size_t)W * H * 4);
  CHECK(mgs_frame_download(scene, img.data(), img.size() * sizeof(uint16_t)));
  std::printf("%llu splats, %u in frustum, %u sorted, %llu bin-list entries, %u shaded pairs, %.3f ms on the GPU\n",
              (unsigned long long)mgs_scene_splat_count(scene), out.frustum_count, out.sorted_count,
              (unsigned long long)out.tile_pairs, out.shaded_count, out.stage_ms[MGS_STAGE_TOTAL]);

  FILE* f = std::fopen(argv[2], "wb");
  if(!f)
  {
    std::perror(argv[2]);
    return 1;
  }
  std::fprintf(f, "P6\n%d %d\n255\n", W, H);
  std::vector<uint8_t> row((size_t)W * 3);
  for(int y = 0; y < H; ++y)
  {
    for(int x = 0; x < W; ++x)
      for(int c = 0; c < 3; ++c)
      {
        const float v          = std::min(1.0f, std::max(0.0f, halfToFloat(img[((size_t)y * W + x) * 4 + c])));
        row[(size_t)x * 3 + c] = (uint8_t)(v * 255.0f + 0.5f);
      }
    std::fwrite(row.data(), 1, row.size(), f);
  }
  std::fclose(f);
  mgs_scene_destroy(scene);
  return 0;
}
