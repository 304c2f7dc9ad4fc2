// mgs_strips — the multi-GPU caller a C++ host would write: one process per GPU, no MPI, no torch.  Plain C++17, links only
// against libmgs.so (RCCL is resolved inside the library on first use):
//   make -C examples
//   examples/mgs_strips scene.ply out.ppm RANK WORLD RENDEZVOUS_FILE [W H]
// Every rank loads the same file (replicated splat buffers), rank 0 creates the RCCL unique id
// (mgs_comm_unique_id) and publishes it through RENDEZVOUS_FILE, the others wait for it; then every rank renders its
// tile-row strip of the same frame and mgs_render_gathered exchanges the strips in place, so that EVERY rank holds the
// whole frame (rank 0 writes it).  WORLD = 1 runs the same code path on one GPU (what the -m gpu test does; two ranks
// on one device are refused by RCCL).  Strips are cost-balanced from the per-row list lengths of one full frame.
#include <algorithm>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "mgs.h"

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

static float halfToFloat(uint16_t h)
{
  const uint32_t s = (h >> 15) & 1u, e = (h >> 10) & 31u, m = h & 1023u;
  if(e == 0)
    return (s ? -1.0f : 1.0f) * (float)m * (1.0f / 16777216.0f);  // subnormal: m * 2^-24
  uint32_t bits = (e == 31) ? ((s << 31) | 0x7F800000u | (m << 13)) : ((s << 31) | ((e + 112u) << 23) | (m << 13));
  float    f;
  std::memcpy(&f, &bits, 4);
  return f;
}

int main(int argc, char** argv)
{
  if(argc < 6)
  {
    std::fprintf(stderr, "usage: mgs_strips scene.ply|.spz|.splat out.ppm RANK WORLD RENDEZVOUS_FILE [W H]\n");
    return 2;
  }
  const std::string path = argv[1], out = argv[2], rdv = argv[5];
  const int         rank = std::atoi(argv[3]), world = std::atoi(argv[4]);
  const int         W = argc > 7 ? std::atoi(argv[6]) : 1280, H = argc > 7 ? std::atoi(argv[7]) : 720;

  // ---- rendezvous: 128 bytes from rank 0 to everybody, through a file renamed into place ----
  unsigned char id[MGS_COMM_ID_BYTES];
  if(rank == 0)
  {
    CHECK(mgs_comm_unique_id(id));
    const std::string tmp = rdv + ".tmp";
    FILE*             f   = std::fopen(tmp.c_str(), "wb");
    if(!f || std::fwrite(id, 1, sizeof(id), f) != sizeof(id))
      return 1;
    std::fclose(f);
    std::rename(tmp.c_str(), rdv.c_str());
  }
  else
  {
    for(int tries = 0;; ++tries)
    {
      FILE* f = std::fopen(rdv.c_str(), "rb");
      if(f && std::fread(id, 1, sizeof(id), f) == sizeof(id))
      {
        std::fclose(f);
        break;
      }
      if(f)
        std::fclose(f);
      if(tries > 6000)
      {
        std::fprintf(stderr, "rank %d: no rendezvous file\n", rank);
        return 1;
      }
      std::this_thread::sleep_for(std::chrono::milliseconds(10));
    }
  }

  MgsSplatSet set;
  CHECK(mgs_splatset_load(path.c_str(), &set));
  MgsScene scene;
  CHECK(mgs_scene_create(/*device*/ world == 1 ? 0 : rank, &scene));
  const float I[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
  int         inst;
  CHECK(mgs_instance_add(scene, set, I, &inst));
  CHECK(mgs_scene_commit(scene, MGS_FORMAT_FLOAT32, MGS_FORMAT_FLOAT32));
  CHECK(mgs_scene_comm_init(scene, rank, world, id));  // collective

  MgsFrameParams p;
  mgs_frame_params_default(&p);
  p.width = W;
  p.height = H;
  const float eye[3] = {4.0f, 1.5f, 0.0f}, ctr[3] = {0, 0, 0}, up[3] = {0, 1, 0};
  mgs_camera_lookat_perspective(eye, ctr, up, 60.0f, 0.1f, 2000.0f, W, H, 0, p.view, p.proj);
  std::memcpy(p.camera_pos, eye, sizeof(eye));

  // cost-balanced strip table: every rank renders the same full frame once and derives the same table
  MgsFrameOut fo;
  CHECK(mgs_render(scene, &p, &fo));
  const int             rows = (H + 15) / 16;
  std::vector<uint32_t> cost((size_t)rows);
  CHECK(mgs_frame_row_costs(scene, cost.data(), cost.size()));
  std::vector<int32_t> bounds((size_t)world + 1, 0);
  {
    double total = 0, mean = 0;
    for(uint32_t c : cost)
      mean += c;
    mean = std::max(mean / rows, 1.0);
    std::vector<double> cum((size_t)rows + 1, 0.0);
    for(int r = 0; r < rows; ++r)
      cum[(size_t)r + 1] = cum[(size_t)r] + cost[(size_t)r] + 0.25 * mean;  // every row costs at least its pixels
    total = cum[(size_t)rows];
    for(int g = 1; g < world; ++g)
    {
      const double target = total * g / world;
      int          b      = (int)(std::lower_bound(cum.begin(), cum.end(), target) - cum.begin());
      bounds[(size_t)g]   = std::max(bounds[(size_t)g - 1], std::min(b, rows));
    }
    bounds[(size_t)world] = rows;
  }
  CHECK(mgs_scene_set_strip_rows(scene, bounds.data(), world + 1));

  // this rank's strip + the exchange: afterwards the frame buffer holds the whole frame on every rank
  CHECK(mgs_render_gathered(scene, &p, &fo));
  CHECK(mgs_sync(scene));
  std::vector<uint16_t> img((size_t)W * H * 4);
  CHECK(mgs_frame_download(scene, img.data(), img.size() * 2));
  if(rank == 0)
  {
    FILE* f = std::fopen(out.c_str(), "wb");
    if(!f)
      return 1;
    std::fprintf(f, "P6\n%d %d\n255\n", W, H);
    std::vector<unsigned char> row((size_t)W * 3);
    for(int y = 0; y < H; ++y)  // rows in frame order (row 0 = NDC y -1), like examples/mgs_render
    {
      for(int x = 0; x < W; ++x)
        for(int c = 0; c < 3; ++c)
          row[(size_t)x * 3 + c] = (unsigned char)(std::min(std::max(halfToFloat(img[((size_t)y * W + x) * 4 + c]), 0.0f), 1.0f) * 255.0f + 0.5f);
      std::fwrite(row.data(), 1, row.size(), f);
    }
    std::fclose(f);
    std::printf("rank 0 of %d: frame %dx%d assembled from strips", world, W, H);
    for(int g = 0; g <= world; ++g)
      std::printf("%s%d", g ? "," : " [", bounds[(size_t)g]);
    std::printf("] written to %s\n", out.c_str());
  }
  CHECK(mgs_scene_comm_destroy(scene));
  mgs_scene_destroy(scene);
  mgs_splatset_destroy(set);
  return 0;
}
