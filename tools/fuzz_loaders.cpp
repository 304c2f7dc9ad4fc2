// Mutation fuzzing of the file readers (csrc/host_model.cpp: .ply / .spz / .splat) under AddressSanitizer + UBSan.
// The readers take untrusted files at the C-ABI boundary (mgs_splatset_load); a reader must return an error or a
// well-formed set, never crash, over-read or allocate without bound.  Build + run: tools/fuzz_loaders.sh
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <random>
#include <string>
#include <vector>
#include "../vk_gaussian_splatting_amd/csrc/host_model.h"

static std::vector<unsigned char> readAll(const std::string& p)
{
  std::ifstream f(p, std::ios::binary);
  return std::vector<unsigned char>((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
}

int main(int argc, char** argv)
{
  if(argc < 4)
  {
    std::fprintf(stderr, "usage: fuzz_loaders <seed file> <iterations> <tmp path with the right extension>\n");
    return 2;
  }
  const std::vector<unsigned char> seed = readAll(argv[1]);
  const int         iters = std::atoi(argv[2]);
  const std::string tmp   = argv[3];
  const std::string ext   = tmp.substr(tmp.find_last_of('.') + 1);
  std::mt19937_64   rng(12345);
  size_t            ok = 0, rejected = 0;
  for(int it = 0; it < iters; ++it)
  {
    std::vector<unsigned char> d = seed;
    const int kind = (int)(rng() % 6);
    if(kind == 0 && !d.empty())  // truncate
      d.resize(rng() % d.size());
    else if(kind == 1)           // flip bytes in the first 512 (headers)
      for(int k = 0; k < 1 + (int)(rng() % 8); ++k)
        d[rng() % std::min<size_t>(d.size(), 512)] ^= (unsigned char)(1u << (rng() % 8));
    else if(kind == 2)           // flip bytes anywhere
      for(int k = 0; k < 1 + (int)(rng() % 32); ++k)
        d[rng() % d.size()] = (unsigned char)rng();
    else if(kind == 3)           // overwrite a 4-byte field with an extreme value
    {
      const uint32_t v[] = {0u, 1u, 0x7FFFFFFFu, 0x80000000u, 0xFFFFFFFFu, 0xFFFFFFF0u, 4000000000u};
      const uint32_t x   = v[rng() % 7];
      const size_t   o   = rng() % (d.size() > 4 ? d.size() - 4 : 1);
      std::memcpy(d.data() + o, &x, std::min<size_t>(4, d.size() - o));
    }
    else if(kind == 4)           // replace a decimal number in the header by a huge / negative one
    {
      const std::string h(d.begin(), d.begin() + std::min<size_t>(d.size(), 2048));
      const size_t      p = h.find("vertex ");
      if(p != std::string::npos)
      {
        const char* reps[] = {"4000000000", "-5", "0", "99999999999999999999", "1e9", ""};
        const std::string r = reps[rng() % 6];
        size_t            e = p + 7;
        while(e < h.size() && h[e] != '\n')
          ++e;
        d.erase(d.begin() + p + 7, d.begin() + e);
        d.insert(d.begin() + p + 7, r.begin(), r.end());
      }
    }
    else                         // append garbage
      for(int k = 0; k < (int)(rng() % 64); ++k)
        d.push_back((unsigned char)rng());
    {
      std::ofstream f(tmp, std::ios::binary | std::ios::trunc);
      f.write(reinterpret_cast<const char*>(d.data()), (std::streamsize)d.size());
    }
    mgs::HostSplatSet s;
    int               rc;
    try
    {
      rc = ext == "ply" ? mgs::loadPly(tmp, s) : (ext == "spz" ? mgs::loadSpz(tmp, s) : mgs::loadSplat(tmp, s));
    }
    catch(const std::exception&)
    {  // the C ABI's guarded() turns these into MGS_ERR_*: acceptable here, crashes and sanitizer reports are not
      rc = -1;
    }
    if(rc == 0)
    {
      const size_t n = s.size();
      if(s.f_dc.size() != 3 * n || s.opacity.size() != n || s.scale.size() != 3 * n || s.rotation.size() != 4 * n
         || (n && s.f_rest.size() % n != 0))
      {
        std::fprintf(stderr, "iteration %d: inconsistent set accepted (n=%zu)\n", it, n);
        return 1;
      }
      ++ok;
    }
    else
      ++rejected;
  }
  std::printf("%s: %d mutated files, %zu accepted, %zu rejected, no crash / sanitizer report\n", argv[1], iters, ok, rejected);
  return 0;
}
