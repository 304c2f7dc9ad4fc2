// host_model.cpp — RAM model, file ingest and the one-off upload transform (host side).
//
// Behavioural spec = the reference loader and uploader:
//   PLY    src/ply_loader_async.cpp:357-441 (property names, "all 45 f_rest or none", RDF->RUB)
//   SPZ    src/ply_loader_async.cpp:304-353 + 3rdparty/spz/src/cc/load-spz.cc:131-139,333-378,467-532
//   .splat src/ply_loader_async.cpp:43-183
//   upload src/splat_set_vk.cpp:85-112,263-288,313-345,356-435
// The readers below are written from the file-format definitions, not from those sources.
#include "host_model.h"

#include <zlib.h>

#include <cmath>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <sstream>

#include "../../include/mgs.h"

namespace mgs {

static thread_local std::string g_error;
void        setError(const std::string& msg) { g_error = msg; }
const char* lastError() { return g_error.c_str(); }

int HostSplatSet::maxShDegree() const
{
  const size_t n = size();
  if(n == 0)
    return -1;
  const size_t perChannel = (uint32_t(f_rest.size()) / n) / 3;
  int          deg        = 0;
  if(perChannel >= 3)
    deg = 1;
  if(perChannel >= 8)
    deg = 2;
  if(perChannel == 15)
    deg = 3;
  return deg;
}

// RDF -> RUB: x kept, y and z mirrored.  Quaternion vector part follows (x keeps its sign because
// it is the product of the two mirrored axes), SH coefficients flip when they are odd in y or z.
void HostSplatSet::convertRdfToRub()
{
  static const float shSign[15] = {-1, -1, +1, -1, +1, +1, -1, +1, -1, +1, -1, -1, +1, -1, +1};
  const size_t       n          = size();
  for(size_t i = 0; i < n; ++i)
  {
    positions[3 * i + 1] = -positions[3 * i + 1];
    positions[3 * i + 2] = -positions[3 * i + 2];
  }
  for(size_t i = 0; i + 3 < rotation.size(); i += 4)
  {
    rotation[i + 2] = -rotation[i + 2];
    rotation[i + 3] = -rotation[i + 3];
  }
  if(n && !f_rest.empty())
  {
    const size_t perSplat = f_rest.size() / n;
    const size_t cpc      = perSplat / 3;
    for(size_t i = 0; i < n; ++i)
    {
      float* s = f_rest.data() + i * perSplat;
      for(size_t c = 0; c < 3; ++c)
        for(size_t j = 0; j < cpc && j < 15; ++j)
          s[c * cpc + j] *= shSign[j];
    }
  }
}

// ------------------------------------------------------------------------------------ PLY
namespace {

enum class PlyType { I8, U8, I16, U16, I32, U32, F32, F64, Bad };

PlyType parsePlyType(const std::string& t)
{
  if(t == "char" || t == "int8") return PlyType::I8;
  if(t == "uchar" || t == "uint8") return PlyType::U8;
  if(t == "short" || t == "int16") return PlyType::I16;
  if(t == "ushort" || t == "uint16") return PlyType::U16;
  if(t == "int" || t == "int32") return PlyType::I32;
  if(t == "uint" || t == "uint32") return PlyType::U32;
  if(t == "float" || t == "float32") return PlyType::F32;
  if(t == "double" || t == "float64") return PlyType::F64;
  return PlyType::Bad;
}
size_t plyTypeSize(PlyType t)
{
  switch(t)
  {
    case PlyType::I8: case PlyType::U8: return 1;
    case PlyType::I16: case PlyType::U16: return 2;
    case PlyType::I32: case PlyType::U32: case PlyType::F32: return 4;
    case PlyType::F64: return 8;
    default: return 0;
  }
}
void byteSwap(uint8_t* p, size_t n)
{
  for(size_t i = 0; i < n / 2; ++i)
    std::swap(p[i], p[n - 1 - i]);
}
double readScalar(const uint8_t* p, PlyType t, bool swap)
{
  uint8_t      tmp[8];
  const size_t n = plyTypeSize(t);
  std::memcpy(tmp, p, n);
  if(swap)
    byteSwap(tmp, n);
  switch(t)
  {
    case PlyType::I8: { int8_t v; std::memcpy(&v, tmp, 1); return v; }
    case PlyType::U8: { uint8_t v; std::memcpy(&v, tmp, 1); return v; }
    case PlyType::I16: { int16_t v; std::memcpy(&v, tmp, 2); return v; }
    case PlyType::U16: { uint16_t v; std::memcpy(&v, tmp, 2); return v; }
    case PlyType::I32: { int32_t v; std::memcpy(&v, tmp, 4); return v; }
    case PlyType::U32: { uint32_t v; std::memcpy(&v, tmp, 4); return v; }
    case PlyType::F32: { float v; std::memcpy(&v, tmp, 4); return v; }
    case PlyType::F64: { double v; std::memcpy(&v, tmp, 8); return v; }
    default: return 0.0;
  }
}

struct PlyProp
{
  std::string name;
  PlyType     type = PlyType::Bad;
  bool        isList = false;
  PlyType     countType = PlyType::Bad;
  size_t      offset = 0;  // byte offset in a fixed-size row
};
struct PlyElement
{
  std::string          name;
  size_t               count = 0;
  std::vector<PlyProp> props;
  bool                 fixed = true;
  size_t               rowSize = 0;
};

}  // namespace

int loadPly(const std::string& path, HostSplatSet& out)
{
  std::ifstream f(path, std::ios::binary);
  if(!f)
  {
    setError("ply: cannot open " + path);
    return MGS_ERR_IO;
  }
  std::string line;
  auto        getLine = [&](std::string& l) {
    if(!std::getline(f, l))
      return false;
    while(!l.empty() && (l.back() == '\r' || l.back() == ' '))
      l.pop_back();
    return true;
  };
  if(!getLine(line) || line != "ply")
  {
    setError("ply: bad magic in " + path);
    return MGS_ERR_FORMAT;
  }
  enum { ASCII, LE, BE } fmt = ASCII;
  std::vector<PlyElement> elements;
  bool                    headerDone = false;
  while(getLine(line))
  {
    std::istringstream ss(line);
    std::string        tok;
    ss >> tok;
    if(tok == "format")
    {
      std::string v;
      ss >> v;
      if(v == "ascii") fmt = ASCII;
      else if(v == "binary_little_endian") fmt = LE;
      else if(v == "binary_big_endian") fmt = BE;
      else
      {
        setError("ply: unknown format " + v);
        return MGS_ERR_FORMAT;
      }
    }
    else if(tok == "element")
    {
      PlyElement e;
      ss >> e.name >> e.count;
      elements.push_back(e);
    }
    else if(tok == "property")
    {
      if(elements.empty())
      {
        setError("ply: property before element");
        return MGS_ERR_FORMAT;
      }
      PlyProp     p;
      std::string t;
      ss >> t;
      if(t == "list")
      {
        std::string ct, it;
        ss >> ct >> it >> p.name;
        p.isList              = true;
        p.countType           = parsePlyType(ct);
        p.type                = parsePlyType(it);
        elements.back().fixed = false;
      }
      else
      {
        p.type = parsePlyType(t);
        ss >> p.name;
      }
      if(p.type == PlyType::Bad)
      {
        setError("ply: unknown property type in: " + line);
        return MGS_ERR_FORMAT;
      }
      p.offset = elements.back().rowSize;
      if(!p.isList)
        elements.back().rowSize += plyTypeSize(p.type);
      elements.back().props.push_back(p);
    }
    else if(tok == "end_header")
    {
      headerDone = true;
      break;
    }
    // comment / obj_info: ignored
  }
  if(!headerDone)
  {
    setError("ply: no end_header in " + path);
    return MGS_ERR_FORMAT;
  }

  // the property groups the reference asks miniply for (ply_loader_async.cpp:383-430)
  static const char* kPos[3]   = {"x", "y", "z"};
  static const char* kScale[3] = {"scale_0", "scale_1", "scale_2"};
  static const char* kRot[4]   = {"rot_0", "rot_1", "rot_2", "rot_3"};
  static const char* kDc[3]    = {"f_dc_0", "f_dc_1", "f_dc_2"};

  const bool swap = (fmt == BE);
  for(PlyElement& e : elements)
  {
    const bool isVertex = (e.name == "vertex");
    if(!isVertex || e.count == 0)
    {
      // skip the element's rows
      if(fmt == ASCII)
      {
        for(size_t r = 0; r < e.count; ++r)
          if(!getLine(line))
            break;
      }
      else if(e.fixed)
        f.seekg((std::streamoff)(e.rowSize * e.count), std::ios::cur);
      else
      {
        for(size_t r = 0; r < e.count; ++r)
          for(const PlyProp& p : e.props)
          {
            uint8_t buf[8];
            if(p.isList)
            {
              f.read((char*)buf, (std::streamsize)plyTypeSize(p.countType));
              const size_t cnt = (size_t)readScalar(buf, p.countType, swap);
              f.seekg((std::streamoff)(cnt * plyTypeSize(p.type)), std::ios::cur);
            }
            else
              f.seekg((std::streamoff)plyTypeSize(p.type), std::ios::cur);
          }
      }
      if(isVertex)
        continue;  // "skipping empty ply element"
      continue;
    }
    if(!e.fixed)
    {
      setError("ply: vertex element with list properties is not a 3DGS file");
      return MGS_ERR_FORMAT;
    }
    const size_t n = e.count;
    if(n > 0xFFFFFFFFull)
    {
      setError("ply: more than 2^32 splats");
      return MGS_ERR_UNSUPPORTED;
    }
    auto find = [&](const std::string& name) -> const PlyProp* {
      for(const PlyProp& p : e.props)
        if(p.name == name)
          return &p;
      return nullptr;
    };
    auto findGroup = [&](const char* const* names, size_t cnt, std::vector<const PlyProp*>& g) {
      g.clear();
      for(size_t i = 0; i < cnt; ++i)
      {
        const PlyProp* p = find(names[i]);
        if(!p)
          return false;
        g.push_back(p);
      }
      return true;
    };
    std::vector<const PlyProp*> gPos, gScale, gRot, gDc, gOp, gRest;
    std::vector<std::string>    restNames;
    std::vector<const char*>    restPtr;
    for(int i = 0; i < 45; ++i)
      restNames.push_back("f_rest_" + std::to_string(i));
    for(auto& s : restNames)
      restPtr.push_back(s.c_str());
    static const char* kOp[1] = {"opacity"};
    const bool hasRest  = findGroup(restPtr.data(), 45, gRest);  // all 45 or none (:383-396)
    const bool hasPos   = findGroup(kPos, 3, gPos);
    const bool hasOp    = findGroup(kOp, 1, gOp);
    const bool hasScale = findGroup(kScale, 3, gScale);
    const bool hasRot   = findGroup(kRot, 4, gRot);
    const bool hasDc    = findGroup(kDc, 3, gDc);
    if(!hasPos || !hasOp || !hasScale || !hasRot || !hasDc)
    {
      setError("ply: invalid 3DGS PLY file (missing x/y/z, opacity, scale_*, rot_* or f_dc_*): " + path);
      return MGS_ERR_FORMAT;
    }
    // the header's count is untrusted: check it against what the file can actually hold BEFORE sizing anything
    // (a 100-byte file declaring 4e9 vertices must be a format error, not an allocation of 1 TB)
    {
      const std::streamoff here = f.tellg();
      f.seekg(0, std::ios::end);
      const std::streamoff fileEnd = f.tellg();
      f.seekg(here, std::ios::beg);
      const uint64_t remaining = (here >= 0 && fileEnd >= here) ? (uint64_t)(fileEnd - here) : 0ull;
      // binary: rowSize bytes per vertex; ascii: at least one digit + one separator per property
      const uint64_t minRow = (fmt == ASCII) ? (uint64_t)e.props.size() * 2ull : (uint64_t)e.rowSize;
      if(minRow == 0 || n > remaining / minRow)
      {
        setError("ply: header declares " + std::to_string(n) + " vertices but only " + std::to_string(remaining)
                 + " bytes of vertex data follow in " + path);
        return MGS_ERR_FORMAT;
      }
    }
    out.positions.resize(n * 3);
    out.opacity.resize(n);
    out.scale.resize(n * 3);
    out.rotation.resize(n * 4);
    out.f_dc.resize(n * 3);
    if(hasRest)
      out.f_rest.resize(n * 45);
    else
      out.f_rest.clear();

    // read rows
    std::vector<double> row(e.props.size());
    std::vector<uint8_t> raw;
    if(fmt != ASCII)
    {
      raw.resize(e.rowSize * n);
      f.read((char*)raw.data(), (std::streamsize)raw.size());
      if((size_t)f.gcount() != raw.size())
      {
        setError("ply: truncated vertex data in " + path);
        return MGS_ERR_FORMAT;
      }
    }
    auto scatter = [&](size_t i, const std::vector<const PlyProp*>& g, float* dst, size_t stride,
                       const uint8_t* rowPtr, const std::vector<double>* asciiRow) {
      for(size_t k = 0; k < g.size(); ++k)
      {
        double v;
        if(asciiRow)
          v = (*asciiRow)[(size_t)(g[k] - e.props.data())];
        else
          v = readScalar(rowPtr + g[k]->offset, g[k]->type, swap);
        dst[i * stride + k] = (float)v;
      }
    };
    if(fmt == ASCII)
    {
      for(size_t i = 0; i < n; ++i)
      {
        if(!getLine(line))
        {
          setError("ply: truncated ascii vertex data in " + path);
          return MGS_ERR_FORMAT;
        }
        std::istringstream ss(line);
        for(size_t k = 0; k < e.props.size(); ++k)
          ss >> row[k];
        scatter(i, gPos, out.positions.data(), 3, nullptr, &row);
        scatter(i, gOp, out.opacity.data(), 1, nullptr, &row);
        scatter(i, gScale, out.scale.data(), 3, nullptr, &row);
        scatter(i, gRot, out.rotation.data(), 4, nullptr, &row);
        scatter(i, gDc, out.f_dc.data(), 3, nullptr, &row);
        if(hasRest)
          scatter(i, gRest, out.f_rest.data(), 45, nullptr, &row);
      }
    }
    else
    {
      parallelBatches(n, [&](size_t i) {
        const uint8_t* rp = raw.data() + i * e.rowSize;
        scatter(i, gPos, out.positions.data(), 3, rp, nullptr);
        scatter(i, gOp, out.opacity.data(), 1, rp, nullptr);
        scatter(i, gScale, out.scale.data(), 3, rp, nullptr);
        scatter(i, gRot, out.rotation.data(), 4, rp, nullptr);
        scatter(i, gDc, out.f_dc.data(), 3, rp, nullptr);
        if(hasRest)
          scatter(i, gRest, out.f_rest.data(), 45, rp, nullptr);
      });
    }
    out.path = path;
    out.convertRdfToRub();  // ply_loader_async.cpp:441
    return MGS_OK;
  }
  setError("ply: invalid 3DGS PLY file (no non-empty vertex element): " + path);
  return MGS_ERR_FORMAT;
}

// ------------------------------------------------------------------------------------ SPZ
// SPZ payload size the 16-byte header promises (0 = header invalid); shared by the inflate cap and the loader
static size_t spzPayloadBytes(const uint8_t* hdr)
{
  uint32_t magic, version, numPoints;
  std::memcpy(&magic, hdr, 4);
  std::memcpy(&version, hdr + 4, 4);
  std::memcpy(&numPoints, hdr + 8, 4);
  const uint8_t shDegree = hdr[12];
  if(magic != 0x5053474eu || version < 1 || version > 3 || shDegree > 3 || numPoints > 10000000u)
    return 0;
  const size_t n = numPoints, shDim = shDegree == 0 ? 0 : shDegree == 1 ? 3 : shDegree == 2 ? 8 : 15;
  return 16 + n * 3 * (version == 1 ? 2 : 3) + n + n * 3 + n * 3 + n * (version >= 3 ? 4 : 3) + n * shDim * 3;
}

// Inflates a gzip stream, never beyond what the SPZ header (the first 16 inflated bytes) says the payload holds:
// the output of an untrusted stream is capped (a few KB of zeros can inflate to gigabytes otherwise).
static bool gunzipSpz(const std::vector<uint8_t>& in, std::vector<uint8_t>& out)
{
  z_stream zs;
  std::memset(&zs, 0, sizeof(zs));
  if(inflateInit2(&zs, 16 + MAX_WBITS) != Z_OK)
    return false;
  zs.next_in  = const_cast<Bytef*>(in.data());
  zs.avail_in = (uInt)in.size();
  std::vector<uint8_t> chunk(1 << 20);
  bool                 ok    = false;
  size_t               limit = 16;  // until the header is known
  bool                 haveLimit = false;
  for(;;)
  {
    zs.next_out  = chunk.data();
    zs.avail_out = (uInt)chunk.size();
    const int rc = inflate(&zs, Z_NO_FLUSH);
    if(rc != Z_OK && rc != Z_STREAM_END)
      break;
    out.insert(out.end(), chunk.data(), chunk.data() + (chunk.size() - zs.avail_out));
    if(!haveLimit && out.size() >= 16)
    {
      limit     = spzPayloadBytes(out.data());
      haveLimit = true;
      if(limit == 0)
      {  // not an SPZ header: the caller reports it; nothing more to inflate
        out.resize(16);
        ok = true;
        break;
      }
    }
    if(rc == Z_STREAM_END)
    {
      ok = true;
      break;
    }
    if(haveLimit && out.size() >= limit)
    {  // everything the header promises is here; whatever follows is not ours to expand
      ok = true;
      break;
    }
    if(zs.avail_in == 0 && zs.avail_out != 0)
      break;  // truncated
  }
  inflateEnd(&zs);
  return ok;
}

int loadSpz(const std::string& path, HostSplatSet& out)
{
  std::ifstream f(path, std::ios::binary | std::ios::ate);
  if(!f)
  {
    setError("spz: cannot open " + path);
    return MGS_ERR_IO;
  }
  const std::streamsize sz = f.tellg();
  f.seekg(0);
  std::vector<uint8_t> gz((size_t)sz);
  f.read((char*)gz.data(), sz);
  std::vector<uint8_t> raw;
  if(!gunzipSpz(gz, raw) || raw.size() < 16)
  {
    setError("spz: not a gzip stream / truncated: " + path);
    return MGS_ERR_FORMAT;
  }
  uint32_t magic, version, numPoints;
  std::memcpy(&magic, raw.data(), 4);
  std::memcpy(&version, raw.data() + 4, 4);
  std::memcpy(&numPoints, raw.data() + 8, 4);
  const uint8_t shDegree = raw[12], fractionalBits = raw[13];
  if(magic != 0x5053474eu)  // "NGSP"
  {
    setError("spz: header not found in " + path);
    return MGS_ERR_FORMAT;
  }
  if(version < 1 || version > 3 || shDegree > 3 || numPoints > 10000000u)
  {
    setError("spz: unsupported version / sh degree / point count in " + path);
    return MGS_ERR_FORMAT;
  }
  const size_t n       = numPoints;
  const size_t shDim   = shDegree == 0 ? 0 : shDegree == 1 ? 3 : shDegree == 2 ? 8 : 15;
  const bool   f16     = (version == 1);
  const bool   small3  = (version >= 3);
  const size_t posB    = n * 3 * (f16 ? 2 : 3);
  const size_t rotB    = n * (small3 ? 4 : 3);
  const size_t need    = 16 + posB + n + n * 3 + n * 3 + rotB + n * shDim * 3;
  if(n == 0 || raw.size() < need)
  {
    setError("spz: read error (payload shorter than header promises) in " + path);
    return MGS_ERR_FORMAT;
  }
  const uint8_t* pPos   = raw.data() + 16;
  const uint8_t* pAlpha = pPos + posB;
  const uint8_t* pColor = pAlpha + n;
  const uint8_t* pScale = pColor + n * 3;
  const uint8_t* pRot   = pScale + n * 3;
  const uint8_t* pSh    = pRot + rotB;

  out.positions.resize(n * 3);
  out.scale.resize(n * 3);
  out.rotation.resize(n * 4);
  out.opacity.resize(n);
  out.f_dc.resize(n * 3);
  out.f_rest.assign(n * shDim * 3, 0.f);

  if(f16)
  {
    for(size_t i = 0; i < n * 3; ++i)
    {
      uint16_t h;
      std::memcpy(&h, pPos + 2 * i, 2);
      out.positions[i] = halfToFloat(h);
    }
  }
  else
  {
    const float s = (float)(1.0 / (double)(1 << fractionalBits));
    for(size_t i = 0; i < n * 3; ++i)
    {
      int32_t v = pPos[3 * i] | (pPos[3 * i + 1] << 8) | (pPos[3 * i + 2] << 16);
      if(v & 0x800000)
        v |= (int32_t)0xff000000;
      out.positions[i] = (float)v * s;
    }
  }
  for(size_t i = 0; i < n * 3; ++i)
    out.scale[i] = pScale[i] / 16.0f - 10.0f;
  const float kColorScale = 0.15f;
  for(size_t i = 0; i < n * 3; ++i)
    out.f_dc[i] = ((pColor[i] / 255.0f) - 0.5f) / kColorScale;
  for(size_t i = 0; i < n; ++i)
  {
    const float a  = pAlpha[i] / 255.0f;
    out.opacity[i] = std::log(a / (1.0f - a));
  }
  const float kSqrtHalf = (float)0.707106781186547524401;
  for(size_t i = 0; i < n; ++i)
  {
    float q[4];  // x y z w
    if(small3)
    {
      const uint8_t* r    = pRot + 4 * i;
      uint32_t       comp = (uint32_t)r[0] | ((uint32_t)r[1] << 8) | ((uint32_t)r[2] << 16) | ((uint32_t)r[3] << 24);
      const int      largest = (int)(comp >> 30);
      float          sumSq   = 0.f;
      for(int k = 3; k >= 0; --k)
      {
        if(k == largest)
          continue;
        const uint32_t mag = comp & 511u;
        const uint32_t neg = (comp >> 9) & 1u;
        comp >>= 10;
        q[k] = kSqrtHalf * (float)mag / 511.0f;
        if(neg)
          q[k] = -q[k];
        sumSq += q[k] * q[k];
      }
      q[largest] = std::sqrt(1.0f - sumSq);
    }
    else
    {
      const uint8_t* r = pRot + 3 * i;
      for(int k = 0; k < 3; ++k)
        q[k] = (float)r[k] * (1.0f / 127.5f) + -1.0f;
      q[3] = std::sqrt(std::max(0.0f, 1.0f - ((q[0] * q[0] + q[1] * q[1]) + q[2] * q[2])));
    }
    // spz stores RUB already; the reference asks for RUB, so no axis flip.  (x,y,z,w) -> (w,x,y,z)
    out.rotation[4 * i + 0] = q[3];
    out.rotation[4 * i + 1] = q[0];
    out.rotation[4 * i + 2] = q[1];
    out.rotation[4 * i + 3] = q[2];
  }
  // SH: spz keeps [coef][rgb]; INRIA layout is channel-major per splat
  for(size_t i = 0; i < n; ++i)
    for(size_t j = 0; j < shDim; ++j)
      for(size_t c = 0; c < 3; ++c)
        out.f_rest[i * shDim * 3 + c * shDim + j] = ((float)pSh[(i * shDim + j) * 3 + c] - 128.0f) / 128.0f;
  out.path = path;
  return MGS_OK;
}

// ------------------------------------------------------------------------------------ .splat
int loadSplat(const std::string& path, HostSplatSet& out)
{
  std::ifstream f(path, std::ios::binary | std::ios::ate);
  if(!f)
  {
    setError("splat: cannot open " + path);
    return MGS_ERR_IO;
  }
  const std::streamsize sz = f.tellg();
  if(sz == 0 || sz % 32 != 0)
  {
    setError("splat: file size is not a positive multiple of 32 bytes: " + path);
    return MGS_ERR_FORMAT;
  }
  const size_t n = (size_t)(sz / 32);
  f.seekg(0);
  std::vector<uint8_t> raw((size_t)sz);
  f.read((char*)raw.data(), sz);
  out.positions.resize(n * 3);
  out.scale.resize(n * 3);
  out.rotation.resize(n * 4);
  out.opacity.resize(n);
  out.f_dc.resize(n * 3);
  out.f_rest.clear();
  const float SH_C0 = 0.28209479177387814f;
  for(size_t i = 0; i < n; ++i)
  {
    const uint8_t* r = raw.data() + 32 * i;
    float          pos[3], scl[3];
    std::memcpy(pos, r, 12);
    std::memcpy(scl, r + 12, 12);
    const uint8_t* col = r + 24;
    const uint8_t* rot = r + 28;
    for(int k = 0; k < 3; ++k)
    {
      out.positions[3 * i + k] = pos[k];
      out.scale[3 * i + k]     = std::log(scl[k]);
      out.f_dc[3 * i + k]      = (col[k] / 255.0f - 0.5f) / SH_C0;
    }
    // the four bytes are taken in file order as the (w,x,y,z) slots of the INRIA layout
    for(int k = 0; k < 4; ++k)
      out.rotation[4 * i + k] = ((float)rot[k] - 128.0f) / 128.0f;
    const float a  = std::min(std::max(col[3] / 255.0f, 1e-6f), 1.0f - 1e-6f);
    out.opacity[i] = -std::log((1.0f / a) - 1.0f);
  }
  out.path = path;
  out.convertRdfToRub();
  return MGS_OK;
}

// ------------------------------------------------------------------------------------ matrices
void mat4Mul(const float a[16], const float b[16], float out[16])
{
  float t[16];
  for(int c = 0; c < 4; ++c)
    for(int r = 0; r < 4; ++r)
      t[c * 4 + r] = ((a[r] * b[c * 4] + a[4 + r] * b[c * 4 + 1]) + a[8 + r] * b[c * 4 + 2]) + a[12 + r] * b[c * 4 + 3];
  std::memcpy(out, t, sizeof(t));
}
void mat4MulVec4(const float m[16], const float v[4], float out[4])
{
  float t[4];
  for(int r = 0; r < 4; ++r)
    t[r] = ((v[0] * m[r] + v[1] * m[4 + r]) + v[2] * m[8 + r]) + v[3] * m[12 + r];
  std::memcpy(out, t, sizeof(t));
}
// adjugate / determinant via 2x2 sub-determinants (the classic closed form, as glm does)
void mat4Inverse(const float m[16], float out[16])
{
  auto        M   = [&](int c, int r) { return m[c * 4 + r]; };
  const float c00 = M(2, 2) * M(3, 3) - M(3, 2) * M(2, 3), c02 = M(1, 2) * M(3, 3) - M(3, 2) * M(1, 3),
              c03 = M(1, 2) * M(2, 3) - M(2, 2) * M(1, 3), c04 = M(2, 1) * M(3, 3) - M(3, 1) * M(2, 3),
              c06 = M(1, 1) * M(3, 3) - M(3, 1) * M(1, 3), c07 = M(1, 1) * M(2, 3) - M(2, 1) * M(1, 3),
              c08 = M(2, 1) * M(3, 2) - M(3, 1) * M(2, 2), c10 = M(1, 1) * M(3, 2) - M(3, 1) * M(1, 2),
              c11 = M(1, 1) * M(2, 2) - M(2, 1) * M(1, 2), c12 = M(2, 0) * M(3, 3) - M(3, 0) * M(2, 3),
              c14 = M(1, 0) * M(3, 3) - M(3, 0) * M(1, 3), c15 = M(1, 0) * M(2, 3) - M(2, 0) * M(1, 3),
              c16 = M(2, 0) * M(3, 2) - M(3, 0) * M(2, 2), c18 = M(1, 0) * M(3, 2) - M(3, 0) * M(1, 2),
              c19 = M(1, 0) * M(2, 2) - M(2, 0) * M(1, 2), c20 = M(2, 0) * M(3, 1) - M(3, 0) * M(2, 1),
              c22 = M(1, 0) * M(3, 1) - M(3, 0) * M(1, 1), c23 = M(1, 0) * M(2, 1) - M(2, 0) * M(1, 1);
  const float f0[4] = {c00, c00, c02, c03}, f1[4] = {c04, c04, c06, c07}, f2[4] = {c08, c08, c10, c11},
              f3[4] = {c12, c12, c14, c15}, f4[4] = {c16, c16, c18, c19}, f5[4] = {c20, c20, c22, c23};
  const float v0[4] = {M(1, 0), M(0, 0), M(0, 0), M(0, 0)}, v1[4] = {M(1, 1), M(0, 1), M(0, 1), M(0, 1)},
              v2[4] = {M(1, 2), M(0, 2), M(0, 2), M(0, 2)}, v3[4] = {M(1, 3), M(0, 3), M(0, 3), M(0, 3)};
  float inv[4][4];
  for(int i = 0; i < 4; ++i)
  {
    const float sa = (i & 1) ? -1.f : 1.f, sb = -sa;
    inv[0][i] = (v1[i] * f0[i] - v2[i] * f1[i] + v3[i] * f2[i]) * sa;
    inv[1][i] = (v0[i] * f0[i] - v2[i] * f3[i] + v3[i] * f4[i]) * sb;
    inv[2][i] = (v0[i] * f1[i] - v1[i] * f3[i] + v3[i] * f5[i]) * sa;
    inv[3][i] = (v0[i] * f2[i] - v1[i] * f4[i] + v2[i] * f5[i]) * sb;
  }
  const float det = (M(0, 0) * inv[0][0] + M(0, 1) * inv[1][0]) + (M(0, 2) * inv[2][0] + M(0, 3) * inv[3][0]);
  const float rd  = 1.0f / det;
  for(int c = 0; c < 4; ++c)
    for(int r = 0; r < 4; ++r)
      out[c * 4 + r] = inv[c][r] * rd;
}

// ------------------------------------------------------------------------------------ upload transform
void buildCov6(const HostSplatSet& s, std::vector<float>& cov6)
{
  const size_t n = s.size();
  cov6.resize(n * 6);
  parallelBatches(n, [&](size_t i) {
    const float sx = std::exp(s.scale[3 * i]), sy = std::exp(s.scale[3 * i + 1]), sz = std::exp(s.scale[3 * i + 2]);
    float       w = s.rotation[4 * i], x = s.rotation[4 * i + 1], y = s.rotation[4 * i + 2], z = s.rotation[4 * i + 3];
    const float len = std::sqrt(((w * w + x * x) + y * y) + z * z);
    if(len <= 0.f) { w = 1.f; x = y = z = 0.f; }
    else { const float r = 1.f / len; w *= r; x *= r; y *= r; z *= r; }
    const float xx = x * x, yy = y * y, zz = z * z, xz = x * z, xy = x * y, yz = y * z, wx = w * x, wy = w * y, wz = w * z;
    // rotation columns scaled by the axis lengths: M = R * diag(s)
    const float m00 = (1.f - 2.f * (yy + zz)) * sx, m10 = (2.f * (xy + wz)) * sx, m20 = (2.f * (xz - wy)) * sx;
    const float m01 = (2.f * (xy - wz)) * sy, m11 = (1.f - 2.f * (xx + zz)) * sy, m21 = (2.f * (yz + wx)) * sy;
    const float m02 = (2.f * (xz + wy)) * sz, m12 = (2.f * (yz - wx)) * sz, m22 = (1.f - 2.f * (xx + yy)) * sz;
    // Sigma = M M^T  (m<row><col>)
    float* c = cov6.data() + 6 * i;
    c[0] = (m00 * m00 + m01 * m01) + m02 * m02;
    c[1] = (m00 * m10 + m01 * m11) + m02 * m12;
    c[2] = (m00 * m20 + m01 * m21) + m02 * m22;
    c[3] = (m10 * m10 + m11 * m11) + m12 * m12;
    c[4] = (m10 * m20 + m11 * m21) + m12 * m22;
    c[5] = (m20 * m20 + m21 * m21) + m22 * m22;
  });
}

void buildRgba(const HostSplatSet& s, std::vector<float>& rgba)
{
  const size_t n = s.size();
  rgba.resize(n * 4);
  const float kC0 = 0.28209479177387814f;
  parallelBatches(n, [&](size_t i) {
    for(int c = 0; c < 3; ++c)
      rgba[4 * i + c] = std::min(std::max(0.5f + kC0 * s.f_dc[3 * i + c], 0.f), 1.f);
    rgba[4 * i + 3] = std::min(std::max(1.0f / (1.0f + std::exp(-s.opacity[i])), 0.f), 1.f);
  });
}

int shStride(uint32_t fRestPerSplat)
{
  const uint32_t cpc = fRestPerSplat / 3;
  int            st  = 0;
  if(cpc >= 3) st += 9;
  if(cpc >= 8) st += 15;
  if(cpc == 15) st += 21;
  return st;
}

void buildShInterleaved(const HostSplatSet& s, std::vector<float>& sh)
{
  const size_t   n      = s.size();
  const uint32_t per    = s.fRestPerSplat();
  const uint32_t cpc    = per / 3;
  const int      stride = shStride(per);
  sh.resize(n * (size_t)stride);
  if(!stride)
    return;
  parallelBatches(n, [&](size_t i) {
    const float* src = s.f_rest.data() + i * per;
    float*       dst = sh.data() + i * (size_t)stride;
    for(int k = 0; k < stride / 3; ++k)
      for(int c = 0; c < 3; ++c)
        dst[3 * k + c] = src[cpc * c + k];
  });
}

// ---- Morton (Z-curve) storage order ------------------------------------------------------------------
// Build-defined, no reference counterpart: the reference keeps file order.  Splats that are close in
// space become close in memory, so a frustum (or a screen strip) keeps or drops whole 2048-splat
// partitions and the survivors of a partition are dense in every cache line of the planar buffers.
static inline uint64_t spread21(uint64_t v)
{
  v &= 0x1fffffull;
  v = (v | (v << 32)) & 0x1f00000000ffffull;
  v = (v | (v << 16)) & 0x1f0000ff0000ffull;
  v = (v | (v << 8)) & 0x100f00f00f00f00full;
  v = (v | (v << 4)) & 0x10c30c30c30c30c3ull;
  v = (v | (v << 2)) & 0x1249249249249249ull;
  return v;
}

void mortonOrder(const HostSplatSet& s, const std::vector<float>* radius, std::vector<uint32_t>& newToOld)
{
  const size_t n = s.size();
  newToOld.resize(n);
  if(n == 0)
    return;
  // quantisation window: mean +- 4 sigma per axis (outliers clamp to the border cells)
  double mean[3] = {0, 0, 0}, var[3] = {0, 0, 0};
  for(size_t i = 0; i < n; ++i)
    for(int a = 0; a < 3; ++a)
      mean[a] += s.positions[3 * i + a];
  for(int a = 0; a < 3; ++a)
    mean[a] /= (double)n;
  for(size_t i = 0; i < n; ++i)
    for(int a = 0; a < 3; ++a)
    {
      const double d = s.positions[3 * i + a] - mean[a];
      var[a] += d * d;
    }
  double lo[3], inv[3];
  for(int a = 0; a < 3; ++a)
  {
    const double sd = std::sqrt(var[a] / (double)n) + 1e-12;
    lo[a]           = mean[a] - 4.0 * sd;
    inv[a]          = 1048575.0 / (8.0 * sd);  // 20 bits per axis, the top bits carry the size class
  }
  // size classes: octaves of the footprint radius around the median (16 classes in the top 4 code bits).
  // Footprints are log-normally distributed over ~2 orders of magnitude; inside a class they differ by < 2x,
  // so a partition's radius bound (and with it the strip-level culling) is tight for the small splats that
  // make up the bulk of a scene.
  float rRef = 0.f;
  if(radius && radius->size() == n && n >= 16)
  {
    std::vector<float> tmp(*radius);
    std::nth_element(tmp.begin(), tmp.begin() + n / 2, tmp.end());
    rRef = tmp[n / 2];
  }
  struct Item
  {
    uint64_t code;
    uint32_t idx;
  };
  std::vector<Item> items(n);
  parallelBatches(n, [&](size_t i) {
    uint64_t q[3];
    for(int a = 0; a < 3; ++a)
    {
      double v = ((double)s.positions[3 * i + a] - lo[a]) * inv[a];
      v        = std::isfinite(v) ? std::min(std::max(v, 0.0), 1048575.0) : 0.0;
      q[a]     = (uint64_t)v;
    }
    uint64_t cls = 0;
    if(radius && radius->size() == n)
    {
      const float r = (*radius)[i];
      int         o = 8;
      if(rRef > 0.f && r > 0.f && std::isfinite(r))
        o = 8 + (int)std::floor(std::log2(r / rRef));
      else if(!(r > 0.f))
        o = 0;
      else
        o = 15;
      cls = (uint64_t)std::min(std::max(o, 0), 15);
    }
    items[i].code = (cls << 60) | spread21(q[0]) | (spread21(q[1]) << 1) | (spread21(q[2]) << 2);
    items[i].idx  = (uint32_t)i;
  });
  std::sort(items.begin(), items.end(), [](const Item& a, const Item& b) { return a.code < b.code || (a.code == b.code && a.idx < b.idx); });
  parallelBatches(n, [&](size_t i) { newToOld[i] = items[i].idx; });
}

uint8_t toUint8(float v, float lo, float hi)
{
  const float t = (v - lo) / (hi - lo);
  return (uint8_t)std::min(std::max(std::round(t * 255.0f), 0.0f), 255.0f);
}

// IEEE binary16 <-> binary32, round to nearest even
uint16_t floatToHalf(float f)
{
  uint32_t x;
  std::memcpy(&x, &f, 4);
  const uint32_t sign = (x >> 16) & 0x8000u;
  x &= 0x7fffffffu;
  if(x >= 0x7f800000u)
    return (uint16_t)(sign | 0x7c00u | (x > 0x7f800000u ? 0x200u : 0u));
  if(x >= 0x477ff000u)
    return (uint16_t)(sign | 0x7c00u);
  if(x <= 0x33000000u)
    return (uint16_t)sign;
  const int      e     = (int)(x >> 23) - 127;
  const uint32_t m     = (x & 0x7fffffu) | 0x800000u;
  const int      shift = (e < -14) ? (13 + (-14 - e)) : 13;
  uint32_t       hm    = m >> shift;
  const uint32_t rem = m & ((1u << shift) - 1u), half = 1u << (shift - 1);
  if(rem > half || (rem == half && (hm & 1u)))
    ++hm;
  const uint32_t h = (e < -14) ? hm : (((uint32_t)(e + 15) << 10) + (hm - 0x400u));
  return (uint16_t)(sign | h);
}
float halfToFloat(uint16_t h)
{
  const uint32_t sign = (uint32_t)(h & 0x8000u) << 16, e = (h >> 10) & 31u, m = h & 0x3ffu;
  uint32_t       x;
  if(e == 0)
  {
    if(m == 0)
      x = sign;
    else
    {
      const float v = std::ldexp((float)m, -24);
      std::memcpy(&x, &v, 4);
      x |= sign;
    }
  }
  else if(e == 31)
    x = sign | 0x7f800000u | (m << 13);
  else
    x = sign | ((e + 112u) << 23) | (m << 13);
  float f;
  std::memcpy(&f, &x, 4);
  return f;
}

}  // namespace mgs
