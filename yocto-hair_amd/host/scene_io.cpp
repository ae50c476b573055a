// scene_io.cpp — host-side scene input for the hair path (plain C++, no device
// code): a minimal JSON + PLY + Radiance-HDR reader producing the flat
// yh_scene_desc of include/yhair.h, and the .pfm / .hdr writers.
//
// It reproduces the load semantics of the reference loader that matter to the
// path (libs/yocto/yocto_sceneio.cpp:1064-1418):
//   * JSON objects iterate in alphabetical key order (nlohmann::json uses
//     std::map), so objects, and therefore lights (pt.cpp:1704-1736), are in
//     alphabetical order of their names;
//   * "lookat" overrides "frame": cameras use lookat_frame(eye, center, up)
//     and focus = |eye - center| (sceneio.cpp:1244-1249); objects and
//     environments use the inv_xz variant (sceneio.cpp:1264,1333;
//     math.h:3229-3239);
//   * shapes come from shapes/<name>.ply; line shapes without a radius get
//     0.001 (add_radius, sceneio.cpp:390-396); faces: any 4-gon makes the
//     whole mesh "quads" and quads (a,b,c,d) become (a,b,d),(c,d,b)
//     (yocto_ply.h:1104-1170, yocto_shape.cpp:2142-2150);
//   * camera film = {film, film/aspect} for aspect >= 1 (pt.cpp:2088-2092);
//   * .hdr texels are mantissa * 2^(e-136) (stb_image.h:6726-6751).
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <zlib.h>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "yhair.h"

namespace {

// ---------------------------------------------------------------------------
// JSON (just enough for the scene files: objects, arrays, numbers, strings,
// true/false/null)
// ---------------------------------------------------------------------------
struct Json {
  enum Kind { Null, Bool, Num, Str, Arr, Obj } kind = Null;
  double                      num = 0;
  bool                        b   = false;
  std::string                 str;
  std::vector<Json>           arr;
  std::map<std::string, Json> obj;  // std::map: alphabetical, as nlohmann
  bool has(const std::string& k) const { return kind == Obj && obj.count(k); }
  const Json& at(const std::string& k) const { return obj.at(k); }
};
struct JsonParser {
  const char* p;
  const char* end;
  void ws() { while (p < end && (*p == ' ' || *p == '\t' || *p == '\n' || *p == '\r')) p++; }
  [[noreturn]] void fail(const char* what) { throw std::runtime_error(std::string("json: ") + what); }
  Json value() {
    ws();
    if (p >= end) fail("unexpected end");
    Json j;
    if (*p == '{') {
      j.kind = Json::Obj;
      p++;
      ws();
      if (*p == '}') { p++; return j; }
      while (true) {
        ws();
        auto k = string();
        ws();
        if (*p != ':') fail("expected ':'");
        p++;
        j.obj[k] = value();
        ws();
        if (*p == ',') { p++; continue; }
        if (*p == '}') { p++; break; }
        fail("expected ',' or '}'");
      }
    } else if (*p == '[') {
      j.kind = Json::Arr;
      p++;
      ws();
      if (*p == ']') { p++; return j; }
      while (true) {
        j.arr.push_back(value());
        ws();
        if (*p == ',') { p++; continue; }
        if (*p == ']') { p++; break; }
        fail("expected ',' or ']'");
      }
    } else if (*p == '"') {
      j.kind = Json::Str;
      j.str  = string();
    } else if (!strncmp(p, "true", 4)) {
      j.kind = Json::Bool, j.b = true, p += 4;
    } else if (!strncmp(p, "false", 5)) {
      j.kind = Json::Bool, j.b = false, p += 5;
    } else if (!strncmp(p, "null", 4)) {
      p += 4;
    } else {
      char* e = nullptr;
      j.kind  = Json::Num;
      j.num   = strtod(p, &e);
      if (e == p) fail("bad number");
      p = e;
    }
    return j;
  }
  std::string string() {
    if (*p != '"') fail("expected string");
    p++;
    std::string s;
    while (p < end && *p != '"') {
      if (*p == '\\' && p + 1 < end) {
        p++;
        switch (*p) {
          case 'n': s += '\n'; break;
          case 't': s += '\t'; break;
          default: s += *p;
        }
        p++;
      } else {
        s += *p++;
      }
    }
    if (p >= end) fail("unterminated string");
    p++;
    return s;
  }
};

bool read_file(const std::string& path, std::string& data) {
  auto f = fopen(path.c_str(), "rb");
  if (!f) return false;
  fseek(f, 0, SEEK_END);
  auto n = ftell(f);
  fseek(f, 0, SEEK_SET);
  data.resize(n);
  auto ok = fread(&data[0], 1, n, f) == (size_t)n;
  fclose(f);
  return ok;
}
bool file_exists(const std::string& path) {
  auto f = fopen(path.c_str(), "rb");
  if (f) fclose(f);
  return f != nullptr;
}

// ---------------------------------------------------------------------------
// small float3 helpers for frames (host only; same operation order as
// math.h:2036-2039, 3229-3239 so that frames are bit-identical)
// ---------------------------------------------------------------------------
struct F3 { float x, y, z; };
F3    sub(F3 a, F3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
F3    neg(F3 a) { return {-a.x, -a.y, -a.z}; }
float dot(F3 a, F3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
F3    cross(F3 a, F3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
F3    normalize(F3 a) {
  auto l = std::sqrt(dot(a, a));
  return (l != 0) ? F3{a.x / l, a.y / l, a.z / l} : a;
}
void lookat_frame(const float* l, bool inv_xz, float* frame) {
  auto eye = F3{l[0], l[1], l[2]}, ctr = F3{l[3], l[4], l[5]}, up = F3{l[6], l[7], l[8]};
  auto w = normalize(sub(eye, ctr));
  auto u = normalize(cross(up, w));
  auto v = normalize(cross(w, u));
  if (inv_xz) w = neg(w), u = neg(u);
  float f[12] = {u.x, u.y, u.z, v.x, v.y, v.z, w.x, w.y, w.z, eye.x, eye.y, eye.z};
  memcpy(frame, f, sizeof(f));
}

bool get_floats(const Json& j, const std::string& key, float* out, int n) {
  if (!j.has(key)) return false;
  auto& a = j.at(key);
  if (a.kind == Json::Num && n == 1) { out[0] = (float)a.num; return true; }
  if (a.kind != Json::Arr || (int)a.arr.size() != n) throw std::runtime_error("json: bad array for " + key);
  for (int i = 0; i < n; i++) out[i] = (float)a.arr[i].num;
  return true;
}

// ---------------------------------------------------------------------------
// PLY (ascii and binary_little_endian; the properties the hair path reads)
// ---------------------------------------------------------------------------
struct ShapeData {
  std::vector<float> positions, normals, radius, texcoords;
  std::vector<int>   lines, triangles;
};
enum PlyType { PLY_I8, PLY_U8, PLY_I16, PLY_U16, PLY_I32, PLY_U32, PLY_F32, PLY_F64, PLY_I64, PLY_U64 };
PlyType ply_type(const std::string& t) {
  if (t == "char" || t == "int8") return PLY_I8;
  if (t == "uchar" || t == "uint8") return PLY_U8;
  if (t == "short" || t == "int16") return PLY_I16;
  if (t == "ushort" || t == "uint16") return PLY_U16;
  if (t == "int" || t == "int32") return PLY_I32;
  if (t == "uint" || t == "uint32") return PLY_U32;
  if (t == "float" || t == "float32") return PLY_F32;
  if (t == "double" || t == "float64") return PLY_F64;
  if (t == "int64") return PLY_I64;
  if (t == "uint64") return PLY_U64;
  throw std::runtime_error("ply: unknown type " + t);
}
inline size_t ply_size(PlyType t) {
  static const size_t sizes[] = {1, 1, 2, 2, 4, 4, 4, 8, 8, 8};
  return sizes[t];
}
struct PlyProp { std::string name; PlyType type = PLY_F32, ltype = PLY_U8; bool list = false; };
struct PlyElem { std::string name; size_t count = 0; std::vector<PlyProp> props; };
inline double ply_read_bin(const unsigned char*& p, PlyType t) {
  double v = 0;
  switch (t) {
    case PLY_I8: v = *(const signed char*)p; break;
    case PLY_U8: v = *p; break;
    case PLY_I16: { int16_t x; memcpy(&x, p, 2); v = x; } break;
    case PLY_U16: { uint16_t x; memcpy(&x, p, 2); v = x; } break;
    case PLY_I32: { int32_t x; memcpy(&x, p, 4); v = x; } break;
    case PLY_U32: { uint32_t x; memcpy(&x, p, 4); v = x; } break;
    case PLY_F32: { float x; memcpy(&x, p, 4); v = x; } break;
    case PLY_F64: { double x; memcpy(&x, p, 8); v = x; } break;
    case PLY_I64: { int64_t x; memcpy(&x, p, 8); v = (double)x; } break;
    case PLY_U64: { uint64_t x; memcpy(&x, p, 8); v = (double)x; } break;
  }
  p += ply_size(t);
  return v;
}
void load_ply(const std::string& path, ShapeData& shape) {
  std::string data;
  if (!read_file(path, data)) throw std::runtime_error(path + ": file not found");
  size_t pos = 0;
  auto   line = [&]() {
    auto e = data.find('\n', pos);
    if (e == std::string::npos) throw std::runtime_error(path + ": bad ply header");
    auto s = data.substr(pos, e - pos);
    pos    = e + 1;
    while (!s.empty() && (s.back() == '\r' || s.back() == ' ')) s.pop_back();
    return s;
  };
  if (line() != "ply") throw std::runtime_error(path + ": not a ply file");
  bool ascii = false;
  std::vector<PlyElem> elems;
  while (true) {
    auto l = line();
    char a[64] = "", b[64] = "", c[64] = "", d[64] = "";
    auto n = sscanf(l.c_str(), "%63s %63s %63s %63s %63s", a, b, c, d, d);
    if (n < 1) continue;
    if (!strcmp(a, "end_header")) break;
    if (!strcmp(a, "format")) {
      if (!strcmp(b, "ascii")) ascii = true;
      else if (!strcmp(b, "binary_little_endian")) ascii = false;
      else throw std::runtime_error(path + ": unsupported ply format");
    } else if (!strcmp(a, "element")) {
      PlyElem e;
      e.name  = b;
      e.count = strtoull(c, nullptr, 10);
      elems.push_back(e);
    } else if (!strcmp(a, "property")) {
      if (elems.empty()) throw std::runtime_error(path + ": property before element");
      PlyProp p;
      if (!strcmp(b, "list")) {
        char t1[64], t2[64], nm[64];
        if (sscanf(l.c_str(), "property list %63s %63s %63s", t1, t2, nm) != 3)
          throw std::runtime_error(path + ": bad list property");
        p.list = true, p.ltype = ply_type(t1), p.type = ply_type(t2), p.name = nm;
      } else {
        p.type = ply_type(b), p.name = c;
      }
      elems.back().props.push_back(p);
    }
  }
  auto bp   = (const unsigned char*)data.data() + pos;
  auto bend = (const unsigned char*)data.data() + data.size();
  auto ap   = data.c_str() + pos;
  auto next = [&](PlyType t) -> double {
    if (ascii) {
      char* e = nullptr;
      auto  v = strtod(ap, &e);
      if (e == ap) throw std::runtime_error(path + ": truncated ascii ply");
      ap = e;
      return v;
    }
    if (bp + ply_size(t) > bend) throw std::runtime_error(path + ": truncated ply");
    return ply_read_bin(bp, t);
  };
  std::vector<std::vector<int>> faces;
  bool has_normals = false, has_radius = false;
  for (auto& e : elems) {
    int ix = -1, iy = -1, iz = -1, inx = -1, iny = -1, inz = -1, ir = -1, iu = -1, iv = -1, is = -1, it = -1;
    for (int k = 0; k < (int)e.props.size(); k++) {
      auto& nm = e.props[k].name;
      if (nm == "x") ix = k; else if (nm == "y") iy = k; else if (nm == "z") iz = k;
      else if (nm == "nx") inx = k; else if (nm == "ny") iny = k; else if (nm == "nz") inz = k;
      else if (nm == "radius") ir = k;
      else if (nm == "u") iu = k; else if (nm == "v") iv = k; else if (nm == "s") is = k; else if (nm == "t") it = k;
    }
    if (iu < 0 || iv < 0) iu = is, iv = it;  // get_texcoords (yocto_ply.h:1083-1094): u v, else s t
    const bool has_uv = iu >= 0 && iv >= 0;
    if (e.name == "vertex") {
      has_normals = inx >= 0 && iny >= 0 && inz >= 0;
      has_radius  = ir >= 0;
      shape.positions.resize(3 * e.count);
      if (has_normals) shape.normals.resize(3 * e.count);
      if (has_radius) shape.radius.resize(e.count);
      if (has_uv) shape.texcoords.resize(2 * e.count);
    }
    std::vector<double> row(e.props.size());
    std::vector<int>    idx;
    const bool is_vertex = e.name == "vertex", is_face = e.name == "face", is_line = e.name == "line";
    // binary rows of plain floats (the hair models: x y z nx ny nz radius): one gather per vertex
    bool all_f32 = !ascii && is_vertex && !e.props.empty();
    for (auto& pr : e.props) all_f32 = all_f32 && !pr.list && pr.type == PLY_F32;
    if (all_f32) {
      if (ix < 0 || iy < 0 || iz < 0) throw std::runtime_error(path + ": vertex without x y z");
      size_t stride = e.props.size() * 4;
      if (bp + stride * e.count > bend) throw std::runtime_error(path + ": truncated ply");
      for (size_t i = 0; i < e.count; i++, bp += stride) {
        auto f = [&](int k) { float x; memcpy(&x, bp + 4 * (size_t)k, 4); return x; };
        shape.positions[3 * i] = f(ix), shape.positions[3 * i + 1] = f(iy), shape.positions[3 * i + 2] = f(iz);
        if (has_normals) shape.normals[3 * i] = f(inx), shape.normals[3 * i + 1] = f(iny), shape.normals[3 * i + 2] = f(inz);
        if (has_radius) shape.radius[i] = f(ir);
        if (has_uv) shape.texcoords[2 * i] = f(iu), shape.texcoords[2 * i + 1] = 1 - f(iv);  // load_shape flips v (yocto_shape.h:745)
      }
      continue;
    }
    if (is_line) shape.lines.reserve(shape.lines.size() + 2 * e.count);
    for (size_t i = 0; i < e.count; i++) {
      for (int k = 0; k < (int)e.props.size(); k++) {
        auto& pr = e.props[k];
        if (pr.list) {
          auto n = (int)next(pr.ltype);
          if (n < 0) throw std::runtime_error(path + ": negative list length");
          idx.resize((size_t)n);
          for (int c = 0; c < n; c++) idx[c] = (int)next(pr.type);
          if (pr.name == "vertex_indices" || pr.name == "vertex_index") {
            if (is_face) faces.push_back(idx);
            else if (is_line)
              for (int c = 1; c < n; c++) shape.lines.push_back(idx[c - 1]), shape.lines.push_back(idx[c]);
          }
        } else {
          row[k] = next(pr.type);
        }
      }
      if (e.name == "vertex") {
        if (ix < 0 || iy < 0 || iz < 0) throw std::runtime_error(path + ": vertex without x y z");
        shape.positions[3 * i] = (float)row[ix], shape.positions[3 * i + 1] = (float)row[iy];
        shape.positions[3 * i + 2] = (float)row[iz];
        if (has_normals) {
          shape.normals[3 * i] = (float)row[inx], shape.normals[3 * i + 1] = (float)row[iny];
          shape.normals[3 * i + 2] = (float)row[inz];
        }
        if (has_radius) shape.radius[i] = (float)row[ir];
        if (has_uv) shape.texcoords[2 * i] = (float)row[iu], shape.texcoords[2 * i + 1] = 1 - (float)row[iv];
      }
    }
  }
  // faces -> triangles with the reference's quad rule
  bool any_quad = false;
  for (auto& f : faces) if (f.size() == 4) any_quad = true;
  auto tri = [&](int a, int b, int c) { shape.triangles.push_back(a), shape.triangles.push_back(b), shape.triangles.push_back(c); };
  for (auto& f : faces) {
    if (any_quad) {
      // get_quads (yocto_ply.h:1122-1144) then quads_to_triangles
      auto quad = [&](int x, int y, int z, int w) {
        tri(x, y, w);
        if (z != w) tri(z, w, y);
      };
      if (f.size() == 4) quad(f[0], f[1], f[2], f[3]);
      else for (size_t c = 2; c < f.size(); c++) quad(f[0], f[c - 1], f[c], f[c]);
    } else {
      for (size_t c = 2; c < f.size(); c++) tri(f[0], f[c - 1], f[c]);
    }
  }
  if (shape.positions.empty()) throw std::runtime_error(path + ": empty shape");
  if (!shape.lines.empty() && shape.radius.empty())
    shape.radius.assign(shape.positions.size() / 3, 0.001f);  // add_radius
}

// ---------------------------------------------------------------------------
// PNG (8-bit grey / grey+alpha / RGB / RGBA / palette, non-interlaced) -> RGB bytes,
// the result the reference gets from stb_image with three requested channels
// (yocto_image.cpp load_image -> image<vec3b>): alpha dropped, grey replicated.
// ---------------------------------------------------------------------------
void load_png(const std::string& path, int& w, int& h, std::vector<unsigned char>& rgb) {
  std::string data;
  if (!read_file(path, data)) throw std::runtime_error(path + ": file not found");
  auto p = (const unsigned char*)data.data();
  auto n = data.size();
  static const unsigned char sig[8] = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
  if (n < 8 || memcmp(p, sig, 8)) throw std::runtime_error(path + ": not a png file");
  auto be32 = [](const unsigned char* q) { return ((unsigned)q[0] << 24) | ((unsigned)q[1] << 16) | ((unsigned)q[2] << 8) | q[3]; };
  size_t pos = 8;
  int    depth = 0, ctype = 0, interlace = 0;
  std::vector<unsigned char> idat, palette;
  w = h = 0;
  while (pos + 12 <= n) {
    unsigned len = be32(p + pos);
    if (pos + 12 + (size_t)len > n) throw std::runtime_error(path + ": truncated png");
    const unsigned char* type = p + pos + 4;
    const unsigned char* body = p + pos + 8;
    if (!memcmp(type, "IHDR", 4) && len >= 13) {
      w = (int)be32(body), h = (int)be32(body + 4), depth = body[8], ctype = body[9], interlace = body[12];
    } else if (!memcmp(type, "PLTE", 4)) {
      palette.assign(body, body + len);
    } else if (!memcmp(type, "IDAT", 4)) {
      idat.insert(idat.end(), body, body + len);
    } else if (!memcmp(type, "IEND", 4)) {
      break;
    }
    pos += 12 + (size_t)len;
  }
  if (w <= 0 || h <= 0 || w > 32768 || h > 32768) throw std::runtime_error(path + ": bad png header");
  if (depth != 8 || interlace != 0) throw std::runtime_error(path + ": only 8-bit non-interlaced png textures are supported");
  int ch = ctype == 0 ? 1 : ctype == 2 ? 3 : ctype == 3 ? 1 : ctype == 4 ? 2 : ctype == 6 ? 4 : 0;
  if (!ch) throw std::runtime_error(path + ": unsupported png colour type");
  size_t stride = (size_t)w * ch;
  std::vector<unsigned char> raw((stride + 1) * (size_t)h);
  uLongf out_len = (uLongf)raw.size();
  if (uncompress(raw.data(), &out_len, idat.data(), (uLong)idat.size()) != Z_OK || out_len != raw.size())
    throw std::runtime_error(path + ": corrupt png data");
  std::vector<unsigned char> img(stride * (size_t)h);
  for (int y = 0; y < h; y++) {  // undo the scanline filters (PNG spec 9.2)
    const unsigned char* src  = raw.data() + (stride + 1) * (size_t)y;
    unsigned char*       dst  = img.data() + stride * (size_t)y;
    const unsigned char* prev = y ? dst - stride : nullptr;
    int                  ft   = src[0];
    for (size_t x = 0; x < stride; x++) {
      int a = x >= (size_t)ch ? dst[x - ch] : 0, b = prev ? prev[x] : 0, c = (prev && x >= (size_t)ch) ? prev[x - ch] : 0;
      int v = src[1 + x];
      switch (ft) {
        case 0: break;
        case 1: v += a; break;
        case 2: v += b; break;
        case 3: v += (a + b) >> 1; break;
        case 4: {
          int pa = abs(b - c), pb = abs(a - c), pc = abs(a + b - 2 * c);
          v += (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
        } break;
        default: throw std::runtime_error(path + ": bad png filter");
      }
      dst[x] = (unsigned char)v;
    }
  }
  rgb.resize((size_t)w * h * 3);
  for (size_t i = 0; i < (size_t)w * h; i++) {
    const unsigned char* q = img.data() + i * ch;
    unsigned char*       o = rgb.data() + i * 3;
    if (ctype == 2 || ctype == 6) {
      o[0] = q[0], o[1] = q[1], o[2] = q[2];
    } else if (ctype == 3) {
      if ((size_t)q[0] * 3 + 2 >= palette.size()) throw std::runtime_error(path + ": palette index out of range");
      o[0] = palette[q[0] * 3], o[1] = palette[q[0] * 3 + 1], o[2] = palette[q[0] * 3 + 2];
    } else {
      o[0] = o[1] = o[2] = q[0];
    }
  }
}

// ---------------------------------------------------------------------------
// Radiance .hdr (RGBE, flat or new-style RLE scanlines)
// ---------------------------------------------------------------------------
void load_hdr(const std::string& path, int& w, int& h, std::vector<float>& rgb) {
  std::string data;
  if (!read_file(path, data)) throw std::runtime_error(path + ": file not found");
  size_t pos = 0;
  auto   line = [&]() {
    auto e = data.find('\n', pos);
    if (e == std::string::npos) throw std::runtime_error(path + ": bad hdr header");
    auto s = data.substr(pos, e - pos);
    pos    = e + 1;
    return s;
  };
  auto first = line();
  if (first != "#?RADIANCE" && first != "#?RGBE") throw std::runtime_error(path + ": not a Radiance file");
  while (true) {
    auto l = line();
    if (l.empty()) break;
  }
  auto dims = line();
  if (sscanf(dims.c_str(), "-Y %d +X %d", &h, &w) != 2) throw std::runtime_error(path + ": unsupported hdr orientation");
  rgb.assign((size_t)w * h * 3, 0.0f);
  auto p   = (const unsigned char*)data.data() + pos;
  auto end = (const unsigned char*)data.data() + data.size();
  auto convert = [&](const unsigned char* c, float* o) {
    if (c[3] != 0) {
      auto f1 = (float)ldexp(1.0f, (int)c[3] - (int)(128 + 8));
      o[0] = c[0] * f1, o[1] = c[1] * f1, o[2] = c[2] * f1;
    } else {
      o[0] = o[1] = o[2] = 0;
    }
  };
  std::vector<unsigned char> scan((size_t)w * 4);
  for (int j = 0; j < h; j++) {
    if (p + 4 > end) throw std::runtime_error(path + ": truncated hdr");
    bool rle = w >= 8 && w < 32768 && p[0] == 2 && p[1] == 2 && !(p[2] & 0x80) && ((p[2] << 8) | p[3]) == w;
    if (!rle) {
      if (p + (size_t)4 * w > end) throw std::runtime_error(path + ": truncated hdr");
      for (int i = 0; i < w; i++) convert(p + 4 * i, &rgb[((size_t)j * w + i) * 3]);
      p += (size_t)4 * w;
      continue;
    }
    p += 4;
    for (int k = 0; k < 4; k++) {
      int i = 0;
      while (i < w) {
        if (p >= end) throw std::runtime_error(path + ": truncated hdr");
        int count = *p++;
        if (count > 128) {
          count -= 128;
          if (p >= end || i + count > w) throw std::runtime_error(path + ": corrupt hdr");
          auto v = *p++;
          for (int z = 0; z < count; z++) scan[(size_t)(i++) * 4 + k] = v;
        } else {
          if (p + count > end || i + count > w || count == 0) throw std::runtime_error(path + ": corrupt hdr");
          for (int z = 0; z < count; z++) scan[(size_t)(i++) * 4 + k] = *p++;
        }
      }
    }
    for (int i = 0; i < w; i++) convert(&scan[(size_t)i * 4], &rgb[((size_t)j * w + i) * 3]);
  }
}

std::string dirname(const std::string& path) {
  auto p = path.find_last_of('/');
  return p == std::string::npos ? std::string(".") : path.substr(0, p);
}

}  // namespace

struct yh_scene_file {
  std::vector<ShapeData>          shape_data;
  std::vector<std::vector<float>> tex_data;
  std::vector<std::vector<unsigned char>> tex_bytes;  // material textures loaded from png
  std::vector<yh_texture>         textures;   // material textures (yh_scene_desc::textures)
  std::vector<yh_shape>           shapes;
  std::vector<yh_material>        materials;
  std::vector<yh_object>          objects;
  std::vector<yh_environment>     environments;
  yh_scene_desc                   desc{};
};

static yh_scene_file* load_scene(const std::string& path, const std::string& camera_name) {
  std::string text;
  if (!read_file(path, text)) throw std::runtime_error(path + ": file not found");
  JsonParser parser{text.data(), text.data() + text.size()};
  auto       js   = parser.value();
  auto       base = dirname(path);
  auto       sf   = std::make_unique<yh_scene_file>();
  const float identity[12] = {1, 0, 0, 0, 1, 0, 0, 0, 1, 0, 0, 0};

  // cameras (sceneio.cpp:1232-1251, get_camera 326-341, set_lens pt.cpp:2088)
  struct Cam { std::string name; float frame[12]; float lens = 0.050f, film = 0.036f, aspect = 1.5f, focus = 10000, aperture = 0; };
  std::vector<Cam> cams;
  if (js.has("cameras"))
    for (auto& [name, e] : js.at("cameras").obj) {
      Cam c;
      c.name = name;
      memcpy(c.frame, identity, sizeof(identity));
      get_floats(e, "frame", c.frame, 12);
      get_floats(e, "lens", &c.lens, 1), get_floats(e, "aspect", &c.aspect, 1);
      get_floats(e, "film", &c.film, 1), get_floats(e, "focus", &c.focus, 1);
      get_floats(e, "aperture", &c.aperture, 1);
      float l[9];
      if (get_floats(e, "lookat", l, 9)) {
        lookat_frame(l, false, c.frame);
        auto d  = sub(F3{l[0], l[1], l[2]}, F3{l[3], l[4], l[5]});
        c.focus = std::sqrt(dot(d, d));
      }
      cams.push_back(c);
    }
  if (cams.empty()) throw std::runtime_error(path + ": scene has no camera");
  const Cam* cam = nullptr;
  for (auto want : {camera_name, std::string("default"), std::string("camera"), std::string("camera1")}) {
    if (cam) break;
    for (auto& c : cams) if (c.name == want) { cam = &c; break; }
  }
  if (!cam) cam = &cams.front();
  memcpy(sf->desc.camera.frame, cam->frame, sizeof(identity));
  sf->desc.camera.lens    = cam->lens;
  sf->desc.camera.film[0] = cam->aspect >= 1 ? cam->film : cam->film * cam->aspect;
  sf->desc.camera.film[1] = cam->aspect >= 1 ? cam->film / cam->aspect : cam->film;
  sf->desc.camera.focus = cam->focus, sf->desc.camera.aperture = cam->aperture;

  // environment textures (sceneio.cpp:1383-1391: textures/<name>.{hdr,exr,png,jpg}, first that exists);
  // 8-bit images become float texels here with the reference's lookup-time conversion
  // srgb_to_rgb(byte / 255) (pt.cpp:147-164, math.h:3742-3745), so environments keep one texel format
  std::map<std::string, int> tex_index;
  struct Tex { int w, h, data; };  // data = index into sf->tex_data
  std::vector<Tex> texs;
  auto get_texture = [&](const std::string& name) {
    auto it = tex_index.find(name);
    if (it != tex_index.end()) return it->second;
    for (const char* ext : {".hdr", ".exr", ".png", ".jpg"}) {
      auto file = base + "/textures/" + name + ext;
      if (!file_exists(file)) continue;
      int w, h;
      sf->tex_data.emplace_back();
      if (!strcmp(ext, ".hdr")) {
        load_hdr(file, w, h, sf->tex_data.back());
      } else if (!strcmp(ext, ".png")) {
        std::vector<unsigned char> bytes;
        load_png(file, w, h, bytes);
        auto srgb_to_rgb = [](float srgb) {
          return (srgb <= 0.04045) ? srgb / 12.92f : std::pow((srgb + 0.055f) / (1.0f + 0.055f), 2.4f);
        };
        auto& f = sf->tex_data.back();
        f.resize(bytes.size());
        for (size_t i = 0; i < bytes.size(); i++) f[i] = srgb_to_rgb(bytes[i] / 255.0f);
      } else {
        throw std::runtime_error(file + ": only .hdr and .png textures are supported");
      }
      texs.push_back({w, h, (int)sf->tex_data.size() - 1});
      return tex_index[name] = (int)texs.size() - 1;
    }
    throw std::runtime_error(base + "/textures/" + name + ".hdr: file not found");
  };

  if (js.has("environments"))
    for (auto& [name, e] : js.at("environments").obj) {
      yh_environment env{};
      memcpy(env.frame, identity, sizeof(identity));
      get_floats(e, "frame", env.frame, 12);
      get_floats(e, "emission", env.emission, 3);
      float l[9];
      if (get_floats(e, "lookat", l, 9)) lookat_frame(l, true, env.frame);
      if (e.has("emission_tex") && !e.at("emission_tex").str.empty()) {
        auto t         = get_texture(e.at("emission_tex").str);
        env.tex_width  = texs[t].w, env.tex_height = texs[t].h;
        env.texels     = (const float*)(intptr_t)(texs[t].data + 1);  // patched below
      }
      sf->environments.push_back(env);
    }

  // colour textures of materials (sceneio.cpp:1383-1391: textures/<name>.{hdr,exr,png,jpg}, first that exists)
  std::map<std::string, int> mat_tex_index;
  auto get_material_texture = [&](const std::string& tname) -> int {
    auto it = mat_tex_index.find(tname);
    if (it != mat_tex_index.end()) return it->second;
    yh_texture t{};
    for (const char* ext : {".hdr", ".exr", ".png", ".jpg"}) {
      auto file = base + "/textures/" + tname + ext;
      if (!file_exists(file)) continue;
      if (!strcmp(ext, ".hdr")) {
        sf->tex_data.emplace_back();
        load_hdr(file, t.width, t.height, sf->tex_data.back());
        t.is_byte = 0, t.pixels = (const void*)(intptr_t)sf->tex_data.size();  // patched below
      } else if (!strcmp(ext, ".png")) {
        sf->tex_bytes.emplace_back();
        load_png(file, t.width, t.height, sf->tex_bytes.back());
        t.is_byte = 1, t.pixels = (const void*)(intptr_t)sf->tex_bytes.size();
      } else {
        throw std::runtime_error(file + ": only .hdr and .png textures are supported");
      }
      sf->textures.push_back(t);
      return mat_tex_index[tname] = (int)sf->textures.size();  // 1-based
    }
    throw std::runtime_error(base + "/textures/" + tname + ".hdr: file not found");
  };

  // materials (sceneio.cpp:1268-1325; defaults yocto_sceneio.h:126-157)
  std::map<std::string, int> material_index;
  if (js.has("materials"))
    for (auto& [name, e] : js.at("materials").obj) {
      yh_material m{};
      m.opacity = 1, m.ior = 1.5f, m.thin = 1, m.trdepth = 0.01f;
      m.beta_m = 0.3f, m.beta_n = 0.3f, m.alpha = 2, m.eta = 1.55f;
      get_floats(e, "eumelanin", &m.eumelanin, 1), get_floats(e, "pheomelanin", &m.pheomelanin, 1);
      get_floats(e, "sigma_a", m.sigma_a, 3);
      get_floats(e, "beta_m", &m.beta_m, 1), get_floats(e, "beta_n", &m.beta_n, 1);
      get_floats(e, "alpha", &m.alpha, 1), get_floats(e, "eta", &m.eta, 1);
      get_floats(e, "emission", m.emission, 3), get_floats(e, "color", m.color, 3);
      get_floats(e, "metallic", &m.metallic, 1), get_floats(e, "specular", &m.specular, 1);
      get_floats(e, "roughness", &m.roughness, 1), get_floats(e, "transmission", &m.transmission, 1);
      get_floats(e, "ior", &m.ior, 1), get_floats(e, "opacity", &m.opacity, 1);
      get_floats(e, "scattering", m.scattering, 3), get_floats(e, "scanisotropy", &m.scanisotropy, 1);
      get_floats(e, "trdepth", &m.trdepth, 1);
      if (e.has("thin")) m.thin = e.at("thin").b ? 1 : 0;
      for (auto& [k, v] : e.obj) {
        if (!(k.size() > 4 && k.substr(k.size() - 4) == "_tex" && !v.str.empty())) continue;
        if (k == "emission_tex") m.emission_tex = get_material_texture(v.str);
        else if (k == "color_tex") m.color_tex = get_material_texture(v.str);
        else if (k == "scattering_tex") m.scattering_tex = get_material_texture(v.str);
        else throw std::runtime_error(path + ": scalar and normal-map textures are not supported (" + name + "." + k + ")");
      }
      material_index[name] = (int)sf->materials.size();
      sf->materials.push_back(m);
    }

  // objects (sceneio.cpp:1327-1343) in alphabetical order; shapes by name
  std::map<std::string, int> shape_index;
  int default_material = -1;
  if (js.has("objects"))
    for (auto& [name, e] : js.at("objects").obj) {
      yh_object o{};
      memcpy(o.frame, identity, sizeof(identity));
      get_floats(e, "frame", o.frame, 12);
      float l[9];
      if (get_floats(e, "lookat", l, 9)) lookat_frame(l, true, o.frame);
      if (e.has("instance") || e.has("subdiv")) throw std::runtime_error(path + ": instances/subdivs are outside the hair path");
      if (e.has("material") && !e.at("material").str.empty()) {
        auto it = material_index.find(e.at("material").str);
        if (it == material_index.end()) throw std::runtime_error(path + ": missing material " + e.at("material").str);
        o.material = it->second;
      } else {  // add_materials (sceneio.cpp:399-408)
        if (default_material < 0) {
          yh_material m{};
          m.opacity = 1, m.ior = 1.5f, m.thin = 1, m.beta_m = 0.3f, m.beta_n = 0.3f, m.alpha = 2, m.eta = 1.55f;
          m.trdepth = 0.01f;
          m.color[0] = m.color[1] = m.color[2] = 0.8f;
          default_material = (int)sf->materials.size();
          sf->materials.push_back(m);
        }
        o.material = default_material;
      }
      if (!e.has("shape") || e.at("shape").str.empty()) throw std::runtime_error(path + ": object without shape: " + name);
      auto sname = e.at("shape").str;
      auto it    = shape_index.find(sname);
      if (it == shape_index.end()) {
        sf->shape_data.emplace_back();
        load_ply(base + "/shapes/" + sname + ".ply", sf->shape_data.back());
        it = shape_index.emplace(sname, (int)sf->shape_data.size() - 1).first;
      }
      o.shape = it->second;
      sf->objects.push_back(o);
    }

  for (auto& sd : sf->shape_data) {
    yh_shape s{};
    s.num_vertices  = (int)sd.positions.size() / 3;
    s.positions     = sd.positions.data();
    s.normals       = sd.normals.empty() ? nullptr : sd.normals.data();
    s.radius        = sd.radius.empty() ? nullptr : sd.radius.data();
    s.num_lines     = (int)sd.lines.size() / 2;
    s.lines         = sd.lines.empty() ? nullptr : sd.lines.data();
    s.num_triangles = s.num_lines ? 0 : (int)sd.triangles.size() / 3;
    s.triangles     = s.num_triangles ? sd.triangles.data() : nullptr;
    s.texcoords     = sd.texcoords.empty() ? nullptr : sd.texcoords.data();
    sf->shapes.push_back(s);
  }
  for (auto& env : sf->environments)
    if (env.texels) env.texels = sf->tex_data[(int)(intptr_t)env.texels - 1].data();

  for (auto& t : sf->textures)  // vectors no longer move
    t.pixels = t.is_byte ? (const void*)sf->tex_bytes[(size_t)(intptr_t)t.pixels - 1].data()
                         : (const void*)sf->tex_data[(size_t)(intptr_t)t.pixels - 1].data();
  sf->desc.num_textures = (int)sf->textures.size(), sf->desc.textures = sf->textures.data();
  sf->desc.num_shapes = (int)sf->shapes.size(), sf->desc.shapes = sf->shapes.data();
  sf->desc.num_materials = (int)sf->materials.size(), sf->desc.materials = sf->materials.data();
  sf->desc.num_objects = (int)sf->objects.size(), sf->desc.objects = sf->objects.data();
  sf->desc.num_environments = (int)sf->environments.size(), sf->desc.environments = sf->environments.data();
  return sf.release();
}

extern "C" {

yh_scene_file* yh_scene_load(const char* json_path, const char* camera, char* error, int error_len) {
  try {
    return load_scene(json_path ? json_path : "", camera ? camera : "");
  } catch (const std::exception& e) {
    if (error && error_len > 0) snprintf(error, error_len, "%s", e.what());
    return nullptr;
  }
}
const yh_scene_desc* yh_scene_get(const yh_scene_file* scene) { return scene ? &scene->desc : nullptr; }
void yh_scene_free(yh_scene_file* scene) { delete scene; }

int yh_save_image(const char* path, int width, int height, const float* rgba, char* error, int error_len) {
  auto fail = [&](const std::string& msg) {
    if (error && error_len > 0) snprintf(error, error_len, "%s", msg.c_str());
    return YH_E_IO;
  };
  std::string p = path ? path : "";
  auto dot_pos = p.rfind('.');
  auto ext     = dot_pos == std::string::npos ? std::string() : p.substr(dot_pos);
  auto f       = fopen(p.c_str(), "wb");
  if (!f) return fail(p + ": cannot open for writing");
  if (ext == ".pfm") {
    // yocto_image.cpp:1527-1556: "PF", "w h", "-1", rows top first, rgb only
    fprintf(f, "PF\n%d %d\n-1\n", width, height);
    for (int i = 0; i < width * height; i++) fwrite(rgba + 4 * i, sizeof(float), 3, f);
  } else if (ext == ".hdr") {
    fprintf(f, "#?RADIANCE\nFORMAT=32-bit_rle_rgbe\n\n-Y %d +X %d\n", height, width);
    for (int i = 0; i < width * height; i++) {
      auto          v = rgba + 4 * i;
      auto          m = std::max(v[0], std::max(v[1], v[2]));
      unsigned char c[4] = {0, 0, 0, 0};
      if (m >= 1e-32f) {
        int  e;
        auto s = (float)frexp(m, &e) * 256.0f / m;
        c[0] = (unsigned char)(v[0] * s), c[1] = (unsigned char)(v[1] * s);
        c[2] = (unsigned char)(v[2] * s), c[3] = (unsigned char)(e + 128);
      }
      fwrite(c, 1, 4, f);
    }
  } else {
    fclose(f);
    return fail(p + ": unsupported image format (use .pfm or .hdr)");
  }
  fclose(f);
  return YH_OK;
}
}
