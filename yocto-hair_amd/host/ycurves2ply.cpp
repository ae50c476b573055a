// ycurves2ply — pbrt `Shape "curve"` statements -> a hair PLY in the layout the
// reference's scenes use (vertex: x y z nx ny nz radius; element line: vertex_indices).
//
// This is the geometry step the reference performs inside its pbrt loader when a
// hair model is converted to its own format (libs/yocto/yocto_pbrt.h:1751-1797,
// reached through apps/ysceneproc): every curve's first four control points become a
// five-vertex strand. The arithmetic runs on the GPU (yh_curves_to_lines); this file
// is only the tokenizer and the PLY writer.
//
//   ycurves2ply [--device N] in.pbrt [more.pbrt ...] out.ply
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <sstream>
#include <string>
#include <vector>

#include "yhair.h"

namespace {
struct Token {
  enum Kind { Word, Str, Open, Close, End } kind = End;
  std::string text;
};
struct Lexer {
  const std::string& s;
  size_t             p = 0;
  explicit Lexer(const std::string& text) : s(text) {}
  Token next() {
    while (p < s.size()) {
      char c = s[p];
      if (c == '#') { while (p < s.size() && s[p] != '\n') p++; continue; }
      if (c == ' ' || c == '\t' || c == '\n' || c == '\r') { p++; continue; }
      break;
    }
    if (p >= s.size()) return {};
    char c = s[p];
    if (c == '[') { p++; return {Token::Open, "["}; }
    if (c == ']') { p++; return {Token::Close, "]"}; }
    if (c == '"') {
      size_t e = s.find('"', p + 1);
      if (e == std::string::npos) throw std::runtime_error("unterminated string");
      Token t{Token::Str, s.substr(p + 1, e - p - 1)};
      p = e + 1;
      return t;
    }
    size_t b = p;
    while (p < s.size() && !strchr(" \t\r\n[]\"#", s[p])) p++;
    return {Token::Word, s.substr(b, p - b)};
  }
};
[[noreturn]] void die(const std::string& msg) {
  std::cerr << "ycurves2ply: " << msg << "\n";
  exit(1);
}
// "float width0" -> width0 ; "point P" / "point3 P" -> P
std::string param_name(const std::string& decl) {
  auto sp = decl.find_last_of(" \t");
  return sp == std::string::npos ? decl : decl.substr(sp + 1);
}
}  // namespace

int main(int argc, char** argv) {
  int                      device = 0;
  std::vector<std::string> files;
  for (int i = 1; i < argc; i++) {
    if (!strcmp(argv[i], "--device") && i + 1 < argc) device = atoi(argv[++i]);
    else files.push_back(argv[i]);
  }
  if (files.size() < 2) die("usage: ycurves2ply [--device N] in.pbrt [more.pbrt ...] out.ply");
  std::string out = files.back();
  files.pop_back();

  std::vector<float> P, w0, w1;
  for (auto& path : files) {
    std::ifstream f(path, std::ios::binary);
    if (!f) die(path + ": file not found");
    std::stringstream ss;
    ss << f.rdbuf();
    std::string text = ss.str();
    Lexer       lex(text);
    try {
      Token t = lex.next();
      while (t.kind != Token::End) {
        if (!(t.kind == Token::Word && t.text == "Shape")) { t = lex.next(); continue; }
        Token type = lex.next();
        bool  curve = type.kind == Token::Str && type.text == "curve";
        std::vector<float> pts;
        float a = -1, b = -1, w = -1;
        t = lex.next();
        while (t.kind == Token::Str) {  // parameter list: "type name" value
          std::string        name = param_name(t.text);
          std::vector<float> vals;
          Token v = lex.next();
          if (v.kind == Token::Open) {
            for (v = lex.next(); v.kind != Token::Close; v = lex.next()) {
              if (v.kind == Token::End) throw std::runtime_error("unterminated [ ]");
              if (v.kind == Token::Word) vals.push_back(strtof(v.text.c_str(), nullptr));
            }
          } else if (v.kind == Token::Word) {
            vals.push_back(strtof(v.text.c_str(), nullptr));
          }
          if (name == "P") pts = vals;
          else if (name == "width0" && !vals.empty()) a = vals[0];
          else if (name == "width1" && !vals.empty()) b = vals[0];
          else if (name == "width" && !vals.empty()) w = vals[0];
          t = lex.next();
        }
        if (!curve) continue;
        if (w >= 0 && a < 0) a = w;  // pbrt's "width" sets both ends
        if (w >= 0 && b < 0) b = w;
        if (pts.size() < 12 || a < 0 || b < 0) die(path + ": curve without P (4 points), width0 and width1");
        P.insert(P.end(), pts.begin(), pts.begin() + 12);
        w0.push_back(a), w1.push_back(b);
      }
    } catch (std::exception& e) {
      die(path + ": " + e.what());
    }
  }
  int n = (int)w0.size();
  if (n == 0) die("no Shape \"curve\" found");

  yh_context* ctx = yh_create(device);
  if (!ctx) die(std::string("no GPU context: ") + yh_last_error(nullptr));
  std::vector<float> pos(15 * (size_t)n), nrm(15 * (size_t)n), rad(5 * (size_t)n);
  std::vector<int>   lines(8 * (size_t)n);
  if (yh_curves_to_lines(ctx, n, P.data(), w0.data(), w1.data(), 0, pos.data(), nrm.data(), rad.data(), lines.data()) != YH_OK)
    die(yh_last_error(ctx));
  yh_destroy(ctx);

  FILE* f = fopen(out.c_str(), "wb");
  if (!f) die(out + ": cannot write");
  size_t nv = 5 * (size_t)n, nl = 4 * (size_t)n;
  fprintf(f,
      "ply\nformat binary_little_endian 1.0\ncomment %d pbrt curves, 4 lines each\nelement vertex %zu\n"
      "property float x\nproperty float y\nproperty float z\nproperty float nx\nproperty float ny\nproperty float nz\n"
      "property float radius\nelement line %zu\nproperty list uchar int vertex_indices\nend_header\n",
      n, nv, nl);
  std::vector<float> row(7 * nv);
  for (size_t v = 0; v < nv; v++) {
    memcpy(&row[7 * v], &pos[3 * v], 12), memcpy(&row[7 * v + 3], &nrm[3 * v], 12);
    row[7 * v + 6] = rad[v];
  }
  fwrite(row.data(), 4, row.size(), f);
  std::vector<unsigned char> lbuf(9 * nl);
  for (size_t l = 0; l < nl; l++) {
    lbuf[9 * l] = 2;
    memcpy(&lbuf[9 * l + 1], &lines[2 * l], 8);
  }
  fwrite(lbuf.data(), 1, lbuf.size(), f);
  fclose(f);
  std::cout << "ycurves2ply: " << n << " curves -> " << nv << " vertices, " << nl << " lines -> " << out << "\n";
  return 0;
}
