// Developer micro-test: the quad-parallel "push hit children by visiting rank" step, isolated from
// any scene memory, checked against the sequential formulation for all masks / sign combinations.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define YH_QUAD_XOR1 0xB1
#define YH_QUAD_XOR2 0x4E
template <int CTRL> __device__ __forceinline__ int dpp_i(int v) { return __builtin_amdgcn_mov_dpp(v, CTRL, 0xF, 0xF, true); }
__global__ void k(const unsigned* in /* per quad: m, lsign, axes, refbase */, unsigned* out /* per quad: cur, sp, s0,s1,s2 */) {
  __shared__ unsigned stk[8 * 64];
  unsigned quad = threadIdx.x >> 2, q = threadIdx.x & 3;
  unsigned gq = blockIdx.x * 64 + quad;
  unsigned m = in[4 * gq], lsign = in[4 * gq + 1], axes = in[4 * gq + 2], ref = in[4 * gq + 3] + q;
  bool h = (m >> q) & 1;
  __attribute__((address_space(3))) unsigned* lstk = (__attribute__((address_space(3))) unsigned*)stk + quad;
  int sp = 2;
  const int STRIDE = 64;
  unsigned pair = q >> 1;
  unsigned sgn  = (lsign >> ((axes >> (2 + 2 * pair)) & 3)) & 1;
  unsigned s0_  = (lsign >> (axes & 3)) & 1;
  unsigned rank = ((pair ^ s0_) << 1) | ((q & 1) ^ sgn);
  unsigned bit  = h ? (1u << rank) : 0u;
  unsigned M    = bit | (unsigned)dpp_i<YH_QUAD_XOR1>((int)bit);
  M |= (unsigned)dpp_i<YH_QUAD_XOR2>((int)M);
  bool     first = h && (M & (bit - 1)) == 0;
  unsigned after = (unsigned)__popc(M >> (rank + 1));
  if (h && !first) lstk[(sp + (int)after) * STRIDE] = ref;
  unsigned mine = first ? ref : 0u;
  mine |= (unsigned)dpp_i<YH_QUAD_XOR1>((int)mine);
  mine |= (unsigned)dpp_i<YH_QUAD_XOR2>((int)mine);
  int nh = __popc(M);
  sp += nh > 0 ? nh - 1 : 0;
  unsigned cur = nh > 0 ? mine : 0xFFFFFFFFu;
  __syncthreads();
  if (q == 0) {
    out[8 * gq] = cur, out[8 * gq + 1] = sp, out[8 * gq + 2] = M;
    for (int k2 = 2; k2 < 5; k2++) out[8 * gq + 1 + k2] = k2 < sp ? lstk[k2 * STRIDE] : 0;
  }
  if (q == 1) out[8 * gq + 6] = cur;
  if (q == 3) out[8 * gq + 7] = cur;
}
int main() {
  std::vector<unsigned> in;
  for (unsigned m = 0; m < 16; m++) for (unsigned ls = 0; ls < 8; ls++) for (unsigned ax = 0; ax < 27; ax++) {
    unsigned axes = (ax % 3) | ((ax / 3 % 3) << 2) | ((ax / 9) << 4);
    in.push_back(m), in.push_back(ls), in.push_back(axes), in.push_back(100);
  }
  int nq = in.size() / 4, nb = (nq + 63) / 64;
  in.resize(nb * 64 * 4, 0);
  unsigned *din, *dout;
  hipMalloc(&din, in.size() * 4), hipMalloc(&dout, nb * 64 * 8 * 4);
  hipMemcpy(din, in.data(), in.size() * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(nb), dim3(256), 0, 0, din, dout);
  std::vector<unsigned> out(nb * 64 * 8);
  hipMemcpy(out.data(), dout, out.size() * 4, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int g = 0; g < nq; g++) {
    unsigned m = in[4 * g], ls = in[4 * g + 1], axes = in[4 * g + 2];
    unsigned s0 = (ls >> (axes & 3)) & 1, sl = (ls >> ((axes >> 2) & 3)) & 1, sr = (ls >> ((axes >> 4) & 3)) & 1;
    unsigned r[4] = {100, 101, 102, 103};
    bool hh[4] = {(bool)(m & 1), (bool)(m & 2), (bool)(m & 4), (bool)(m & 8)};
    unsigned la = sl ? r[1] : r[0], lb = sl ? r[0] : r[1]; bool ha = sl ? hh[1] : hh[0], hb = sl ? hh[0] : hh[1];
    unsigned ra = sr ? r[3] : r[2], rb = sr ? r[2] : r[3]; bool hc = sr ? hh[3] : hh[2], hd = sr ? hh[2] : hh[3];
    unsigned o[4] = {s0 ? ra : la, s0 ? rb : lb, s0 ? la : ra, s0 ? lb : rb};
    bool gg[4] = {s0 ? hc : ha, s0 ? hd : hb, s0 ? ha : hc, s0 ? hb : hd};
    std::vector<unsigned> st; unsigned nxt = 0xFFFFFFFFu;
    if (gg[3]) nxt = o[3];
    for (int k2 = 2; k2 >= 0; k2--) if (gg[k2]) { if (nxt != 0xFFFFFFFFu) st.push_back(nxt); nxt = o[k2]; }
    bool ok = out[8 * g] == nxt && out[8 * g + 1] == 2 + st.size() && out[8 * g + 6] == nxt && out[8 * g + 7] == nxt;
    for (size_t i = 0; i < st.size(); i++) ok = ok && out[8 * g + 3 + i] == st[i];
    if (!ok && bad++ < 10) printf("MISMATCH m=%u ls=%u axes=%u: got cur %u sp %u M %u stack %u %u %u (lane1 cur %u lane3 cur %u), want cur %u stack size %zu\n", m, ls, axes,
        out[8 * g], out[8 * g + 1], out[8 * g + 2], out[8 * g + 3], out[8 * g + 4], out[8 * g + 5], out[8 * g + 6], out[8 * g + 7], nxt, st.size());
  }
  printf("%d quads checked, %d mismatches\n", nq, bad);
  return 0;
}
