// Same-box A/B of the decode layer: the five-launch split-K form (decode.hip) against the single-writer form (decode_sw.hip), both
// through the library's C ABI, 28 layers of 1.5B-shape random weights, 16 rows, captured graphs.  Also checks every single-writer
// kernel of layer 0 against a CPU evaluation of the same arithmetic.  Build: tools/probes/build_decode_sw_probe.sh.
//   usage: decode_sw_probe [pos=266] [reps=20]
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <string>
#include <vector>
#include <functional>
#include "unigen_hip.h"
#include <algorithm>
extern "C" void ug_decode_sw_set_trace(unsigned long long* p);   // probe builds of the library only (tools/probes/decode_sw_trace.patch)

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)
#define UG(x) do { int r_ = (x); if (r_ != 0) { printf("ug error %d (%s) at %s:%d\n", r_, ug_last_error(), __FILE__, __LINE__); exit(1); } } while (0)

typedef unsigned short bf16_t;
static float bf2f(bf16_t v) { uint32_t u = ((uint32_t)v) << 16; float f; memcpy(&f, &u, 4); return f; }
static bf16_t f2bf(float f) { uint32_t u; memcpy(&u, &f, 4); if ((u & 0x7fffffff) > 0x7f800000) return 0x7fc0; u += 0x7fff + ((u >> 16) & 1); return (bf16_t)(u >> 16); }
static float rb(float f) { return bf2f(f2bf(f)); }
static uint64_t rng_s = 0x9E3779B97F4A7C15ull;
static float urand() { rng_s ^= rng_s << 13; rng_s ^= rng_s >> 7; rng_s ^= rng_s << 17; return (float)((rng_s >> 11) * (1.0 / 9007199254740992.0)); }
static float nrand() { float a = urand(), b = urand(); return sqrtf(-2.f * logf(a + 1e-12f)) * cosf(6.2831853f * b); }

template <typename T> T* dalloc(size_t n) { T* p; CK(hipMalloc(&p, n * sizeof(T))); CK(hipMemset(p, 0, n * sizeof(T))); return p; }
static bf16_t* upload_bf16(const std::vector<bf16_t>& v) { bf16_t* p = dalloc<bf16_t>(v.size()); CK(hipMemcpy(p, v.data(), v.size() * 2, hipMemcpyHostToDevice)); return p; }
static float* upload_f32(const std::vector<float>& v) { float* p = dalloc<float>(v.size()); CK(hipMemcpy(p, v.data(), v.size() * 4, hipMemcpyHostToDevice)); return p; }
// (uniform with the requested standard deviation: 1.4 G values are drawn, Box-Muller would take most of a minute)
static std::vector<bf16_t> rand_bf16(size_t n, float scale) { std::vector<bf16_t> v(n); for (auto& e : v) e = f2bf((urand() * 2.f - 1.f) * 1.7320508f * scale); return v; }

constexpr int R = 16, H = 1536, I = 8960, HQ = 12, HK = 2, HD = 128, NQKV = (HQ + 2 * HK) * HD, V = 8192, L = 28, TMAX = 512, MAXPOS = 512;
const float EPS = 1e-6f;

struct Layer { bf16_t *wqkv, *bqkv, *wo, *wgu, *wdown; float *ln1, *ln2; bf16_t *ck, *cv; };

static double rel_err(const std::vector<float>& a, const std::vector<float>& b) {
  double num = 0, den = 0;
  for (size_t i = 0; i < a.size(); ++i) { num += (double)(a[i] - b[i]) * (a[i] - b[i]); den += (double)b[i] * b[i]; }
  return sqrt(num / (den + 1e-30));
}

int main(int argc, char** argv) {
  const int pos = argc > 1 ? atoi(argv[1]) : 266;
  const int reps = argc > 2 ? atoi(argv[2]) : 20;
  hipStream_t st; CK(hipStreamCreate(&st));
  std::vector<Layer> ly(L);
  // host copies of layer 0 for the CPU check
  std::vector<bf16_t> h_wqkv, h_bqkv, h_wo, h_wgu, h_wdown, h_ck, h_cv; std::vector<float> h_ln1, h_ln2;
  for (int l = 0; l < L; ++l) {
    auto wqkv = rand_bf16((size_t)NQKV * H, 0.03f), bq = rand_bf16(NQKV, 0.1f), wo = rand_bf16((size_t)H * H, 0.03f);
    auto wgu = rand_bf16((size_t)2 * I * H, 0.03f), wd = rand_bf16((size_t)H * I, 0.012f);
    std::vector<float> l1(H), l2(H); for (auto& e : l1) e = 1.f + 0.1f * nrand(); for (auto& e : l2) e = 1.f + 0.1f * nrand();
    auto ck = rand_bf16((size_t)R * HK * TMAX * HD, 1.f), cv = rand_bf16((size_t)R * HK * TMAX * HD, 1.f);
    ly[l] = Layer{upload_bf16(wqkv), upload_bf16(bq), upload_bf16(wo), upload_bf16(wgu), upload_bf16(wd), upload_f32(l1), upload_f32(l2), upload_bf16(ck), upload_bf16(cv)};
    if (l == 0) { h_wqkv = wqkv; h_bqkv = bq; h_wo = wo; h_wgu = wgu; h_wdown = wd; h_ln1 = l1; h_ln2 = l2; h_ck = ck; h_cv = cv; }
  }
  auto whead = rand_bf16((size_t)V * H, 0.03f);
  bf16_t* d_whead = upload_bf16(whead);
  std::vector<float> lnf(H); for (auto& e : lnf) e = 1.f + 0.1f * nrand();
  float* d_lnf = upload_f32(lnf);
  std::vector<float> cs((size_t)MAXPOS * 64), sn((size_t)MAXPOS * 64);
  for (int p = 0; p < MAXPOS; ++p) for (int i = 0; i < 64; ++i) { const double f = p * pow(1e6, -(double)i / 64.0); cs[p * 64 + i] = (float)cos(f); sn[p * 64 + i] = (float)sin(f); }
  float *d_cs = upload_f32(cs), *d_sn = upload_f32(sn);
  std::vector<float> h0((size_t)R * H); for (auto& e : h0) e = nrand();
  float* d_h = upload_f32(h0);
  float* d_h0 = upload_f32(h0);
  int* d_pos = dalloc<int>(1); CK(hipMemcpy(d_pos, &pos, 4, hipMemcpyHostToDevice));
  bf16_t *d_q = dalloc<bf16_t>((size_t)R * HQ * HD), *d_o = dalloc<bf16_t>((size_t)R * HQ * HD), *d_act = dalloc<bf16_t>((size_t)R * I);
  float* d_logits = dalloc<float>((size_t)R * V);
  // old-form scratch
  float *acc_qkv = dalloc<float>((size_t)R * NQKV), *acc_gu = dalloc<float>((size_t)R * 2 * I), *acc_o = dalloc<float>((size_t)R * H), *acc_down = dalloc<float>((size_t)R * H);
  float *x_mid = dalloc<float>((size_t)R * H), *ss_attn = dalloc<float>(32), *ss_mlp = dalloc<float>(32), *acc_head = dalloc<float>((size_t)R * V);
  bf16_t* d_hn = dalloc<bf16_t>((size_t)R * H);
  const float scale = 1.f / sqrtf((float)HD);

  // ------------------------------------------------------------ correctness: layer 0 + head, kernel by kernel, vs CPU
  {
    std::vector<float> h = h0;
    // qkv
    UG(ug_decode_sw_qkv(d_h, nullptr, 0, nullptr, ly[0].ln1, EPS, R, H, ly[0].wqkv, H, ly[0].bqkv, d_cs, d_sn, d_pos, d_q, HQ * HD, ly[0].ck, ly[0].cv, HQ, HK, HD, TMAX, MAXPOS, st));
    CK(hipStreamSynchronize(st));
    std::vector<float> xn((size_t)R * H);
    auto norm = [&](const std::vector<float>& hh, const std::vector<float>& w) {
      for (int r = 0; r < R; ++r) { double ss = 0; for (int k = 0; k < H; ++k) ss += (double)hh[r * H + k] * hh[r * H + k];
        const float rs = 1.f / sqrtf((float)(ss / H) + EPS); for (int k = 0; k < H; ++k) xn[r * H + k] = rb(w[k] * (hh[r * H + k] * rs)); }
    };
    norm(h, h_ln1);
    std::vector<float> qkv((size_t)R * NQKV);
    for (int r = 0; r < R; ++r) for (int n = 0; n < NQKV; ++n) { double a = 0; for (int k = 0; k < H; ++k) a += (double)xn[r * H + k] * bf2f(h_wqkv[(size_t)n * H + k]); qkv[r * NQKV + n] = rb((float)a + bf2f(h_bqkv[n])); }
    for (int r = 0; r < R; ++r) for (int hd = 0; hd < HQ + HK; ++hd) for (int i = 0; i < 64; ++i) {
      float& x1 = qkv[r * NQKV + hd * HD + i]; float& x2 = qkv[r * NQKV + hd * HD + i + 64];
      const float c = cs[pos * 64 + i], s = sn[pos * 64 + i]; const float o1 = rb(x1 * c - x2 * s), o2 = rb(x2 * c + x1 * s); x1 = o1; x2 = o2; }
    std::vector<bf16_t> gq((size_t)R * HQ * HD), gk((size_t)R * HK * TMAX * HD), gv(gk.size());
    CK(hipMemcpy(gq.data(), d_q, gq.size() * 2, hipMemcpyDeviceToHost)); CK(hipMemcpy(gk.data(), ly[0].ck, gk.size() * 2, hipMemcpyDeviceToHost)); CK(hipMemcpy(gv.data(), ly[0].cv, gv.size() * 2, hipMemcpyDeviceToHost));
    std::vector<float> a1, b1;
    for (int r = 0; r < R; ++r) for (int c = 0; c < HQ * HD; ++c) { a1.push_back(bf2f(gq[r * HQ * HD + c])); b1.push_back(qkv[r * NQKV + c]); }
    printf("check qkv  q rel err %.3e\n", rel_err(a1, b1));
    a1.clear(); b1.clear();
    for (int r = 0; r < R; ++r) for (int hk = 0; hk < HK; ++hk) for (int d = 0; d < HD; ++d) {
      a1.push_back(bf2f(gk[(((size_t)r * HK + hk) * TMAX + pos) * HD + d])); b1.push_back(qkv[r * NQKV + (HQ + hk) * HD + d]);
      a1.push_back(bf2f(gv[(((size_t)r * HK + hk) * TMAX + pos) * HD + d])); b1.push_back(qkv[r * NQKV + (HQ + HK + hk) * HD + d]); }
    printf("check qkv  k/v rel err %.3e\n", rel_err(a1, b1));
    // untouched cache rows must be untouched
    size_t changed = 0; for (size_t i = 0; i < gk.size(); ++i) { const size_t t = (i / HD) % TMAX; if ((int)t != pos && (gk[i] != h_ck[i] || gv[i] != h_cv[i])) ++changed; }
    printf("check qkv  cache rows other than pos changed: %zu\n", changed);
    // attention on the GPU's q and cache
    UG(ug_attn_decode_q(d_q, HQ * HD, ly[0].ck, ly[0].cv, nullptr, d_o, HQ * HD, R, HQ, HK, HD, TMAX, d_pos, scale, st));
    CK(hipStreamSynchronize(st));
    std::vector<bf16_t> go((size_t)R * HQ * HD); CK(hipMemcpy(go.data(), d_o, go.size() * 2, hipMemcpyDeviceToHost));
    a1.clear(); b1.clear();
    std::vector<float> oref((size_t)R * HQ * HD);
    for (int r = 0; r < R; ++r) for (int hq = 0; hq < HQ; ++hq) {
      const int hk = hq / (HQ / HK); std::vector<double> sc(pos + 1); double mx = -1e30;
      for (int t = 0; t <= pos; ++t) { double d = 0; for (int e = 0; e < HD; ++e) d += (double)bf2f(gq[r * HQ * HD + hq * HD + e]) * bf2f(gk[(((size_t)r * HK + hk) * TMAX + t) * HD + e]); sc[t] = d * scale; mx = fmax(mx, sc[t]); }
      double lsum = 0; for (int t = 0; t <= pos; ++t) { sc[t] = exp(sc[t] - mx); lsum += sc[t]; }
      for (int e = 0; e < HD; ++e) { double o = 0; for (int t = 0; t <= pos; ++t) o += (double)rb((float)sc[t]) * bf2f(gv[(((size_t)r * HK + hk) * TMAX + t) * HD + e]); oref[r * HQ * HD + hq * HD + e] = (float)(o / lsum); }
    }
    for (size_t i = 0; i < go.size(); ++i) { a1.push_back(bf2f(go[i])); b1.push_back(oref[i]); }
    printf("check attn o rel err %.3e\n", rel_err(a1, b1));
    // o projection + residual (on the GPU's o)
    UG(ug_decode_sw_resid(d_o, HQ * HD, R, ly[0].wo, H, H, HQ * HD, d_h, st));
    CK(hipStreamSynchronize(st));
    std::vector<float> gh((size_t)R * H); CK(hipMemcpy(gh.data(), d_h, gh.size() * 4, hipMemcpyDeviceToHost));
    std::vector<float> href = h;
    for (int r = 0; r < R; ++r) for (int n = 0; n < H; ++n) { double a = 0; for (int k = 0; k < HQ * HD; ++k) a += (double)bf2f(go[r * HQ * HD + k]) * bf2f(h_wo[(size_t)n * H + k]); href[r * H + n] += rb((float)a); }
    printf("check o_proj h rel err %.3e\n", rel_err(gh, href));
    // gate_up on the GPU's h
    UG(ug_decode_sw_gate_up(d_h, nullptr, 0, nullptr, ly[0].ln2, EPS, R, H, ly[0].wgu, H, I, d_act, I, st));
    CK(hipStreamSynchronize(st));
    std::vector<bf16_t> gact((size_t)R * I); CK(hipMemcpy(gact.data(), d_act, gact.size() * 2, hipMemcpyDeviceToHost));
    norm(gh, h_ln2);
    a1.clear(); b1.clear();
    for (int r = 0; r < R; ++r) for (int c = 0; c < I; ++c) {
      double g = 0, u = 0; for (int k = 0; k < H; ++k) { g += (double)xn[r * H + k] * bf2f(h_wgu[(size_t)c * H + k]); u += (double)xn[r * H + k] * bf2f(h_wgu[(size_t)(I + c) * H + k]); }
      const float gb = rb((float)g), ub = rb((float)u); const float sl = rb(gb / (1.f + expf(-gb)));
      a1.push_back(bf2f(gact[(size_t)r * I + c])); b1.push_back(rb(sl * ub)); }
    printf("check gate_up act rel err %.3e\n", rel_err(a1, b1));
    // down on the GPU's act
    UG(ug_decode_sw_resid(d_act, I, R, ly[0].wdown, I, H, I, d_h, st));
    CK(hipStreamSynchronize(st));
    std::vector<float> gh2((size_t)R * H); CK(hipMemcpy(gh2.data(), d_h, gh2.size() * 4, hipMemcpyDeviceToHost));
    href = gh;
    for (int r = 0; r < R; ++r) for (int n = 0; n < H; ++n) { double a = 0; for (int k = 0; k < I; ++k) a += (double)bf2f(gact[(size_t)r * I + k]) * bf2f(h_wdown[(size_t)n * I + k]); href[r * H + n] += rb((float)a); }
    printf("check down h rel err %.3e\n", rel_err(gh2, href));
    // head
    UG(ug_decode_sw_head(d_h, nullptr, 0, nullptr, d_lnf, EPS, R, H, d_whead, H, V, d_logits, V, nullptr, nullptr, st));
    CK(hipStreamSynchronize(st));
    std::vector<float> gl((size_t)R * V); CK(hipMemcpy(gl.data(), d_logits, gl.size() * 4, hipMemcpyDeviceToHost));
    norm(gh2, lnf);
    std::vector<float> lref((size_t)R * V);
    for (int r = 0; r < R; ++r) for (int n = 0; n < V; ++n) { double a = 0; for (int k = 0; k < H; ++k) a += (double)xn[r * H + k] * bf2f(whead[(size_t)n * H + k]); lref[(size_t)r * V + n] = (float)a; }
    printf("check head logits rel err %.3e\n", rel_err(gl, lref));
    // pending-accumulator input: gate_up(h, pend) == gate_up(h + float(bf16(pend))), x_out = that sum, bit for bit
    {
      std::vector<float> pend((size_t)R * H), hs((size_t)R * H); for (auto& e : pend) e = nrand() * 0.5f;
      for (size_t i = 0; i < hs.size(); ++i) hs[i] = h0[i] + rb(pend[i]);
      float *d_pend = upload_f32(pend), *d_hs = upload_f32(hs), *d_xo = dalloc<float>((size_t)R * H);
      bf16_t* d_act2 = dalloc<bf16_t>((size_t)R * I);
      UG(ug_decode_sw_gate_up(d_h0, d_pend, H, d_xo, ly[0].ln2, EPS, R, H, ly[0].wgu, H, I, d_act, I, st));
      UG(ug_decode_sw_gate_up(d_hs, nullptr, 0, nullptr, ly[0].ln2, EPS, R, H, ly[0].wgu, H, I, d_act2, I, st));
      CK(hipStreamSynchronize(st));
      std::vector<bf16_t> a1v((size_t)R * I), a2v((size_t)R * I); std::vector<float> xo((size_t)R * H);
      CK(hipMemcpy(a1v.data(), d_act, a1v.size() * 2, hipMemcpyDeviceToHost)); CK(hipMemcpy(a2v.data(), d_act2, a2v.size() * 2, hipMemcpyDeviceToHost)); CK(hipMemcpy(xo.data(), d_xo, xo.size() * 4, hipMemcpyDeviceToHost));
      size_t da = 0, dx = 0; for (size_t i = 0; i < a1v.size(); ++i) da += a1v[i] != a2v[i]; for (size_t i = 0; i < xo.size(); ++i) dx += memcmp(&xo[i], &hs[i], 4) != 0;
      printf("check pend  act values differing %zu of %zu, x_out values differing %zu of %zu\n", da, a1v.size(), dx, xo.size());
    }
    fflush(stdout);
  }

  // ------------------------------------------------------------ timing
  auto new_layer = [&](int l) {
    UG(ug_decode_sw_qkv(d_h, nullptr, 0, nullptr, ly[l].ln1, EPS, R, H, ly[l].wqkv, H, ly[l].bqkv, d_cs, d_sn, d_pos, d_q, HQ * HD, ly[l].ck, ly[l].cv, HQ, HK, HD, TMAX, MAXPOS, st));
    UG(ug_attn_decode_q(d_q, HQ * HD, ly[l].ck, ly[l].cv, nullptr, d_o, HQ * HD, R, HQ, HK, HD, TMAX, d_pos, scale, st));
    UG(ug_decode_sw_resid(d_o, HQ * HD, R, ly[l].wo, H, H, HQ * HD, d_h, st));
    UG(ug_decode_sw_gate_up(d_h, nullptr, 0, nullptr, ly[l].ln2, EPS, R, H, ly[l].wgu, H, I, d_act, I, st));
    UG(ug_decode_sw_resid(d_act, I, R, ly[l].wdown, I, H, I, d_h, st));
  };
  auto old_layer = [&](int l) {
    UG(ug_decode_gemv_resid_norm(d_h, acc_down, H, ly[l].ln1, x_mid, ss_attn, R, ly[l].wqkv, H, acc_qkv, NQKV, NQKV, H, acc_gu, (int64_t)R * 2 * I, nullptr, 0, ss_mlp, st));
    UG(ug_attn_decode_fused(acc_qkv, NQKV, ss_attn, EPS, H, ly[l].bqkv, d_cs, d_sn, d_pos, ly[l].ck, ly[l].cv, nullptr, d_o, HQ * HD, R, HQ, HK, HD, TMAX, MAXPOS, scale, st));
    UG(ug_decode_gemv(d_o, HQ * HD, R, ly[l].wo, H, acc_o, H, H, HQ * HD, acc_qkv, (int64_t)R * NQKV, acc_down, (int64_t)R * H, ss_attn, st));
    UG(ug_decode_gemv_resid_norm(x_mid, acc_o, H, ly[l].ln2, d_h, ss_mlp, R, ly[l].wgu, H, acc_gu, 2 * I, 2 * I, H, nullptr, 0, nullptr, 0, nullptr, st));
    UG(ug_decode_gemv_swiglu(acc_gu, 2 * I, ss_mlp, EPS, H, R, ly[l].wdown, I, acc_down, H, H, I, acc_o, (int64_t)R * H, nullptr, 0, nullptr, st));
  };
  // mixed layer: split-K q/k/v, attention and o projection as shipped; single-writer gate/up fed by (x_mid, acc_o); split-K down on the
  // finished bf16 act (clears acc_o); the next layer's q/k/v consumes (x, acc_down) as before
  auto mixed_layer = [&](int l) {
    UG(ug_decode_gemv_resid_norm(d_h, acc_down, H, ly[l].ln1, x_mid, ss_attn, R, ly[l].wqkv, H, acc_qkv, NQKV, NQKV, H, nullptr, 0, nullptr, 0, nullptr, st));
    UG(ug_attn_decode_fused(acc_qkv, NQKV, ss_attn, EPS, H, ly[l].bqkv, d_cs, d_sn, d_pos, ly[l].ck, ly[l].cv, nullptr, d_o, HQ * HD, R, HQ, HK, HD, TMAX, MAXPOS, scale, st));
    UG(ug_decode_gemv(d_o, HQ * HD, R, ly[l].wo, H, acc_o, H, H, HQ * HD, acc_qkv, (int64_t)R * NQKV, acc_down, (int64_t)R * H, ss_attn, st));
    UG(ug_decode_sw_gate_up(x_mid, acc_o, H, d_h, ly[l].ln2, EPS, R, H, ly[l].wgu, H, I, d_act, I, st));
    UG(ug_decode_gemv(d_act, I, R, ly[l].wdown, I, acc_down, H, H, I, acc_o, (int64_t)R * H, nullptr, 0, nullptr, st));
  };
  // G1 layer: split-K q/k/v + attention as shipped; single-writer o (h finished in place) and gate/up; split-K down on the bf16 act.
  float* accD[2] = {acc_down, acc_o};                 // the down projection's accumulators alternate by layer; acc_o is free in this chain
  float* xbuf[2] = {d_h, x_mid};
  float* zeros = dalloc<float>((size_t)R * H);
  auto g1_layer = [&](int l) {
    float* xin = xbuf[l & 1]; float* xout = xbuf[(l + 1) & 1];
    const float* pend = l == 0 ? zeros : accD[(l - 1) & 1];
    UG(ug_decode_gemv_resid_norm(xin, pend, H, ly[l].ln1, xout, ss_attn, R, ly[l].wqkv, H, acc_qkv, NQKV, NQKV, H, nullptr, 0, nullptr, 0, nullptr, st));
    UG(ug_attn_decode_fused(acc_qkv, NQKV, ss_attn, EPS, H, ly[l].bqkv, d_cs, d_sn, d_pos, ly[l].ck, ly[l].cv, nullptr, d_o, HQ * HD, R, HQ, HK, HD, TMAX, MAXPOS, scale, st));
    UG(ug_decode_sw_resid(d_o, HQ * HD, R, ly[l].wo, H, H, HQ * HD, xout, st));
    UG(ug_decode_sw_gate_up(xout, nullptr, 0, nullptr, ly[l].ln2, EPS, R, H, ly[l].wgu, H, I, d_act, I, st));
    UG(ug_decode_gemv(d_act, I, R, ly[l].wdown, I, accD[l & 1], H, H, I, acc_qkv, (int64_t)R * NQKV, l == 0 ? nullptr : accD[(l - 1) & 1], l == 0 ? 0 : (int64_t)R * H, ss_attn, st));
  };
  struct Case { const char* name; std::function<void()> body; double units; };
  auto time_graph = [&](const char* name, const std::function<void()>& body, double per) {
    CK(hipMemcpyAsync(d_h, d_h0, (size_t)R * H * 4, hipMemcpyDeviceToDevice, st));
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
    body();
    CK(hipStreamEndCapture(st, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    for (int i = 0; i < 3; ++i) CK(hipGraphLaunch(ge, st));
    CK(hipStreamSynchronize(st));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e30f, sum = 0;
    for (int i = 0; i < reps; ++i) {
      CK(hipEventRecord(e0, st)); CK(hipGraphLaunch(ge, st)); CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1)); best = fminf(best, ms); sum += ms;
    }
    printf("%-28s best %8.2f us  mean %8.2f us   per unit: best %6.2f mean %6.2f us\n", name, best * 1e3, sum / reps * 1e3, best * 1e3 / per, sum / reps * 1e3 / per);
    fflush(stdout);
    CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
  };
  const char* only = getenv("PROBE_ONLY");
  auto want = [&](const char* n) { return !only || strstr(only, n); };
  if (want("chain")) {
    time_graph("old chain (28 layers)", [&] { for (int l = 0; l < L; ++l) old_layer(l); }, L);
    time_graph("new chain (28 layers)", [&] { for (int l = 0; l < L; ++l) new_layer(l); }, L);
    time_graph("old chain (28 layers)", [&] { for (int l = 0; l < L; ++l) old_layer(l); }, L);
    time_graph("new chain (28 layers)", [&] { for (int l = 0; l < L; ++l) new_layer(l); }, L);
    CK(hipMemsetAsync(acc_down, 0, (size_t)R * H * 4, st)); CK(hipMemsetAsync(acc_o, 0, (size_t)R * H * 4, st)); CK(hipMemsetAsync(acc_qkv, 0, (size_t)R * NQKV * 4, st));
    time_graph("mixed chain (28 layers)", [&] { for (int l = 0; l < L; ++l) mixed_layer(l); }, L);
    time_graph("mixed chain (28 layers)", [&] { for (int l = 0; l < L; ++l) mixed_layer(l); }, L);
    CK(hipMemsetAsync(acc_down, 0, (size_t)R * H * 4, st)); CK(hipMemsetAsync(acc_o, 0, (size_t)R * H * 4, st)); CK(hipMemsetAsync(acc_qkv, 0, (size_t)R * NQKV * 4, st)); CK(hipMemsetAsync(ss_attn, 0, 128, st));
    time_graph("G1 chain (28 layers)", [&] { for (int l = 0; l < L; ++l) g1_layer(l); }, L);
    time_graph("G1 chain (28 layers)", [&] { for (int l = 0; l < L; ++l) g1_layer(l); }, L);
    time_graph("new head", [&] { for (int l = 0; l < L; ++l) UG(ug_decode_sw_head(d_h, nullptr, 0, nullptr, d_lnf, EPS, R, H, d_whead, H, V, d_logits, V, nullptr, nullptr, st)); }, L);
    time_graph("old finish+head gemv", [&] { for (int l = 0; l < L; ++l) { UG(ug_decode_finish_resid_norm(acc_down, H, d_h, d_lnf, d_hn, R, H, EPS, nullptr, nullptr, st));
                                                                         UG(ug_decode_gemv(d_hn, H, R, d_whead, H, acc_head, V, V, H, nullptr, 0, nullptr, 0, nullptr, st)); } }, L);
  }
  if (want("each")) {
    // one kernel type x 28 layers back to back (independent weights; each launch still waits for its predecessor)
    time_graph("new qkv x28", [&] { for (int l = 0; l < L; ++l) UG(ug_decode_sw_qkv(d_h, nullptr, 0, nullptr, ly[l].ln1, EPS, R, H, ly[l].wqkv, H, ly[l].bqkv, d_cs, d_sn, d_pos, d_q, HQ * HD, ly[l].ck, ly[l].cv, HQ, HK, HD, TMAX, MAXPOS, st)); }, L);
    time_graph("new attn x28", [&] { for (int l = 0; l < L; ++l) UG(ug_attn_decode_q(d_q, HQ * HD, ly[l].ck, ly[l].cv, nullptr, d_o, HQ * HD, R, HQ, HK, HD, TMAX, d_pos, scale, st)); }, L);
    time_graph("new o x28", [&] { for (int l = 0; l < L; ++l) UG(ug_decode_sw_resid(d_o, HQ * HD, R, ly[l].wo, H, H, HQ * HD, d_h, st)); }, L);
    time_graph("new gate_up x28", [&] { for (int l = 0; l < L; ++l) UG(ug_decode_sw_gate_up(d_h, nullptr, 0, nullptr, ly[l].ln2, EPS, R, H, ly[l].wgu, H, I, d_act, I, st)); }, L);
    time_graph("new down x28", [&] { for (int l = 0; l < L; ++l) UG(ug_decode_sw_resid(d_act, I, R, ly[l].wdown, I, H, I, d_h, st)); }, L);
    time_graph("old down-as-bf16-gemv x28", [&] { for (int l = 0; l < L; ++l) UG(ug_decode_gemv(d_act, I, R, ly[l].wdown, I, acc_down, H, H, I, nullptr, 0, nullptr, 0, nullptr, st)); }, L);
    time_graph("old qkv x28", [&] { for (int l = 0; l < L; ++l) UG(ug_decode_gemv_resid_norm(d_h, acc_down, H, ly[l].ln1, x_mid, ss_attn, R, ly[l].wqkv, H, acc_qkv, NQKV, NQKV, H, nullptr, 0, nullptr, 0, nullptr, st)); }, L);
    time_graph("old attn x28", [&] { for (int l = 0; l < L; ++l) UG(ug_attn_decode_fused(acc_qkv, NQKV, ss_attn, EPS, H, ly[l].bqkv, d_cs, d_sn, d_pos, ly[l].ck, ly[l].cv, nullptr, d_o, HQ * HD, R, HQ, HK, HD, TMAX, MAXPOS, scale, st)); }, L);
    time_graph("old o x28", [&] { for (int l = 0; l < L; ++l) UG(ug_decode_gemv(d_o, HQ * HD, R, ly[l].wo, H, acc_o, H, H, HQ * HD, nullptr, 0, nullptr, 0, nullptr, st)); }, L);
    time_graph("old gate_up x28", [&] { for (int l = 0; l < L; ++l) UG(ug_decode_gemv_resid_norm(x_mid, acc_o, H, ly[l].ln2, d_h, ss_mlp, R, ly[l].wgu, H, acc_gu, 2 * I, 2 * I, H, nullptr, 0, nullptr, 0, nullptr, st)); }, L);
    time_graph("old down x28", [&] { for (int l = 0; l < L; ++l) UG(ug_decode_gemv_swiglu(acc_gu, 2 * I, ss_mlp, EPS, H, R, ly[l].wdown, I, acc_down, H, H, I, nullptr, 0, nullptr, 0, nullptr, st)); }, L);
  }
  // ------------------------------------------------------------ per-stage timeline of the single-writer kernels (trace build only)
  if (getenv("PROBE_TRACE")) {
    unsigned long long* d_tr = dalloc<unsigned long long>(256 * 16 * 8);
    struct K { const char* name; std::function<void(int)> run; int nw; };
    std::vector<K> ks = {
      {"qkv", [&](int l) { UG(ug_decode_sw_qkv(d_h, nullptr, 0, nullptr, ly[l].ln1, EPS, R, H, ly[l].wqkv, H, ly[l].bqkv, d_cs, d_sn, d_pos, d_q, HQ * HD, ly[l].ck, ly[l].cv, HQ, HK, HD, TMAX, MAXPOS, st)); }, 6},
      {"o", [&](int l) { UG(ug_decode_sw_resid(d_o, HQ * HD, R, ly[l].wo, H, H, HQ * HD, d_h, st)); }, 6},
      {"gate_up", [&](int l) { UG(ug_decode_sw_gate_up(d_h, nullptr, 0, nullptr, ly[l].ln2, EPS, R, H, ly[l].wgu, H, I, d_act, I, st)); }, 6},
      {"down", [&](int l) { UG(ug_decode_sw_resid(d_act, I, R, ly[l].wdown, I, H, I, d_h, st)); }, 12},
      {"head", [&](int l) { UG(ug_decode_sw_head(d_h, nullptr, 0, nullptr, d_lnf, EPS, R, H, d_whead, H, V, d_logits, V, nullptr, nullptr, st)); }, 6}};
    const char* stage[8] = {"entry", "loads+ring issued", "operand landed", "operand built", "tile 0 landed", "mfma done", "partials met", "stores acked"};
    for (auto& k : ks) {
      // run the chain up to layer 5 untraced, trace the kernel in layer 6 (warm, in situ), keep going
      std::vector<std::vector<double>> acc(8);
      for (int rep = 0; rep < 5; ++rep) {
        CK(hipMemsetAsync(d_tr, 0, 256 * 16 * 8 * 8, st));
        for (int l = 0; l < 6; ++l) new_layer(l);
        ug_decode_sw_set_trace(d_tr);
        k.run(6);
        ug_decode_sw_set_trace(nullptr);
        for (int l = 7; l < 9; ++l) new_layer(l);
        CK(hipStreamSynchronize(st));
        std::vector<unsigned long long> tr(256 * 16 * 8); CK(hipMemcpy(tr.data(), d_tr, tr.size() * 8, hipMemcpyDeviceToHost));
        unsigned long long t0 = ~0ull; for (int b = 0; b < 256; ++b) for (int w = 0; w < k.nw; ++w) if (tr[(b * 16 + w) * 8]) t0 = std::min(t0, tr[(b * 16 + w) * 8]);
        for (int i = 0; i < 8; ++i) for (int b = 0; b < 256; ++b) for (int w = 0; w < k.nw; ++w) { const unsigned long long v = tr[(b * 16 + w) * 8 + i]; if (v) acc[i].push_back((double)(v - t0) * 0.01); }
      }
      printf("trace %-8s (us after the first wave's entry; min / median / max over waves x 5 runs)\n", k.name);
      for (int i = 0; i < 8; ++i) { auto& v = acc[i]; if (v.empty()) continue; std::sort(v.begin(), v.end()); printf("   %-20s %6.2f %6.2f %6.2f\n", stage[i], v.front(), v[v.size() / 2], v.back()); }
      fflush(stdout);
    }
  }
  // reproducibility of the single-writer chain: two replays from the same h must agree bit for bit
  {
    std::vector<float> r1((size_t)R * H), r2((size_t)R * H);
    for (int rep = 0; rep < 2; ++rep) {
      CK(hipMemcpyAsync(d_h, d_h0, (size_t)R * H * 4, hipMemcpyDeviceToDevice, st));
      for (int l = 0; l < L; ++l) new_layer(l);
      CK(hipStreamSynchronize(st));
      CK(hipMemcpy(rep ? r2.data() : r1.data(), d_h, r1.size() * 4, hipMemcpyDeviceToHost));
    }
    size_t diff = 0; bool finite = true; for (size_t i = 0; i < r1.size(); ++i) { diff += memcmp(&r1[i], &r2[i], 4) != 0; finite = finite && isfinite(r1[i]); }
    printf("new chain twice from the same input: %zu of %zu values differ, finite=%d\n", diff, r1.size(), (int)finite);
  }
  return 0;
}
