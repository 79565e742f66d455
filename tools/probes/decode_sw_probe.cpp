// Same-box A/B of the decode layer through the library's C ABI: the five-launch split-K form (decode.hip) against the shipped mix
// (split-K q/k/v + attention + down, single-writer o + gate/up, decode_sw.hip), 28 layers of 1.5B-shape random weights, 16 rows,
// captured graphs; every single-writer kernel of layer 0 is checked against a CPU evaluation of the same arithmetic.
// The all-single-writer layer (q/k/v with a RoPE epilogue, attention on a finished q, down with full-K workgroups), the deferred-rstd
// and partial-staging variants and the per-wave trace stamps measured in round 6 are in tools/probes/decode_sw_full.hip.inc
// (commit b353337 builds them; outputs: profiles/r6_probe1..5.txt, summary profiles/r06_decode_forms.md).
// Build: tools/probes/build_decode_sw_probe.sh.   usage: decode_sw_probe [pos=266] [reps=20]
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <string>
#include <vector>
#include <algorithm>
#include <functional>
#include "unigen_hip.h"
#include <algorithm>

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
    std::vector<float> xn((size_t)R * H);
    auto norm = [&](const std::vector<float>& hh, const std::vector<float>& w) {
      for (int r = 0; r < R; ++r) { double ss = 0; for (int k = 0; k < H; ++k) ss += (double)hh[r * H + k] * hh[r * H + k];
        const float rs = 1.f / sqrtf((float)(ss / H) + EPS); for (int k = 0; k < H; ++k) xn[r * H + k] = rb(w[k] * (hh[r * H + k] * rs)); }
    };
    std::vector<float> a1, b1;
    std::vector<bf16_t> go = rand_bf16((size_t)R * HQ * HD, 1.f);
    CK(hipMemcpy(d_o, go.data(), go.size() * 2, hipMemcpyHostToDevice));
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
    // down on the GPU's act: split-K in k-blocks of seven slabs, partial tiles pre-reduced in LDS, one atomic per element and k-block
    CK(hipMemsetAsync(acc_down, 0, (size_t)R * H * 4, st));
    UG(ug_decode_sw_kblock(d_act, I, R, ly[0].wdown, I, acc_down, H, H, I, nullptr, 0, nullptr, 0, nullptr, st));
    CK(hipStreamSynchronize(st));
    std::vector<float> gd((size_t)R * H); CK(hipMemcpy(gd.data(), acc_down, gd.size() * 4, hipMemcpyDeviceToHost));
    std::vector<float> dref((size_t)R * H);
    for (int r = 0; r < R; ++r) for (int n = 0; n < H; ++n) { double a = 0; for (int k = 0; k < I; ++k) a += (double)bf2f(gact[(size_t)r * I + k]) * bf2f(h_wdown[(size_t)n * I + k]); dref[r * H + n] = (float)a; }
    printf("check down (k-blocks) acc rel err %.3e\n", rel_err(gd, dref));
    std::vector<float> gh2 = gh;
    for (size_t i = 0; i < gh2.size(); ++i) gh2[i] += rb(gd[i]);
    CK(hipMemcpy(d_h, gh2.data(), gh2.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemsetAsync(acc_down, 0, (size_t)R * H * 4, st));
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
#ifdef UG_HAVE_OGU   // needs the library built with tools/probes/decode_ogu_r6.patch applied (measured and rejected: profiles/r06_decode_forms.md)
    // fused o + gate/up launch against the two separate launches, bit for bit, over many repetitions (any stale read of the in-launch
    // gather would show as a differing value)
    {
      uint32_t* d_flags = dalloc<uint32_t>(256); uint32_t* d_err = dalloc<uint32_t>(16 + 256 * 32);
      float *d_ha = dalloc<float>((size_t)R * H), *d_hb = dalloc<float>((size_t)R * H);
      bf16_t *d_acta = dalloc<bf16_t>((size_t)R * I), *d_actb = dalloc<bf16_t>((size_t)R * I);
      printf("fused o+gate/up supported: %d\n", ug_decode_sw_o_gate_up_supported(H, I, HQ * HD));
      size_t bad_h = 0, bad_a = 0; const int NREP = 200;
      std::vector<float> ha((size_t)R * H), hb((size_t)R * H); std::vector<bf16_t> aa((size_t)R * I), ab((size_t)R * I);
      for (int rep = 0; rep < NREP; ++rep) {
        const int l = rep % L;
        std::vector<bf16_t> go2 = rand_bf16((size_t)R * HQ * HD, 1.f);
        CK(hipMemcpyAsync(d_o, go2.data(), go2.size() * 2, hipMemcpyHostToDevice, st));
        CK(hipMemcpyAsync(d_ha, d_h0, (size_t)R * H * 4, hipMemcpyDeviceToDevice, st)); CK(hipMemcpyAsync(d_hb, d_h0, (size_t)R * H * 4, hipMemcpyDeviceToDevice, st));
        if (rep % 28 == 0) CK(hipMemsetAsync(d_flags, 0, 1024, st));
        UG(ug_decode_sw_resid(d_o, HQ * HD, R, ly[l].wo, H, H, HQ * HD, d_ha, st));
        UG(ug_decode_sw_gate_up(d_ha, nullptr, 0, nullptr, ly[l].ln2, EPS, R, H, ly[l].wgu, H, I, d_acta, I, st));
        UG(ug_decode_sw_o_gate_up(d_o, HQ * HD, ly[l].wo, H, d_hb, ly[l].ln2, EPS, R, H, ly[l].wgu, H, I, d_actb, I, d_flags, (uint32_t)(rep % 28) + 1, d_err, st));
        CK(hipStreamSynchronize(st));
        CK(hipMemcpy(ha.data(), d_ha, ha.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(hb.data(), d_hb, hb.size() * 4, hipMemcpyDeviceToHost));
        CK(hipMemcpy(aa.data(), d_acta, aa.size() * 2, hipMemcpyDeviceToHost)); CK(hipMemcpy(ab.data(), d_actb, ab.size() * 2, hipMemcpyDeviceToHost));
        for (size_t i = 0; i < ha.size(); ++i) bad_h += memcmp(&ha[i], &hb[i], 4) != 0;
        for (size_t i = 0; i < aa.size(); ++i) bad_a += aa[i] != ab[i];
      }
      uint32_t err = 0; CK(hipMemcpy(&err, d_err, 4, hipMemcpyDeviceToHost));
      printf("check fused o+gate/up vs separate over %d runs: h values differing %zu, act values differing %zu, err word %u\n", NREP, bad_h, bad_a, err);
    }
#endif
    fflush(stdout);
  }

  // ------------------------------------------------------------ timing
  auto old_layer = [&](int l) {
    UG(ug_decode_gemv_resid_norm(d_h, acc_down, H, ly[l].ln1, x_mid, ss_attn, R, ly[l].wqkv, H, acc_qkv, NQKV, NQKV, H, acc_gu, (int64_t)R * 2 * I, nullptr, 0, ss_mlp, st));
    UG(ug_attn_decode_fused(acc_qkv, NQKV, ss_attn, EPS, H, ly[l].bqkv, d_cs, d_sn, d_pos, ly[l].ck, ly[l].cv, nullptr, d_o, HQ * HD, R, HQ, HK, HD, TMAX, MAXPOS, scale, st));
    UG(ug_decode_gemv(d_o, HQ * HD, R, ly[l].wo, H, acc_o, H, H, HQ * HD, acc_qkv, (int64_t)R * NQKV, acc_down, (int64_t)R * H, ss_attn, st));
    UG(ug_decode_gemv_resid_norm(x_mid, acc_o, H, ly[l].ln2, d_h, ss_mlp, R, ly[l].wgu, H, acc_gu, 2 * I, 2 * I, H, nullptr, 0, nullptr, 0, nullptr, st));
    UG(ug_decode_gemv_swiglu(acc_gu, 2 * I, ss_mlp, EPS, H, R, ly[l].wdown, I, acc_down, H, H, I, acc_o, (int64_t)R * H, nullptr, 0, nullptr, st));
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
  auto g2_layer = [&](int l) {                        // G1 with the down projection in k-blocks
    float* xin = xbuf[l & 1]; float* xout = xbuf[(l + 1) & 1];
    const float* pend = l == 0 ? zeros : accD[(l - 1) & 1];
    UG(ug_decode_gemv_resid_norm(xin, pend, H, ly[l].ln1, xout, ss_attn, R, ly[l].wqkv, H, acc_qkv, NQKV, NQKV, H, nullptr, 0, nullptr, 0, nullptr, st));
    UG(ug_attn_decode_fused(acc_qkv, NQKV, ss_attn, EPS, H, ly[l].bqkv, d_cs, d_sn, d_pos, ly[l].ck, ly[l].cv, nullptr, d_o, HQ * HD, R, HQ, HK, HD, TMAX, MAXPOS, scale, st));
    UG(ug_decode_sw_resid(d_o, HQ * HD, R, ly[l].wo, H, H, HQ * HD, xout, st));
    UG(ug_decode_sw_gate_up(xout, nullptr, 0, nullptr, ly[l].ln2, EPS, R, H, ly[l].wgu, H, I, d_act, I, st));
    UG(ug_decode_sw_kblock(d_act, I, R, ly[l].wdown, I, accD[l & 1], H, H, I, acc_qkv, (int64_t)R * NQKV, l == 0 ? nullptr : accD[(l - 1) & 1], l == 0 ? 0 : (int64_t)R * H, ss_attn, st));
  };
#ifdef UG_HAVE_OGU
  uint32_t* g_flags = dalloc<uint32_t>(256); uint32_t* g_err = dalloc<uint32_t>(16 + 256 * 32);      // (+ the trace of a -DUG_OGU_TRACE build)
  auto dump_trace = [&](const char* what) {
    std::vector<unsigned long long> tr(256 * 16); CK(hipMemcpy(tr.data(), g_err + 16, tr.size() * 8, hipMemcpyDeviceToHost));
    if (!tr[0]) return;
    unsigned long long t0 = ~0ull; for (int b = 0; b < 256; ++b) { t0 = std::min(t0, tr[b * 16 + 0]); t0 = std::min(t0, tr[b * 16 + 4]); }
    const char* names[11] = {"svc start", "svc partials parked (#1)", "svc stores drained", "svc poll done", "wrk start", "wrk o tile landed", "wrk after #1", "wrk released (#2)", "wrk stream gathered", "wrk MFMA loop done", "wrk after #4"};
    printf("timeline of the last fused launch (%s), us from the first workgroup's start: min / median / max over 256 workgroups\n", what);
    for (int k = 0; k < 11; ++k) { std::vector<double> v; for (int b = 0; b < 256; ++b) v.push_back((double)(tr[b * 16 + k] - t0) * 0.01); std::sort(v.begin(), v.end());
      printf("  %-28s %6.2f %6.2f %6.2f\n", names[k], v[0], v[128], v[255]); }
    { std::vector<double> v, n; for (int b = 0; b < 256; ++b) { v.push_back((double)(tr[b * 16 + 12] - t0) * 0.01); n.push_back((double)tr[b * 16 + 13]); } std::sort(v.begin(), v.end()); std::sort(n.begin(), n.end());
      printf("  %-28s %6.2f %6.2f %6.2f   passes %g %g %g\n", "svc first poll back", v[0], v[128], v[255], n[0], n[128], n[255]); }
  };
  auto g3_layer = [&](int l) {                        // G2 with the o projection and gate/up in one launch
    float* xin = xbuf[l & 1]; float* xout = xbuf[(l + 1) & 1];
    const float* pend = l == 0 ? zeros : accD[(l - 1) & 1];
    if (l == 0) CK(hipMemsetAsync(g_flags, 0, 1024, st));
    UG(ug_decode_gemv_resid_norm(xin, pend, H, ly[l].ln1, xout, ss_attn, R, ly[l].wqkv, H, acc_qkv, NQKV, NQKV, H, nullptr, 0, nullptr, 0, nullptr, st));
    UG(ug_attn_decode_fused(acc_qkv, NQKV, ss_attn, EPS, H, ly[l].bqkv, d_cs, d_sn, d_pos, ly[l].ck, ly[l].cv, nullptr, d_o, HQ * HD, R, HQ, HK, HD, TMAX, MAXPOS, scale, st));
    UG(ug_decode_sw_o_gate_up(d_o, HQ * HD, ly[l].wo, H, xout, ly[l].ln2, EPS, R, H, ly[l].wgu, H, I, d_act, I, g_flags, (uint32_t)l + 1, g_err, st));
    UG(ug_decode_sw_kblock(d_act, I, R, ly[l].wdown, I, accD[l & 1], H, H, I, acc_qkv, (int64_t)R * NQKV, l == 0 ? nullptr : accD[(l - 1) & 1], l == 0 ? 0 : (int64_t)R * H, ss_attn, st));
  };
#endif
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
    time_graph("old chain (28 layers)", [&] { for (int l = 0; l < L; ++l) old_layer(l); }, L);
    CK(hipMemsetAsync(acc_down, 0, (size_t)R * H * 4, st)); CK(hipMemsetAsync(acc_o, 0, (size_t)R * H * 4, st)); CK(hipMemsetAsync(acc_qkv, 0, (size_t)R * NQKV * 4, st));
    CK(hipMemsetAsync(acc_down, 0, (size_t)R * H * 4, st)); CK(hipMemsetAsync(acc_o, 0, (size_t)R * H * 4, st)); CK(hipMemsetAsync(acc_qkv, 0, (size_t)R * NQKV * 4, st)); CK(hipMemsetAsync(ss_attn, 0, 128, st));
    time_graph("G1 chain (28 layers)", [&] { for (int l = 0; l < L; ++l) g1_layer(l); }, L);
    time_graph("G1 chain (28 layers)", [&] { for (int l = 0; l < L; ++l) g1_layer(l); }, L);
    time_graph("G2 chain (28 layers)", [&] { for (int l = 0; l < L; ++l) g2_layer(l); }, L);
    time_graph("G2 chain (28 layers)", [&] { for (int l = 0; l < L; ++l) g2_layer(l); }, L);
#ifdef UG_HAVE_OGU
    time_graph("G3 chain (28 layers)", [&] { for (int l = 0; l < L; ++l) g3_layer(l); }, L);
    time_graph("G3 chain (28 layers)", [&] { for (int l = 0; l < L; ++l) g3_layer(l); }, L);
    { uint32_t err = 0; CK(hipMemcpy(&err, g_err, 4, hipMemcpyDeviceToHost)); printf("G3 err word %u\n", err); }
    dump_trace("G3 chain");
#endif
    time_graph("new head", [&] { for (int l = 0; l < L; ++l) UG(ug_decode_sw_head(d_h, nullptr, 0, nullptr, d_lnf, EPS, R, H, d_whead, H, V, d_logits, V, nullptr, nullptr, st)); }, L);
    time_graph("old finish+head gemv", [&] { for (int l = 0; l < L; ++l) { UG(ug_decode_finish_resid_norm(acc_down, H, d_h, d_lnf, d_hn, R, H, EPS, nullptr, nullptr, st));
                                                                         UG(ug_decode_gemv(d_hn, H, R, d_whead, H, acc_head, V, V, H, nullptr, 0, nullptr, 0, nullptr, st)); } }, L);
  }
  if (want("each")) {
    // one kernel type x 28 layers back to back (independent weights; each launch still waits for its predecessor)
    time_graph("new o x28", [&] { for (int l = 0; l < L; ++l) UG(ug_decode_sw_resid(d_o, HQ * HD, R, ly[l].wo, H, H, HQ * HD, d_h, st)); }, L);
    time_graph("new gate_up x28", [&] { for (int l = 0; l < L; ++l) UG(ug_decode_sw_gate_up(d_h, nullptr, 0, nullptr, ly[l].ln2, EPS, R, H, ly[l].wgu, H, I, d_act, I, st)); }, L);
#ifdef UG_HAVE_OGU
    time_graph("fused o+gate_up x28", [&] { CK(hipMemsetAsync(g_flags, 0, 1024, st)); for (int l = 0; l < L; ++l) UG(ug_decode_sw_o_gate_up(d_o, HQ * HD, ly[l].wo, H, d_h, ly[l].ln2, EPS, R, H, ly[l].wgu, H, I, d_act, I, g_flags, (uint32_t)l + 1, g_err, st)); }, L);
    dump_trace("fused x28");
#endif
    time_graph("o then gate_up x28", [&] { for (int l = 0; l < L; ++l) { UG(ug_decode_sw_resid(d_o, HQ * HD, R, ly[l].wo, H, H, HQ * HD, d_h, st)); UG(ug_decode_sw_gate_up(d_h, nullptr, 0, nullptr, ly[l].ln2, EPS, R, H, ly[l].wgu, H, I, d_act, I, st)); } }, L);
    time_graph("down in k-blocks x28", [&] { for (int l = 0; l < L; ++l) UG(ug_decode_sw_kblock(d_act, I, R, ly[l].wdown, I, acc_down, H, H, I, nullptr, 0, nullptr, 0, nullptr, st)); }, L);
    time_graph("old down-as-bf16-gemv x28", [&] { for (int l = 0; l < L; ++l) UG(ug_decode_gemv(d_act, I, R, ly[l].wdown, I, acc_down, H, H, I, nullptr, 0, nullptr, 0, nullptr, st)); }, L);
    time_graph("old qkv x28", [&] { for (int l = 0; l < L; ++l) UG(ug_decode_gemv_resid_norm(d_h, acc_down, H, ly[l].ln1, x_mid, ss_attn, R, ly[l].wqkv, H, acc_qkv, NQKV, NQKV, H, nullptr, 0, nullptr, 0, nullptr, st)); }, L);
    time_graph("old attn x28", [&] { for (int l = 0; l < L; ++l) UG(ug_attn_decode_fused(acc_qkv, NQKV, ss_attn, EPS, H, ly[l].bqkv, d_cs, d_sn, d_pos, ly[l].ck, ly[l].cv, nullptr, d_o, HQ * HD, R, HQ, HK, HD, TMAX, MAXPOS, scale, st)); }, L);
    time_graph("old o x28", [&] { for (int l = 0; l < L; ++l) UG(ug_decode_gemv(d_o, HQ * HD, R, ly[l].wo, H, acc_o, H, H, HQ * HD, nullptr, 0, nullptr, 0, nullptr, st)); }, L);
    time_graph("old gate_up x28", [&] { for (int l = 0; l < L; ++l) UG(ug_decode_gemv_resid_norm(x_mid, acc_o, H, ly[l].ln2, d_h, ss_mlp, R, ly[l].wgu, H, acc_gu, 2 * I, 2 * I, H, nullptr, 0, nullptr, 0, nullptr, st)); }, L);
    time_graph("old down x28", [&] { for (int l = 0; l < L; ++l) UG(ug_decode_gemv_swiglu(acc_gu, 2 * I, ss_mlp, EPS, H, R, ly[l].wdown, I, acc_down, H, H, I, nullptr, 0, nullptr, 0, nullptr, st)); }, L);
  }
  // two runs of the shipped chain from the same input
  {
    std::vector<float> r1((size_t)R * H), r2((size_t)R * H);
    for (int rep = 0; rep < 2; ++rep) {
      CK(hipMemcpyAsync(d_h, d_h0, (size_t)R * H * 4, hipMemcpyDeviceToDevice, st));
      CK(hipMemsetAsync(acc_down, 0, (size_t)R * H * 4, st)); CK(hipMemsetAsync(acc_o, 0, (size_t)R * H * 4, st)); CK(hipMemsetAsync(acc_qkv, 0, (size_t)R * NQKV * 4, st)); CK(hipMemsetAsync(ss_attn, 0, 128, st));
      for (int l = 0; l < L; ++l) g1_layer(l);
      CK(hipStreamSynchronize(st));
      CK(hipMemcpy(rep ? r2.data() : r1.data(), d_h, r1.size() * 4, hipMemcpyDeviceToHost));
    }
    size_t diff = 0; bool finite = true; for (size_t i = 0; i < r1.size(); ++i) { diff += memcmp(&r1[i], &r2[i], 4) != 0; finite = finite && isfinite(r1[i]); }
    printf("shipped chain twice from the same input (its split-K launches sum by fp32 atomics): %zu of %zu values differ, finite=%d\n", diff, r1.size(), (int)finite);
  }
  return 0;
}
