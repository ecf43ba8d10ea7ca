// Train-step glue around the hot path: the gradient-modulation block of the reference's training loop
// as ONE device kernel, so that nothing between loss.backward() and optimizer.step() touches the host.
//
// Replaces train_test.py:87-184.  Task types diag2021 / grade / subtype (softmax scores) and, since round 4, 'survival': the branch scores
// are then the concordance indices of risk = -sum_t S_t, S = cumprod(1 - sigmoid(out)) (:99-102,121-128), which the reference obtains on the
// host from scikit-survival's concordance_index_censored(event = 1 - censor, time, risk, tied_tol = 1e-8) (utils/utils.py:315-317; the
// package is absent from this image: its published pair rule is restated here - PARITY UNPINNED for this branch, oracle/trainstep.py):
//   out_t = feat_t W[:, :hs]^T + b / 2,  out_i = feat_i W[:, hs:]^T + b / 2                         (:90-93)
//   score_x = sum_b softmax(out_x[b])[label[b]]   (Python sum over b in order)                     (:119-120)
//   ratio_t = score_t / score_i, ratio_i = 1 / ratio_t                                            (:150-152)
//   per class row r of classifier.weight.grad [C, 2 hs] with g_t = row[:hs], g_i = row[hs:]:        (:158-183)
//     sim = g_t . g_i / (|g_t| |g_i|); if sim < 0:
//       if ratio_t < 1:   ps = g_t . g_i / |g_i|^2;  a = g_t - ps g_i;  perpen = a - ps g_i;  row[:hs] = |a| perpen / |perpen|
//       elif ratio_i < 1: the same with the roles of g_t and g_i exchanged, written to row[hs:]
// The reference evaluates ~50 scalar expressions with .item()-style host syncs and per-sample Python loops; here one
// workgroup does all of it: wave-level dot products, LDS for the [B, C] logits, one wave per class row.
#include "smml_common.h"

namespace {

constexpr int GM_MAX_C = 16;

// censor / survtime non-null: the survival scores (C-index per branch); label is then unused
__global__ __launch_bounds__(256) void grad_modulate_kernel(const float* __restrict__ ft, const float* __restrict__ fi,
                                                            const float* __restrict__ W, const float* __restrict__ bias,
                                                            const long long* __restrict__ label, const float* __restrict__ censor,
                                                            const float* __restrict__ survtime, float* __restrict__ G,
                                                            float* __restrict__ info, int B, int C, int hs) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* outs = sm;                       // [2][B][C]
  float* pl = sm + 2 * B * C;             // [2][B] probability of the labelled class
  float* ratio = pl + 2 * B;              // [2]
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int ld = 2 * hs;
  // ---- phase 1: the two [B, C] logit blocks, one wave per (side, b, c) item ----
  for (int item = wave; item < 2 * B * C; item += 4) {
    const int side = item / (B * C), rem = item - side * B * C, b = rem / C, c = rem - b * C;
    const float* f = (side ? fi : ft) + (size_t)b * hs;
    const float* w = W + (size_t)c * ld + side * hs;
    float s = 0.f;
    for (int k = lane; k < hs; k += 64) s = fmaf(f[k], w[k], s);
    s = wave_sum(s);
    if (lane == 0) outs[item] = s + 0.5f * bias[c];
  }
  __syncthreads();
  if (!censor) {
    // ---- phase 2: softmax probability of the labelled class per sample ----
    for (int i = tid; i < 2 * B; i += 256) {
      const int side = i / B, b = i - side * B;
      const float* o = outs + (side * B + b) * C;
      float m = o[0];
      for (int c = 1; c < C; ++c) m = fmaxf(m, o[c]);
      float den = 0.f;
      for (int c = 0; c < C; ++c) den += expf(o[c] - m);
      long long lb = label[b];
      lb = lb < 0 ? 0 : (lb >= C ? C - 1 : lb);      // the reference would raise on an out-of-range label; never fault here
      pl[i] = expf(o[lb] - m) / den;
    }
    __syncthreads();
    if (tid == 0) {                                   // Python's sum([...]): left to right, starting from 0
      float st = 0.f, si = 0.f;
      for (int b = 0; b < B; ++b) { st += pl[b]; si += pl[B + b]; }
      const float rt = st / si;
      ratio[0] = rt; ratio[1] = 1.f / rt;
      if (info) { info[0] = st; info[1] = si; info[2] = rt; info[3] = 1.f / rt; }
    }
  } else {
    // ---- phase 2 (survival): risk_b = -sum_t cumprod_t (1 - sigmoid(out_b)) per branch (:99-102,123-124) ----
    for (int i = tid; i < 2 * B; i += 256) {
      const float* o = outs + i * C;
      float S = 1.f, acc = 0.f;
      for (int c = 0; c < C; ++c) { S *= 1.f - 1.f / (1.f + expf(-o[c])); acc += S; }
      pl[i] = -acc;
    }
    __syncthreads();
    // concordance index per branch, scikit-survival's rule: sample i with an event (censor == 0) is comparable with every j whose time is
    // later, and with every j CENSORED at the same time; a comparable pair is concordant if risk_i > risk_j, tied if |risk_i - risk_j| <= 1e-8
    // (counted 1/2), else discordant; c = (concordant + tied / 2) / comparable.  One thread per (branch, i), integer counts, then a serial sum.
    float* cnt = outs;                              // reuse the logits' LDS: [2][B][2] (numerator x 2, comparable)
    __syncthreads();
    for (int i = tid; i < 2 * B; i += 256) {
      const int side = i / B, a = i - side * B;
      int num2 = 0, den = 0;                        // numerator in halves
      if (censor[a] == 0.f) {
        const float ta = survtime[a], ra = pl[side * B + a];
        for (int j = 0; j < B; ++j) {
          if (j == a) continue;
          const float tj = survtime[j];
          if (tj > ta || (tj == ta && censor[j] != 0.f)) {
            const float rj = pl[side * B + j];
            ++den;
            num2 += (fabsf(ra - rj) <= 1e-8f) ? 1 : ((ra > rj) ? 2 : 0);
          }
        }
      }
      cnt[2 * i] = (float)num2; cnt[2 * i + 1] = (float)den;
    }
    __syncthreads();
    if (tid == 0) {
      float n2[2] = {0.f, 0.f}, dn[2] = {0.f, 0.f};
      float cm = 0.f;
      for (int b = 0; b < B; ++b) cm += censor[b];
      for (int sd = 0; sd < 2; ++sd)
        for (int b = 0; b < B; ++b) { n2[sd] += cnt[2 * (sd * B + b)]; dn[sd] += cnt[2 * (sd * B + b) + 1]; }
      // all samples censored (:127-133), or no comparable pair (scikit-survival raises there): no modulation - ratios that take no branch
      const bool ok = (cm != (float)B) && dn[0] > 0.f && dn[1] > 0.f;
      const float ct = ok ? 0.5f * n2[0] / dn[0] : 1.f, ci = ok ? 0.5f * n2[1] / dn[1] : 1.f;
      const float rt = ok ? ct / ci : 1.f;
      ratio[0] = rt; ratio[1] = ok ? 1.f / rt : 1.f;
      // info[0..1]: the two concordance indices, or - no modulation happened - the reason: -2 all samples censored (the reference prints
      // and skips, :127-133), -1 no comparable pair in a branch (scikit-survival raises there: the caller may do the same)
      if (info) {
        const float why = (cm == (float)B) ? -2.f : -1.f;
        info[0] = ok ? ct : why; info[1] = ok ? ci : why; info[2] = ratio[0]; info[3] = ratio[1];
      }
    }
  }
  __syncthreads();
  const float ratio_t = ratio[0], ratio_i = ratio[1];
  // ---- phase 3: one wave per class row ----
  for (int r = wave; r < C; r += 4) {
    float* gt = G + (size_t)r * ld;
    float* gi = gt + hs;
    float dot = 0.f, nt2 = 0.f, ni2 = 0.f;
    for (int k = lane; k < hs; k += 64) { const float a = gt[k], b = gi[k]; dot = fmaf(a, b, dot); nt2 = fmaf(a, a, nt2); ni2 = fmaf(b, b, ni2); }
    dot = wave_sum(dot); nt2 = wave_sum(nt2); ni2 = wave_sum(ni2);
    const float sim = dot / (sqrtf(nt2) * sqrtf(ni2));
    int branch = 0;
    if (sim < 0.f) branch = (ratio_t < 1.f) ? 1 : ((ratio_i < 1.f) ? 2 : 0);     // NaN sim (a zero gradient row) compares false
    if (info && lane == 0) { info[4 + 2 * r] = sim; info[5 + 2 * r] = (float)branch; }
    if (branch == 0) continue;                       // wave-uniform
    float* x = (branch == 1) ? gt : gi;              // the row half that is rewritten
    const float* y = (branch == 1) ? gi : gt;        // the half it is projected against
    const float yn = sqrtf((branch == 1) ? ni2 : nt2);
    const float ps = dot / (yn * yn);                // dot / y.norm() ** 2
    float a2 = 0.f, p2 = 0.f;
    for (int k = lane; k < hs; k += 64) {
      const float pc = ps * y[k];
      const float a = x[k] - pc, p = a - pc;
      a2 = fmaf(a, a, a2); p2 = fmaf(p, p, p2);
    }
    a2 = wave_sum(a2); p2 = wave_sum(p2);
    const float an = sqrtf(a2), pn = sqrtf(p2);
    for (int k = lane; k < hs; k += 64) {
      const float pc = ps * y[k];
      const float p = (x[k] - pc) - pc;
      x[k] = an * (p / pn);
    }
  }
}

}  // namespace

extern "C" {

// info (nullable): [4 + 2 C] floats = score_t, score_i, ratio_t, ratio_i, then (sim, branch taken 0 / 1 / 2) per class row
int smml_grad_modulate_f32(const float* feat_t, const float* feat_i, const float* weight, const float* bias,
                           const long long* label, float* weight_grad, float* info, int B, int C, int hs, void* stream) {
  SMML_REQUIRE(feat_t && feat_i && weight && bias && label && weight_grad, "smml_grad_modulate_f32: null pointer");
  SMML_REQUIRE(B > 0 && B <= 1024 && C > 0 && C <= GM_MAX_C && hs > 0, "smml_grad_modulate_f32: need 0 < B <= 1024, 0 < C <= %d, hs > 0",
               GM_MAX_C);
  const size_t lds = ((size_t)2 * B * C + 2 * B + 2) * sizeof(float);
  SMML_REQUIRE(lds <= 64 * 1024, "smml_grad_modulate_f32: B x C = %d x %d needs %zu bytes of LDS (limit 64 KiB: B (C + 1) <= 8191)", B, C, lds);
  hipLaunchKernelGGL(grad_modulate_kernel, dim3(1), dim3(256), lds, (hipStream_t)stream, feat_t, feat_i, weight, bias, label,
                     (const float*)nullptr, (const float*)nullptr, weight_grad, info, B, C, hs);
  SMML_LAUNCH_CHECK("smml_grad_modulate_f32");
  return SMML_OK;
}

// task_type 'survival' (train_test.py:99-102,121-149): the branch scores are concordance indices of risk = -sum_t S_t against
// (censor, survtime) [B] fp32 each (censor 1 = censored); info[0..1] = cindex_t, cindex_i.  All censored / no comparable pair: no edit and
// info[0] = info[1] = -2 / -1 (the reference skips the first case and raises, through scikit-survival, in the second).
int smml_grad_modulate_survival_f32(const float* feat_t, const float* feat_i, const float* weight, const float* bias,
                                    const float* censor, const float* survtime, float* weight_grad, float* info, int B, int C, int hs,
                                    void* stream) {
  SMML_REQUIRE(feat_t && feat_i && weight && bias && censor && survtime && weight_grad, "smml_grad_modulate_survival_f32: null pointer");
  SMML_REQUIRE(B > 0 && B <= 1024 && C >= 2 && C <= GM_MAX_C && hs > 0, "smml_grad_modulate_survival_f32: need 0 < B <= 1024, 2 <= C <= %d, hs > 0",
               GM_MAX_C);
  const size_t lds = ((size_t)2 * B * C + 2 * B + 2) * sizeof(float);
  SMML_REQUIRE(lds <= 64 * 1024, "smml_grad_modulate_survival_f32: B x C = %d x %d needs %zu bytes of LDS (limit 64 KiB)", B, C, lds);
  hipLaunchKernelGGL(grad_modulate_kernel, dim3(1), dim3(256), lds, (hipStream_t)stream, feat_t, feat_i, weight, bias,
                     (const long long*)nullptr, censor, survtime, weight_grad, info, B, C, hs);
  SMML_LAUNCH_CHECK("smml_grad_modulate_survival_f32");
  return SMML_OK;
}

}  // extern "C"
