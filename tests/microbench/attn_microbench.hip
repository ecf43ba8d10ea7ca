// Stand-alone timing harness for the fused attention kernels (no Python / torch start-up): compiles
// csrc/deform_attn.hip with the tuning macros given on the command line and times forward and backward
// at the bench shape.  Build + run: tests/microbench/run.sh   (on the GPU box)
#include "../../subspace-multimodal-learning_amd/csrc/deform_attn.hip"
#include "../../subspace-multimodal-learning_amd/csrc/capi.hip"

#include <stdlib.h>
#include <vector>

#ifndef VARIANT
#define VARIANT "default"
#endif

static float* dev_rand(size_t n, float lo, float hi, unsigned seed) {
  std::vector<float> h(n);
  unsigned s = seed * 2654435761u + 12345u;
  for (size_t i = 0; i < n; ++i) {
    s = s * 1664525u + 1013904223u;
    h[i] = lo + (hi - lo) * ((s >> 8) * (1.0f / 16777216.0f));
  }
  float* d = nullptr;
  hipMalloc(&d, n * sizeof(float));
  hipMemcpy(d, h.data(), n * sizeof(float), hipMemcpyHostToDevice);
  return d;
}
static float* dev_zero(size_t n) {
  float* d = nullptr;
  hipMalloc(&d, n * sizeof(float));
  hipMemset(d, 0, n * sizeof(float));
  return d;
}
static double checksum(const float* d, size_t n) {
  std::vector<float> h(n);
  hipMemcpy(h.data(), d, n * sizeof(float), hipMemcpyDeviceToHost);
  double s = 0;
  for (size_t i = 0; i < n; ++i) s += (double)h[i] * (double)((i % 97) + 1);
  return s;
}

int main(int argc, char** argv) {
  const int B = argc > 1 ? atoi(argv[1]) : 4, S = argc > 2 ? atoi(argv[2]) : 100, H = 8, G = 8, PD = 2;
  const float drop_p = argc > 3 ? (float)atof(argv[3]) : 0.f;
  const int N = S * S, t = (S + 2 - 6) / 4 + 1, J = t * t, HD = H * 64;
  const int nst = smml_deform_attn_nst(N);
  float* q = dev_rand((size_t)B * N * HD, -0.5f, 0.5f, 1);
  float* k = dev_rand((size_t)B * J * HD, -0.5f, 0.5f, 2);
  float* v = dev_rand((size_t)B * J * HD, -1.f, 1.f, 3);
  float* vs = dev_rand((size_t)B * G * J * PD, -1.1f, 1.1f, 4);
  float* gq = dev_rand((size_t)N * PD, -1.f, 1.f, 5);
  float* w1 = dev_rand(32 * PD, -0.7f, 0.7f, 6); float* b1 = dev_rand(32, -0.1f, 0.1f, 7);
  float* w2 = dev_rand(32 * 32, -0.2f, 0.2f, 8); float* b2 = dev_rand(32, -0.1f, 0.1f, 9);
  float* w3 = dev_rand(32, -0.2f, 0.2f, 10); float* b3 = dev_rand(1, -0.1f, 0.1f, 11);
  float* dout = dev_rand((size_t)B * N * HD, -1.f, 1.f, 12);
  float* out = dev_zero((size_t)B * N * HD); float* lse = dev_zero((size_t)B * H * N);
  float* lt = dev_zero((size_t)B * H * J * nst); float* dlt = dev_zero((size_t)B * H * J * nst);
  unsigned short* mk = (unsigned short*)dev_zero((size_t)B * H * J * nst);   // [B, H, J, 2, nst] uint16
  float* dq = dev_zero((size_t)B * N * HD); float* dk = dev_zero((size_t)B * J * HD); float* dv = dev_zero((size_t)B * J * HD);
  float* dvs = dev_zero((size_t)B * G * J * PD);
  float* dw1 = dev_zero(64); float* db1 = dev_zero(32); float* dw2 = dev_zero(1024); float* db2 = dev_zero(32);
  float* dw3 = dev_zero(32); float* db3 = dev_zero(4);
  const size_t wsb = smml_deform_attn_bwd_workspace_bytes(B, N, J, H);
  float* ws = dev_zero(wsb / 4 + 4);
  hipEvent_t e0, e1, c0, c1;
  hipEventCreate(&e0); hipEventCreate(&e1); hipEventCreate(&c0); hipEventCreate(&c1);
  const float scale = 0.125f;
  const double pairs = (double)B * H * N * J;
  float fwd_ms = 0, bwd_ms = 0, cpb_ms = 0;
  const int reps = 3;
  for (int it = 0; it < reps + 1; ++it) {
    hipEventRecord(e0, 0);
    int rc = smml_deform_attn_fwd_f32(q, k, v, vs, gq, w1, b1, w2, b2, w3, b3, out, lse, lt, mk, B, N, J, H, G, PD, scale,
                                      drop_p, 77ull, nullptr, nullptr, nullptr);
    hipEventRecord(e1, 0);
    if (rc) { printf("fwd error: %s\n", smml_last_error()); return 1; }
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (it > 0) fwd_ms += ms;
    hipEventRecord(e0, 0);
    rc = smml_deform_attn_bwd_f32(q, k, v, vs, gq, w1, b1, w2, b2, w3, b3, out, dout, lse, lt, mk, dlt, dq, dk, dv, dvs, dw1,
                                  db1, dw2, db2, dw3, db3, ws, wsb, B, N, J, H, G, PD, scale, drop_p, 77ull, c0, c1, nullptr);
    hipEventRecord(e1, 0);
    if (rc) { printf("bwd error: %s\n", smml_last_error()); return 1; }
    hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
    float cms; hipEventElapsedTime(&cms, c0, c1);
    if (it > 0) { bwd_ms += ms; cpb_ms += cms; }
  }
  fwd_ms /= reps; bwd_ms /= reps; cpb_ms /= reps;
  printf("%-28s B=%d N=%d J=%d p=%.2f | fwd %7.3f ms %6.1f TF | cpb_bwd %7.3f ms %6.1f TF | bwd total %7.3f ms | chk out %.6e dw2 %.6e dvs %.6e dq %.6e\n",
         VARIANT, B, N, J, drop_p, fwd_ms, pairs * 2496 / (fwd_ms * 1e-3) / 1e12, cpb_ms, pairs * 4480 / (cpb_ms * 1e-3) / 1e12, bwd_ms,
         checksum(out, (size_t)B * N * HD), checksum(dw2, 1024), checksum(dvs, (size_t)B * G * J * PD),
         checksum(dq, (size_t)B * N * HD));
  return 0;
}
