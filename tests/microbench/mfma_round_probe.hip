// How does v_mfma_f32_32x32x16_f16 round?  D = C + sum_k a_k b_k with all rows / columns alike; every case prints the result
// as a multiple of u = ulp(1.5) = 2^-23 beside what round-to-nearest-even of the exact sum, truncation toward zero and
// truncation toward -inf of the exact sum would give.  Products: a_k = 2^-12, b_k = m 2^-13  ->  m/4 u each (exact in fp16).
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float floatx16 __attribute__((ext_vector_type(16)));
struct Case { float c; float a[16]; float b[16]; };
__global__ void k(const Case* cs, float* out, int n) {
  const int lane = threadIdx.x, hf = lane >> 5;
  for (int i = 0; i < n; ++i) {
    half8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (_Float16)cs[i].a[8 * hf + j]; b[j] = (_Float16)cs[i].b[8 * hf + j]; }
    floatx16 acc;
    for (int r = 0; r < 16; ++r) acc[r] = cs[i].c;
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
    if (lane == 0) out[i] = acc[0];
  }
}
int main() {
  const double u = ldexp(1.0, -23);
  const int NC = 22;
  Case h[NC] = {};
  const char* name[NC];
  auto small = [&](Case& c, int kidx, double m) { c.a[kidx] = (float)ldexp(1.0, -12); c.b[kidx] = (float)(m * ldexp(1.0, -13)); };  // m/4 u
  int n = 0;
  name[n] = "C=1.5, one product +0.75u"; h[n].c = 1.5f; small(h[n], 0, 3); ++n;
  name[n] = "C=1.5, one product -0.25u"; h[n].c = 1.5f; small(h[n], 0, -1); ++n;
  name[n] = "C=1.5, one product -0.75u"; h[n].c = 1.5f; small(h[n], 0, -3); ++n;
  name[n] = "C=-1.5, one product +0.25u"; h[n].c = -1.5f; small(h[n], 0, 1); ++n;
  name[n] = "C=-1.5, one product +0.75u"; h[n].c = -1.5f; small(h[n], 0, 3); ++n;
  name[n] = "C=1.5, 16 products +0.25u (sum 4u)"; h[n].c = 1.5f; for (int j = 0; j < 16; ++j) small(h[n], j, 1); ++n;
  name[n] = "C=1.5, 16 products -0.25u (sum -4u)"; h[n].c = 1.5f; for (int j = 0; j < 16; ++j) small(h[n], j, -1); ++n;
  name[n] = "C=-1.5, 16 products +0.25u (sum 4u)"; h[n].c = -1.5f; for (int j = 0; j < 16; ++j) small(h[n], j, 1); ++n;
  name[n] = "C=1.5, 16 products +0.03125u (sum 0.5u)"; h[n].c = 1.5f; for (int j = 0; j < 16; ++j) small(h[n], j, 0.125); ++n;
  name[n] = "C=1.5, 16 products +0.046875u (sum 0.75u)"; h[n].c = 1.5f; for (int j = 0; j < 16; ++j) small(h[n], j, 0.1875); ++n;
  name[n] = "C=0, product 1.5 + 15 products +0.25u (sum 3.75u)"; h[n].c = 0.f; h[n].a[0] = 1.5f; h[n].b[0] = 1.f; for (int j = 1; j < 16; ++j) small(h[n], j, 1); ++n;
  name[n] = "C=0, product -1.5 + 15 products +0.25u"; h[n].c = 0.f; h[n].a[0] = -1.5f; h[n].b[0] = 1.f; for (int j = 1; j < 16; ++j) small(h[n], j, 1); ++n;
  name[n] = "C=1.5, products +1.5 -3 (cancel) + one +0.25u"; h[n].c = 1.5f; h[n].a[0] = 1.5f; h[n].b[0] = 1.f; h[n].a[1] = -3.f; h[n].b[1] = 1.f; small(h[n], 2, 1); ++n;
  name[n] = "C=2^-30, 16 products +0.25u: nothing large"; h[n].c = (float)ldexp(1.0, -30); for (int j = 0; j < 16; ++j) small(h[n], j, 1); ++n;
  name[n] = "C=0, product +1.5 + 15 products -0.25u (sum -3.75u)"; h[n].c = 0.f; h[n].a[0] = 1.5f; h[n].b[0] = 1.f; for (int j = 1; j < 16; ++j) small(h[n], j, -1); ++n;
  name[n] = "C=0, product -1.5 + 15 products -0.25u"; h[n].c = 0.f; h[n].a[0] = -1.5f; h[n].b[0] = 1.f; for (int j = 1; j < 16; ++j) small(h[n], j, -1); ++n;
  name[n] = "C=0, product +1.5 + 7 products +0.75u in its block"; h[n].c = 0.f; h[n].a[0] = 1.5f; h[n].b[0] = 1.f; for (int j = 1; j < 8; ++j) small(h[n], j, 3); ++n;
  name[n] = "C=0, product +1.5 + 7 products -0.75u in its block"; h[n].c = 0.f; h[n].a[0] = 1.5f; h[n].b[0] = 1.f; for (int j = 1; j < 8; ++j) small(h[n], j, -3); ++n;
  name[n] = "C=0, product +1.5 + 7 products +1.25u in its block"; h[n].c = 0.f; h[n].a[0] = 1.5f; h[n].b[0] = 1.f; for (int j = 1; j < 8; ++j) small(h[n], j, 5); ++n;
  name[n] = "C=0, product +1.5 + 7 products -1.25u in its block"; h[n].c = 0.f; h[n].a[0] = 1.5f; h[n].b[0] = 1.f; for (int j = 1; j < 8; ++j) small(h[n], j, -5); ++n;
  name[n] = "C=0, +1.5 in block 0, 8 products +0.25u in block 1"; h[n].c = 0.f; h[n].a[0] = 1.5f; h[n].b[0] = 1.f; for (int j = 8; j < 16; ++j) small(h[n], j, 1); ++n;
  name[n] = "C=1.5, 8 products +0.046875u block 0 only (0.375u)"; h[n].c = 1.5f; for (int j = 0; j < 8; ++j) small(h[n], j, 0.1875); ++n;
  Case* dc; float* dout;
  hipMalloc(&dc, sizeof(h)); hipMalloc(&dout, NC * 4);
  hipMemcpy(dc, h, sizeof(h), hipMemcpyHostToDevice);
  k<<<1, 64>>>(dc, dout, n);
  float r[NC]; hipMemcpy(r, dout, NC * 4, hipMemcpyDeviceToHost);
  for (int i = 0; i < n; ++i) {
    double ex = h[i].c;
    for (int j = 0; j < 16; ++j) ex += (double)(float)(_Float16)h[i].a[j] * (double)(float)(_Float16)h[i].b[j];
    const float rn = (float)ex;
    const double q = ldexp(1.0, ilogb(ex == 0 ? 1.0 : ex) - 23);
    const double rz = trunc(ex / q) * q, rd = floor(ex / q) * q;
    const double base = (fabs(h[i].c) > 1 ? h[i].c : (fabs(ex) > 1 ? (ex > 0 ? 1.5 : -1.5) : 0.0));
    printf("%-52s got %+8.4f u | exact %+8.4f  RN %+8.4f  RZ %+8.4f  floor %+8.4f   (offsets from %.1f in u = 2^-23)\n", name[i],
           (r[i] - base) / u, (ex - base) / u, (rn - base) / u, (rz - base) / u, (rd - base) / u, base);
  }
  return 0;
}
