// Issue cost of the vector instructions the position-bias kernels are made of, on gfx950, with ONE and with TWO resident
// waves per SIMD: ns per instruction per SIMD for 8 independent chains.  Answers "which replacements are worth it":
// packed-fp32, conversions, v_fma_mix, v_dot2c (bf16), compare / select, transcendental.
// Build: hipcc --offload-arch=gfx950 -O3 valu_mix_probe.hip -o bin/valu_mix_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float float2v __attribute__((ext_vector_type(2)));

// eight independent chains in ONE asm statement (between separate asm statements the compiler inserts an s_nop)
#define I8(A) A(0) A(1) A(2) A(3) A(4) A(5) A(6) A(7)
#define PROBE(NAME, TYPE, ASM)                                                                           \
  __global__ __launch_bounds__(64) void NAME(float* out, int iters) {                                    \
    TYPE x[8];                                                                                           \
    for (int i = 0; i < 8; ++i) x[i] = (TYPE)((float)threadIdx.x * 0.001f + i + 1.f);                    \
    TYPE a = (TYPE)1.0001f, b = (TYPE)0.001f;                                                            \
    asm volatile("" : "+v"(a), "+v"(b));                                                                 \
    for (int it = 0; it < iters; ++it) {                                                                 \
      _Pragma("unroll") for (int u = 0; u < 8; ++u)                                                      \
        asm volatile(I8(ASM) : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]),   \
                     "+v"(x[6]), "+v"(x[7]) : "v"(a), "v"(b) : "vcc", "s40", "s41");                     \
    }                                                                                                    \
    TYPE s = x[0];                                                                                       \
    for (int i = 1; i < 8; ++i) s += x[i];                                                               \
    out[blockIdx.x * 64 + threadIdx.x] = *(float*)&s;                                                    \
  }
#define S(x) #x
#define A_k_fma(n) "v_fma_f32 %" S(n) ", %" S(n) ", %8, %9\n\t"
PROBE(k_fma, float, A_k_fma)
#define A_k_max(n) "v_max_f32 %" S(n) ", 0, %" S(n) "\n\t"
PROBE(k_max, float, A_k_max)
#define A_k_and(n) "v_and_b32 %" S(n) ", 0xffff0000, %" S(n) "\n\t"
PROBE(k_and, float, A_k_and)
#define A_k_lshl(n) "v_lshlrev_b32 %" S(n) ", 16, %" S(n) "\n\t"
PROBE(k_lshl, float, A_k_lshl)
#define A_k_bfi(n) "v_bfi_b32 %" S(n) ", %8, %" S(n) ", %9\n\t"
PROBE(k_bfi, float, A_k_bfi)
#define A_k_perm(n) "v_perm_b32 %" S(n) ", %" S(n) ", %8, %9\n\t"
PROBE(k_perm, float, A_k_perm)
#define A_k_pkfma(n) "v_pk_fma_f32 %" S(n) ", %" S(n) ", %8, %9\n\t"
PROBE(k_pkfma, float2v, A_k_pkfma)
#define A_k_pkmul(n) "v_pk_mul_f32 %" S(n) ", %" S(n) ", %8\n\t"
PROBE(k_pkmul, float2v, A_k_pkmul)
#define A_k_pkadd(n) "v_pk_add_f32 %" S(n) ", %" S(n) ", %8\n\t"
PROBE(k_pkadd, float2v, A_k_pkadd)
#define A_k_cvtbf(n) "v_cvt_pk_bf16_f32 %" S(n) ", %" S(n) ", %8\n\t"
PROBE(k_cvtbf, float, A_k_cvtbf)
#define A_k_cvtf16(n) "v_cvt_pk_f16_f32 %" S(n) ", %" S(n) ", %8\n\t"
PROBE(k_cvtf16, float, A_k_cvtf16)
#define A_k_fmamix(n) "v_fma_mix_f32 %" S(n) ", %8, -1.0, %" S(n) " op_sel_hi:[1,0,0]\n\t"
PROBE(k_fmamix, float, A_k_fmamix)
#define A_k_dot2c(n) "v_dot2c_f32_bf16 %" S(n) ", %8, %9\n\t"
PROBE(k_dot2c, float, A_k_dot2c)
#define A_k_cmp(n) "v_cmp_gt_f32 vcc, %" S(n) ", %8\n\t"
PROBE(k_cmp, float, A_k_cmp)
#define A_k_cmp64(n) "v_cmp_gt_f32 s[40:41], %" S(n) ", %8\n\t"
PROBE(k_cmp64, float, A_k_cmp64)
#define A_k_cnd(n) "v_cndmask_b32 %" S(n) ", %" S(n) ", %8, vcc\n\t"
PROBE(k_cnd, float, A_k_cnd)
#define A_k_cnd64(n) "v_cndmask_b32 %" S(n) ", %" S(n) ", %8, s[40:41]\n\t"
PROBE(k_cnd64, float, A_k_cnd64)
#define A_k_fmaclamp(n) "v_fma_f32 %" S(n) ", %" S(n) ", %8, %9 clamp\n\t"
PROBE(k_fmaclamp, float, A_k_fmaclamp)
#define A_k_log(n) "v_log_f32 %" S(n) ", %" S(n) "\n\t"
PROBE(k_log, float, A_k_log)
#define A_k_rcp(n) "v_rcp_f32 %" S(n) ", %" S(n) "\n\t"
PROBE(k_rcp, float, A_k_rcp)
#define A_k_swap(n) "v_permlane32_swap_b32 %" S(n) ", %" S(n) "\n\t"
PROBE(k_swap, float, A_k_swap)
#define A_k_mov(n) "v_mov_b32 %" S(n) ", %8\n\t"
PROBE(k_mov, float, A_k_mov)
#define A_k_nop(n) "s_nop 0\n\t"
PROBE(k_nop, float, A_k_nop)
#define A_k_nop3(n) "s_nop 3\n\t"
PROBE(k_nop3, float, A_k_nop3)
#define A_k_nop11(n) "s_nop 11\n\t"
PROBE(k_nop11, float, A_k_nop11)
#define A_k_fmanop(n) "v_fma_f32 %" S(n) ", %" S(n) ", %8, %9\n\ts_nop 0\n\t"
PROBE(k_fmanop, float, A_k_fmanop)
#define A_k_pkfmanop(n) "v_pk_fma_f32 %" S(n) ", %" S(n) ", %8, %9\n\ts_nop 0\n\t"
PROBE(k_pkfmanop, float2v, A_k_pkfmanop)
#define A_k_fmasalu(n) "v_fma_f32 %" S(n) ", %" S(n) ", %8, %9\n\ts_add_u32 s40, s40, 1\n\t"
PROBE(k_fmasalu, float, A_k_fmasalu)

#define A_k_maxi(n) "v_max_i32 %" S(n) ", 0, %" S(n) "\n\t"
PROBE(k_maxi, float, A_k_maxi)
#define A_k_addf(n) "v_add_f32 %" S(n) ", %" S(n) ", %8\n\t"
PROBE(k_addf, float, A_k_addf)
#define A_k_mulf(n) "v_mul_f32 %" S(n) ", %" S(n) ", %8\n\t"
PROBE(k_mulf, float, A_k_mulf)
#define A_k_mulclamp(n) "v_mul_f32_e64 %" S(n) ", %" S(n) ", %8 clamp\n\t"
PROBE(k_mulclamp, float, A_k_mulclamp)
#define A_k_fmaabs(n) "v_fma_f32 %" S(n) ", |%" S(n) "|, %8, %9\n\t"
PROBE(k_fmaabs, float, A_k_fmaabs)
#define A_k_or(n) "v_or_b32 %" S(n) ", %" S(n) ", %8\n\t"
PROBE(k_or, float, A_k_or)
#define A_k_andr(n) "v_and_b32 %" S(n) ", %" S(n) ", %8\n\t"
PROBE(k_andr, float, A_k_andr)
#define A_k_ashr(n) "v_ashrrev_i32 %" S(n) ", 31, %" S(n) "\n\t"
PROBE(k_ashr, float, A_k_ashr)
#define A_k_addu(n) "v_add_u32 %" S(n) ", %" S(n) ", %8\n\t"
PROBE(k_addu, float, A_k_addu)
#define A_k_med3(n) "v_med3_f32 %" S(n) ", %" S(n) ", %8, %9\n\t"
PROBE(k_med3, float, A_k_med3)
#define A_k_pkmaxh(n) "v_pk_max_f16 %" S(n) ", %" S(n) ", %8\n\t"
PROBE(k_pkmaxh, float, A_k_pkmaxh)
#define A_k_maxe64(n) "v_max_f32_e64 %" S(n) ", %" S(n) ", %8\n\t"
PROBE(k_maxe64, float, A_k_maxe64)
#define A_k_maxfma(n) "v_max_f32 %" S(n) ", 0, %" S(n) "\n\tv_fma_f32 %" S(n) ", %" S(n) ", %8, %9\n\t"
PROBE(k_maxfma, float, A_k_maxfma)
#define A_k_cvtfma(n) "v_cvt_pk_bf16_f32 %" S(n) ", %" S(n) ", %8\n\tv_fma_f32 %" S(n) ", %" S(n) ", %8, %9\n\t"
PROBE(k_cvtfma, float, A_k_cvtfma)

template <typename K>
static void run(const char* name, K kern) {
  float* out; hipMalloc(&out, 4096 * 64 * 4);
  const int iters = 4000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  double ns[3];
  for (int w = 0; w < 3; ++w) {
    const int waves = 1024 << w;                      // 1, 2 or 4 waves per SIMD (256 CUs x 4 SIMDs)
    kern<<<waves, 64>>>(out, 10);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    kern<<<waves, 64>>>(out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    ns[w] = ms * 1e6 / ((double)iters * 64 * (1 << w));   // per instruction per SIMD
  }
  printf("%-36s ns / instr / SIMD at 1, 2, 4 waves per SIMD: %6.2f %6.2f %6.2f\n", name, ns[0], ns[1], ns[2]);
  hipFree(out);
}
int main() {
  run("v_fma_f32", k_fma); run("v_max_f32", k_max); run("v_and_b32 (literal)", k_and); run("v_lshlrev_b32", k_lshl);
  run("v_bfi_b32", k_bfi); run("v_perm_b32", k_perm); run("v_mov_b32", k_mov);
  run("v_pk_fma_f32", k_pkfma); run("v_pk_mul_f32", k_pkmul); run("v_pk_add_f32", k_pkadd);
  run("v_cvt_pk_bf16_f32", k_cvtbf); run("v_cvt_pk_f16_f32", k_cvtf16); run("v_fma_mix_f32", k_fmamix);
  run("v_dot2c_f32_bf16", k_dot2c); run("v_cmp_gt_f32 -> vcc", k_cmp); run("v_cmp_gt_f32 -> sgpr pair", k_cmp64);
  run("v_cndmask_b32 (vcc)", k_cnd); run("v_cndmask_b32 (sgpr pair)", k_cnd64); run("v_fma_f32 clamp", k_fmaclamp);
  run("v_log_f32", k_log); run("v_rcp_f32", k_rcp); run("v_permlane32_swap_b32", k_swap); run("s_nop 0", k_nop); run("s_nop 3", k_nop3); run("s_nop 11", k_nop11);
  run("v_fma_f32 + s_nop 0 (per pair)", k_fmanop); run("v_pk_fma_f32 + s_nop 0 (per pair)", k_pkfmanop);
  run("v_max_i32 (relu on the bits)", k_maxi);
  run("v_add_f32", k_addf);
  run("v_mul_f32", k_mulf);
  run("v_mul_f32 clamp", k_mulclamp);
  run("v_fma_f32 |src0|", k_fmaabs);
  run("v_or_b32", k_or);
  run("v_and_b32 (register)", k_andr);
  run("v_ashrrev_i32", k_ashr);
  run("v_add_u32", k_addu);
  run("v_med3_f32", k_med3);
  run("v_pk_max_f16", k_pkmaxh);
  run("v_max_f32 (two registers)", k_maxe64);
  run("v_max_f32 + v_fma_f32 (per pair)", k_maxfma);
  run("v_cvt_pk_bf16 + v_fma_f32 (per pair)", k_cvtfma);
  return 0;
}
