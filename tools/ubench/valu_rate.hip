// Throughput of single VALU instructions on gfx950 at 1 / 2 / 4 waves per SIMD: cycles of the SIMD per
// wave-instruction (independent streams, 8 destination registers).  Build: hipcc --offload-arch=gfx950 -O2
// -o valu_rate valu_rate.hip ; run on an MI355X.  Diagnostic only (not part of the library).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <vector>

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

// every test body issues 64 instructions: 8 x REP8
#define KERNEL(NAME, DECL, ONE, SINK)                                                              \
    __global__ void __launch_bounds__(1024) NAME(long long* out, int iters, int seed) {            \
        DECL;                                                                                      \
        long long t0 = __builtin_amdgcn_s_memtime();                                               \
        for (int it = 0; it < iters; ++it) {                                                       \
            REP8(ONE) REP8(ONE) REP8(ONE) REP8(ONE) REP8(ONE) REP8(ONE) REP8(ONE) REP8(ONE)        \
        }                                                                                          \
        long long t1 = __builtin_amdgcn_s_memtime();                                               \
        SINK;                                                                                      \
        if ((threadIdx.x & 63) == 0) out[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = t1 - t0; \
    }

#define DECL_U32                                                                       \
    unsigned a[8], b = threadIdx.x * 2654435761u + seed, c = seed | 1;                 \
    unsigned long long m = 0x5555555555555555ull * seed; unsigned sg[8] = {0,0,0,0,0,0,0,0};      \
    asm volatile("v_cmp_lt_u32 vcc, 77, %0" : : "v"(b) : "vcc");                           \
    for (int i = 0; i < 8; ++i) a[i] = b + i
#define SINK_U32                                                   \
    unsigned s = 0;                                                \
    for (int i = 0; i < 8; ++i) s ^= a[i];                         \
    for (int i = 0; i < 8; ++i) s ^= sg[i];                        \
    if (s == 0x12345678u && seed == -1) out[0] = s + m
#define DECL_F64                                                                   \
    double a[8], b = 1.0 + 1e-9 * threadIdx.x + seed * 1e-12, c = 0.999999 + seed * 1e-13; \
    double sc = 0.5 + seed; unsigned long long sm[8] = {0,0,0,0,0,0,0,0};                  \
    for (int i = 0; i < 8; ++i) a[i] = b + i
#define SINK_F64                                                   \
    double s = 0;                                                  \
    for (int i = 0; i < 8; ++i) s += a[i];                         \
    for (int i = 0; i < 8; ++i) s += (double)sm[i];                \
    if (s == 0.12345 && seed == -1) out[0] = (long long)(s + sc)
#define DECL_F32                                                                   \
    float a[8], b = 1.0f + 1e-4f * threadIdx.x + seed * 1e-6f, c = 0.9999f + seed * 1e-7f; \
    for (int i = 0; i < 8; ++i) a[i] = b + i
#define SINK_F32                                                   \
    float s = 0;                                                   \
    for (int i = 0; i < 8; ++i) s += a[i];                         \
    if (s == 0.12345f && seed == -1) out[0] = (long long)s
#define DECL_MIX                                                   \
    DECL_F64;                                                      \
    unsigned u[8];                                                 \
    float f[8];                                                    \
    unsigned long long q[8];                                       \
    for (int i = 0; i < 8; ++i) { u[i] = threadIdx.x + i + seed; f[i] = (float)a[i]; q[i] = u[i]; }
#define SINK_MIX                                                   \
    double s = 0;                                                  \
    for (int i = 0; i < 8; ++i) s += a[i] + u[i] + f[i] + (double)q[i]; \
    if (s == 0.12345 && seed == -1) out[0] = (long long)s

#define I_XOR(i) asm volatile("v_xor_b32 %0, %1, %0" : "+v"(a[i]) : "v"(b));
#define I_MOV(i) asm volatile("v_mov_b32 %0, %1" : "+v"(a[i]) : "v"(b));
#define I_CNDMASK(i) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(b));
#define I_CNDMASK64(i) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "s"(m));
#define I_CNDMASK_E64VCC(i) asm volatile("v_cndmask_b32_e64 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(b));
#define I_CMP_CND_VCC(i) asm volatile("v_cmp_lt_u32 vcc, %1, %2\n v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(b), "v"(c) : "vcc");
#define I_CMP_CND_SGPR(i) asm volatile("v_cmp_lt_u32_e64 %1, %2, %3\n v_cndmask_b32_e64 %0, %0, %2, %1" : "+v"(a[i]), "=&s"(m) : "v"(b), "v"(c));
#define I_READLANE(i) asm volatile("v_readlane_b32 %0, %1, 3" : "=s"(sg[i]) : "v"(a[i]));
#define I_XOR_CHAIN(i) asm volatile("v_xor_b32 %0, %1, %0" : "+v"(a[0]) : "v"(b));
#define I_ADD3_CHAIN(i) asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(a[0]) : "v"(b), "v"(c));
#define I_DPP_CHAIN(i) asm volatile("v_mov_b32_dpp %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a[0]));
#define I_BPERM(i) asm volatile("ds_bpermute_b32 %0, %1, %0\n s_waitcnt lgkmcnt(0)" : "+v"(a[i]) : "v"(b));
#define I_BPERM_NOWAIT(i) asm volatile("ds_bpermute_b32 %0, %1, %0" : "+v"(a[i]) : "v"(b));

#define I_DPP(i) asm volatile("v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a[i]) : "v"(b));
#define I_ADD3(i) asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
#define I_LSHLADD(i) asm volatile("v_lshl_add_u32 %0, %0, 1, %1" : "+v"(a[i]) : "v"(b));
#define I_MULLO(i) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
#define I_MULHI(i) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
#define I_MUL24(i) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(a[i]) : "v"(c));
#define I_MAD24(i) asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(a[i]) : "v"(c), "v"(b));
#define I_PKMIN(i) asm volatile("v_pk_min_u16 %0, %0, %1" : "+v"(a[i]) : "v"(b));
#define I_PERM(i) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
#define I_ALIGNBIT(i) asm volatile("v_alignbit_b32 %0, %0, %0, 13" : "+v"(a[i]));
#define I_BFE(i) asm volatile("v_bfe_u32 %0, %0, 3, 9" : "+v"(a[i]));

#define I_FMA64(i) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a[i]) : "v"(c), "v"(b));
#define I_ADD64(i) asm volatile("v_add_f64 %0, %0, %1" : "+v"(a[i]) : "v"(b));
#define I_MUL64(i) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a[i]) : "v"(c));
#define I_MAX64(i) asm volatile("v_max_f64 %0, %0, %1" : "+v"(a[i]) : "v"(b));
#define I_RCP64(i) asm volatile("v_rcp_f64 %0, %0" : "+v"(a[i]));
#define I_FLOOR64(i) asm volatile("v_floor_f64 %0, %0" : "+v"(a[i]));
#define I_LDEXP64(i) asm volatile("v_ldexp_f64 %0, %0, 1" : "+v"(a[i]));
#define I_CMP64(i) asm volatile("v_cmp_lt_f64 vcc, %0, %1" : : "v"(a[i]), "v"(b) : "vcc");
#define I_FMA64_CHAIN(i) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a[0]) : "v"(c), "v"(b));
#define I_ADD64_CHAIN(i) asm volatile("v_add_f64 %0, %0, %1" : "+v"(a[0]) : "v"(b));
#define I_RCP64_CHAIN(i) asm volatile("v_rcp_f64 %0, %0" : "+v"(a[0]));
#define I_CMP64_S(i) asm volatile("v_cmp_lt_f64_e64 %0, %1, %2" : "=s"(sm[i]) : "v"(a[i]), "v"(b));
#define I_FMA64_S(i) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a[i]) : "s"(sc), "v"(b));
#define I_MOV64(i) asm volatile("v_mov_b64 %0, %1" : "+v"(a[i]) : "v"(b));
#define I_CVT_F64_F32(i) asm volatile("v_cvt_f64_f32 %0, %1" : "+v"(a[i]) : "v"(f[i]));
#define I_CVT_F32_F64(i) asm volatile("v_cvt_f32_f64 %0, %1" : "+v"(f[i]) : "v"(a[i]));
#define I_CVT_F64_I32(i) asm volatile("v_cvt_f64_i32 %0, %1" : "+v"(a[i]) : "v"(u[i]));
#define I_CVT_I32_F64(i) asm volatile("v_cvt_i32_f64 %0, %1" : "+v"(u[i]) : "v"(a[i]));
#define I_MAD64(i) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, 0" : "+v"(q[i]) : "v"(u[i]), "v"(u[(i + 1) & 7]) : "vcc");
#define I_LSHLADD64(i) asm volatile("v_lshl_add_u64 %0, %0, 1, %1" : "+v"(q[i]) : "v"(q[(i + 1) & 7]));

#define I_FMA32(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(c), "v"(b));
#define I_LOG32(i) asm volatile("v_log_f32 %0, %0" : "+v"(a[i]));
#define I_SQRT32(i) asm volatile("v_sqrt_f32 %0, %0" : "+v"(a[i]));
#define I_SIN32(i) asm volatile("v_sin_f32 %0, %0" : "+v"(a[i]));
#define I_CVTU32(i) asm volatile("v_cvt_f32_u32 %0, %0" : "+v"(a[i]));

KERNEL(k_xor, DECL_U32, I_XOR, SINK_U32)
KERNEL(k_mov, DECL_U32, I_MOV, SINK_U32)
KERNEL(k_cndmask, DECL_U32, I_CNDMASK, SINK_U32)
KERNEL(k_dpp, DECL_U32, I_DPP, SINK_U32)
KERNEL(k_cndmask64, DECL_U32, I_CNDMASK64, SINK_U32)
KERNEL(k_cndmask_e64vcc, DECL_U32, I_CNDMASK_E64VCC, SINK_U32)
KERNEL(k_cmp_cnd_vcc, DECL_U32, I_CMP_CND_VCC, SINK_U32)
KERNEL(k_cmp_cnd_sgpr, DECL_U32, I_CMP_CND_SGPR, SINK_U32)
KERNEL(k_readlane, DECL_U32, I_READLANE, SINK_U32)
KERNEL(k_xor_chain, DECL_U32, I_XOR_CHAIN, SINK_U32)
KERNEL(k_add3_chain, DECL_U32, I_ADD3_CHAIN, SINK_U32)
KERNEL(k_dpp_chain, DECL_U32, I_DPP_CHAIN, SINK_U32)
KERNEL(k_bperm, DECL_U32, I_BPERM, SINK_U32)
KERNEL(k_bperm_nowait, DECL_U32, I_BPERM_NOWAIT, SINK_U32)
KERNEL(k_add3, DECL_U32, I_ADD3, SINK_U32)
KERNEL(k_lshladd, DECL_U32, I_LSHLADD, SINK_U32)
KERNEL(k_mullo, DECL_U32, I_MULLO, SINK_U32)
KERNEL(k_mulhi, DECL_U32, I_MULHI, SINK_U32)
KERNEL(k_mul24, DECL_U32, I_MUL24, SINK_U32)
KERNEL(k_mad24, DECL_U32, I_MAD24, SINK_U32)
KERNEL(k_pkmin, DECL_U32, I_PKMIN, SINK_U32)
KERNEL(k_perm, DECL_U32, I_PERM, SINK_U32)
KERNEL(k_alignbit, DECL_U32, I_ALIGNBIT, SINK_U32)
KERNEL(k_bfe, DECL_U32, I_BFE, SINK_U32)
KERNEL(k_fma64, DECL_F64, I_FMA64, SINK_F64)
KERNEL(k_add64, DECL_F64, I_ADD64, SINK_F64)
KERNEL(k_mul64, DECL_F64, I_MUL64, SINK_F64)
KERNEL(k_max64, DECL_F64, I_MAX64, SINK_F64)
KERNEL(k_rcp64, DECL_F64, I_RCP64, SINK_F64)
KERNEL(k_floor64, DECL_F64, I_FLOOR64, SINK_F64)
KERNEL(k_ldexp64, DECL_F64, I_LDEXP64, SINK_F64)
KERNEL(k_cmp64, DECL_F64, I_CMP64, SINK_F64)
KERNEL(k_mov64, DECL_F64, I_MOV64, SINK_F64)
KERNEL(k_fma64_chain, DECL_F64, I_FMA64_CHAIN, SINK_F64)
KERNEL(k_add64_chain, DECL_F64, I_ADD64_CHAIN, SINK_F64)
KERNEL(k_rcp64_chain, DECL_F64, I_RCP64_CHAIN, SINK_F64)
KERNEL(k_cmp64_s, DECL_F64, I_CMP64_S, SINK_F64)
KERNEL(k_fma64_s, DECL_F64, I_FMA64_S, SINK_F64)
KERNEL(k_cvt_f64_f32, DECL_MIX, I_CVT_F64_F32, SINK_MIX)
KERNEL(k_cvt_f32_f64, DECL_MIX, I_CVT_F32_F64, SINK_MIX)
KERNEL(k_cvt_f64_i32, DECL_MIX, I_CVT_F64_I32, SINK_MIX)
KERNEL(k_cvt_i32_f64, DECL_MIX, I_CVT_I32_F64, SINK_MIX)
KERNEL(k_mad_u64_u32, DECL_MIX, I_MAD64, SINK_MIX)
KERNEL(k_lshl_add_u64, DECL_MIX, I_LSHLADD64, SINK_MIX)
KERNEL(k_fma32, DECL_F32, I_FMA32, SINK_F32)
KERNEL(k_log32, DECL_F32, I_LOG32, SINK_F32)
KERNEL(k_sqrt32, DECL_F32, I_SQRT32, SINK_F32)
KERNEL(k_sin32, DECL_F32, I_SIN32, SINK_F32)
KERNEL(k_cvt_f32_u32, DECL_F32, I_CVTU32, SINK_F32)

struct Test {
    const char* name;
    void (*fn)(long long*, int, int);
};

int main() {
    Test tests[] = {
        {"v_xor_b32", k_xor}, {"v_mov_b32", k_mov}, {"v_cndmask_b32", k_cndmask}, {"v_mov_b32_dpp", k_dpp}, {"v_cndmask_b32_e64 sgpr", k_cndmask64}, {"v_cndmask_b32_e64 vcc", k_cndmask_e64vcc}, {"v_cmp vcc + v_cndmask_e32 (pair)", k_cmp_cnd_vcc}, {"v_cmp_e64 sgpr + v_cndmask_e64 (pair)", k_cmp_cnd_sgpr}, {"v_readlane_b32", k_readlane}, {"v_xor_b32 dependent", k_xor_chain}, {"v_add3_u32 dependent", k_add3_chain}, {"v_mov_b32_dpp dependent", k_dpp_chain}, {"ds_bpermute_b32+wait", k_bperm}, {"ds_bpermute_b32", k_bperm_nowait},
        {"v_add3_u32", k_add3}, {"v_lshl_add_u32", k_lshladd}, {"v_mul_lo_u32", k_mullo}, {"v_mul_hi_u32", k_mulhi},
        {"v_mul_u32_u24", k_mul24}, {"v_mad_u32_u24", k_mad24}, {"v_pk_min_u16", k_pkmin}, {"v_perm_b32", k_perm},
        {"v_alignbit_b32", k_alignbit}, {"v_bfe_u32", k_bfe}, {"v_fma_f64", k_fma64}, {"v_add_f64", k_add64},
        {"v_mul_f64", k_mul64}, {"v_max_f64", k_max64}, {"v_rcp_f64", k_rcp64}, {"v_floor_f64", k_floor64},
        {"v_ldexp_f64", k_ldexp64}, {"v_cmp_lt_f64", k_cmp64}, {"v_mov_b64", k_mov64}, {"v_fma_f64 dependent", k_fma64_chain}, {"v_add_f64 dependent", k_add64_chain}, {"v_rcp_f64 dependent", k_rcp64_chain}, {"v_cmp_lt_f64_e64 sgpr", k_cmp64_s}, {"v_fma_f64 sgpr operand", k_fma64_s},
        {"v_cvt_f64_f32", k_cvt_f64_f32}, {"v_cvt_f32_f64", k_cvt_f32_f64}, {"v_cvt_f64_i32", k_cvt_f64_i32},
        {"v_cvt_i32_f64", k_cvt_i32_f64}, {"v_mad_u64_u32", k_mad_u64_u32}, {"v_lshl_add_u64", k_lshl_add_u64},
        {"v_fma_f32", k_fma32}, {"v_log_f32", k_log32}, {"v_sqrt_f32", k_sqrt32}, {"v_sin_f32", k_sin32},
        {"v_cvt_f32_u32", k_cvt_f32_u32},
    };
    const int iters = 2000, blocks = 256;
    long long* d;
    hipMalloc(&d, sizeof(long long) * blocks * 16);
    std::vector<long long> h(blocks * 16);
    printf("{\"unit\": \"SIMD cycles per wave-instruction (s_memtime ticks / (instructions x waves per SIMD))\", \"rows\": {\n");
    bool first = true;
    for (auto& t : tests) {
        printf("%s  \"%s\": {", first ? "" : ",\n", t.name);
        first = false;
        int k = 0;
        for (int wps : {1, 2, 4}) {
            int threads = 64 * 4 * wps;  // one workgroup per CU: 4 SIMDs x wps waves
            hipLaunchKernelGGL(t.fn, dim3(blocks), dim3(threads), 0, 0, d, 50, 1);
            hipLaunchKernelGGL(t.fn, dim3(blocks), dim3(threads), 0, 0, d, iters, 1);
            hipDeviceSynchronize();
            int n = blocks * threads / 64;
            hipMemcpy(h.data(), d, sizeof(long long) * n, hipMemcpyDeviceToHost);
            std::sort(h.begin(), h.begin() + n);
            double cyc = (double)h[n / 2] / ((double)iters * 64.0 * wps);
            printf("%s\"w%d\": %.2f", k++ ? ", " : "", wps, cyc);
        }
        printf("}");
    }
    printf("\n}}\n");
    hipFree(d);
    return 0;
}
