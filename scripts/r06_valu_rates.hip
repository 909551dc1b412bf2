// r06_valu_rates.hip -- what one instruction of the pixel kernels' inner loops costs a SIMD on gfx950, relative to v_add_u32: the packed
// 16-bit, dot-product, SDWA, saturate and sum-of-absolute-differences instructions of k_resample_420 / k_resample_hv1 / k_area_420(t)
// (lp_kernels_pixel.hip, lp_area_core.h). 256 workgroups x 1024 threads = four waves per SIMD on every CU, eight independent chains per
// lane, no memory traffic: the time per instruction is the SIMD's ISSUE cost of that instruction (dependency latency hidden by the
// chains and the waves). Used by profiles/r06_resample_floor.md to price the kernels' instruction streams.
// Build: hipcc -O3 --offload-arch=gfx950 -o r06_valu_rates r06_valu_rates.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

// OP8(text with %N placeholders): eight copies on registers %0..%7, second source %8, third %9
#define BODY8(I0, I1, I2, I3, I4, I5, I6, I7) asm volatile(I0 "\n" I1 "\n" I2 "\n" I3 "\n" I4 "\n" I5 "\n" I6 "\n" I7 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c))

#define KERNEL(NAME, I0, I1, I2, I3, I4, I5, I6, I7)                                                                         \
    __global__ void __launch_bounds__(1024) NAME(uint32_t* out, int iters)                                                   \
    {                                                                                                                        \
        uint32_t a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7; \
        uint32_t b = (blockIdx.x | 1) * 0x00010001u, c = threadIdx.x * 0x01010101u;                                          \
        for (int i = 0; i < iters; i++) {                                                                                    \
            _Pragma("unroll") for (int u = 0; u < 4; u++) BODY8(I0, I1, I2, I3, I4, I5, I6, I7);                             \
        }                                                                                                                    \
        out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;                                  \
    }
#define K1(NAME, OP) KERNEL(NAME, OP(0), OP(1), OP(2), OP(3), OP(4), OP(5), OP(6), OP(7))

#define OP_ADD(n) "v_add_u32 %" #n ", %" #n ", %8"
#define OP_PKMUL(n) "v_pk_mul_lo_u16 %" #n ", %" #n ", %8"
#define OP_PKADD(n) "v_pk_add_u16 %" #n ", %" #n ", %8"
#define OP_PKMAD(n) "v_pk_mad_u16 %" #n ", %" #n ", %8, %9"
#define OP_PKSHR(n) "v_pk_lshrrev_b16 %" #n ", %8, %" #n
#define OP_DOT2(n) "v_dot2_u32_u16 %" #n ", %" #n ", %8, %9"
#define OP_DOT4(n) "v_dot4_u32_u8 %" #n ", %" #n ", %8, %9"
#define OP_SDWA(n) "v_add_u16_sdwa %" #n ", %8, %9 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1 src1_sel:BYTE_1"
#define OP_SAT(n) "v_sat_pk_u8_i16 %" #n ", %" #n
#define OP_SATS(n) "v_sat_pk_u8_i16_sdwa %" #n ", %8 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD"
#define OP_SAD(n) "v_sad_u8 %" #n ", %8, 0, %" #n
#define OP_PERM(n) "v_perm_b32 %" #n ", %" #n ", %8, %9"
#define OP_CVTB(n) "v_cvt_f32_ubyte1 %" #n ", %" #n
#define OP_FMUL(n) "v_mul_f32 %" #n ", %" #n ", %8"
#define OP_FADD(n) "v_add_f32 %" #n ", %" #n ", %8"
#define OP_BFE(n) "v_bfe_u32 %" #n ", %" #n ", 3, 9"
#define OP_ALIGN(n) "v_alignbit_b32 %" #n ", %" #n ", %8, %9"
#define OP_LSHLADD(n) "v_lshl_add_u32 %" #n ", %" #n ", 2, %8"
#define OP_AND(n) "v_and_b32 %" #n ", %" #n ", %8"
#define OP_LSHL(n) "v_lshlrev_b32 %" #n ", 3, %" #n
#define OP_SUB(n) "v_sub_u32 %" #n ", %" #n ", %8"
#define OP_MIN(n) "v_min_u32 %" #n ", %" #n ", %8"
#define OP_FMA(n) "v_fma_f32 %" #n ", %" #n ", %8, %9"
#define OP_MAD24(n) "v_mad_u32_u24 %" #n ", %" #n ", %8, %9"
#define OP_MUL24(n) "v_mul_u32_u24 %" #n ", %" #n ", %8"
#define OP_CNDM(n) "v_cndmask_b32 %" #n ", %" #n ", %8, vcc"
#define OP_CVTI(n) "v_cvt_f32_u32 %" #n ", %" #n
#define OP_ADD3(n) "v_add3_u32 %" #n ", %" #n ", %8, %9"
#define OP_MOV(n) "v_mov_b32 %" #n ", %8"
#define OP_XOR(n) "v_xor_b32 %" #n ", %" #n ", %8"
#define OP_ADDC(n) "v_add_co_u32 %" #n ", vcc, %" #n ", %8"

K1(k_add, OP_ADD) K1(k_pkmul, OP_PKMUL) K1(k_pkadd, OP_PKADD) K1(k_pkmad, OP_PKMAD) K1(k_pkshr, OP_PKSHR) K1(k_dot2, OP_DOT2) K1(k_dot4, OP_DOT4)
K1(k_sdwa, OP_SDWA) K1(k_sat, OP_SAT) K1(k_sats, OP_SATS) K1(k_sad, OP_SAD) K1(k_perm, OP_PERM) K1(k_cvtb, OP_CVTB) K1(k_fmul, OP_FMUL) K1(k_fadd, OP_FADD)
K1(k_bfe, OP_BFE) K1(k_align, OP_ALIGN) K1(k_lshladd, OP_LSHLADD)
K1(k_and, OP_AND) K1(k_lshl, OP_LSHL) K1(k_sub, OP_SUB) K1(k_min, OP_MIN) K1(k_fma, OP_FMA) K1(k_mad24, OP_MAD24) K1(k_mul24, OP_MUL24) K1(k_cndm, OP_CNDM) K1(k_cvti, OP_CVTI) K1(k_add3, OP_ADD3) K1(k_mov, OP_MOV) K1(k_xor, OP_XOR)

// v_pk_mul_f32 / v_pk_add_f32 work on register PAIRS (two floats per lane and instruction): eight independent pairs
template <int WHICH>
__global__ void __launch_bounds__(1024) k_pkf32(float* out, int iters)
{
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 a0 = {1.f + threadIdx.x, 2.f}, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f, a4 = a0 + 4.f, a5 = a0 + 5.f, a6 = a0 + 6.f, a7 = a0 + 7.f;
    const f2 b = {1.0000001f, 0.9999999f};
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int u = 0; u < 4; u++) {
            if (WHICH == 0)
                asm volatile("v_pk_mul_f32 %0, %0, %8\n v_pk_mul_f32 %1, %1, %8\n v_pk_mul_f32 %2, %2, %8\n v_pk_mul_f32 %3, %3, %8\n v_pk_mul_f32 %4, %4, %8\n v_pk_mul_f32 %5, %5, %8\n v_pk_mul_f32 %6, %6, %8\n v_pk_mul_f32 %7, %7, %8"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));
            else
                asm volatile("v_pk_add_f32 %0, %0, %8\n v_pk_add_f32 %1, %1, %8\n v_pk_add_f32 %2, %2, %8\n v_pk_add_f32 %3, %3, %8\n v_pk_add_f32 %4, %4, %8\n v_pk_add_f32 %5, %5, %8\n v_pk_add_f32 %6, %6, %8\n v_pk_add_f32 %7, %7, %8"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));
        }
    }
    const f2 r = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    out[blockIdx.x * blockDim.x + threadIdx.x] = r.x + r.y;
}
#define OP_CMP(n) "v_cmp_eq_u32 vcc, %" #n ", %8"
#define OP_LSHR(n) "v_lshrrev_b32 %" #n ", 3, %" #n
#define OP_ASHR(n) "v_ashrrev_i32 %" #n ", 3, %" #n
#define OP_CNDS(n) "v_cndmask_b32 %" #n ", %" #n ", %8, s[20:21]"
K1(k_cmp, OP_CMP) K1(k_lshr, OP_LSHR) K1(k_ashr, OP_ASHR)

int main()
{
    uint32_t* d_out;
    CK(hipMalloc(&d_out, 256 * 1024 * 4));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int iters = 4000;
    struct { const char* name; void (*k)(uint32_t*, int); } ks[] = {
        {"v_add_u32", k_add}, {"v_pk_mul_lo_u16", k_pkmul}, {"v_pk_add_u16", k_pkadd}, {"v_pk_mad_u16", k_pkmad}, {"v_pk_lshrrev_b16", k_pkshr},
        {"v_dot2_u32_u16", k_dot2}, {"v_dot4_u32_u8", k_dot4}, {"v_add_u16_sdwa (dst WORD_1, preserve)", k_sdwa}, {"v_sat_pk_u8_i16", k_sat},
        {"v_sat_pk_u8_i16_sdwa (dst WORD_1, preserve)", k_sats}, {"v_sad_u8", k_sad}, {"v_perm_b32", k_perm}, {"v_cvt_f32_ubyte1", k_cvtb},
        {"v_mul_f32", k_fmul}, {"v_add_f32", k_fadd}, {"v_bfe_u32", k_bfe}, {"v_alignbit_b32", k_align}, {"v_lshl_add_u32", k_lshladd},
        {"v_and_b32", k_and}, {"v_lshlrev_b32", k_lshl}, {"v_sub_u32", k_sub}, {"v_min_u32", k_min}, {"v_fma_f32", k_fma}, {"v_mad_u32_u24", k_mad24}, {"v_mul_u32_u24", k_mul24},
        {"v_cvt_f32_u32", k_cvti}, {"v_add3_u32", k_add3}, {"v_mov_b32", k_mov}, {"v_xor_b32", k_xor}, {"v_cmp_eq_u32 (vcc)", k_cmp}, {"v_lshrrev_b32", k_lshr}, {"v_ashrrev_i32", k_ashr},
        {"v_pk_mul_f32 (two floats)", (void (*)(uint32_t*, int))k_pkf32<0>}, {"v_pk_add_f32 (two floats)", (void (*)(uint32_t*, int))k_pkf32<1>}};
    double base = 0;
    for (auto& e : ks) {
        e.k<<<256, 1024>>>(d_out, 50);
        CK(hipDeviceSynchronize());
        float best = 1e9f;
        for (int rep = 0; rep < 3; rep++) {
            CK(hipEventRecord(e0));
            e.k<<<256, 1024>>>(d_out, iters);
            CK(hipEventRecord(e1));
            CK(hipDeviceSynchronize());
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            best = ms < best ? ms : best;
        }
        const double per_simd_ns = best * 1e6 / ((double)iters * 32 * 4); // four waves per SIMD
        if (!base) base = per_simd_ns;
        printf("%-46s %8.3f ms  %6.3f ns per instruction per SIMD  = %5.2f x v_add_u32\n", e.name, best, per_simd_ns, per_simd_ns / base);
    }
    return 0;
}
