// Micro-benchmark: issue cost of single VALU / LDS-crossbar instructions of the render kernels on gfx950, in SIMD cycles per wave64 instruction.
// Every kernel runs the same loop shape -- 64 copies of ONE instruction on eight independent register chains per trip -- on every SIMD of the
// chip with W waves per SIMD (W = 1, 3: the render kernels run three), and the cost is  elapsed x clock x SIMDs / instructions.
// What it is for: round 2 measured v_fma_f32 at ~2 cycles and v_pk_fma_f32 at ~4 (profiles/r02_micro_pk_fma_rate.txt), so an
// instruction COUNT is not a cycle count; this table prices the instruction classes the level-3 kernel is made of.
//   hipcc --offload-arch=gfx950 -O3 -o valu_cost valu_cost.hip && ./valu_cost
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
#define REP64(X) REP8(X) REP8(X) REP8(X) REP8(X) REP8(X) REP8(X) REP8(X) REP8(X)

// a[0..7]: float chains, b: a second float operand, i[0..7]: integer chains
#define KERNEL(NAME, ASMLINE)                                                                                                  \
    __global__ void __launch_bounds__(768) NAME(float *out, int iters)                                                         \
    {                                                                                                                           \
        float a0 = threadIdx.x * 1e-3f, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7; \
        float b = 0.999f, c = 0.25f;                                                                                            \
        int lane = threadIdx.x & 63;                                                                                            \
        int addr = ((lane * 5) & 63) << 2;                                                                                      \
        asm volatile("" : "+v"(b), "+v"(c), "+v"(addr));                                                                       \
        for (int it = 0; it < iters; ++it) {                                                                                    \
            REP64(ASMLINE)                                                                                                      \
        }                                                                                                                       \
        asm volatile("s_waitcnt lgkmcnt(0)");                                                                                  \
        out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;                                    \
    }

#define L_FMA(k) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a##k) : "v"(b), "v"(c));
#define L_FMAC(k) asm volatile("v_fmac_f32_e32 %0, %1, %2" : "+v"(a##k) : "v"(b), "v"(c));
#define L_MUL(k) asm volatile("v_mul_f32_e32 %0, %0, %1" : "+v"(a##k) : "v"(b));
#define L_ADD(k) asm volatile("v_add_f32_e32 %0, %0, %1" : "+v"(a##k) : "v"(c));
#define L_MAX(k) asm volatile("v_max_f32_e32 %0, %0, %1" : "+v"(a##k) : "v"(c));
#define L_MED3(k) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(a##k) : "v"(c), "v"(b));
#define L_MOV(k) asm volatile("v_mov_b32_e32 %0, %1" : "=v"(a##k) : "v"(b));
// (no "vcc" clobber on the readers: with it hipcc pads every statement with an s_nop; nothing here writes vcc inside the loop)
#define L_CNDMASK(k) asm volatile("v_cndmask_b32_e32 %0, %0, %1, vcc" : "+v"(a##k) : "v"(c));
#define L_CNDMASK64(k) asm volatile("v_cndmask_b32_e64 %0, %0, %1, s[20:21]" : "+v"(a##k) : "v"(c));
#define L_CMP(k) asm volatile("v_cmp_gt_f32_e32 vcc, %0, %1" : : "v"(a##k), "v"(c) : "vcc");
#define L_CMP64(k) asm volatile("v_cmp_gt_f32_e64 s[22:23], %0, %1" : : "v"(a##k), "v"(c) : "s22", "s23");
#define L_CMPSEL(k) asm volatile("v_cmp_gt_f32_e32 vcc, %0, %1\n\tv_cndmask_b32_e32 %0, %0, %1, vcc" : "+v"(a##k) : "v"(c) : "vcc");
#define L_BPERMW(k) asm volatile("ds_bpermute_b32 %0, %1, %0\n\ts_waitcnt lgkmcnt(0)" : "+v"(a##k) : "v"(addr));
#define L_SWAP(k) asm volatile("v_permlane32_swap_b32_e32 %0, %1" : "+v"(a##k), "+v"(b));
#define L_ADDCO(k) asm volatile("v_add_co_u32_e32 %0, vcc, %0, %1" : "+v"(a##k) : "v"(addr) : "vcc");
#define L_MAXI(k) asm volatile("v_max_i32_e32 %0, %0, %1" : "+v"(a##k) : "v"(addr));
#define L_MADU64(k) asm volatile("v_mad_u64_u32 %0, s[22:23], %1, 48, %0" : "+v"(p##k) : "v"(addr) : "s22", "s23");
#define L_CMPX(k) asm volatile("v_cmpx_lt_f32_e32 vcc, %0, %1\n\ts_mov_b64 exec, -1" : : "v"(c), "v"(b) : "vcc");
#define L_ADDU(k) asm volatile("v_add_u32_e32 %0, %0, %1" : "+v"(a##k) : "v"(addr));
#define L_LSHLADD(k) asm volatile("v_lshl_add_u32 %0, %0, 2, %1" : "+v"(a##k) : "v"(addr));
#define L_LSHLOR(k) asm volatile("v_lshl_or_b32 %0, %0, 8, %1" : "+v"(a##k) : "v"(addr));
#define L_MADU24(k) asm volatile("v_mad_u32_u24 %0, %0, %1, %1" : "+v"(a##k) : "v"(addr));
#define L_AND(k) asm volatile("v_and_b32_e32 %0, %0, %1" : "+v"(a##k) : "v"(addr));
#define L_FLOOR(k) asm volatile("v_floor_f32_e32 %0, %0" : "+v"(a##k));
#define L_CVTI(k) asm volatile("v_cvt_i32_f32_e32 %0, %0" : "+v"(a##k));
#define L_CVTF(k) asm volatile("v_cvt_f32_i32_e32 %0, %0" : "+v"(a##k));
#define L_EXP(k) asm volatile("v_exp_f32_e32 %0, %0" : "+v"(a##k));
#define L_LOG(k) asm volatile("v_log_f32_e32 %0, %0" : "+v"(a##k));
#define L_RCP(k) asm volatile("v_rcp_f32_e32 %0, %0" : "+v"(a##k));
#define L_DPPMUL(k) asm volatile("v_mul_f32_dpp %0, %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a##k) : "v"(b));
#define L_DPPMOV(k) asm volatile("v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a##k) : "v"(b));
#define L_BPERM(k) asm volatile("ds_bpermute_b32 %0, %1, %0" : "+v"(a##k) : "v"(addr));
#define L_READLANE(k) asm volatile("v_readlane_b32 s20, %0, 3" : : "v"(a##k) : "s20");
#define L_READFIRST(k) asm volatile("v_readfirstlane_b32 s20, %0" : : "v"(a##k) : "s20");
#define L_ACCREAD(k) asm volatile("v_accvgpr_read_b32 %0, a0" : "=v"(a##k));
#define L_FMA_S(k) asm volatile("v_fma_f32 %0, %0, s20, %1" : "+v"(a##k) : "v"(c) : "s20");
#define L_ADD3(k) asm volatile("v_add3_u32 %0, %0, %1, %1" : "+v"(a##k) : "v"(addr));
#define L_BFE(k) asm volatile("v_bfe_u32 %0, %0, 4, 4" : "+v"(a##k));
#define L_SUBREV(k) asm volatile("v_sub_f32_e32 %0, 1.0, %0" : "+v"(a##k));
#define L_MULLIT(k) asm volatile("v_mul_f32_e32 %0, 0x3fb8aa3b, %0" : "+v"(a##k));
#define L_SNOP(k) asm volatile("s_nop 0");
#define L_SMOV(k) asm volatile("s_mov_b32 s20, s21" ::: "s20");

KERNEL(k_fma, L_FMA) KERNEL(k_fmac, L_FMAC) KERNEL(k_mul, L_MUL) KERNEL(k_add, L_ADD) KERNEL(k_max, L_MAX) KERNEL(k_med3, L_MED3) KERNEL(k_mov, L_MOV)
KERNEL(k_cndmask, L_CNDMASK) KERNEL(k_cmp, L_CMP) KERNEL(k_cmpx, L_CMPX) KERNEL(k_addu, L_ADDU) KERNEL(k_lshladd, L_LSHLADD) KERNEL(k_lshlor, L_LSHLOR)
KERNEL(k_madu24, L_MADU24) KERNEL(k_and, L_AND) KERNEL(k_floor, L_FLOOR) KERNEL(k_cvti, L_CVTI) KERNEL(k_cvtf, L_CVTF) KERNEL(k_exp, L_EXP) KERNEL(k_log, L_LOG)
KERNEL(k_rcp, L_RCP) KERNEL(k_dppmul, L_DPPMUL) KERNEL(k_dppmov, L_DPPMOV) KERNEL(k_bperm, L_BPERM) KERNEL(k_readlane, L_READLANE) KERNEL(k_readfirst, L_READFIRST)
KERNEL(k_accread, L_ACCREAD) KERNEL(k_fma_s, L_FMA_S) KERNEL(k_add3, L_ADD3) KERNEL(k_bfe, L_BFE) KERNEL(k_subrev, L_SUBREV) KERNEL(k_mullit, L_MULLIT)
KERNEL(k_snop, L_SNOP) KERNEL(k_smov, L_SMOV) KERNEL(k_cndmask64, L_CNDMASK64) KERNEL(k_cmp64, L_CMP64) KERNEL(k_cmpsel, L_CMPSEL) KERNEL(k_bpermw, L_BPERMW)
KERNEL(k_swap, L_SWAP) KERNEL(k_addco, L_ADDCO) KERNEL(k_maxi, L_MAXI)

// packed fp32 on register pairs
__global__ void __launch_bounds__(768) k_pkfma(float *out, int iters)
{
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    f32x2 a0 = {1.f, 2.f}, a1 = a0, a2 = a0, a3 = a0, a4 = a0, a5 = a0, a6 = a0, a7 = a0, b = {0.999f, 0.998f}, c = {0.25f, 0.5f};
    asm volatile("" : "+v"(b), "+v"(c));
#define L_PKFMA(k) asm volatile("v_pk_fma_f32 %0, %0, %1, %2 op_sel_hi:[1,0,1]" : "+v"(a##k) : "v"(b), "v"(c));
    for (int it = 0; it < iters; ++it) { REP64(L_PKFMA) }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0[0] + a1[1] + a2[0] + a3[1] + a4[0] + a5[1] + a6[0] + a7[1];
}

typedef void (*kern_t)(float *, int);
struct Entry { const char *name; kern_t k; };

int main(int argc, char **argv)
{
    float *out;
    (void)hipMalloc(&out, 256 * 768 * sizeof(float));
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    int clk_khz = 2400000;
    (void)hipDeviceGetAttribute(&clk_khz, hipDeviceAttributeClockRate, 0);
    const Entry es[] = {{"v_fma_f32 (VOP3)", k_fma}, {"v_fmac_f32_e32", k_fmac}, {"v_fma_f32 with an SGPR source", k_fma_s}, {"v_pk_fma_f32 (2 FMAs / lane)", k_pkfma}, {"v_mul_f32", k_mul},
                        {"v_mul_f32 literal", k_mullit}, {"v_add_f32", k_add}, {"v_sub_f32 1.0 - x", k_subrev}, {"v_max_f32", k_max}, {"v_med3_f32", k_med3}, {"v_mov_b32", k_mov},
                        {"v_cndmask_b32 (vcc)", k_cndmask}, {"v_cndmask_b32_e64 (s[20:21])", k_cndmask64}, {"v_cmp_gt_f32 -> vcc", k_cmp}, {"v_cmp_gt_f32_e64 -> s[22:23]", k_cmp64},
                        {"v_cmp + v_cndmask (pair)", k_cmpsel}, {"ds_bpermute + s_waitcnt (each)", k_bpermw}, {"v_permlane32_swap", k_swap}, {"v_add_co_u32", k_addco}, {"v_max_i32", k_maxi}, {"v_cmpx + s_mov exec", k_cmpx}, {"v_add_u32", k_addu}, {"v_add3_u32", k_add3},
                        {"v_lshl_add_u32", k_lshladd}, {"v_lshl_or_b32", k_lshlor}, {"v_mad_u32_u24", k_madu24}, {"v_and_b32", k_and}, {"v_bfe_u32", k_bfe}, {"v_floor_f32", k_floor},
                        {"v_cvt_i32_f32", k_cvti}, {"v_cvt_f32_i32", k_cvtf}, {"v_exp_f32", k_exp}, {"v_log_f32", k_log}, {"v_rcp_f32", k_rcp}, {"v_mul_f32_dpp row_shr", k_dppmul},
                        {"v_mov_b32_dpp row_shr", k_dppmov}, {"ds_bpermute_b32", k_bperm}, {"v_readlane_b32", k_readlane}, {"v_readfirstlane_b32", k_readfirst},
                        {"v_accvgpr_read_b32", k_accread}, {"s_nop 0", k_snop}, {"s_mov_b32", k_smov}};
    printf("clock %.2f GHz (device attribute); cycles per wave64 instruction per SIMD = elapsed x clock x 1024 SIMDs / instructions\n", clk_khz / 1e6);
    printf("%-32s %12s %12s %12s\n", "instruction", "1 wave/SIMD", "2 waves/SIMD", "3 waves/SIMD");
    const int iters = 4000;
    for (const Entry &e : es) {
        printf("%-32s", e.name);
        for (int wps = 1; wps <= 3; ++wps) {
            const int threads = 256 * wps;          // 4 SIMDs x wps waves
            hipLaunchKernelGGL(e.k, dim3(256), dim3(threads), 0, 0, out, 10);
            (void)hipDeviceSynchronize();
            float best = 1e30f;
            for (int r = 0; r < 3; ++r) {
                (void)hipEventRecord(e0);
                hipLaunchKernelGGL(e.k, dim3(256), dim3(threads), 0, 0, out, iters);
                (void)hipEventRecord(e1);
                (void)hipEventSynchronize(e1);
                float ms = 0;
                (void)hipEventElapsedTime(&ms, e0, e1);
                if (ms < best) best = ms;
            }
            const double instr_per_simd = (double)iters * 64 * wps;
            printf(" %12.2f", best * 1e-3 * (clk_khz * 1e3) / instr_per_simd);
        }
        printf("\n");
    }
    return 0;
}
