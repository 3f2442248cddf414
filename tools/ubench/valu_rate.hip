// Instruction issue rates on gfx950: cycles per wave-instruction per SIMD with 1, 2 and 4 waves a SIMD, 128 instructions (8 independent
// registers x 16) per loop iteration. hipcc --offload-arch=gfx950 -O3 -o tools/ab/valu_rate tools/ubench/valu_rate.hip ; run on the GPU box
// (profiles/r04_valu_rates.txt, r05_valu_rates.txt). Cycles are at the nominal clock the runtime reports; the chip runs below it under load.
#include <hip/hip_runtime.h>
#include <cstdio>

#define REP16(x) x x x x x x x x x x x x x x x x
#define OUT8 "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7])
#define OUT8D "+v"(d[0]), "+v"(d[1]), "+v"(d[2]), "+v"(d[3]), "+v"(d[4]), "+v"(d[5]), "+v"(d[6]), "+v"(d[7])
// I(n): one instruction on register n; %8 = a VGPR operand, %9 = an SGPR operand, %10 = an SGPR pair
#define EIGHT(I) I("%0") I("%1") I("%2") I("%3") I("%4") I("%5") I("%6") I("%7")

#define KERNEL(NAME, I)                                                                                                  \
    __global__ __launch_bounds__(256) void NAME(float *out, int iters, float seed) {                                     \
        float r[8];                                                                                                      \
        for (int i = 0; i < 8; ++i) r[i] = seed + i + threadIdx.x;                                                       \
        float s = seed * 0.5f;                                                                                           \
        float ss = __builtin_amdgcn_readfirstlane(__float_as_int(seed));                                                 \
        unsigned long long mask = 0x5555aaaa5555aaaaull + (unsigned long long)__builtin_amdgcn_readfirstlane(iters);    \
        for (int it = 0; it < iters; ++it) asm volatile(REP16(EIGHT(I)) : OUT8 : "v"(s), "s"(ss), "s"(mask) : "vcc");     \
        float acc = 0;                                                                                                   \
        for (int i = 0; i < 8; ++i) acc += r[i];                                                                         \
        if (acc == 123.456f) out[blockIdx.x * blockDim.x + threadIdx.x] = acc;                                           \
    }
#define KERNELD(NAME, I)                                                                                                 \
    __global__ __launch_bounds__(256) void NAME(float *out, int iters, float seed) {                                     \
        double d[8];                                                                                                     \
        for (int i = 0; i < 8; ++i) d[i] = seed + i + threadIdx.x;                                                       \
        double s = seed * 0.5f;                                                                                          \
        for (int it = 0; it < iters; ++it) asm volatile(REP16(EIGHT(I)) : OUT8D : "v"(s));                               \
        double acc = 0;                                                                                                  \
        for (int i = 0; i < 8; ++i) acc += d[i];                                                                         \
        if (acc == 123.456) out[blockIdx.x * blockDim.x + threadIdx.x] = (float)acc;                                     \
    }

#define I_ADD(n) "v_add_f32_e32 " n ", %8, " n "\n"
#define I_ADD_S(n) "v_add_f32_e32 " n ", %9, " n "\n"
#define I_ADD_LIT(n) "v_add_f32_e32 " n ", 0x3f012345, " n "\n"
#define I_ADD_INL(n) "v_add_f32_e32 " n ", 0.5, " n "\n"
#define I_MUL(n) "v_mul_f32_e32 " n ", %8, " n "\n"
#define I_FMA(n) "v_fma_f32 " n ", %8, " n ", " n "\n"
#define I_FMAC(n) "v_fmac_f32_e32 " n ", %8, %8\n"
#define I_MIN(n) "v_min_f32_e32 " n ", %8, " n "\n"
#define I_MIN3(n) "v_min3_f32 " n ", %8, " n ", " n "\n"
#define I_AND(n) "v_and_b32_e32 " n ", %8, " n "\n"
#define I_ADDU(n) "v_add_u32_e32 " n ", %8, " n "\n"
#define I_LSHLOR(n) "v_lshl_or_b32 " n ", " n ", 3, %8\n"
#define I_BFI(n) "v_bfi_b32 " n ", %8, " n ", " n "\n"
#define I_MULLO(n) "v_mul_lo_u32 " n ", %8, " n "\n"
#define I_CVT(n) "v_cvt_f32_u32_e32 " n ", " n "\n"
#define I_RCP(n) "v_rcp_f32_e32 " n ", " n "\n"
#define I_CND_VCC(n) "v_cndmask_b32_e32 " n ", %8, " n ", vcc\n"
#define I_CND_S(n) "v_cndmask_b32_e64 " n ", %8, " n ", %10\n"
#define I_CMP_VCC(n) "v_cmp_lt_f32_e32 vcc, %8, " n "\n"
#define I_CMP_S(n) "v_cmp_lt_f32_e64 s[20:21], %8, " n "\n"
#define I_CMP_CND(n) "v_cmp_lt_f32_e32 vcc, %8, " n "\n v_add_f32_e32 " n ", %8, " n "\n v_add_f32_e32 " n ", %8, " n "\n v_cndmask_b32_e32 " n ", %8, " n ", vcc\n"
#define I_DPP_WSHR(n) "v_mov_b32_dpp " n ", " n " wave_shr:1 row_mask:0xf bank_mask:0xf\n"
#define I_DPP_RSHR(n) "v_mov_b32_dpp " n ", " n " row_shr:1 row_mask:0xf bank_mask:0xf\n"
#define I_ADD_DPP(n) "v_add_f32_dpp " n ", %8, " n " wave_shr:1 row_mask:0xf bank_mask:0xf\n"
#define I_ADD_SDWA(n) "v_add_f32_sdwa " n ", %8, " n " dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD\n"
#define I_READLANE(n) "v_readlane_b32 s22, " n ", 5\n"
#define I_SADD(n) "s_add_i32 s22, s22, 1\n"
#define I_SNOP(n) "s_nop 0\n"
#define I_PKADD(n) "v_pk_add_f32 " n ", %8, " n "\n"
#define I_PKFMA(n) "v_pk_fma_f32 " n ", %8, " n ", " n "\n"
#define I_PKMUL(n) "v_pk_mul_f32 " n ", %8, " n "\n"
#define I_PKADDU16(n) "v_pk_add_u16 " n ", %8, " n "\n"
#define I_PKSUBU16(n) "v_pk_sub_u16 " n ", " n ", %8\n"
#define I_PKMADU16(n) "v_pk_mad_u16 " n ", %8, " n ", " n "\n"
#define I_PKMULLOU16(n) "v_pk_mul_lo_u16 " n ", %8, " n "\n"
#define I_PKLSHR16(n) "v_pk_lshrrev_b16 " n ", 3, " n "\n"
#define I_PERM(n) "v_perm_b32 " n ", %8, " n ", %9\n"
#define I_SDWA_B0(n) "v_add_u32_sdwa " n ", %8, " n " dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\n"
#define I_SDWA_MOV(n) "v_mov_b32_sdwa " n ", " n " dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2\n"
#define I_BFE(n) "v_bfe_u32 " n ", " n ", 8, 8\n"
#define I_MUL24(n) "v_mul_u32_u24_e32 " n ", %8, " n "\n"
#define I_MAD24(n) "v_mad_u32_u24 " n ", %8, " n ", " n "\n"
#define I_MULHI(n) "v_mul_hi_u32 " n ", %8, " n "\n"
#define I_LSHR(n) "v_lshrrev_b32_e32 " n ", 3, " n "\n"
#define I_ADD3(n) "v_add3_u32 " n ", %8, " n ", " n "\n"
#define I_SUBU(n) "v_sub_u32_e32 " n ", " n ", %8\n"
#define I_ADD_DPP_U(n) "v_add_u32_dpp " n ", %8, " n " row_shr:1 row_mask:0xf bank_mask:0xf\n"
#define I_ADD64(n) "v_add_f64 " n ", %8, " n "\n"
#define I_FMA64(n) "v_fma_f64 " n ", %8, " n ", " n "\n"

KERNEL(k_add, I_ADD)
KERNEL(k_add_s, I_ADD_S)
KERNEL(k_add_lit, I_ADD_LIT)
KERNEL(k_add_inl, I_ADD_INL)
KERNEL(k_mul, I_MUL)
KERNEL(k_fma, I_FMA)
KERNEL(k_fmac, I_FMAC)
KERNEL(k_min, I_MIN)
KERNEL(k_min3, I_MIN3)
KERNEL(k_and, I_AND)
KERNEL(k_addu, I_ADDU)
KERNEL(k_lshlor, I_LSHLOR)
KERNEL(k_bfi, I_BFI)
KERNEL(k_mullo, I_MULLO)
KERNEL(k_cvt, I_CVT)
KERNEL(k_rcp, I_RCP)
KERNEL(k_cnd_vcc, I_CND_VCC)
KERNEL(k_cnd_s, I_CND_S)
KERNEL(k_cmp_vcc, I_CMP_VCC)
KERNEL(k_cmp_s, I_CMP_S)
KERNEL(k_cmp_cnd, I_CMP_CND)
KERNEL(k_dpp_wshr, I_DPP_WSHR)
KERNEL(k_dpp_rshr, I_DPP_RSHR)
KERNEL(k_add_dpp, I_ADD_DPP)
KERNEL(k_add_sdwa, I_ADD_SDWA)
KERNEL(k_readlane, I_READLANE)
KERNEL(k_sadd, I_SADD)
KERNEL(k_snop, I_SNOP)
KERNEL(k_pk_add_u16, I_PKADDU16)
KERNEL(k_pk_sub_u16, I_PKSUBU16)
KERNEL(k_pk_mad_u16, I_PKMADU16)
KERNEL(k_pk_mul_lo_u16, I_PKMULLOU16)
KERNEL(k_pk_lshr_b16, I_PKLSHR16)
KERNEL(k_perm_b32, I_PERM)
KERNEL(k_add_u32_sdwa_byte, I_SDWA_B0)
KERNEL(k_mov_sdwa_byte, I_SDWA_MOV)
KERNEL(k_bfe_u32, I_BFE)
KERNEL(k_mul_u32_u24, I_MUL24)
KERNEL(k_mad_u32_u24, I_MAD24)
KERNEL(k_mul_hi_u32, I_MULHI)
KERNEL(k_lshrrev_b32, I_LSHR)
KERNEL(k_add3_u32, I_ADD3)
KERNEL(k_sub_u32, I_SUBU)
KERNEL(k_add_u32_dpp, I_ADD_DPP_U)
KERNELD(k_pkadd, I_PKADD)
KERNELD(k_pkfma, I_PKFMA)
KERNELD(k_pkmul, I_PKMUL)
KERNELD(k_add64, I_ADD64)
KERNELD(k_fma64, I_FMA64)

static void run(const char *name, void (*k)(float *, int, float), float *out, int clock_khz, int cus, int per_rep = 1) {
    const int iters = 1000;
    printf("%-34s", name);
    for (int waves_per_simd : {1, 2, 3, 4}) {
        const int blocks = cus * waves_per_simd;  // 256 threads = 4 waves = one per SIMD
        hipEvent_t e0, e1;
        (void)hipEventCreate(&e0);
        (void)hipEventCreate(&e1);
        hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, out, 10, 1.0f);
        (void)hipDeviceSynchronize();
        (void)hipEventRecord(e0, 0);
        hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0f);
        (void)hipEventRecord(e1, 0);
        (void)hipEventSynchronize(e1);
        float ms = 0;
        (void)hipEventElapsedTime(&ms, e0, e1);
        const double instr_per_wave = (double)iters * 128 * per_rep;
        const double cycles = ms * 1e-3 * clock_khz * 1e3;
        printf("  %dw: %6.2f", waves_per_simd, cycles / (instr_per_wave * waves_per_simd));
    }
    printf("   cycles per wave-instruction per SIMD\n");
}

int main() {
    int clock_khz = 0, cus = 0;
    (void)hipDeviceGetAttribute(&clock_khz, hipDeviceAttributeClockRate, 0);
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
    printf("CUs %d, clock %d kHz (nominal)\n", cus, clock_khz);
    float *out;
    (void)hipMalloc(&out, 1 << 24);
#define RUN(k) run(#k, k, out, clock_khz, cus)
    RUN(k_add); RUN(k_add_s); RUN(k_add_lit); RUN(k_add_inl); RUN(k_mul); RUN(k_fma); RUN(k_fmac); RUN(k_min); RUN(k_min3); RUN(k_and); RUN(k_addu);
    RUN(k_lshlor); RUN(k_bfi); RUN(k_mullo); RUN(k_cvt); RUN(k_rcp); RUN(k_cnd_vcc); RUN(k_cnd_s); RUN(k_cmp_vcc); RUN(k_cmp_s);
    run("k_cmp_cnd (cmp,add,add,cnd)", k_cmp_cnd, out, clock_khz, cus, 4);
    RUN(k_dpp_wshr); RUN(k_dpp_rshr); RUN(k_add_dpp); RUN(k_add_sdwa); RUN(k_readlane); RUN(k_sadd); RUN(k_snop);
    RUN(k_pk_add_u16); RUN(k_pk_sub_u16); RUN(k_pk_mad_u16); RUN(k_pk_mul_lo_u16); RUN(k_pk_lshr_b16); RUN(k_perm_b32); RUN(k_add_u32_sdwa_byte); RUN(k_mov_sdwa_byte);
    RUN(k_bfe_u32); RUN(k_mul_u32_u24); RUN(k_mad_u32_u24); RUN(k_mul_hi_u32); RUN(k_lshrrev_b32); RUN(k_add3_u32); RUN(k_sub_u32); RUN(k_add_u32_dpp);
    RUN(k_pkadd); RUN(k_pkfma); RUN(k_pkmul); RUN(k_add64); RUN(k_fma64);
    return 0;
}
