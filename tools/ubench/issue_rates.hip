// Issue cost of single VALU instructions on gfx950, measured with hand-written instruction streams (inline
// asm, so the compiler cannot fold or hoist anything): 4 independent register chains, 64 instructions per
// asm block, 4 waves per SIMD on every CU.  Prints cycles per wave-instruction per SIMD.
//   hipcc --offload-arch=gfx950 -O3 -o issue_rates issue_rates.hip && ./issue_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

constexpr int kIters = 2048;

#define REP4(x) x x x x
#define REP16(x) REP4(x) REP4(x) REP4(x) REP4(x)

#define KERNEL32(name, INSTR)                                                                         \
    __global__ __launch_bounds__(256) void name(uint32_t* out, unsigned long long* clk) {             \
        uint32_t a = threadIdx.x, b = threadIdx.x * 3u + 1u, c = threadIdx.x * 5u + 2u, d = 7u;       \
        uint32_t m = 0xD2511F53u;                                                                     \
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();                                   \
        const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();                               \
        for (int it = 0; it < kIters; ++it) {                                                         \
            asm volatile(REP16(INSTR) : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(m));                 \
        }                                                                                             \
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();                                   \
        const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();                               \
        out[blockIdx.x * blockDim.x + threadIdx.x] = a ^ b ^ c ^ d;                                   \
        if (threadIdx.x == 0 && blockIdx.x == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }              \
    }

#define KERNEL64(name, INSTR)                                                                         \
    __global__ __launch_bounds__(256) void name(uint32_t* out, unsigned long long* clk) {             \
        double a = 1.0 + threadIdx.x * 1e-9, b = 1.1, c = 1.2, d = 1.3;                               \
        double m = 1.0000001;                                                                         \
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();                                   \
        const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();                               \
        for (int it = 0; it < kIters; ++it) {                                                         \
            asm volatile(REP16(INSTR) : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(m));                 \
        }                                                                                             \
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();                                   \
        const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();                               \
        out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)(a + b + c + d);                       \
        if (threadIdx.x == 0 && blockIdx.x == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }              \
    }

// each INSTR string holds 4 instructions, one per chain
KERNEL32(k_xor, "v_xor_b32 %0, %0, %4\n v_xor_b32 %1, %1, %4\n v_xor_b32 %2, %2, %4\n v_xor_b32 %3, %3, %4\n")
KERNEL32(k_add, "v_add_u32 %0, %0, %4\n v_add_u32 %1, %1, %4\n v_add_u32 %2, %2, %4\n v_add_u32 %3, %3, %4\n")
KERNEL32(k_mov, "v_mov_b32 %0, %1\n v_mov_b32 %1, %2\n v_mov_b32 %2, %3\n v_mov_b32 %3, %0\n")
KERNEL32(k_cnd, "v_cndmask_b32 %0, %0, %4, vcc\n v_cndmask_b32 %1, %1, %4, vcc\n v_cndmask_b32 %2, %2, %4, vcc\n v_cndmask_b32 %3, %3, %4, vcc\n")
KERNEL32(k_mullo, "v_mul_lo_u32 %0, %0, %4\n v_mul_lo_u32 %1, %1, %4\n v_mul_lo_u32 %2, %2, %4\n v_mul_lo_u32 %3, %3, %4\n")
KERNEL32(k_mulhi, "v_mul_hi_u32 %0, %0, %4\n v_mul_hi_u32 %1, %1, %4\n v_mul_hi_u32 %2, %2, %4\n v_mul_hi_u32 %3, %3, %4\n")
KERNEL32(k_alignbit, "v_alignbit_b32 %0, %0, %4, 12\n v_alignbit_b32 %1, %1, %4, 12\n v_alignbit_b32 %2, %2, %4, 12\n v_alignbit_b32 %3, %3, %4, 12\n")
KERNEL32(k_cmp, "v_cmp_eq_u32 vcc, %0, %4\n v_cmp_eq_u32 vcc, %1, %4\n v_cmp_eq_u32 vcc, %2, %4\n v_cmp_eq_u32 vcc, %3, %4\n")
KERNEL64(k_fma64, "v_fma_f64 %0, %0, %4, %4\n v_fma_f64 %1, %1, %4, %4\n v_fma_f64 %2, %2, %4, %4\n v_fma_f64 %3, %3, %4, %4\n")
KERNEL64(k_add64, "v_add_f64 %0, %0, %4\n v_add_f64 %1, %1, %4\n v_add_f64 %2, %2, %4\n v_add_f64 %3, %3, %4\n")
KERNEL64(k_mul64, "v_mul_f64 %0, %0, %4\n v_mul_f64 %1, %1, %4\n v_mul_f64 %2, %2, %4\n v_mul_f64 %3, %3, %4\n")
KERNEL64(k_mov64, "v_mov_b64 %0, %1\n v_mov_b64 %1, %2\n v_mov_b64 %2, %3\n v_mov_b64 %3, %0\n")
__global__ __launch_bounds__(256) void k_mad64(uint32_t* out, unsigned long long* clk) {
    unsigned long long a = threadIdx.x, b = 1, c = 2, d = 3;
    uint32_t m = 0xD2511F53u, x = threadIdx.x * 7u + 1u;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < kIters; ++it) {
        asm volatile(REP16("v_mad_u64_u32 %0, vcc, %4, %5, 0\n v_mad_u64_u32 %1, vcc, %4, %5, 0\n v_mad_u64_u32 %2, vcc, %4, %5, 0\n v_mad_u64_u32 %3, vcc, %4, %5, 0\n")
                     : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(m), "v"(x) : "vcc");
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)(a ^ b ^ c ^ d);
    if (threadIdx.x == 0 && blockIdx.x == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
}
KERNEL64(k_rcp64, "v_rcp_f64 %0, %0\n v_rcp_f64 %1, %1\n v_rcp_f64 %2, %2\n v_rcp_f64 %3, %3\n")
KERNEL64(k_rndne64, "v_rndne_f64 %0, %0\n v_rndne_f64 %1, %1\n v_rndne_f64 %2, %2\n v_rndne_f64 %3, %3\n")
KERNEL64(k_cmp64, "v_cmp_gt_f64 vcc, %0, %4\n v_cmp_gt_f64 vcc, %1, %4\n v_cmp_gt_f64 vcc, %2, %4\n v_cmp_gt_f64 vcc, %3, %4\n")
KERNEL64(k_fma64dep, "v_fma_f64 %0, %0, %4, %4\n v_fma_f64 %0, %0, %4, %4\n v_fma_f64 %0, %0, %4, %4\n v_fma_f64 %0, %0, %4, %4\n")
KERNEL32(k_xordep, "v_xor_b32 %0, %0, %4\n v_xor_b32 %0, %0, %4\n v_xor_b32 %0, %0, %4\n v_xor_b32 %0, %0, %4\n")

// v_cndmask variants: mask in VCC written once per block by a compare, and mask in an SGPR pair
__global__ __launch_bounds__(256) void k_cnd_vcc(uint32_t* out, unsigned long long* clk) {
    uint32_t a = threadIdx.x, b = threadIdx.x * 3u + 1u, c = threadIdx.x * 5u + 2u, d = 7u, m = 0xD2511F53u;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < kIters; ++it) {
        asm volatile("v_cmp_gt_u32 vcc, %0, %4\n s_nop 4\n" REP16("v_cndmask_b32 %0, %0, %4, vcc\n v_cndmask_b32 %1, %1, %4, vcc\n v_cndmask_b32 %2, %2, %4, vcc\n v_cndmask_b32 %3, %3, %4, vcc\n")
                     : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(m) : "vcc");
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = a ^ b ^ c ^ d;
    if (threadIdx.x == 0 && blockIdx.x == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
}
__global__ __launch_bounds__(256) void k_cnd_sgpr(uint32_t* out, unsigned long long* clk) {
    uint32_t a = threadIdx.x, b = threadIdx.x * 3u + 1u, c = threadIdx.x * 5u + 2u, d = 7u, m = 0xD2511F53u;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < kIters; ++it) {
        asm volatile("v_cmp_gt_u32 s[20:21], %0, %4\n s_nop 4\n" REP16("v_cndmask_b32 %0, %0, %4, s[20:21]\n v_cndmask_b32 %1, %1, %4, s[20:21]\n v_cndmask_b32 %2, %2, %4, s[20:21]\n v_cndmask_b32 %3, %3, %4, s[20:21]\n")
                     : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(m) : "s20", "s21");
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = a ^ b ^ c ^ d;
    if (threadIdx.x == 0 && blockIdx.x == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
}
// 32-bit VOP2 with an SGPR operand and with a literal (the forms Philox key xors take)
KERNEL32(k_xor_sgpr, "v_xor_b32 %0, s20, %0\n v_xor_b32 %1, s20, %1\n v_xor_b32 %2, s20, %2\n v_xor_b32 %3, s20, %3\n")
KERNEL32(k_xor_lit, "v_xor_b32 %0, 0x9E3779B9, %0\n v_xor_b32 %1, 0x9E3779B9, %1\n v_xor_b32 %2, 0x9E3779B9, %2\n v_xor_b32 %3, 0x9E3779B9, %3\n")
KERNEL32(k_bitop3, "v_bitop3_b32 %0, %0, %4, %1 bitop3:0x96\n v_bitop3_b32 %1, %1, %4, %2 bitop3:0x96\n v_bitop3_b32 %2, %2, %4, %3 bitop3:0x96\n v_bitop3_b32 %3, %3, %4, %0 bitop3:0x96\n")
KERNEL32(k_bitop3_sgpr, "v_bitop3_b32 %0, %0, %4, s20 bitop3:0x96\n v_bitop3_b32 %1, %1, %4, s20 bitop3:0x96\n v_bitop3_b32 %2, %2, %4, s20 bitop3:0x96\n v_bitop3_b32 %3, %3, %4, s20 bitop3:0x96\n")
KERNEL32(k_lshl_add, "v_lshl_add_u32 %0, %0, 3, %4\n v_lshl_add_u32 %1, %1, 3, %4\n v_lshl_add_u32 %2, %2, 3, %4\n v_lshl_add_u32 %3, %3, 3, %4\n")
KERNEL32(k_lshr, "v_lshrrev_b32 %0, 5, %0\n v_lshrrev_b32 %1, 5, %1\n v_lshrrev_b32 %2, 5, %2\n v_lshrrev_b32 %3, 5, %3\n")
KERNEL64(k_fmac64, "v_fmac_f64 %0, %4, %4\n v_fmac_f64 %1, %4, %4\n v_fmac_f64 %2, %4, %4\n v_fmac_f64 %3, %4, %4\n")
KERNEL64(k_ldexp64, "v_ldexp_f64 %0, %0, 1\n v_ldexp_f64 %1, %1, 1\n v_ldexp_f64 %2, %2, 1\n v_ldexp_f64 %3, %3, 1\n")
KERNEL64(k_rsq64, "v_rsq_f64 %0, %0\n v_rsq_f64 %1, %1\n v_rsq_f64 %2, %2\n v_rsq_f64 %3, %3\n")

__global__ __launch_bounds__(256) void k_mad64_sgpr(uint32_t* out, unsigned long long* clk) {
    unsigned long long a = threadIdx.x, b = 1, c = 2, d = 3;
    uint32_t x = threadIdx.x * 7u + 1u;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < kIters; ++it) {
        asm volatile("s_mov_b32 s20, 0xD2511F53\n" REP16("v_mad_u64_u32 %0, s[22:23], %4, s20, 0\n v_mad_u64_u32 %1, s[22:23], %4, s20, 0\n v_mad_u64_u32 %2, s[22:23], %4, s20, 0\n v_mad_u64_u32 %3, s[22:23], %4, s20, 0\n")
                     : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(x) : "s20", "s22", "s23");
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)(a ^ b ^ c ^ d);
    if (threadIdx.x == 0 && blockIdx.x == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
}
KERNEL64(k_fma64_sgpr, "v_fma_f64 %0, %0, s[20:21], %4\n v_fma_f64 %1, %1, s[20:21], %4\n v_fma_f64 %2, %2, s[20:21], %4\n v_fma_f64 %3, %3, s[20:21], %4\n")
KERNEL64(k_fma64_inl, "v_fma_f64 %0, %0, %4, 1.0\n v_fma_f64 %1, %1, %4, 1.0\n v_fma_f64 %2, %2, %4, 1.0\n v_fma_f64 %3, %3, %4, 1.0\n")
KERNEL32(k_cnd_e64vcc, "v_cndmask_b32_e64 %0, %0, %4, vcc\n v_cndmask_b32_e64 %1, %1, %4, vcc\n v_cndmask_b32_e64 %2, %2, %4, vcc\n v_cndmask_b32_e64 %3, %3, %4, vcc\n")
KERNEL32(k_add_lit, "v_add_u32 %0, 0x9E3779B9, %0\n v_add_u32 %1, 0x9E3779B9, %1\n v_add_u32 %2, 0x9E3779B9, %2\n v_add_u32 %3, 0x9E3779B9, %3\n")
KERNEL32(k_add_inl, "v_add_u32 %0, 1, %0\n v_add_u32 %1, 1, %1\n v_add_u32 %2, 1, %2\n v_add_u32 %3, 1, %3\n")
KERNEL32(k_xor_mix, "v_xor_b32 %0, %0, %4\n v_mul_hi_u32 %1, %1, %4\n v_xor_b32 %2, %2, %4\n v_mul_hi_u32 %3, %3, %4\n")

template <class K>
void run(const char* name, K kern, int waves_per_simd, uint32_t* d_out, unsigned long long* d_clk) {
    const int grid = 256 * waves_per_simd;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    kern<<<grid, 256>>>(d_out, d_clk);
    hipEventRecord(e0);
    kern<<<grid, 256>>>(d_out, d_clk);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    unsigned long long clk[2];
    hipMemcpy(clk, d_clk, sizeof clk, hipMemcpyDeviceToHost);
    const double ghz = (double)clk[0] / ((double)clk[1] * 10.0);
    const double n = (double)kIters * 64.0 * waves_per_simd;
    printf("%-22s waves/SIMD %d  %7.3f ms  clock %.2f GHz  %5.2f cycles per wave-instruction per SIMD\n", name, waves_per_simd, ms,
           ghz, ms * 1e-3 * ghz * 1e9 / n);
}

// Does an instruction cost less when only part of the wave's lanes are enabled?  The same fp64 fma / 32-bit streams
// with EXEC restricted to the low 32, 16 and 1 lanes for the duration of the loop.
#define KERNEL64_EXEC(name, MASK)                                                                     \
    __global__ __launch_bounds__(256) void name(uint32_t* out, unsigned long long* clk) {             \
        double a = 1.0 + threadIdx.x * 1e-9, b = 1.1, c = 1.2, d = 1.3;                               \
        double m = 1.0000001;                                                                         \
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();                                   \
        const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();                               \
        if ((MASK >> (threadIdx.x & 63)) & 1ull) {                                                    \
            for (int it = 0; it < kIters; ++it) {                                                     \
                asm volatile(REP16("v_fma_f64 %0, %0, %4, %4\n v_fma_f64 %1, %1, %4, %4\n v_fma_f64 %2, %2, %4, %4\n v_fma_f64 %3, %3, %4, %4\n") \
                             : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(m));                          \
            }                                                                                         \
        }                                                                                             \
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();                                   \
        const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();                               \
        out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)(a + b + c + d);                       \
        if (threadIdx.x == 0 && blockIdx.x == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }              \
    }
KERNEL64_EXEC(k_fma64_exec32, 0x00000000FFFFFFFFull)
KERNEL64_EXEC(k_fma64_exec16, 0x000000000000FFFFull)
KERNEL64_EXEC(k_fma64_exec1, 0x0000000000000001ull)
KERNEL64_EXEC(k_fma64_exec_hi32, 0xFFFFFFFF00000000ull)
KERNEL64_EXEC(k_fma64_exec_even, 0x5555555555555555ull)
KERNEL64_EXEC(k_fma64_exec2, 0x0000000000000003ull)
KERNEL64_EXEC(k_fma64_exec4, 0x000000000000000Full)
KERNEL64_EXEC(k_fma64_exec8, 0x00000000000000FFull)
KERNEL64_EXEC(k_fma64_exec1_lane37, 0x0000002000000000ull)
KERNEL64_EXEC(k_fma64_exec2_far, 0x0000000100000001ull)
KERNEL64_EXEC(k_fma64_exec12, 0x0000000000000FFFull)
KERNEL64_EXEC(k_fma64_exec9, 0x00000000000001FFull)
KERNEL64_EXEC(k_fma64_exec8_spread, 0x0101010101010101ull)
KERNEL64_EXEC(k_fma64_exec16_spread, 0x1111111111111111ull)
KERNEL64_EXEC(k_fma64_exec15, 0x0000000000007FFFull)
#define KERNEL32_EXEC(name, MASK, INSTR)                                                              \
    __global__ __launch_bounds__(256) void name(uint32_t* out, unsigned long long* clk) {             \
        uint32_t a = threadIdx.x, b = threadIdx.x * 3u + 1u, c = threadIdx.x * 5u + 2u, d = 7u;       \
        uint32_t m = 0xD2511F53u;                                                                     \
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();                                   \
        const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();                               \
        if ((MASK >> (threadIdx.x & 63)) & 1ull) {                                                    \
            for (int it = 0; it < kIters; ++it) {                                                     \
                asm volatile(REP16(INSTR) : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(m));             \
            }                                                                                         \
        }                                                                                             \
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();                                   \
        const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();                               \
        out[blockIdx.x * blockDim.x + threadIdx.x] = a ^ b ^ c ^ d;                                   \
        if (threadIdx.x == 0 && blockIdx.x == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }              \
    }
KERNEL32_EXEC(k_mullo_exec4, 0xFull, "v_mul_lo_u32 %0, %0, %4\n v_mul_lo_u32 %1, %1, %4\n v_mul_lo_u32 %2, %2, %4\n v_mul_lo_u32 %3, %3, %4\n")
KERNEL32_EXEC(k_xor_exec4, 0xFull, "v_xor_b32 %0, %0, %4\n v_xor_b32 %1, %1, %4\n v_xor_b32 %2, %2, %4\n v_xor_b32 %3, %3, %4\n")
KERNEL32_EXEC(k_fma32_exec4, 0xFull, "v_fma_f32 %0, %0, %4, %4\n v_fma_f32 %1, %1, %4, %4\n v_fma_f32 %2, %2, %4, %4\n v_fma_f32 %3, %3, %4, %4\n")
KERNEL32_EXEC(k_fma32_exec64, 0xFFFFFFFFFFFFFFFFull, "v_fma_f32 %0, %0, %4, %4\n v_fma_f32 %1, %1, %4, %4\n v_fma_f32 %2, %2, %4, %4\n v_fma_f32 %3, %3, %4, %4\n")

// Conversions between fp64 and 32-bit integers / fp32 (no dependency chain: the issue rate is what is measured).
#define KERNEL_CVT(name, INSTR, TO, TI, INIT)                                                         \
    __global__ __launch_bounds__(256) void name(uint32_t* out, unsigned long long* clk) {             \
        TO x = 0, y = 0, z = 0, w = 0;                                                                \
        TI a = (TI)(INIT + threadIdx.x), b = (TI)(INIT + 1), c = (TI)(INIT + 2), d = (TI)(INIT + 3);  \
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();                                   \
        const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();                               \
        for (int it = 0; it < kIters; ++it) {                                                         \
            asm volatile(REP16(INSTR " %0, %4\n " INSTR " %1, %5\n " INSTR " %2, %6\n " INSTR " %3, %7\n")              \
                         : "+v"(x), "+v"(y), "+v"(z), "+v"(w) : "v"(a), "v"(b), "v"(c), "v"(d));      \
        }                                                                                             \
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();                                   \
        const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();                               \
        out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)x ^ (uint32_t)y ^ (uint32_t)z ^ (uint32_t)w; \
        if (threadIdx.x == 0 && blockIdx.x == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }              \
    }
KERNEL_CVT(k_cvt_i32_f64, "v_cvt_i32_f64", uint32_t, double, 1.5)
KERNEL_CVT(k_cvt_u32_f64, "v_cvt_u32_f64", uint32_t, double, 1.5)
KERNEL_CVT(k_cvt_f64_i32, "v_cvt_f64_i32", double, uint32_t, 3)
KERNEL_CVT(k_cvt_f64_u32, "v_cvt_f64_u32", double, uint32_t, 3)
KERNEL_CVT(k_cvt_f32_f64, "v_cvt_f32_f64", float, double, 1.5)
KERNEL_CVT(k_cvt_f64_f32, "v_cvt_f64_f32", double, float, 1.5)
KERNEL_CVT(k_floor_f64, "v_floor_f64", double, double, 1.5)
KERNEL_CVT(k_fract_f64, "v_fract_f64", double, double, 1.5)
KERNEL_CVT(k_frexp_mant_f64, "v_frexp_mant_f64", double, double, 1.5)
KERNEL_CVT(k_frexp_exp_f64, "v_frexp_exp_i32_f64", uint32_t, double, 1.5)
KERNEL_CVT(k_sqrt_f64, "v_sqrt_f64", double, double, 1.5)
KERNEL_CVT(k_rcp_f32, "v_rcp_f32", float, float, 1.5)

int main() {
    uint32_t* d_out;
    unsigned long long* d_clk;
    hipMalloc(&d_out, sizeof(uint32_t) * 256 * 8 * 256);
    hipMalloc(&d_clk, 16);
#define RUN(k) run(#k, k, 4, d_out, d_clk)
    RUN(k_xor); RUN(k_add); RUN(k_mov); RUN(k_cnd); RUN(k_alignbit); RUN(k_cmp); RUN(k_mullo); RUN(k_mulhi);
    RUN(k_mad64); RUN(k_fma64); RUN(k_add64); RUN(k_mul64); RUN(k_mov64); RUN(k_rcp64); RUN(k_rndne64); RUN(k_cmp64);
    RUN(k_bitop3); RUN(k_bitop3_sgpr); RUN(k_cnd_vcc); RUN(k_cnd_sgpr); RUN(k_xor_sgpr); RUN(k_xor_lit); RUN(k_lshl_add); RUN(k_lshr); RUN(k_fmac64); RUN(k_ldexp64); RUN(k_rsq64);
    RUN(k_mad64_sgpr); RUN(k_fma64_sgpr); RUN(k_fma64_inl); RUN(k_cnd_e64vcc); RUN(k_add_lit); RUN(k_add_inl); RUN(k_xor_mix);
    run("k_fma64dep (1 wave)", k_fma64dep, 1, d_out, d_clk);
    run("k_fma64dep", k_fma64dep, 4, d_out, d_clk);
    run("k_xordep (1 wave)", k_xordep, 1, d_out, d_clk);
    run("k_xordep", k_xordep, 4, d_out, d_clk);
    run("k_fma64 (1 wave)", k_fma64, 1, d_out, d_clk);
    run("k_fma64 (2 waves)", k_fma64, 2, d_out, d_clk);
    RUN(k_fma64_exec32); RUN(k_fma64_exec16); RUN(k_fma64_exec1); RUN(k_fma64_exec_hi32); RUN(k_fma64_exec_even); RUN(k_fma64_exec2); RUN(k_fma64_exec4); RUN(k_fma64_exec8); RUN(k_fma64_exec1_lane37); RUN(k_fma64_exec2_far); RUN(k_fma64_exec12); RUN(k_fma64_exec9); RUN(k_fma64_exec15); RUN(k_fma64_exec8_spread); RUN(k_fma64_exec16_spread); RUN(k_mullo_exec4); RUN(k_xor_exec4); RUN(k_fma32_exec4); RUN(k_fma32_exec64);
    RUN(k_cvt_i32_f64); RUN(k_cvt_u32_f64); RUN(k_cvt_f64_i32); RUN(k_cvt_f64_u32); RUN(k_cvt_f32_f64); RUN(k_cvt_f64_f32);
    RUN(k_floor_f64); RUN(k_fract_f64); RUN(k_frexp_mant_f64); RUN(k_frexp_exp_f64); RUN(k_sqrt_f64); RUN(k_rcp_f32);
    return 0;
}
