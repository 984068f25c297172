// Issue-rate microbenchmark for the VALU instructions that dominate the walk kernel (gfx950).
// Each kernel runs a long unrolled stream of one instruction with 8 independent chains per lane,
// 4 waves per SIMD on every CU; reports cycles per wave-instruction per SIMD from the wall time at the
// measured clock (s_memtime / s_memrealtime).
//   hipcc --offload-arch=gfx950 -O3 -o valu_rates valu_rates.hip && ./valu_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

constexpr int kIters = 4096;
constexpr int kChains = 8;

template <int OP>
__global__ __launch_bounds__(256) void k(uint64_t* out, unsigned long long* clk) {
    uint32_t a[kChains];
    uint64_t p[kChains];
    double d[kChains];
    for (int i = 0; i < kChains; ++i) {
        a[i] = threadIdx.x * 2654435761u + i;
        p[i] = a[i];
        d[i] = 1.0 + a[i] * 1e-9;
    }
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < kIters; ++it) {
#pragma unroll
        for (int i = 0; i < kChains; ++i) {
            if (OP == 0) p[i] = (uint64_t)(uint32_t)p[i] * 0xD2511F53u + (p[i] >> 32);          // v_mad_u64_u32
            if (OP == 1) d[i] = __builtin_fma(d[i], 1.0000001, 1e-9);                             // v_fma_f64
            if (OP == 2) a[i] = a[i] ^ (a[i] >> 3);                                               // v_lshr + v_xor
            if (OP == 3) a[i] = a[i] * 0xCD9E8D57u + 1u;                                          // v_mul_lo_u32 (+add)
            if (OP == 4) a[i] = __umulhi(a[i], 0xCD9E8D57u) + 7u;                                 // v_mul_hi_u32 (+add)
            if (OP == 5) d[i] = d[i] + 1e-9;                                                      // v_add_f64
            if (OP == 6) a[i] = __mul24((int)(a[i] & 0xFFFFFF), 0x9E3779) + 3u;                   // v_mad_u32_u24
            if (OP == 7) { d[i] = (double)a[i]; a[i] = (uint32_t)__double2hiint(d[i]); }          // v_cvt_f64_u32
            if (OP == 8) d[i] = __builtin_amdgcn_ldexp(d[i], (int)(a[i] & 1u));                    // v_ldexp_f64 (+v_and hoisted)
            if (OP == 9) d[i] = __builtin_amdgcn_rcp(d[i]);                                       // v_rcp_f64
            if (OP == 10) d[i] = __builtin_amdgcn_rsq(d[i]);                                      // v_rsq_f64
            if (OP == 11) d[i] = 1.0000001 / d[i];                                                // IEEE division sequence
            if (OP == 12) d[i] = __builtin_sqrt(d[i]) + 1.0;                                      // IEEE sqrt sequence + add
            if (OP == 13) { a[i] = (uint32_t)(int)d[i]; d[i] = __hiloint2double((int)(a[i] | 0x3ff00000u), (int)a[i]); }  // v_cvt_i32_f64 + or
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
    uint64_t s = 0;
    for (int i = 0; i < kChains; ++i) s += p[i] + a[i] + (uint64_t)d[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
}

template <int OP>
void run(const char* name, int instr_per_step, uint64_t* d_out, unsigned long long* d_clk) {
    const int grid = 256 * 4;  // 4 blocks of 4 waves per CU -> 4 waves per SIMD
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    k<OP><<<grid, 256>>>(d_out, d_clk);
    hipEventRecord(e0);
    k<OP><<<grid, 256>>>(d_out, d_clk);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long clk[2]; hipMemcpy(clk, d_clk, sizeof clk, hipMemcpyDeviceToHost);
    const double ghz = (double)clk[0] / ((double)clk[1] * 10.0) ;  // memrealtime ticks at 100 MHz
    const double wave_instr_per_simd = (double)kIters * kChains * instr_per_step * 4.0;  // 4 waves per SIMD
    const double cycles = ms * 1e-3 * ghz * 1e9;
    printf("%-28s %8.3f ms  clock %.2f GHz  %.2f cycles per wave-instruction per SIMD (counting %d instr/step)\n", name, ms, ghz,
           cycles / wave_instr_per_simd, instr_per_step);
}

int main() {
    uint64_t* d_out; unsigned long long* d_clk;
    hipMalloc(&d_out, sizeof(uint64_t) * 256 * 4 * 256);
    hipMalloc(&d_clk, 16);
    run<0>("v_mad_u64_u32", 1, d_out, d_clk);
    run<1>("v_fma_f64", 1, d_out, d_clk);
    run<5>("v_add_f64", 1, d_out, d_clk);
    run<2>("v_lshrrev + v_xor", 2, d_out, d_clk);
    run<3>("v_mul_lo_u32 + v_add", 2, d_out, d_clk);
    run<4>("v_mul_hi_u32 + v_add", 2, d_out, d_clk);
    run<6>("v_mul_u32_u24 + v_add", 2, d_out, d_clk);
    run<7>("v_cvt_f64_u32", 1, d_out, d_clk);
    run<8>("v_ldexp_f64", 1, d_out, d_clk);
    run<9>("v_rcp_f64", 1, d_out, d_clk);
    run<10>("v_rsq_f64", 1, d_out, d_clk);
    run<11>("fp64 division (sequence)", 1, d_out, d_clk);
    run<12>("fp64 sqrt (sequence) + add", 1, d_out, d_clk);
    run<13>("v_cvt_i32_f64 + v_or", 2, d_out, d_clk);
    return 0;
}
