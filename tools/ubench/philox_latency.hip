// Is a chain of Philox4x32-10 calls bound by instruction issue or by dependency latency?  One wave runs a
// serial chain of calls (each call's output is the next call's counter); the launch puts W waves on every
// SIMD (W = 1, 2, 4, 8).  Issue-bound: cycles per call per SIMD stay flat once W covers the pipeline.
//   hipcc --offload-arch=gfx950 -O3 -o philox_latency philox_latency.hip && ./philox_latency
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

constexpr int kCalls = 2048;

__device__ __forceinline__ void philox(uint32_t& c0, uint32_t& c1, uint32_t& c2, uint32_t& c3, uint32_t k0, uint32_t k1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        c1 = (uint32_t)p1;
        c3 = (uint32_t)p0;
        c0 = n0;
        c2 = n2;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
}

template <int CHAINS>
__global__ __launch_bounds__(256) void k(uint32_t* out, unsigned long long* clk, uint32_t k0, uint32_t k1) {
    uint32_t c[CHAINS][4];
    for (int j = 0; j < CHAINS; ++j) {
        c[j][0] = threadIdx.x + j;
        c[j][1] = blockIdx.x;
        c[j][2] = 7 * j;
        c[j][3] = 1;
    }
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < kCalls; ++it) {
#pragma unroll
        for (int j = 0; j < CHAINS; ++j) philox(c[j][0], c[j][1], c[j][2], c[j][3], k0, k1);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
    uint32_t s = 0;
    for (int j = 0; j < CHAINS; ++j) s += c[j][0] ^ c[j][1] ^ c[j][2] ^ c[j][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        clk[0] = t1 - t0;
        clk[1] = r1 - r0;
    }
}

template <int CHAINS>
void run(int waves_per_simd, uint32_t* d_out, unsigned long long* d_clk) {
    const int grid = 256 * waves_per_simd;  // one block = 4 waves = one wave per SIMD of a CU
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    k<CHAINS><<<grid, 256>>>(d_out, d_clk, 0x5EED2026u, 0u);
    hipEventRecord(e0);
    k<CHAINS><<<grid, 256>>>(d_out, d_clk, 0x5EED2026u, 0u);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    unsigned long long clk[2];
    hipMemcpy(clk, d_clk, sizeof clk, hipMemcpyDeviceToHost);
    const double ghz = (double)clk[0] / ((double)clk[1] * 10.0);
    const double cycles = ms * 1e-3 * ghz * 1e9;
    const double calls_per_simd = (double)kCalls * CHAINS * waves_per_simd;
    printf("chains/lane %d  waves/SIMD %d: %7.3f ms  clock %.2f GHz  %6.1f cycles per call per SIMD  (one wave: %6.1f cycles per call)\n",
           CHAINS, waves_per_simd, ms, ghz, cycles / calls_per_simd, cycles / ((double)kCalls * CHAINS));
}

int main() {
    uint32_t* d_out;
    unsigned long long* d_clk;
    hipMalloc(&d_out, sizeof(uint32_t) * 256 * 8 * 256);
    hipMalloc(&d_clk, 16);
    for (int w : {1, 2, 3, 4, 5, 6, 8}) run<1>(w, d_out, d_clk);
    for (int w : {1, 2, 4}) run<2>(w, d_out, d_clk);
    return 0;
}
