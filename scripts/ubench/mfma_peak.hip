// Micro-benchmark: what the fp32 matrix pipe sustains on this chip (v_mfma_f32_32x32x2_f32), as a function of
// waves per SIMD, a barrier every 64 MFMAs, and operand data (zeros vs random: DVFS).  Build:
//   hipcc --offload-arch=gfx950 -O3 -o mfma_peak mfma_peak.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <bool BARRIER, bool PRIO>
__global__ __launch_bounds__(256) void mfma_loop(const float* __restrict__ in, float* __restrict__ out, int iters, unsigned long long* ticks) {
    if (PRIO) {   // co-resident waves of one SIMD get different issue priorities (HW_ID.wave_id parity)
        const unsigned slot = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 4);
        if (slot & 1) __builtin_amdgcn_s_setprio(2); else __builtin_amdgcn_s_setprio(0);
    }
    const unsigned long long t_begin = __builtin_amdgcn_s_memtime();
    f32x16 acc[4];
    for (int a = 0; a < 4; ++a)
        for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
    // 16 distinct A and 16 distinct B operand registers per lane: every MFMA sees new operand values
    // (operand toggling is what sets the power draw, hence the sustained clock)
    float av[16], bv[16];
    for (int j = 0; j < 16; ++j) {
        av[j] = in[(threadIdx.x * 7 + 64 * j) & 4095];
        bv[j] = in[(threadIdx.x * 3 + 64 * j + 17) & 4095];
    }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[4 * g + j], bv[4 * g + j], acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[4 * g + j], bv[(4 * g + j + 8) & 15], acc[1], 0, 0, 0);
                acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[(4 * g + j + 8) & 15], bv[4 * g + j], acc[2], 0, 0, 0);
                acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[(4 * g + j + 8) & 15], bv[(4 * g + j + 8) & 15], acc[3], 0, 0, 0);
            }
        }
        if (BARRIER) __syncthreads();
    }
    float s = 0.f;
    for (int a = 0; a < 4; ++a)
        for (int r = 0; r < 16; ++r) s += acc[a][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) { ticks[0] = __builtin_amdgcn_s_memtime() - t_begin; }
}

int main() {
    const int iters = 8000;          // 64 MFMAs per iteration per wave
    float *in, *out;
    hipMalloc(&in, 4096 * 4);
    hipMalloc(&out, 256 * 8 * 256 * 4);
    unsigned long long* ticks; hipMalloc(&ticks, 8); unsigned long long hticks = 0;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int data = 1; data < 2; ++data) {
        std::vector<float> h(4096);
        for (auto& v : h) v = data ? (float)rand() / RAND_MAX * 2.f - 1.f : 0.f;
        hipMemcpy(in, h.data(), 4096 * 4, hipMemcpyHostToDevice);
        for (int bar = 0; bar < 2; ++bar)   // 0 free-running, 1 barrier/64 MFMAs, 2 free + slot priority, 3 barrier + slot priority
            for (int bpc = 1; bpc <= 2; ++bpc) {          // blocks per CU -> waves per SIMD
                const int blocks = 256 * bpc;
                for (int rep = 0; rep < 2; ++rep) {
                    hipEventRecord(e0);
                    if (bar == 1) hipLaunchKernelGGL((mfma_loop<true, false>), dim3(blocks), dim3(256), 0, 0, in, out, iters, ticks);
                    else if (bar == 0) hipLaunchKernelGGL((mfma_loop<false, false>), dim3(blocks), dim3(256), 0, 0, in, out, iters, ticks);
                    else if (bar == 2) hipLaunchKernelGGL((mfma_loop<false, true>), dim3(blocks), dim3(256), 0, 0, in, out, iters, ticks);
                    else hipLaunchKernelGGL((mfma_loop<true, true>), dim3(blocks), dim3(256), 0, 0, in, out, iters, ticks);
                    hipEventRecord(e1);
                    hipEventSynchronize(e1);
                    float ms;
                    hipEventElapsedTime(&ms, e0, e1);
                    const double flops = (double)blocks * 4 * iters * 64 * 4096.0;
                    hipMemcpy(&hticks, ticks, 8, hipMemcpyDeviceToHost);
                    if (rep) printf("data=%s barrier=%d waves/SIMD=%d : %.2f ms  %.1f TFLOP/s | block0 wave0: %llu ticks = %.1f ticks/MFMA(own), %.1f ticks/us of kernel time\n", data ? "rand" : "zero", bar, bpc, ms, flops / ms / 1e9, hticks, (double)hticks / (iters * 64.0), hticks / (ms * 1e3));
                }
            }
    }
    return 0;
}
