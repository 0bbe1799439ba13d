// Probe: (1) rounding of the float -> bf16 cast hipcc emits (v_cvt_pk_bf16_f32) against round-to-nearest-even;
// (2) v_mfma_f32_32x32x16_bf16 operand / result layout and accumulation against a host fp64 product.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ void cvt(const float* in, unsigned short* out, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { __bf16 b = (__bf16)in[i]; out[i] = __builtin_bit_cast(unsigned short, b); }
}
// one wave: C[32][32] = A[32][16] * B[16][32], A row-major [i][k], B given as [j][k]
__global__ void mm(const unsigned short* A, const unsigned short* B, float* C) {
    const int l = threadIdx.x, l31 = l & 31, half = l >> 5;
    bf16x8 a, b;
    for (int e = 0; e < 8; ++e) {
        a[e] = __builtin_bit_cast(__bf16, A[l31 * 16 + half * 8 + e]);
        b[e] = __builtin_bit_cast(__bf16, B[l31 * 16 + half * 8 + e]);
    }
    f32x16 acc = {};
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
    for (int r = 0; r < 16; ++r) C[((r & 3) + 8 * (r >> 2) + 4 * half) * 32 + l31] = acc[r];
}
static unsigned short rne(float f) {
    uint32_t u; memcpy(&u, &f, 4);
    u += 0x7fff + ((u >> 16) & 1);
    return (unsigned short)(u >> 16);
}
static float tofloat(unsigned short h) { uint32_t u = (uint32_t)h << 16; float f; memcpy(&f, &u, 4); return f; }
int main() {
    const int n = 1 << 20;
    std::vector<float> h(n);
    for (auto& v : h) v = ((float)rand() / RAND_MAX * 2.f - 1.f) * 3.f;
    float* d; unsigned short* o;
    hipMalloc(&d, n * 4); hipMalloc(&o, n * 2);
    hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(cvt, dim3(n / 256), dim3(256), 0, 0, d, o, n);
    std::vector<unsigned short> r(n);
    hipMemcpy(r.data(), o, n * 2, hipMemcpyDeviceToHost);
    int bad = 0, trunc = 0;
    for (int i = 0; i < n; ++i) { if (r[i] != rne(h[i])) ++bad; uint32_t u; memcpy(&u, &h[i], 4); if (r[i] == (unsigned short)(u >> 16)) ++trunc; }
    printf("cvt: %d of %d differ from round-to-nearest-even; %d equal plain truncation\n", bad, n, trunc);
    std::vector<unsigned short> A(32 * 16), B(32 * 16);
    for (auto& v : A) v = rne((float)rand() / RAND_MAX * 2.f - 1.f);
    for (auto& v : B) v = rne((float)rand() / RAND_MAX * 2.f - 1.f);
    unsigned short *dA, *dB; float* dC;
    hipMalloc(&dA, 1024); hipMalloc(&dB, 1024); hipMalloc(&dC, 4096);
    hipMemcpy(dA, A.data(), 1024, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), 1024, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(mm, dim3(1), dim3(64), 0, 0, dA, dB, dC);
    std::vector<float> C(1024);
    hipMemcpy(C.data(), dC, 4096, hipMemcpyDeviceToHost);
    double maxerr = 0;
    for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) {
        double s = 0; for (int k = 0; k < 16; ++k) s += (double)tofloat(A[i * 16 + k]) * tofloat(B[j * 16 + k]);
        maxerr = fmax(maxerr, fabs(s - C[i * 32 + j]));
    }
    printf("mfma 32x32x16 bf16: max |err| vs fp64 = %.3e\n", maxerr);
    return 0;
}
