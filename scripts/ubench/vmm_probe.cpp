// Probe: which uses of the HIP virtual-memory API are sound on this stack (for tests/guard/guard_alloc.cpp)?
//   hipcc --offload-arch=gfx950 -O2 -o vmm_probe vmm_probe.cpp
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("FAIL %s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__global__ void axpy(const float* a, float* o, int n) { int i = blockIdx.x * 256 + threadIdx.x; if (i < n) o[i] = 2.f * a[i] + 1.f; }

static int run(size_t map_off_granules, bool ptr_at_end, bool async_copy, int rounds, int free_mode, bool recommended) {
    hipMemAllocationProp prop; memset(&prop, 0, sizeof(prop));
    prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = 0;
    size_t g = 0; CK(hipMemGetAllocationGranularity(&g, &prop, recommended ? hipMemAllocationGranularityRecommended : hipMemAllocationGranularityMinimum));
    int bad = 0;
    for (int r = 0; r < rounds; ++r) {
        const int n = 1000 + 977 * r;
        const size_t bytes = n * sizeof(float), mapped = (bytes + g - 1) / g * g;
        void* bases[2]; hipMemGenericAllocationHandle_t hs[2]; float* p[2];
        for (int k = 0; k < 2; ++k) {
            CK(hipMemAddressReserve(&bases[k], mapped + 2 * g, g, nullptr, 0));
            CK(hipMemCreate(&hs[k], mapped, &prop, 0));
            char* lo = (char*)bases[k] + map_off_granules * g;
            CK(hipMemMap(lo, mapped, 0, hs[k], 0));
            hipMemAccessDesc acc; memset(&acc, 0, sizeof(acc)); acc.location = prop.location; acc.flags = hipMemAccessFlagsProtReadWrite;
            CK(hipMemSetAccess(lo, mapped, &acc, 1));
            p[k] = (float*)(ptr_at_end ? lo + mapped - (bytes + 15) / 16 * 16 : lo);
        }
        std::vector<float> h(n), o(n);
        for (int i = 0; i < n; ++i) h[i] = (float)(i % 1000) * 0.5f + r;
        if (async_copy) { CK(hipMemcpyAsync(p[0], h.data(), bytes, hipMemcpyHostToDevice, 0)); }
        else CK(hipMemcpy(p[0], h.data(), bytes, hipMemcpyHostToDevice));
        axpy<<<(n + 255) / 256, 256>>>(p[0], p[1], n);
        CK(hipGetLastError());
        if (async_copy) { CK(hipMemcpyAsync(o.data(), p[1], bytes, hipMemcpyDeviceToHost, 0)); CK(hipStreamSynchronize(0)); }
        else CK(hipMemcpy(o.data(), p[1], bytes, hipMemcpyDeviceToHost));
        int wrong = 0;
        for (int i = 0; i < n; ++i) wrong += (o[i] != 2.f * h[i] + 1.f);
        bad += wrong != 0;
        CK(hipDeviceSynchronize());
        for (int k = 0; k < 2; ++k) {
            char* lo = (char*)bases[k] + map_off_granules * g;
            if (free_mode >= 1) { CK(hipMemUnmap(lo, mapped)); CK(hipMemRelease(hs[k])); }
            if (free_mode >= 2) CK(hipMemAddressFree(bases[k], mapped + 2 * g));
        }
    }
    printf("free_mode %d (0 leak, 1 unmap+release, 2 + address free)  granularity %zu  map offset %zu granule(s)  ptr %s  %s copies: %d of %d rounds wrong\n", free_mode, g, map_off_granules,
           ptr_at_end ? "at end" : "at start", async_copy ? "async" : "sync", bad, rounds);
    return 0;
}
int main() {
    for (int rec = 0; rec < 2; ++rec) for (int fm = 0; fm < 3; ++fm) for (int off = 0; off < 2; ++off) for (int end = 0; end < 2; ++end)
        if (run(off, end, false, 40, fm, rec)) return 1;
    return 0;
}
