// Probe: HW_ID / XCC_ID of every wave of a 512-workgroup launch with 73.7 KB LDS per workgroup (2 per CU).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <map>
#include <vector>
#include <tuple>
__global__ __launch_bounds__(256) void probe(unsigned* out) {
    __shared__ float big[18432];
    big[threadIdx.x] = 1.f;
    __syncthreads();
    const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);     // HW_REG_HW_ID, all 32 bits
    const unsigned xcc = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20);    // HW_REG_XCC_ID[3:0]
    const unsigned lds = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 6);    // HW_REG_LDS_ALLOC
    // keep the block alive long enough for all 512 to be co-resident
    float s = big[(threadIdx.x * 7) & 255];
    for (int i = 0; i < 20000; ++i) s = s * 1.0001f + 0.5f;
    if ((threadIdx.x & 63) == 0) {
        unsigned* o = out + (blockIdx.x * 4 + (threadIdx.x >> 6)) * 4;
        o[0] = hw; o[1] = xcc; o[2] = lds; o[3] = (unsigned)s;
    }
}
int main() {
    const int blocks = 512;
    unsigned* d; hipMalloc(&d, blocks * 16 * 4);
    hipLaunchKernelGGL(probe, dim3(blocks), dim3(256), 0, 0, d);
    std::vector<unsigned> h(blocks * 16);
    hipMemcpy(h.data(), d, blocks * 16 * 4, hipMemcpyDeviceToHost);
    std::map<std::tuple<unsigned, unsigned, unsigned, unsigned, unsigned>, std::vector<unsigned>> m;   // (xcc,se,sh,cu,simd) -> wave ids
    std::map<unsigned, int> ldsbase;
    for (int b = 0; b < blocks; ++b)
        for (int w = 0; w < 4; ++w) {
            const unsigned hw = h[(b * 4 + w) * 4], xcc = h[(b * 4 + w) * 4 + 1], lds = h[(b * 4 + w) * 4 + 2];
            m[{xcc, (hw >> 13) & 7, (hw >> 12) & 1, (hw >> 8) & 15, (hw >> 4) & 3}].push_back(hw & 15);
            ldsbase[lds]++;
            if (b < 4) printf("block %d wave %d: hw=%08x wave_id=%u simd=%u cu=%u sh=%u se=%u xcc=%u lds_alloc=%08x\n", b, w, hw, hw & 15, (hw >> 4) & 3, (hw >> 8) & 15, (hw >> 12) & 1, (hw >> 13) & 7, xcc, lds);
        }
    std::map<std::vector<unsigned>, int> pat;
    for (auto& kv : m) pat[kv.second]++;
    printf("distinct SIMDs seen: %zu\n", m.size());
    for (auto& kv : pat) { printf("wave-id set {"); for (auto v : kv.first) printf("%u ", v); printf("} on %d SIMDs\n", kv.second); }
    for (auto& kv : ldsbase) printf("lds_alloc %08x : %d waves\n", kv.first, kv.second);
    return 0;
}
