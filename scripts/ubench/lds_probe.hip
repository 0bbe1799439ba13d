// Probe of two gfx950 facts the bf16-storage conv kernels (csrc/dpig_conv_bf16.hip) are built on:
//  (1) LDS-DMA through a buffer descriptor (buffer_load_dwordx4 ... lds): what lands in LDS for a lane whose offset is
//      out of the descriptor's range (zeros?  nothing?), and the lane -> LDS address rule (base + lane * 16);
//  (2) ds_read_b64_tr_b16: which lane's memory element ends up in which lane / register slot.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>
typedef short s16x4 __attribute__((ext_vector_type(4)));

__global__ void dma_probe(const unsigned* src, unsigned src_bytes, unsigned* out, int mode) {
    __shared__ __attribute__((aligned(16))) unsigned lds[1024];          // 4 KB
    const int l = threadIdx.x;
    for (int i = l; i < 1024; i += 64) lds[i] = 0xABABABABu;
    __syncthreads();
    // lanes 0..47 in range (lane l reads 16 B at offset 16 * (63 - l): reversed, to show the LDS side is lane-linear),
    // lanes 48..63 out of range
    const unsigned off = (l < 48) ? 16u * (63 - l) : 0x7fffffffu;
    if (mode == 0) {
        __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned*>(src), 0, (int)src_bytes, 0x00020000);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(lds + 256), 16, (int)off, 0, 0, 0);
    } else {
        const unsigned* p = (l < 48) ? src + 4 * (63 - l) : src;          // flat form: every lane must be in range
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)p,
                                         (__attribute__((address_space(3))) void*)(lds + 256), 16, 0, 0);
    }
    __builtin_amdgcn_s_waitcnt(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = l; i < 1024; i += 64) out[i] = lds[i];
}

__global__ void tr_probe(unsigned short* out) {
    __shared__ __attribute__((aligned(16))) unsigned short lds[512];
    const int l = threadIdx.x;
    for (int i = l; i < 512; i += 64) lds[i] = (unsigned short)i;        // element value = its index
    __syncthreads();
    // lane l supplies the address of elements 4l .. 4l+3 (8 bytes)
    s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(lds + 4 * l));
    for (int e = 0; e < 4; ++e) out[l * 4 + e] = (unsigned short)v[e];
}

int main() {
    std::vector<unsigned> h(256);
    for (int i = 0; i < 256; ++i) h[i] = 0x1000u + i;
    unsigned *src, *out;
    hipMalloc(&src, 1024); hipMalloc(&out, 4096);
    hipMemcpy(src, h.data(), 1024, hipMemcpyHostToDevice);
    for (int mode = 0; mode < 2; ++mode) {
        hipLaunchKernelGGL(dma_probe, dim3(1), dim3(64), 0, 0, src, 1024u, out, mode);
        std::vector<unsigned> r(1024);
        hipMemcpy(r.data(), out, 4096, hipMemcpyDeviceToHost);
        printf("== LDS-DMA mode %d (%s) ==\n", mode, mode == 0 ? "raw_buffer_load_lds, lanes 48..63 out of range" : "global_load_lds");
        int before = 0, after = 0;
        for (int i = 0; i < 256; ++i) before += r[i] != 0xABABABABu;
        for (int i = 512; i < 1024; ++i) after += r[i] != 0xABABABABu;
        printf("words changed outside the 1 KB target: %d before, %d after\n", before, after);
        for (int l = 0; l < 64; l += (l < 46 ? 15 : 1)) {
            printf("lane %2d slot: %08x %08x %08x %08x\n", l, r[256 + 4 * l], r[256 + 4 * l + 1], r[256 + 4 * l + 2], r[256 + 4 * l + 3]);
        }
    }
    unsigned short* o2;
    hipMalloc(&o2, 512);
    hipLaunchKernelGGL(tr_probe, dim3(1), dim3(64), 0, 0, o2);
    std::vector<unsigned short> t(256);
    hipMemcpy(t.data(), o2, 512, hipMemcpyDeviceToHost);
    printf("== ds_read_b64_tr_b16: lane l supplied elements 4l..4l+3; lane: received element indices ==\n");
    for (int l = 0; l < 64; ++l) printf("lane %2d: %3d %3d %3d %3d\n", l, t[4 * l], t[4 * l + 1], t[4 * l + 2], t[4 * l + 3]);
    return 0;
}
