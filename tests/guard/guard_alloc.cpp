// Guard-page device allocator (TEST TOOL, not product): plugged into torch through
// torch.cuda.memory.CUDAPluggableAllocator so that EVERY device allocation of a process -- kernel inputs, outputs, workspaces, the
// flat parameter / gradient / Adam buffers -- is its own virtual-memory reservation
//
//      [ unmapped granule | mapped, size rounded up to the granularity | unmapped granule ]
//
// with the tensor placed so that its LAST 16-byte piece is the last mapped piece (DPIG_GUARD_MODE=hi, default) or its first byte is
// the first mapped byte (=lo).  A kernel that reads or writes past the end (hi) or before the start (lo) of any operand takes a
// "Memory access fault by GPU" at once, in one process, every time -- the ordinary allocator maps whole 2-MB+ segments, so the same
// access lands in a neighbouring tensor and nothing is seen.  With AMD_SERIALIZE_KERNEL=3 the fault arrives while
// the host is still inside the launching call; faulthandler then names the C-ABI entry point.
//
//   DPIG_GUARD_MODE  hi | lo
//   DPIG_GUARD_FILL  byte value (0..255) the mapped range is filled with before it is handed out; 255 = fp32 / bf16 NaNs, so a kernel
//                    that consumes memory it was told to overwrite (beta = 0 first-touch contracts) shows up as NaN; unset = no fill
//   DPIG_GUARD_LOG   1 = one line per allocation on stderr
//
// Frees synchronise the device first (an unmap under a running kernel would itself fault) and keep the address range reserved
// (never reused: use-after-free faults too).  Build: __graft_entry__.build_guard().
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <unordered_map>

namespace {

struct Rec {
    void* base;
    size_t reserved, mapped;
    hipMemGenericAllocationHandle_t handle;
};

std::mutex g_mu;
std::unordered_map<void*, Rec> g_live;
size_t g_gran = 0;
long g_count = 0, g_bytes = 0;

void die(const char* what, hipError_t e) {
    std::fprintf(stderr, "[guard_alloc] %s failed: %s\n", what, hipGetErrorString(e));
    std::abort();
}
#define GCHECK(call)                                   \
    do {                                               \
        hipError_t e_ = (call);                        \
        if (e_ != hipSuccess) die(#call, e_);          \
    } while (0)

}  // namespace

extern "C" void* dpig_guard_alloc(ssize_t size, int device, hipStream_t stream) {
    if (size <= 0) size = 1;
    std::lock_guard<std::mutex> lk(g_mu);
    hipMemAllocationProp prop;
    std::memset(&prop, 0, sizeof(prop));
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = device;
    if (!g_gran) GCHECK(hipMemGetAllocationGranularity(&g_gran, &prop, hipMemAllocationGranularityMinimum));
    const size_t g = g_gran;
    Rec r;
    r.mapped = ((size_t)size + g - 1) / g * g;
    r.reserved = r.mapped + 2 * g;
    GCHECK(hipMemAddressReserve(&r.base, r.reserved, g, nullptr, 0));
    GCHECK(hipMemCreate(&r.handle, r.mapped, &prop, 0));
    char* lo = (char*)r.base + g;
    GCHECK(hipMemMap(lo, r.mapped, 0, r.handle, 0));
    hipMemAccessDesc acc;
    std::memset(&acc, 0, sizeof(acc));
    acc.location = prop.location;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    GCHECK(hipMemSetAccess(lo, r.mapped, &acc, 1));
    const char* fill = std::getenv("DPIG_GUARD_FILL");
    if (fill && *fill) {
        GCHECK(hipMemsetAsync(lo, std::atoi(fill) & 255, r.mapped, stream));
    }
    const char* mode = std::getenv("DPIG_GUARD_MODE");
    const bool at_start = mode && std::strcmp(mode, "lo") == 0;
    const size_t sz16 = ((size_t)size + 15) / 16 * 16;
    void* p = at_start ? (void*)lo : (void*)(lo + r.mapped - sz16);
    g_live[p] = r;
    ++g_count;
    g_bytes += (long)r.mapped;
    const char* lg = std::getenv("DPIG_GUARD_LOG");
    if (lg && *lg == '1')
        std::fprintf(stderr, "[guard_alloc] #%ld %zd B -> %p (mapped %p..%p)\n", g_count, size, p, (void*)lo, (void*)(lo + r.mapped));
    return p;
}

extern "C" void dpig_guard_free(void* ptr, ssize_t size, int device, hipStream_t stream) {
    (void)size; (void)device; (void)stream;
    if (!ptr) return;
    Rec r;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        auto it = g_live.find(ptr);
        if (it == g_live.end()) {
            std::fprintf(stderr, "[guard_alloc] free of unknown pointer %p\n", ptr);
            std::abort();
        }
        r = it->second;
        g_live.erase(it);
        g_bytes -= (long)r.mapped;
    }
    (void)hipDeviceSynchronize();          // nothing may still be running on this range
    char* lo = (char*)r.base + g_gran;
    GCHECK(hipMemUnmap(lo, r.mapped));
    GCHECK(hipMemRelease(r.handle));
    // The address range is deliberately NOT returned (no hipMemAddressFree): on this stack (ROCm 7.2, gfx950) a range that is freed and
    // handed out again by a later hipMemAddressReserve reads and writes wrong data (scripts/ubench/vmm_probe.cpp: 34-40 of 40 rounds
    // wrong with the address free, 0 of 40 without it).  A 48-bit address space does not run out in a test process, and a side effect
    // worth having: a freed tensor's addresses stay unmapped for good, so a kernel launched with a stale pointer faults as well.
}

extern "C" long dpig_guard_live_bytes() { return g_bytes; }
extern "C" long dpig_guard_alloc_count() { return g_count; }
extern "C" size_t dpig_guard_granularity() { return g_gran; }
