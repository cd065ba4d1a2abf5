// development: what does device memory cost to get?  hipMalloc / hipFree of large buffers (one call, several threads, first time and again),
// and the same range built from smaller physical allocations mapped into one reserved address range (hipMemCreate / hipMemMap).
// build: hipcc --offload-arch=gfx950 -O2 -o tools/ubench/alloc_rate tools/ubench/alloc_rate.hip -pthread
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <thread>
#include <vector>

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void stream_kernel(const uint4 *__restrict__ a, uint4 *__restrict__ b, size_t n)
{
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) { uint4 v = a[i]; v.x += 1; b[i] = v; }
}
// random 16-byte reads over the whole range (address translation: what page size did the mapping get?)
__global__ void gather_kernel(const uint4 *__restrict__ a, unsigned *__restrict__ out, size_t n, size_t per_lane)
{
    size_t x = (blockIdx.x * (size_t)blockDim.x + threadIdx.x) * 0x9E3779B97F4A7C15ull;
    unsigned acc = 0;
    for (size_t i = 0; i < per_lane; ++i) { x = x * 6364136223846793005ull + 1442695040888963407ull; acc += a[(x >> 20) % n].x; }
    if (acc == 0x12345678u) out[0] = acc;
}

static int bandwidth(const char *what, char *base, size_t bytes)
{
    const size_t half = bytes / 2, n = half / 16;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    stream_kernel<<<256 * 16, 256>>>((const uint4 *)base, (uint4 *)(base + half), n);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int i = 0; i < 3; ++i) stream_kernel<<<256 * 16, 256>>>((const uint4 *)base, (uint4 *)(base + half), n);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    printf("%s: stream copy of %.1f GB: %.2f TB/s", what, half / 1e9, 3.0 * 2.0 * half / (ms * 1e-3) / 1e12);
    unsigned *out = nullptr;
    CK(hipMalloc(&out, 4));
    gather_kernel<<<256 * 8, 256>>>((const uint4 *)base, out, bytes / 16, 64);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    gather_kernel<<<256 * 8, 256>>>((const uint4 *)base, out, bytes / 16, 256);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    CK(hipEventElapsedTime(&ms, e0, e1));
    printf(", random 16-byte reads: %.2f G/s\n", 256.0 * 8 * 256 * 256 / (ms * 1e-3) / 1e9);
    CK(hipFree(out));
    return 0;
}

int main()
{
    (void)hipFree(nullptr);
    const size_t GB = 1ull << 30;
    for (size_t gb : {32, 36, 40, 44, 48, 56, 64, 100, 160}) {
        void *p = nullptr;
        double t0 = now();
        hipError_t e = hipMalloc(&p, gb * GB);
        double t1 = now();
        (void)hipFree(p);
        double t2 = now();
        printf("hipMalloc %3zu GB: %.3f s (%s), hipFree %.3f s\n", gb, t1 - t0, hipGetErrorString(e), t2 - t1);
        fflush(stdout);
    }
    {
        void *p[8];
        double t0 = now();
        for (int i = 0; i < 8; ++i) CK(hipMalloc(&p[i], 25 * GB));
        double t1 = now();
        for (int i = 0; i < 8; ++i) CK(hipFree(p[i]));
        printf("8 x 25 GB, one thread: hipMalloc %.3f s, hipFree %.3f s\n", t1 - t0, now() - t1);
    }
    {   // one address range, many physical pieces
        const size_t total = 120 * GB, piece = 8 * GB;
        hipMemAllocationProp prop = {};
        prop.type = hipMemAllocationTypePinned;
        prop.location.type = hipMemLocationTypeDevice;
        prop.location.id = 0;
        size_t gran = 0;
        CK(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended));
        printf("recommended granularity %zu bytes\n", gran);
        double t0 = now();
        void *va = nullptr;
        CK(hipMemAddressReserve(&va, total, 0, nullptr, 0));
        std::vector<hipMemGenericAllocationHandle_t> hs;
        for (size_t o = 0; o < total; o += piece) {
            hipMemGenericAllocationHandle_t h;
            CK(hipMemCreate(&h, piece, &prop, 0));
            CK(hipMemMap((char *)va + o, piece, 0, h, 0));
            hs.push_back(h);
        }
        hipMemAccessDesc acc = {};
        acc.location = prop.location;
        acc.flags = hipMemAccessFlagsProtReadWrite;
        CK(hipMemSetAccess(va, total, &acc, 1));
        double t1 = now();
        printf("120 GB as 15 mapped pieces of 8 GB: %.3f s\n", t1 - t0);
        if (bandwidth("mapped pieces", (char *)va, 32 * GB)) return 1;
        if (bandwidth("mapped pieces, across a piece boundary", (char *)va + 4 * GB, 32 * GB)) return 1;
        double t2 = now();
        CK(hipMemUnmap(va, total));
        for (auto h : hs) CK(hipMemRelease(h));
        CK(hipMemAddressFree(va, total));
        printf("unmap + release: %.3f s\n", now() - t2);
    }
    {
        void *p = nullptr;
        CK(hipMalloc(&p, 32 * GB));
        if (bandwidth("hipMalloc", (char *)p, 32 * GB)) return 1;
        CK(hipFree(p));
    }
    return 0;
}
