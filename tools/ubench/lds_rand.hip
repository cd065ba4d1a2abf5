// LDS operation cost with the access patterns of sk_bucket_kernel (development tool): random slots of a 4096 x 8-byte table,
// two 512-lane blocks per CU (4 waves per SIMD), all lanes or a fraction of them active.
//   hipcc --offload-arch=gfx950 -O3 -o lds_rand tools/ubench/lds_rand.hip && ./lds_rand
// Reports shader cycles per wave-instruction per CU (all 16 waves of the CU issuing the same operation back to back; the address
// arithmetic -- one multiply-add and a mask per operation -- is in the loop of every variant, incl. the empty one).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned long long u64;
typedef unsigned int u32;
#define ITER 2048
template <int OP, int ACTIVE_PCT>
__global__ void __launch_bounds__(512, 4) k(u64 *out, u32 seed)
{
    __shared__ u64 tab[4096 + 1024];                   // 40 KB: two blocks per CU
    __shared__ u32 cnt[4096];
    for (int i = threadIdx.x; i < 4096 + 1024; i += 512) tab[i] = ~0ull;
    for (int i = threadIdx.x; i < 4096; i += 512) cnt[i] = 0;
    __syncthreads();
    u32 x = threadIdx.x * 2654435761u + seed + blockIdx.x;
    u64 acc = 0;
    const bool act = (x >> 8) % 100 < ACTIVE_PCT;
    u64 t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < ITER; ++it) {
        x = x * 1664525u + 1013904223u;
        const u32 s = (x >> 12) & 4095;
        if (OP == 0) { acc += s; }
        if (OP == 1) { if (act) acc += atomicCAS(&tab[s], ~0ull, (u64)x | 1ull << 40); }
        if (OP == 2) { if (act) acc += tab[s]; }
        if (OP == 3) { if (act) atomicAdd(&cnt[s], 1u); }
        if (OP == 4) { if (act) acc += atomicAdd((unsigned long long *)&tab[s], 1ull); }
        if (OP == 5) { if (act) { const ulonglong2 v = *(const ulonglong2 *)&tab[s & ~1u]; acc += v.x + v.y; } }
        if (OP == 6) { if (act) atomicAdd((unsigned long long *)&tab[s], 1ull); }
        if (OP == 7) { if (act) acc += atomicCAS(&tab[threadIdx.x & 4095], ~0ull, (u64)x | 1ull << 40); }       // conflict-free
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    u64 t1 = __builtin_amdgcn_s_memtime();
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 8 + (threadIdx.x >> 6)] = t1 - t0;
    if (acc == 0x123456789ull) out[1 << 20] = 1;
}
template <int OP, int PCT>
static double run(u64 *d, int n_cu)
{
    const int blocks = 2 * n_cu;
    hipLaunchKernelGGL((k<OP, PCT>), dim3(blocks), dim3(512), 0, 0, d, 7u);
    hipDeviceSynchronize();
    hipLaunchKernelGGL((k<OP, PCT>), dim3(blocks), dim3(512), 0, 0, d, 9u);
    hipDeviceSynchronize();
    std::vector<u64> h(blocks * 8);
    hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost);
    double s = 0;
    for (u64 v : h) s += (double)v;
    return s / h.size() / ITER / 16.0;       // cycles per wave-iteration, divided by the CU's 16 waves = per wave-instruction per CU
}
int main()
{
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    u64 *d;
    hipMalloc(&d, ((1 << 20) + 8) * 8);
    const int cu = p.multiProcessorCount;
    printf("%s, %d CUs; cycles per wave-instruction per CU (16 waves per CU issuing), random slots of a 4096-entry table\n", p.name, cu);
    const double base = run<0, 100>(d, cu);
    printf("%-44s %7.2f\n", "loop + address arithmetic only", base);
#define ROW(NAME, OP) printf("%-44s %7.2f %7.2f %7.2f   (100 %% / 35 %% / 10 %% of the lanes active; loop subtracted)\n", NAME, run<OP, 100>(d, cu) - base, run<OP, 35>(d, cu) - base, run<OP, 10>(d, cu) - base);
    ROW("ds_cmpst_rtn_b64 (atomicCAS u64)", 1)
    ROW("ds_cmpst_rtn_b64, conflict-free slots", 7)
    ROW("ds_read_b64", 2)
    ROW("ds_read_b128 (slot pair)", 5)
    ROW("ds_add_u32 (no return)", 3)
    ROW("ds_add_u64 (no return)", 6)
    ROW("ds_add_rtn_u64", 4)
    return 0;
}
