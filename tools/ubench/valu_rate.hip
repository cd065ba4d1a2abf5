// Issue-rate microbenchmark for the integer VALU / LDS instructions the counting kernels are made of (development tool).
//   hipcc --offload-arch=gfx950 -O3 -o valu_rate tools/ubench/valu_rate.hip && ./valu_rate
// For each instruction: one block per CU of 256 / 512 / 1024 lanes (1, 2, 4 waves per SIMD), every wave runs ITER x 32 independent
// copies; reports shader cycles (s_memtime) per wave-instruction per SIMD = cycles / (instructions x waves per SIMD).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <string>
typedef unsigned long long u64;
typedef unsigned int u32;
#define ITER 256
#define REP8(x) x x x x x x x x
#define BODY(NAME, ASM4)                                                                                          \
    __global__ void NAME(u64 *out, u32 seed)                                                                      \
    {                                                                                                             \
        __shared__ u32 lds[16384];                                                                                \
        for (int i = threadIdx.x; i < 16384; i += blockDim.x) lds[i] = i * seed;                                  \
        __syncthreads();                                                                                          \
        u32 a0 = threadIdx.x * seed, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, b0 = seed, b1 = seed * 3, b2 = 5, b3 = 7; \
        u64 q0 = a0, q1 = a1, q2 = a2, q3 = a3;                                                                    \
        u32 ad = (threadIdx.x * 4) & 65535;                                                                        \
        u64 t0 = __builtin_amdgcn_s_memtime();                                                                    \
        for (int it = 0; it < ITER; ++it) {                                                                       \
            REP8(asm volatile(ASM4 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(q0), "+v"(q1), "+v"(q2), "+v"(q3) : "v"(b0), "v"(b1), "v"(b2), "v"(b3), "v"(ad) : "memory", "vcc");) \
        }                                                                                                         \
        asm volatile("s_waitcnt lgkmcnt(0) vmcnt(0)" ::: "memory");                                               \
        u64 t1 = __builtin_amdgcn_s_memtime();                                                                    \
        if ((threadIdx.x & 63) == 0) out[blockIdx.x * 16 + (threadIdx.x >> 6)] = t1 - t0;                         \
        if (a0 + a1 + a2 + a3 + (u32)q0 + (u32)q1 + (u32)q2 + (u32)q3 == 0x12345) out[4096] = 1;                   \
    }
// %0-%3: 32-bit regs, %4-%7: 64-bit regs, %8-%11: 32-bit inputs, %12: LDS address
BODY(k_add, "v_add_u32 %0, %0, %8\n v_add_u32 %1, %1, %9\n v_add_u32 %2, %2, %10\n v_add_u32 %3, %3, %11\n")
BODY(k_xor, "v_xor_b32 %0, %0, %8\n v_xor_b32 %1, %1, %9\n v_xor_b32 %2, %2, %10\n v_xor_b32 %3, %3, %11\n")
BODY(k_mul_lo, "v_mul_lo_u32 %0, %0, %8\n v_mul_lo_u32 %1, %1, %9\n v_mul_lo_u32 %2, %2, %10\n v_mul_lo_u32 %3, %3, %11\n")
BODY(k_mul_u24, "v_mul_u32_u24 %0, %0, %8\n v_mul_u32_u24 %1, %1, %9\n v_mul_u32_u24 %2, %2, %10\n v_mul_u32_u24 %3, %3, %11\n")
BODY(k_mad_u24, "v_mad_u32_u24 %0, %0, %8, %1\n v_mad_u32_u24 %1, %1, %9, %2\n v_mad_u32_u24 %2, %2, %10, %3\n v_mad_u32_u24 %3, %3, %11, %0\n")
BODY(k_shl64, "v_lshlrev_b64 %4, 2, %4\n v_lshlrev_b64 %5, 2, %5\n v_lshlrev_b64 %6, 2, %6\n v_lshlrev_b64 %7, 2, %7\n")
BODY(k_shr64, "v_lshrrev_b64 %4, 2, %4\n v_lshrrev_b64 %5, 2, %5\n v_lshrrev_b64 %6, 2, %6\n v_lshrrev_b64 %7, 2, %7\n")
BODY(k_alignbit, "v_alignbit_b32 %0, %0, %8, 30\n v_alignbit_b32 %1, %1, %9, 30\n v_alignbit_b32 %2, %2, %10, 30\n v_alignbit_b32 %3, %3, %11, 30\n")
BODY(k_lshl_or, "v_lshl_or_b32 %0, %0, 2, %8\n v_lshl_or_b32 %1, %1, 2, %9\n v_lshl_or_b32 %2, %2, 2, %10\n v_lshl_or_b32 %3, %3, 2, %11\n")
BODY(k_and_or, "v_and_or_b32 %0, %0, %8, %9\n v_and_or_b32 %1, %1, %9, %10\n v_and_or_b32 %2, %2, %10, %11\n v_and_or_b32 %3, %3, %11, %8\n")
BODY(k_bfe, "v_bfe_u32 %0, %0, 3, 13\n v_bfe_u32 %1, %1, 3, 13\n v_bfe_u32 %2, %2, 3, 13\n v_bfe_u32 %3, %3, 3, 13\n")
BODY(k_bfrev, "v_bfrev_b32 %0, %0\n v_bfrev_b32 %1, %1\n v_bfrev_b32 %2, %2\n v_bfrev_b32 %3, %3\n")
BODY(k_min, "v_min_u32 %0, %0, %8\n v_min_u32 %1, %1, %9\n v_min_u32 %2, %2, %10\n v_min_u32 %3, %3, %11\n")
BODY(k_min3, "v_min3_u32 %0, %0, %8, %9\n v_min3_u32 %1, %1, %9, %10\n v_min3_u32 %2, %2, %10, %11\n v_min3_u32 %3, %3, %11, %8\n")
BODY(k_cmp64_cnd, "v_cmp_lt_u64 vcc, %4, %5\n v_cndmask_b32 %0, %0, %8, vcc\n v_cmp_lt_u64 vcc, %6, %7\n v_cndmask_b32 %1, %1, %9, vcc\n")
BODY(k_cmp32_cnd, "v_cmp_lt_u32 vcc, %0, %8\n v_cndmask_b32 %1, %1, %9, vcc\n v_cmp_lt_u32 vcc, %2, %10\n v_cndmask_b32 %3, %3, %11, vcc\n")
BODY(k_perm, "v_perm_b32 %0, %0, %8, %9\n v_perm_b32 %1, %1, %9, %10\n v_perm_b32 %2, %2, %10, %11\n v_perm_b32 %3, %3, %11, %8\n")
BODY(k_xad, "v_xad_u32 %0, %0, %8, %9\n v_xad_u32 %1, %1, %9, %10\n v_xad_u32 %2, %2, %10, %11\n v_xad_u32 %3, %3, %11, %8\n")
BODY(k_add3, "v_add3_u32 %0, %0, %8, %9\n v_add3_u32 %1, %1, %9, %10\n v_add3_u32 %2, %2, %10, %11\n v_add3_u32 %3, %3, %11, %8\n")
BODY(k_mbcnt, "v_mbcnt_lo_u32_b32 %0, %8, %0\n v_mbcnt_hi_u32_b32 %1, %9, %1\n v_mbcnt_lo_u32_b32 %2, %10, %2\n v_mbcnt_hi_u32_b32 %3, %11, %3\n")
BODY(k_ds_read32, "ds_read_b32 %0, %12\n ds_read_b32 %1, %12 offset:256\n ds_read_b32 %2, %12 offset:512\n ds_read_b32 %3, %12 offset:768\n")
BODY(k_ds_read64, "ds_read_b64 %4, %12\n ds_read_b64 %5, %12 offset:512\n ds_read_b64 %6, %12 offset:1024\n ds_read_b64 %7, %12 offset:1536\n")
BODY(k_ds_add, "ds_add_u32 %12, %8\n ds_add_u32 %12, %9 offset:256\n ds_add_u32 %12, %10 offset:512\n ds_add_u32 %12, %11 offset:768\n")
BODY(k_ds_add_rtn, "ds_add_rtn_u32 %0, %12, %8\n ds_add_rtn_u32 %1, %12, %9 offset:256\n ds_add_rtn_u32 %2, %12, %10 offset:512\n ds_add_rtn_u32 %3, %12, %11 offset:768\n")
BODY(k_ds_write32, "ds_write_b32 %12, %8\n ds_write_b32 %12, %9 offset:256\n ds_write_b32 %12, %10 offset:512\n ds_write_b32 %12, %11 offset:768\n")
BODY(k_ds_bperm, "ds_bpermute_b32 %0, %12, %8\n ds_bpermute_b32 %1, %12, %9\n ds_bpermute_b32 %2, %12, %10\n ds_bpermute_b32 %3, %12, %11\n")
BODY(k_readlane_mix, "v_add_u32 %0, %0, %8\n s_nop 0\n v_add_u32 %1, %1, %9\n s_nop 0\n")
BODY(k_salu_mix, "v_add_u32 %0, %0, %8\n s_add_u32 s20, s20, 1\n v_add_u32 %1, %1, %9\n s_add_u32 s21, s21, 1\n")

struct K { const char *name; void (*fn)(u64 *, u32); int n_per_asm; };
int main()
{
    hipDeviceProp_t pr;
    hipGetDeviceProperties(&pr, 0);
    const int ncu = pr.multiProcessorCount;
    u64 *d;
    hipMalloc(&d, 8 * 8192);
    K ks[] = {{"v_add_u32", k_add, 4}, {"v_xor_b32", k_xor, 4}, {"v_mul_lo_u32", k_mul_lo, 4}, {"v_mul_u32_u24", k_mul_u24, 4}, {"v_mad_u32_u24", k_mad_u24, 4},
              {"v_lshlrev_b64", k_shl64, 4}, {"v_lshrrev_b64", k_shr64, 4}, {"v_alignbit_b32", k_alignbit, 4}, {"v_lshl_or_b32", k_lshl_or, 4},
              {"v_and_or_b32", k_and_or, 4}, {"v_bfe_u32", k_bfe, 4}, {"v_bfrev_b32", k_bfrev, 4}, {"v_min_u32", k_min, 4}, {"v_min3_u32", k_min3, 4},
              {"v_cmp_lt_u64+cndmask", k_cmp64_cnd, 4}, {"v_cmp_lt_u32+cndmask", k_cmp32_cnd, 4}, {"v_perm_b32", k_perm, 4}, {"v_xad_u32", k_xad, 4},
              {"v_add3_u32", k_add3, 4}, {"v_mbcnt", k_mbcnt, 4}, {"ds_read_b32", k_ds_read32, 4}, {"ds_read_b64", k_ds_read64, 4}, {"ds_add_u32", k_ds_add, 4},
              {"ds_add_rtn_u32", k_ds_add_rtn, 4}, {"ds_write_b32", k_ds_write32, 4}, {"ds_bpermute_b32", k_ds_bperm, 4},
              {"v_add+s_nop", k_readlane_mix, 2}, {"v_add+s_add", k_salu_mix, 2}};
    printf("device %s, %d CUs\n%-24s %10s %10s %10s   (cycles per wave-instruction per SIMD at 1 / 2 / 4 waves per SIMD)\n", pr.name, ncu, "instruction", "1w", "2w", "4w");
    for (auto &k : ks) {
        double res[3];
        int idx = 0;
        for (int threads : {256, 512, 1024}) {
            hipMemset(d, 0, 8 * 8192);
            for (int r = 0; r < 3; ++r) hipLaunchKernelGGL(k.fn, dim3(ncu), dim3(threads), 0, 0, d, 2654435761u);
            hipDeviceSynchronize();
            std::vector<u64> h(16 * ncu);
            hipMemcpy(h.data(), d, 8 * 16 * ncu, hipMemcpyDeviceToHost);
            double sum = 0; int n = 0;
            for (int b = 0; b < ncu; ++b) for (int w = 0; w < threads / 64; ++w) { sum += (double)h[b * 16 + w]; ++n; }
            const double cyc = sum / n;                                 // cycles one wave took for ITER * 8 * n_per_asm instructions
            const int wps = threads / 256;
            res[idx++] = cyc / ((double)ITER * 8 * k.n_per_asm) / wps;
        }
        printf("%-24s %10.2f %10.2f %10.2f\n", k.name, res[0], res[1], res[2]);
    }
    return 0;
}
