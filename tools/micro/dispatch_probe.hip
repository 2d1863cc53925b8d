// Where does the hardware put the workgroups of a GEMM-shaped launch?  (round 4, diagnostic only)
// 256 threads, 72 KB of dynamic LDS (two workgroups per CU, as k_gemm_f16x3t16): every workgroup records its XCC, its CU
// (HW_ID: cu_id [11:8], sh_id [12], se_id [15:13]) and its start time, then stays resident for ~30 us so that the whole first
// round is placed before any slot frees up.  Prints, per XCD-local index q = blockIdx.x >> 3, the CU key and the start offset.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <map>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ __launch_bounds__(256, 2) void k_probe(unsigned *out, unsigned long long *t0) {
    extern __shared__ char smem[];
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    const unsigned long long s = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) {
        out[2 * blockIdx.x] = hw; out[2 * blockIdx.x + 1] = xcc; t0[blockIdx.x] = s;
        smem[0] = 1;
    }
    while (__builtin_amdgcn_s_memrealtime() - s < 3000ull) __builtin_amdgcn_s_sleep(16);    // 30 us
}

int main() {
    const int grid = 1024;
    unsigned *out; unsigned long long *t0;
    CHECK(hipMalloc(&out, grid * 8)); CHECK(hipMalloc(&t0, grid * 8));
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_probe), hipFuncAttributeMaxDynamicSharedMemorySize, 73728));
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(k_probe, dim3(grid), dim3(256), 73728, 0, out, t0);
        CHECK(hipDeviceSynchronize());
    }
    std::vector<unsigned> h(2 * grid); std::vector<unsigned long long> ht(grid);
    CHECK(hipMemcpy(h.data(), out, grid * 8, hipMemcpyDeviceToHost));
    CHECK(hipMemcpy(ht.data(), t0, grid * 8, hipMemcpyDeviceToHost));
    unsigned long long tmin = ~0ull;
    for (auto v : ht) tmin = v < tmin ? v : tmin;
    // XCD 0's workgroups in launch order
    for (int x = 0; x < 2; ++x) {
        printf("blockIdx %% 8 == %d: q -> (xcc, se, sh, cu, start in 10-ns ticks)\n", x);
        std::map<unsigned, std::vector<int>> by_cu;
        for (int b = x; b < grid; b += 8) {
            const unsigned hw = h[2 * b], xcc = h[2 * b + 1] & 15;
            const unsigned cu = (hw >> 8) & 15, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
            if ((b >> 3) < 72) printf("  q %3d: xcc %u se %u sh %u cu %2u  t %llu\n", b >> 3, xcc, se, sh, cu, ht[b] - tmin);
            by_cu[(xcc << 12) | (se << 8) | (sh << 4) | cu].push_back(b >> 3);
        }
        printf("  residents per CU (q values):");
        for (auto &kv : by_cu) { printf(" ["); for (int q : kv.second) printf("%d ", q); printf("]"); }
        printf("\n");
    }
    return 0;
}
