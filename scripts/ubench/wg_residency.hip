// Do 256 (512) workgroups of 1024 threads with 145 KB of LDS each run side by side on the 256 CUs?  Every workgroup
// records when it started and ended (wall_clock64, 100 MHz) and where it ran (XCC_ID, HW_ID).
// build: hipcc --offload-arch=gfx950 -O3 -o wg_residency wg_residency.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

__global__ void __launch_bounds__(1024) k(unsigned long long* rec, int spin)
{
    extern __shared__ float lds[];
    const unsigned long long t0 = wall_clock64();
    float x = threadIdx.x;
    for (int i = 0; i < spin; i++) { x = x * 1.0001f + 0.5f; lds[(threadIdx.x + i) & 1023] = x; }
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned hw, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        rec[blockIdx.x * 4 + 0] = t0; rec[blockIdx.x * 4 + 1] = wall_clock64(); rec[blockIdx.x * 4 + 2] = hw; rec[blockIdx.x * 4 + 3] = xcc + (unsigned long long)(lds[5] > 1e30f);
    }
}

int main()
{
    for (int wgs : {256, 512}) {
        unsigned long long* rec; if (hipMalloc(&rec, wgs * 32) != hipSuccess) return 1;
        const size_t lds = 145 * 1024;
        if (hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return 2;
        hipLaunchKernelGGL(k, dim3(wgs), dim3(1024), lds, 0, rec, 1000);
        hipLaunchKernelGGL(k, dim3(wgs), dim3(1024), lds, 0, rec, 200000);
        if (hipDeviceSynchronize() != hipSuccess) return 3;
        std::vector<unsigned long long> h(wgs * 4);
        if (hipMemcpy(h.data(), rec, wgs * 32, hipMemcpyDeviceToHost) != hipSuccess) return 4;
        unsigned long long first = ~0ull, last = 0;
        for (int i = 0; i < wgs; i++) { first = std::min(first, h[4 * i]); last = std::max(last, h[4 * i + 1]); }
        int late = 0; std::vector<int> per_xcc(8, 0);
        std::vector<double> starts;
        for (int i = 0; i < wgs; i++) {
            const double s = (h[4 * i] - first) / 100.0, e = (h[4 * i + 1] - first) / 100.0;   // microseconds
            starts.push_back(s);
            if (s > 50.0) late++;
            per_xcc[h[4 * i + 3] & 7]++;
        }
        std::sort(starts.begin(), starts.end());
        printf("%d workgroups: kernel %.1f us; started more than 50 us after the first: %d; start quantiles (us): 50%% %.1f 90%% %.1f 99%% %.1f max %.1f; per XCC:", wgs,
               (last - first) / 100.0, late, starts[wgs / 2], starts[wgs * 9 / 10], starts[wgs * 99 / 100], starts.back());
        for (int x = 0; x < 8; x++) printf(" %d", per_xcc[x]);
        printf("\n");
        // distinct (xcc, se, cu) places used by the first 256
        std::vector<unsigned long long> places;
        for (int i = 0; i < wgs; i++) if ((h[4 * i] - first) / 100.0 <= 50.0) places.push_back(((h[4 * i + 3] & 7) << 32) | (h[4 * i + 2] & 0xffffff00ull));
        std::sort(places.begin(), places.end()); places.erase(std::unique(places.begin(), places.end()), places.end());
        printf("   distinct (XCC, HW_ID without wave/simd bits) among the early ones: %zu\n", places.size());
        (void)hipFree(rec);
    }
    return 0;
}
