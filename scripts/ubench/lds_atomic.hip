// Microbenchmark: rate of float additions into LDS rows of 64 floats by 16 wavefronts of one workgroup per CU, 16 lanes per
// row (4 rows per wave-instruction), in the access shapes the owner mode of the update kernel could use.
// build: hipcc --offload-arch=gfx950 -O3 -o lds_atomic lds_atomic.hip ; run: ./lds_atomic
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

constexpr int ROWS = 391, W = 64, ITERS = 20000;

template <int MODE>
__global__ void __launch_bounds__(1024) k(float* out, unsigned seed)
{
    __shared__ __align__(16) float acc[ROWS * W + ROWS];
    for (int i = threadIdx.x; i < ROWS; i += 1024) acc[ROWS * W + i] = 0;
    for (int i = threadIdx.x; i < ROWS * W; i += 1024) acc[i] = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63, g = lane & 15, q = lane >> 4;
    unsigned s = seed ^ (blockIdx.x * 977u + (threadIdx.x >> 4) * 7919u);
    float v0 = 1.f, v1 = 2.f, v2 = 3.f, v3 = 4.f;
    for (int it = 0; it < ITERS; it++) {
        s = s * 1664525u + 1013904223u;
        const int row = (s >> 8) % ROWS;      // same for the 16 lanes of a group
        float* a = acc + row * W;
        if (MODE == 0) {           // 4 ds_add_f32, lane holds 4g..4g+3, all groups on the same 16 banks per instruction
            atomicAdd(a + 4 * g + 0, v0); atomicAdd(a + 4 * g + 1, v1); atomicAdd(a + 4 * g + 2, v2); atomicAdd(a + 4 * g + 3, v3);
        } else if (MODE == 1) {    // rotated by group: conflict-free banks
            atomicAdd(a + 4 * g + ((0 + q) & 3), v0); atomicAdd(a + 4 * g + ((1 + q) & 3), v1);
            atomicAdd(a + 4 * g + ((2 + q) & 3), v2); atomicAdd(a + 4 * g + ((3 + q) & 3), v3);
        } else if (MODE == 2) {    // non-atomic read-modify-write, 16 bytes per lane
            float4 x = *reinterpret_cast<float4*>(a + 4 * g);
            x.x += v0; x.y += v1; x.z += v2; x.w += v3;
            *reinterpret_cast<float4*>(a + 4 * g) = x;
        } else if (MODE == 3) {    // plain 16-byte store
            *reinterpret_cast<float4*>(a + 4 * g) = make_float4(v0, v1, v2, v3);
        } else if (MODE == 4) {    // one ds_add_f32 per lane only (a quarter of the data)
            atomicAdd(a + 4 * g + (q & 3), v0);
        } else if (MODE == 6) {    // fixed point: 4 ds_add_u32
            int* ia = (int*)a;
            atomicAdd(ia + 4 * g + 0, __float2int_rn(v0 * 67108864.f)); atomicAdd(ia + 4 * g + 1, __float2int_rn(v1 * 67108864.f));
            atomicAdd(ia + 4 * g + 2, __float2int_rn(v2 * 67108864.f)); atomicAdd(ia + 4 * g + 3, __float2int_rn(v3 * 67108864.f));
        } else if (MODE == 7) {    // fixed point, rotated banks
            int* ia = (int*)a;
            atomicAdd(ia + 4 * g + ((0 + q) & 3), __float2int_rn(v0 * 67108864.f)); atomicAdd(ia + 4 * g + ((1 + q) & 3), __float2int_rn(v1 * 67108864.f));
            atomicAdd(ia + 4 * g + ((2 + q) & 3), __float2int_rn(v2 * 67108864.f)); atomicAdd(ia + 4 * g + ((3 + q) & 3), __float2int_rn(v3 * 67108864.f));
        } else if (MODE == 8) {    // 2 ds_add_u64
            unsigned long long* la = (unsigned long long*)a;
            atomicAdd(la + 2 * g + 0, (unsigned long long)__float2int_rn(v0 * 67108864.f)); atomicAdd(la + 2 * g + 1, (unsigned long long)__float2int_rn(v1 * 67108864.f));
        } else if (MODE == 9) {    // try-lock per row (integer compare-and-swap by the group's first lane) + 16-byte read-modify-write
            int* lock = (int*)(acc + ROWS * W) + row;
            bool done = false;
            while (__ballot(!done)) {
                int got = 1;
                if (!done && g == 0) got = atomicCAS(lock, 0, 1);
                got = __shfl(got, 0, 16);
                if (!done && got == 0) {
                    float4 x = *reinterpret_cast<float4*>(a + 4 * g);
                    x.x += v0; x.y += v1; x.z += v2; x.w += v3;
                    *reinterpret_cast<float4*>(a + 4 * g) = x;
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                    if (g == 0) atomicExch(lock, 0);
                    done = true;
                }
            }
        } else if (MODE == 5) {    // 64-bit packed? two ds_add_f32 on consecutive lanes' dwords: lane -> dword (lane of 64 in row)
            atomicAdd(a + lane, v0);   // all 64 lanes on one row (the row of group 0)
        }
        v0 += 1e-3f;
    }
    __syncthreads();
    float sum = 0;
    for (int i = threadIdx.x; i < ROWS * W; i += 1024) sum += acc[i];
    out[blockIdx.x * 1024 + threadIdx.x] = sum;
}

template <int MODE> void run(const char* name, float* out)
{
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(1024), 0, 0, out, 1u);
    hipEventRecord(a, 0);
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(1024), 0, 0, out, 2u);
    hipEventRecord(b, 0);
    hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    // row additions (64 floats) per second chip-wide
    const double rows = 256.0 * 1024 / 16 * ITERS * (MODE == 4 ? 0.25 : MODE == 5 ? 0.25 : 1.0);
    printf("%-28s %8.3f ms  %7.2f G row-adds/s  (%.1f ms for 97.9M rows)\n", name, ms, rows / ms / 1e6, 97.9e6 / (rows / ms));
}

int main()
{
    float* out; if (hipMalloc(&out, 256 * 1024 * 4) != hipSuccess) return 1;
    run<0>("ds_add x4 (same banks)", out);
    run<1>("ds_add x4 (rotated)", out);
    run<2>("read+add+write b128", out);
    run<3>("write b128", out);
    run<4>("ds_add x1", out);
    run<5>("ds_add x1 whole wave one row", out);
    run<6>("ds_add_u32 x4 fixed point", out);
    run<7>("ds_add_u32 x4 rotated", out);
    run<8>("ds_add_u64 x2 (half data)", out);
    run<9>("try-lock + rmw b128", out);
    hipDeviceSynchronize();
    return 0;
}
