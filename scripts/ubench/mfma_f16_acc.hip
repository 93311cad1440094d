// How v_mfma_f32_16x16x32_f16 accumulates: the error of a chain of MFMAs against the exact sum, in units of
// 2^-24 * sum |terms| (the scale the f16 screen's margin is written in: it allows 2 * (number of terms) of them), on
// random operands and on patterns built to show truncation or narrow alignment inside the matrix pipe:
//   "tail":   one product of 1 and 31 products of 2^-24 per MFMA  (half a unit in the last place of the big one each)
//   "tail2":  the same with products of 2^-23 (one unit in the last place each: any 24-bit datapath keeps them)
//   "cancel": +1, -1 and 30 products of 2^-20  (what is left after a cancellation must be exact)
//   "subnormal": f16 subnormal inputs (2^-20 * 2^-4): are they flushed?
// build: hipcc --offload-arch=gfx950 -O2 -o mfma_f16_acc mfma_f16_acc.hip ; run: ./mfma_f16_acc
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef _Float16 half8_t __attribute__((ext_vector_type(8)));
typedef float float4_t __attribute__((ext_vector_type(4)));

// A [16][32 n] row-major, B [32 n][16] row-major (f16), C [16][16] f32 initial value -> D
__global__ void chain(const _Float16* A, const _Float16* B, const float* C, float* D, int n)
{
    const int lane = threadIdx.x, rc = lane & 15, quad = lane >> 4;
    float4_t acc;
    for (int r = 0; r < 4; r++) acc[r] = C[(4 * quad + r) * 16 + rc];
    for (int s = 0; s < n; s++) {
        half8_t a, b;
        for (int j = 0; j < 8; j++) {
            const int k = 32 * s + 8 * quad + j;
            a[j] = A[rc * 32 * n + k];
            b[j] = B[k * 16 + rc];
        }
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc, 0, 0, 0);
    }
    for (int r = 0; r < 4; r++) D[(4 * quad + r) * 16 + rc] = acc[r];
}

static double run(const char* name, int n, const std::vector<float>& a, const std::vector<float>& b, const std::vector<float>& c)
{
    const int K = 32 * n;
    std::vector<_Float16> ah(16 * K), bh(K * 16);
    for (int i = 0; i < 16 * K; i++) { ah[i] = (_Float16)a[i]; bh[i] = (_Float16)b[i]; }
    _Float16 *dA, *dB; float *dC, *dD;
    hipMalloc(&dA, ah.size() * 2); hipMalloc(&dB, bh.size() * 2); hipMalloc(&dC, 1024); hipMalloc(&dD, 1024);
    hipMemcpy(dA, ah.data(), ah.size() * 2, hipMemcpyHostToDevice); hipMemcpy(dB, bh.data(), bh.size() * 2, hipMemcpyHostToDevice);
    hipMemcpy(dC, c.data(), 1024, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(chain, dim3(1), dim3(64), 0, 0, dA, dB, dC, dD, n);
    std::vector<float> d(256);
    hipMemcpy(d.data(), dD, 1024, hipMemcpyDeviceToHost);
    double worst = 0, worst_abs = 0;
    for (int i = 0; i < 16; i++)
        for (int j = 0; j < 16; j++) {
            long double exact = c[i * 16 + j], mag = fabs((double)c[i * 16 + j]);
            for (int k = 0; k < K; k++) {
                const long double p = (long double)(float)ah[i * K + k] * (long double)(float)bh[k * 16 + j];
                exact += p; mag += fabsl(p);
            }
            const double err = fabs((double)((long double)d[i * 16 + j] - exact));
            if (mag > 0) worst = fmax(worst, err / ((double)mag * ldexp(1.0, -24)));
            worst_abs = fmax(worst_abs, err);
        }
    printf("%-10s n = %d (%3d terms + C): worst error %.3f x 2^-24 sum|terms| (allowed %d), absolute %.3g; D[0][0] = %.9g\n", name, n, K, worst,
           2 * (K + 1), worst_abs, d[0]);
    hipFree(dA); hipFree(dB); hipFree(dC); hipFree(dD);
    return worst;
}

int main()
{
    srand(7);
    for (int n : {1, 2, 4, 8}) {
        const int K = 32 * n;
        std::vector<float> a(16 * K), b(K * 16), c(256);
        for (auto& x : a) x = (float)(_Float16)((rand() / (float)RAND_MAX * 2 - 1) * 100.f);
        for (auto& x : b) x = (float)(_Float16)((rand() / (float)RAND_MAX * 2 - 1) * 100.f);
        for (auto& x : c) x = (rand() / (float)RAND_MAX * 2 - 1) * 1000.f;
        run("random", n, a, b, c);
        // tail: per MFMA step one product of 1, the others 2^-12 * 2^-12
        for (int i = 0; i < 16; i++) for (int k = 0; k < K; k++) { a[i * K + k] = (k % 32 == 0) ? 1.f : ldexpf(1.f, -12); }
        for (int k = 0; k < K; k++) for (int j = 0; j < 16; j++) { b[k * 16 + j] = (k % 32 == 0) ? 1.f : ldexpf(1.f, -12); }
        for (auto& x : c) x = 0.f;
        run("tail", n, a, b, c);
        for (int i = 0; i < 16; i++) for (int k = 0; k < K; k++) { a[i * K + k] = (k % 32 == 0) ? 1.f : ldexpf(1.f, -11); }
        run("tail2", n, a, b, c);
        for (int i = 0; i < 16; i++) for (int k = 0; k < K; k++) { a[i * K + k] = (k % 32 == 0) ? 1.f : (k % 32 == 1) ? -1.f : ldexpf(1.f, -10); }
        for (int k = 0; k < K; k++) for (int j = 0; j < 16; j++) { b[k * 16 + j] = (k % 32 <= 1) ? 1.f : ldexpf(1.f, -10); }
        run("cancel", n, a, b, c);
        for (int i = 0; i < 16; i++) for (int k = 0; k < K; k++) a[i * K + k] = ldexpf(1.f, -20);   // f16 subnormal
        for (int k = 0; k < K; k++) for (int j = 0; j < 16; j++) b[k * 16 + j] = ldexpf(1.f, -4);
        run("subnormal", n, a, b, c);
    }
    return 0;
}
