// Partition of a launch's tuples by the slice of the catalog one of their items lies in (a counting sort with at most
// 1024 bins): histogram per segment, offsets, scatter.  The owner mode of the Hogwild update kernel (sgd_hogwild.hip)
// consumes the result: slice s of the output is applied by workgroup s.
//
// A block owns a segment of PART_SEG consecutive tuples in both passes.  The scatter pass sorts a tile of PART_TILE
// tuples by bin inside LDS first, so that a bin's tuples of a tile leave as one contiguous run (16 tuples = 256 B on
// average with 512 bins) instead of as single 16-byte stores to 256 different places.
#include "common.h"

constexpr int PART_BINS = 1024;              // at most; the launch says how many are in use
constexpr int PART_THREADS = 1024;
constexpr int PART_SEG = 28672;              // tuples per block (a multiple of every tile size)
// tuples sorted in LDS at a time: 4096 (72 KB with the bin numbers, two blocks per CU) or 7168 (126 KB, one block per CU,
// runs 1.75 times as long)
static_assert(PART_SEG % 4096 == 0 && PART_SEG % 7168 == 0, "whole tiles per segment");

__device__ __forceinline__ int part_bin(const int4& t, int field, int slice_rows)
{
    return (field == 2 ? t.z : t.w) / slice_rows;
}

// counts[bin * P + block] = tuples of the block's segment that fall into the bin
__global__ void __launch_bounds__(PART_THREADS) part_hist_kernel(const int4* __restrict__ src, int64_t n, int field, int slice_rows,
                                                                 int* __restrict__ counts, int P, int bins)
{
    __shared__ int hist[PART_BINS];
    if (threadIdx.x < bins) hist[threadIdx.x] = 0;
    __syncthreads();
    const int64_t s0 = (int64_t)blockIdx.x * PART_SEG;
    const int64_t s1 = s0 + PART_SEG < n ? s0 + PART_SEG : n;
    for (int64_t i = s0 + threadIdx.x; i < s1; i += PART_THREADS) atomicAdd(&hist[part_bin(src[i], field, slice_rows)], 1);
    __syncthreads();
    if (threadIdx.x < bins) counts[(int64_t)threadIdx.x * P + blockIdx.x] = hist[threadIdx.x];
}

// block b: exclusive scan of counts[b * P .. (b + 1) * P) in place; total[b] = their sum
__global__ void __launch_bounds__(256) part_scan_bin_kernel(int* __restrict__ counts, int P, int64_t* __restrict__ total)
{
    __shared__ int wsum[4];
    __shared__ int carry_s;
    int* c = counts + (int64_t)blockIdx.x * P;
    if (threadIdx.x == 0) carry_s = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (int t0 = 0; t0 < P; t0 += 256) {
        const int i = t0 + threadIdx.x;
        const int x = i < P ? c[i] : 0;
        int inc = x;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const int y = __shfl_up(inc, o, 64); if (lane >= o) inc += y; }
        if (lane == 63) wsum[wv] = inc;
        __syncthreads();
        int base = carry_s;
        for (int q = 0; q < wv; q++) base += wsum[q];
        if (i < P) c[i] = base + inc - x;
        __syncthreads();
        if (threadIdx.x == 255) carry_s = base + inc;
        __syncthreads();
    }
    if (threadIdx.x == 0) total[blockIdx.x] = carry_s;
}

// qoff[0 .. bins] = exclusive scan of total[0 .. bins)
__global__ void part_scan_total_kernel(const int64_t* __restrict__ total, int bins, int64_t* __restrict__ qoff)
{
    if (threadIdx.x == 0) {
        int64_t run = 0;
        for (int b = 0; b < bins; b++) { qoff[b] = run; run += total[b]; }
        qoff[bins] = run;
    }
}

template <int BINS, int PART_TILE>
__global__ void __launch_bounds__(PART_THREADS) part_scatter_kernel(const int4* __restrict__ src, int4* __restrict__ dst, int64_t n, int field,
                                                                    int slice_rows, const int* __restrict__ counts, int P,
                                                                    const int64_t* __restrict__ qoff)
{
    constexpr int bins = BINS;
    extern __shared__ __align__(16) unsigned char part_lds[];
    int4* stage = (int4*)part_lds;                                          // [PART_TILE] the tile, ordered by bin
    uint16_t* sbin = (uint16_t*)(part_lds + sizeof(int4) * PART_TILE);     // [PART_TILE] bin of each staged tuple
    __shared__ int hist[BINS], loff[BINS];
    __shared__ int64_t gbase[BINS];                                   // where the block's next tuple of each bin goes
    __shared__ int wsum[BINS / 64];
    if (threadIdx.x < bins) {
        gbase[threadIdx.x] = qoff[threadIdx.x] + counts[(int64_t)threadIdx.x * P + blockIdx.x];
        hist[threadIdx.x] = 0;
    }
    __syncthreads();
    constexpr int PER = PART_TILE / PART_THREADS;
    const int64_t s0 = (int64_t)blockIdx.x * PART_SEG;
    for (int tile = 0; tile < PART_SEG / PART_TILE; tile++) {
        const int64_t t0 = s0 + (int64_t)tile * PART_TILE;
        if (t0 >= n) break;
        const int cnt = (int)(n - t0 < PART_TILE ? n - t0 : PART_TILE);
        int4 t[PER]; int bin[PER], rank[PER];
#pragma unroll
        for (int q = 0; q < PER; q++) {
            const int i = threadIdx.x + q * PART_THREADS;
            bin[q] = -1;
            if (i < cnt) { t[q] = src[t0 + i]; bin[q] = part_bin(t[q], field, slice_rows); rank[q] = atomicAdd(&hist[bin[q]], 1); }
        }
        __syncthreads();
        // exclusive scan of the bin counts by the first `bins` threads (a multiple of 64)
        if (threadIdx.x < bins) {
            const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
            const int x = hist[threadIdx.x];
            int inc = x;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) { const int y = __shfl_up(inc, o, 64); if (lane >= o) inc += y; }
            if (lane == 63) wsum[wv] = inc;
            loff[threadIdx.x] = inc - x;   // within the wavefront so far
        }
        __syncthreads();
        if (threadIdx.x < bins) {
            const int wv = threadIdx.x >> 6;
            int base = 0;
            for (int q = 0; q < wv; q++) base += wsum[q];
            loff[threadIdx.x] += base;
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < PER; q++)
            if (bin[q] >= 0) { const int at = loff[bin[q]] + rank[q]; stage[at] = t[q]; sbin[at] = (uint16_t)bin[q]; }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < PER; q++) {
            const int i = threadIdx.x + q * PART_THREADS;
            if (i < cnt) { const int b = sbin[i]; dst[gbase[b] + (i - loff[b])] = stage[i]; }
        }
        __syncthreads();
        if (threadIdx.x < bins) { gbase[threadIdx.x] += hist[threadIdx.x]; hist[threadIdx.x] = 0; }
        __syncthreads();
    }
}

static int part_blocks(int64_t n) { return (int)((n + PART_SEG - 1) / PART_SEG); }

size_t partition_scratch_bytes(int64_t n, int slices)
{
    return sizeof(int) * (size_t)slices * (size_t)part_blocks(n) + sizeof(int64_t) * (size_t)(slices + 8);
}

template <int BINS, int TILE>
static int launch_scatter(const int4* src, int4* dst, int64_t n, int field, int slice_rows, const int* counts, int P, const int64_t* qoff, hipStream_t st)
{
    const size_t lds = (sizeof(int4) + sizeof(uint16_t)) * TILE;
    HIP_TRY(hipFuncSetAttribute((const void*)part_scatter_kernel<BINS, TILE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL((part_scatter_kernel<BINS, TILE>), dim3(P), dim3(PART_THREADS), lds, st, src, dst, n, field, slice_rows, counts, P, qoff);
    return 0;
}

// slices: 256, 512 or 1024.  tile: 4096 or 7168 tuples sorted in LDS at a time (0: the default for that many slices)
int launch_partition_by_slice(const int4* src, int4* dst, int64_t n, int field, int slice_rows, int slices, void* scratch,
                              int64_t* qoff, hipStream_t st, int tile)
{
    if (n <= 0) return 0;
    if ((slices != 256 && slices != 512 && slices != 1024) || slice_rows < 1 || (field != 2 && field != 3) || (tile != 0 && tile != 4096 && tile != 7168)) {
        rec_set_error("partition: unsupported geometry"); return -1;
    }
    if (tile == 0) tile = slices == 256 ? 4096 : 7168;
    const int P = part_blocks(n);
    int* counts = (int*)scratch;
    int64_t* total = (int64_t*)((char*)scratch + sizeof(int) * (size_t)slices * (size_t)P);
    hipLaunchKernelGGL(part_hist_kernel, dim3(P), dim3(PART_THREADS), 0, st, src, n, field, slice_rows, counts, P, slices);
    hipLaunchKernelGGL(part_scan_bin_kernel, dim3(slices), dim3(256), 0, st, counts, P, total);
    hipLaunchKernelGGL(part_scan_total_kernel, dim3(1), dim3(64), 0, st, (const int64_t*)total, slices, qoff);
    int rc;
    if (slices == 256) rc = tile == 4096 ? launch_scatter<256, 4096>(src, dst, n, field, slice_rows, counts, P, qoff, st) : launch_scatter<256, 7168>(src, dst, n, field, slice_rows, counts, P, qoff, st);
    else if (slices == 512) rc = tile == 4096 ? launch_scatter<512, 4096>(src, dst, n, field, slice_rows, counts, P, qoff, st) : launch_scatter<512, 7168>(src, dst, n, field, slice_rows, counts, P, qoff, st);
    else rc = tile == 4096 ? launch_scatter<1024, 4096>(src, dst, n, field, slice_rows, counts, P, qoff, st) : launch_scatter<1024, 7168>(src, dst, n, field, slice_rows, counts, P, qoff, st);
    if (rc) return rc;
    HIP_TRY(hipGetLastError());
    return 0;
}
