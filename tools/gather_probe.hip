// Random gather / scatter rate of the memory system against granule size (diagnostic for the cell-order
// pipeline's scatter and un-permute passes, which move 16-byte records at random positions of 160 MB arrays).
//   gather:  out[i] (coalesced) = table[idx[i]]   granule G bytes
//   scatter: table[idx[i]] = in[i] (coalesced)
// build + run on the GPU box:  hipcc -O3 --offload-arch=gfx950 tools/gather_probe.hip -o /tmp/gp && /tmp/gp
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include <numeric>
#include <random>

template <int W>   // W = 16-byte words per granule
__global__ void gather(const uint4 *__restrict__ table, const unsigned *__restrict__ idx, uint4 *__restrict__ out, long long n)
{
    const long long stride = (long long)gridDim.x * blockDim.x * 4;
    for (long long i0 = (long long)blockIdx.x * blockDim.x * 4 + threadIdx.x; i0 < n; i0 += stride) {
        unsigned p[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) { const long long i = i0 + (long long)k * blockDim.x; p[k] = idx[i < n ? i : n - 1]; }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const long long i = i0 + (long long)k * blockDim.x;
            if (i < n) {
#pragma unroll
                for (int w = 0; w < W; ++w) out[(size_t)i * W + w] = table[(size_t)p[k] * W + w];
            }
        }
    }
}
template <int W>
__global__ void scatter(uint4 *__restrict__ table, const unsigned *__restrict__ idx, const uint4 *__restrict__ in, long long n)
{
    const long long stride = (long long)gridDim.x * blockDim.x * 4;
    for (long long i0 = (long long)blockIdx.x * blockDim.x * 4 + threadIdx.x; i0 < n; i0 += stride) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const long long i = i0 + (long long)k * blockDim.x;
            if (i < n) {
                const unsigned p = idx[i];
#pragma unroll
                for (int w = 0; w < W; ++w) table[(size_t)p * W + w] = in[(size_t)i * W + w];
            }
        }
    }
}

template <int W>
void run(size_t bytes)
{
    const long long n = (long long)(bytes / (16 * W));
    std::vector<unsigned> h(n);
    std::iota(h.begin(), h.end(), 0u);
    std::mt19937 rng(1);
    std::shuffle(h.begin(), h.end(), rng);
    unsigned *idx; uint4 *a, *b;
    hipMalloc(&idx, 4 * n); hipMalloc(&a, bytes); hipMalloc(&b, bytes);
    hipMemcpy(idx, h.data(), 4 * n, hipMemcpyHostToDevice);
    hipMemset(a, 1, bytes); hipMemset(b, 2, bytes);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int mode = 0; mode < 2; ++mode) {
        for (int it = 0; it < 3; ++it) {
            if (mode == 0) hipLaunchKernelGGL(gather<W>, dim3(2048), dim3(256), 0, 0, a, idx, b, n);
            else hipLaunchKernelGGL(scatter<W>, dim3(2048), dim3(256), 0, 0, a, idx, b, n);
        }
        hipEventRecord(e0);
        for (int it = 0; it < 10; ++it) {
            if (mode == 0) hipLaunchKernelGGL(gather<W>, dim3(2048), dim3(256), 0, 0, a, idx, b, n);
            else hipLaunchKernelGGL(scatter<W>, dim3(2048), dim3(256), 0, 0, a, idx, b, n);
        }
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 10;
        printf("%s granule %4d B, %6.1f MB table: %8.1f us  %6.1f G granules/s  %6.2f TB/s of granule bytes (x2 with the coalesced side)\n",
               mode ? "scatter" : "gather ", 16 * W, bytes / 1e6, ms * 1e3, n / ms / 1e6, (double)bytes / ms / 1e9);
    }
    hipFree(idx); hipFree(a); hipFree(b);
}

int main()
{
    for (size_t mb : {160, 20}) {
        run<1>(mb << 20); run<2>(mb << 20); run<4>(mb << 20); run<8>(mb << 20);
    }
    return 0;
}
