// probe: do misaligned ds_read_b64 / ds_read_b128 (4-byte aligned addresses) return the right bytes on gfx950, and at what rate?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void probe(int mode, int iters, int shift, float *out, unsigned long long *cyc)
{
    __shared__ float s[4096 + 64];
    for (int i = threadIdx.x; i < 4096 + 64; i += blockDim.x) s[i] = (float)i;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    // address: 4 lanes share, different blocks adjacent; +shift floats of misalignment
    unsigned a = (unsigned)(size_t)s + (unsigned)(((lane >> 2) * 8 + (lane & 3) * 40 + shift) * 4);
    float acc = 0.f;
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        float v0, v1, v2, v3, v4;
        if (mode == 0) {
            asm volatile("ds_read_b32 %0, %5\n ds_read_b32 %1, %5 offset:4\n ds_read_b32 %2, %5 offset:8\n ds_read_b32 %3, %5 offset:12\n ds_read_b32 %4, %5 offset:16\n s_waitcnt lgkmcnt(0)"
                         : "=&v"(v0), "=&v"(v1), "=&v"(v2), "=&v"(v3), "=&v"(v4) : "v"(a) : "memory");
        } else if (mode == 1) {
            float2 p, q;
            asm volatile("ds_read_b64 %0, %3\n ds_read_b64 %1, %3 offset:8\n ds_read_b32 %2, %3 offset:16\n s_waitcnt lgkmcnt(0)"
                         : "=&v"(p), "=&v"(q), "=&v"(v4) : "v"(a) : "memory");
            v0 = p.x; v1 = p.y; v2 = q.x; v3 = q.y;
        } else if (mode == 2) {
            float4 p;
            asm volatile("ds_read_b128 %0, %2\n ds_read_b32 %1, %2 offset:16\n s_waitcnt lgkmcnt(0)"
                         : "=&v"(p), "=&v"(v4) : "v"(a) : "memory");
            v0 = p.x; v1 = p.y; v2 = p.z; v3 = p.w;
        } else {
            float2 p, q;
            asm volatile("ds_read2_b32 %0, %3 offset0:0 offset1:1\n ds_read2_b32 %1, %3 offset0:2 offset1:3\n ds_read_b32 %2, %3 offset:16\n s_waitcnt lgkmcnt(0)"
                         : "=&v"(p), "=&v"(q), "=&v"(v4) : "v"(a) : "memory");
            v0 = p.x; v1 = p.y; v2 = q.x; v3 = q.y;
        }
        acc += v0 + 2.f * v1 + 3.f * v2 + 4.f * v3 + 5.f * v4;
        a ^= (unsigned)((it & 1) << 7);
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
int main()
{
    float *out; unsigned long long *cyc;
    const int B = 256 * 4, T = 256, iters = 2000;
    hipMalloc(&out, B * T * 4); hipMalloc(&cyc, B * 8);
    for (int shift = 0; shift < 4; ++shift)
        for (int mode = 0; mode < 4; ++mode) {
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            probe<<<B, T>>>(mode, iters, shift, out, cyc);
            hipDeviceSynchronize();
            hipEventRecord(e0);
            probe<<<B, T>>>(mode, iters, shift, out, cyc);
            hipEventRecord(e1); hipDeviceSynchronize();
            float ms; hipEventElapsedTime(&ms, e0, e1);
            std::vector<float> h(T); hipMemcpy(h.data(), out, T * 4, hipMemcpyDeviceToHost);
            // expected for lane l, iteration-invariant part: check lane 5 by recomputation on host
            double exp = 0; 
            for (int l : {5}) {
                int base = (l >> 2) * 8 + (l & 3) * 40 + shift;
                for (int it = 0; it < iters; ++it) { int b = base ^ ((it > 0 ? ((((it - 1) & 1) ? 0 : 0)) : 0)); (void)b; }
            }
            // simpler: compare the modes against mode 0 on the host
            static std::vector<float> ref;
            if (mode == 0) ref = h;
            bool same = true; for (int i = 0; i < T; ++i) same &= (h[i] == ref[i]);
            printf("shift %d mode %d: %.3f ms  same_as_b32=%d  lane5=%g\n", shift, mode, ms, (int)same, h[5]);
        }
    return 0;
}
