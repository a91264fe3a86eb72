// Building blocks of the LDS-resident kernels (gfx950): explicit LDS reads with counted waits,
// the table-image descriptor, and the recursion's table-row requests.
//
// Why explicit reads (measured on MI355X, profiles/):
//  * per point the kernels read prod(order) * nDep coefficients (48 fp64 = 384 B for the bicubic
//    case) from the LDS table at a random span -> the LDS pipe, not HBM (40 B per point), is the
//    scarce resource;
//  * hipcc fuses neighbouring 8-byte LDS reads into ds_read2_b64, which runs at half the
//    bandwidth of ds_read_b64 (128 vs 256 B/clk/CU).
// So every table read is an explicit ds_read_b64 / ds_read_b32 (inline asm, counted s_waitcnt).
// RULE: hipcc does not track asm operands - it neither waits for an asm read's destination nor
// refrains from copying / spilling it before the wait.  Kernels built on these helpers must not
// spill (check_spills.py enforces it) and keep the issue-to-wait windows short.
#pragma once
#include "bsk_device.hpp"

namespace bsk {

constexpr int TILE = 1024;          // threads per workgroup of the LDS-resident kernels

// -------------------------------------------------------------------------------------
// explicit LDS reads
// -------------------------------------------------------------------------------------
template <typename T>
struct LdsRead;

template <>
struct LdsRead<double> {
    template <int OFF>
    static __device__ __forceinline__ double at(unsigned addr)
    {
        double v;
        asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF) : "memory");
        return v;
    }
};

template <>
struct LdsRead<float> {
    template <int OFF>
    static __device__ __forceinline__ float at(unsigned addr)
    {
        float v;
        asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF) : "memory");
        return v;
    }
};

// Wait until at most CNT of this wave's LDS reads are outstanding and tie the first N values
// of `v` to the wait so no consumer can be scheduled above it.  LDS returns in order, so
// with CNT reads issued after those N, they are complete.
template <int CNT, int N, typename T, int CAP>
__device__ __forceinline__ void lds_wait_n(T (&v)[CAP])
{
    static_assert(N >= 1 && N <= 8 && N <= CAP, "row length");
    static_assert(CNT >= 0 && CNT <= 15, "lgkmcnt is 4 bits");
    if constexpr (N == 1) asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(v[0]) : "n"(CNT) : "memory");
    else if constexpr (N == 2) asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(v[0]), "+v"(v[1]) : "n"(CNT) : "memory");
    else if constexpr (N == 3)
        asm volatile("s_waitcnt lgkmcnt(%3)" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]) : "n"(CNT) : "memory");
    else if constexpr (N == 4)
        asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]) : "n"(CNT) : "memory");
    else if constexpr (N == 5)
        asm volatile("s_waitcnt lgkmcnt(%5)"
                     : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]) : "n"(CNT) : "memory");
    else if constexpr (N == 6)
        asm volatile("s_waitcnt lgkmcnt(%6)"
                     : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]) : "n"(CNT) : "memory");
    else if constexpr (N == 7)
        asm volatile("s_waitcnt lgkmcnt(%7)"
                     : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6])
                     : "n"(CNT) : "memory");
    else
        asm volatile("s_waitcnt lgkmcnt(%8)"
                     : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7])
                     : "n"(CNT) : "memory");
}

template <int CNT, typename T, int N>
__device__ __forceinline__ void lds_wait_row(T (&v)[N])
{
    lds_wait_n<CNT, N>(v);
}

// Issue N reads of consecutive elements starting at byte address addr into v[0..N).
template <typename T, int N, int CAP, int K = 0>
__device__ __forceinline__ void lds_issue_n(unsigned addr, T (&v)[CAP])
{
    if constexpr (K < N) {
        v[K] = LdsRead<T>::template at<K * (int)sizeof(T)>(addr);
        lds_issue_n<T, N, CAP, K + 1>(addr, v);
    }
}

template <typename T, int O>
__device__ __forceinline__ void lds_issue_row(unsigned addr, T (&v)[O])
{
    lds_issue_n<T, O, O>(addr, v);
}

// -------------------------------------------------------------------------------------
// axis tables in LDS, read with explicit instructions
// -------------------------------------------------------------------------------------
// Cox-de Boor recursion, compile-time order, tables read by explicit LDS instructions.
// tab_addr: byte address of this variable's axis table (knots, then reciprocal rows).
// All reads of the span - the O-1 knots ix-O+1 .. ix-1 and, per level D, the D reciprocals
// r[D][ix-D .. ix-1] - are issued up front; each level then waits only for its own row.
template <typename T, int O, int D>
__device__ __forceinline__ void basis_issue(unsigned tab_addr, int nk, int ix, T (&rc)[O][O])
{
    if constexpr (D < O) {
        lds_issue_n<T, D, O>(tab_addr + (unsigned)(D * nk + ix - D) * (unsigned)sizeof(T), rc[D]);
        basis_issue<T, O, D + 1>(tab_addr, nk, ix, rc);
    }
}

// -------------------------------------------------------------------------------------
// descriptor extension for the tile kernels
// -------------------------------------------------------------------------------------
// LDS image (bytes, all offsets 16-byte aligned):
//   [axis tables: Desc::tab_len x T] [bucket tables: lut_len x u32] [coefficients: coef_len x T]
template <typename T>
struct TileDesc {
    int lut_off[MAXI];   // first bucket of variable iv (u32 index)
    int lut_m[MAXI];     // number of buckets
    int lut_steps[MAXI]; // binary steps after the bucket lookup
    T lut_scale[MAXI];   // buckets per unit parameter
    int lut_len;
    unsigned tab_bytes, lut_bytes, coef_bytes;   // rounded up to 16
};

// Span search through the bucket table of variable iv (same result as find_span): one table read brackets the
// span, `lut_steps` bisection steps on the knots finish it (1 for near-uniform knots, against ceil(log2(spans)))
template <typename T, typename KP, typename LP>
__device__ __forceinline__ int find_span_lut(KP knots, LP lut, const TileDesc<T> &td, int iv, T lo, int ncoef, T u)
{
    int b = (int)((u - lo) * td.lut_scale[iv]);
    b = min(max(b, 0), td.lut_m[iv] - 1);
    const unsigned e = lut[td.lut_off[iv] + b];
    int l = (int)(e & 0xffffu), h = (int)(e >> 16);
    for (int s = 0; s < td.lut_steps[iv]; ++s) {
        const int mid = (l + h) >> 1;
        const T km = knots[mid];
        const bool open = l < h;
        const bool right = open && (km <= u);
        const bool left = open && !right;
        l = right ? mid + 1 : l;
        h = left ? mid : h;
    }
    return (u != u) ? ncoef : l;
}

// The same search with a compile-time number of bisection steps (STEPS >= lut_steps[iv]: a step on a closed bracket
// changes nothing): straight-line code, so that the searches of a lane's several points interleave their dependent
// LDS reads - behind the run-time loop above they run one after the other.
template <typename T, int STEPS, typename KP, typename LP>
__device__ __forceinline__ int find_span_lut_n(KP knots, LP lut, const TileDesc<T> &td, int iv, T lo, int ncoef, T u)
{
    if constexpr (STEPS <= 0) {
        return find_span_lut<T>(knots, lut, td, iv, lo, ncoef, u);
    } else {
        int b = (int)((u - lo) * td.lut_scale[iv]);
        b = min(max(b, 0), td.lut_m[iv] - 1);
        const unsigned e = lut[td.lut_off[iv] + b];
        int l = (int)(e & 0xffffu), h = (int)(e >> 16);
#pragma unroll
        for (int s = 0; s < STEPS; ++s) {
            const int mid = (l + h) >> 1;
            const T km = knots[mid];
            const bool open = l < h;
            const bool right = open && (km <= u);
            const bool left = open && !right;
            l = right ? mid + 1 : l;
            h = left ? mid : h;
        }
        return (u != u) ? ncoef : l;
    }
}

// Descriptor of the uniform-knot surface kernels (bsk_uniform.hpp): domain, span width, image layout.
template <typename T>
struct UniDesc {
    T lo[2], hi[2];        // domain
    T inv_h[2];            // spans per unit parameter
    T eps;                 // bias of the span estimate (see uni_spans2)
    int ns[2];             // spans per variable
    int ncoef[2];
    int nDep;
    int rs;                // row stride of the coefficient image (elements), = 32 / order (mod 32)
    unsigned kn_off[2];    // byte offsets inside the image
    unsigned coef_off;
    unsigned img_bytes;    // multiple of 16
};

// The same for curves, surfaces of other orders and volumes (eval_stream_uni / jac_stream_uni).
template <typename T>
struct UniDescN {
    T lo[3], hi[3];
    T inv_h[3];
    T eps;
    int ns[3];
    unsigned kn_off[3];    // byte offsets inside the image
    unsigned coef_off;
    unsigned img_bytes;    // multiple of 16
};

}  // namespace bsk
