// LDS-resident tile kernels (gfx950): the hot path when the spline's coefficient table fits
// in one CU's LDS (BASELINE cfg2: 64 x 64 x 3 fp64 = 96 KiB).
//
// Why this kernel looks the way it does (measured on MI355X, profiles/):
//  * per point the kernel reads prod(order) * nDep coefficients (48 fp64 = 384 B for the
//    bicubic case) from the LDS table at a random span -> the LDS pipe, not HBM (40 B per
//    point), is the scarce resource;
//  * random per-lane LDS addresses conflict ~3.5-way (32 lanes of a half-wave over 32
//    8-byte bank pairs), and hipcc fuses neighbouring 8-byte reads into ds_read2_b64, which
//    runs at half the bandwidth of ds_read_b64.
// So: (1) every table read is an explicit ds_read_b64 / ds_read_b32 (inline asm, counted
// s_waitcnt), (2) the span search goes through a bucket table (one 4-byte read + usually one
// knot read instead of log2(n) knot reads), and (3) within each 1024-point tile the points
// are re-assigned to lanes by LDS *bank class* of their coefficient window (class = window
// offset mod 32), so the 32 lanes of a half-wave read 32 different banks: every
// coefficient read of the tile is then (nearly) conflict free.  Results are un-permuted
// through LDS so global loads and stores stay fully coalesced.
#pragma once
#include "bsk_device.hpp"

namespace bsk {

constexpr int TILE = 1024;          // points per tile = threads per workgroup
constexpr int NCLASS = 32;          // LDS bank classes
constexpr int ROWS = TILE / NCLASS; // half-wave rows per tile

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
    static_assert(N >= 1 && N <= 6 && N <= CAP, "row length");
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
    else
        asm volatile("s_waitcnt lgkmcnt(%6)"
                     : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]) : "n"(CNT) : "memory");
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

// One slab = O rows of O contiguous coefficients (rows `rstride` bytes apart), contracted
// with b_in (within a row) then b_out (across rows): the last two variables of the window.
// Reads are issued for the whole slab, rows are consumed as they arrive.
template <typename T, int O, int A = 0>
__device__ __forceinline__ void slab_wait_fma(T (&c)[O][O], const T (&b_out)[O], const T (&b_in)[O], T &acc)
{
    if constexpr (A < O) {
        constexpr int left = (O - 1 - A) * O;                 // reads issued after row A
        lds_wait_row<(left > 15 ? 15 : left)>(c[A]);
        if constexpr (left > 15) lds_wait_row<15>(c[A]);      // (only O == 6: 30, 24, 18 -> clamp)
        T t = T(0);
#pragma unroll
        for (int k = 0; k < O; ++k) t += c[A][k] * b_in[k];
        acc += t * b_out[A];
        slab_wait_fma<T, O, A + 1>(c, b_out, b_in, acc);
    }
}

template <typename T, int O>
__device__ __forceinline__ T slab_contract(unsigned addr, unsigned rstride, const T (&b_out)[O], const T (&b_in)[O])
{
    T c[O][O];
#pragma unroll
    for (int a = 0; a < O; ++a) lds_issue_row<T, O>(addr + a * rstride, c[a]);
    T acc = T(0);
    if constexpr (O * O > 16) {
        // more than 15 younger reads cannot be expressed in lgkmcnt: drain, then consume
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int a = 0; a < O; ++a) {
            lds_wait_row<0>(c[a]);
            T t = T(0);
#pragma unroll
            for (int k = 0; k < O; ++k) t += c[a][k] * b_in[k];
            acc += t * b_out[a];
        }
    } else {
        slab_wait_fma<T, O>(c, b_out, b_in, acc);
    }
    return acc;
}

template <typename T, int O>
__device__ __forceinline__ T row_contract(unsigned addr, const T (&b)[O])
{
    T c[O];
    lds_issue_row<T, O>(addr, c);
    lds_wait_row<0>(c);
    T acc = T(0);
#pragma unroll
    for (int k = 0; k < O; ++k) acc += c[k] * b[k];
    return acc;
}

// -------------------------------------------------------------------------------------
// axis tables in LDS, read with explicit instructions
// -------------------------------------------------------------------------------------
// Span search through the bucket table (layout: TileDesc).  e = lo | hi << 16 brackets the
// span of every u that falls in the bucket; `steps` binary steps close the bracket.
template <typename T>
__device__ __forceinline__ int find_span_lut(unsigned knots_addr, unsigned lut_addr, int m, int steps, T lo, T scale,
                                             int ncoef, T u)
{
    int b = (int)((u - lo) * scale);
    b = b < 0 ? 0 : (b > m - 1 ? m - 1 : b);
    unsigned e;
    asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(e) : "v"(lut_addr + 4u * (unsigned)b) : "memory");
    int l = (int)(e & 0xffffu), h = (int)(e >> 16);
    for (int s = 0; s < steps; ++s) {
        const int mid = (l + h) >> 1;
        T km[1];
        km[0] = LdsRead<T>::template at<0>(knots_addr + (unsigned)mid * (unsigned)sizeof(T));
        lds_wait_row<0>(km);
        const bool open = l < h;
        const bool right = open && (km[0] <= u);
        const bool left = open && !right;
        l = right ? mid + 1 : l;
        h = left ? mid : h;
    }
    return (u != u) ? ncoef : l;
}

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

template <typename T, int O, int D>
__device__ __forceinline__ void basis_levels(T u, int wrt, const T (&kn)[O], T (&rc)[O][O], T (&b)[O])
{
    if constexpr (D < O) {
        constexpr int younger = (O * (O - 1) - D * (D + 1)) / 2;      // reads issued after row D
        lds_wait_n<younger, D>(rc[D]);
        if (D < O - wrt) {                                             // value level (reference :12-18)
#pragma unroll
            for (int j = 0; j < D; ++j) {
                const int bi = O - D + j;
                const T alpha = (u - kn[(O - 1) - D + j]) * rc[D][j];   // knots[ix - D + j]
                b[bi - 1] += (T(1) - alpha) * b[bi];
                b[bi] *= alpha;
            }
        } else {                                                       // derivative level (reference :19-26)
#pragma unroll
            for (int j = 0; j < D; ++j) {
                const int bi = O - D + j;
                const T alpha = T(D) * rc[D][j];
                b[bi - 1] -= alpha * b[bi];
                b[bi] *= alpha;
            }
        }
        basis_levels<T, O, D + 1>(u, wrt, kn, rc, b);
    }
}

template <typename T, int O>
__device__ __forceinline__ void basis_lds(unsigned tab_addr, int nk, int ix, T u, int wrt, T (&b)[O])
{
#pragma unroll
    for (int k = 0; k < O; ++k) b[k] = T(0);
    b[O - 1] = T(1);
    if constexpr (O > 1) {
        T kn[O];          // kn[j] = knots[ix - (O-1) + j], j < O-1
        T rc[O][O];       // rc[D][j] = r[D][ix - D + j], j < D
        lds_issue_n<T, O - 1, O>(tab_addr + (unsigned)(ix - (O - 1)) * (unsigned)sizeof(T), kn);
        basis_issue<T, O, 1>(tab_addr, nk, ix, rc);
        lds_wait_n<(O * (O - 1)) / 2, O - 1>(kn);
        basis_levels<T, O, 1>(u, wrt, kn, rc, b);
    }
    if (wrt >= O) {
#pragma unroll
        for (int k = 0; k < O; ++k) b[k] = T(0);                       // reference :9-10
    }
}

// -------------------------------------------------------------------------------------
// descriptor extension for the tile kernels
// -------------------------------------------------------------------------------------
// LDS image (bytes, all offsets 16-byte aligned):
//   [axis tables: Desc::tab_len x T] [bucket tables: lut_len x u32] [coefficients: coef_len x T]
//   [staging: see eval_tile]
template <typename T>
struct TileDesc {
    int lut_off[MAXI];   // first bucket of variable iv (u32 index)
    int lut_m[MAXI];     // number of buckets
    int lut_steps[MAXI]; // binary steps after the bucket lookup
    T lut_scale[MAXI];   // buckets per unit parameter
    int lut_len;
    unsigned tab_bytes, lut_bytes, coef_bytes;   // rounded up to 16
};

// -------------------------------------------------------------------------------------
// the tile kernel
// -------------------------------------------------------------------------------------
// PERM: bank-class permutation of the points of each tile (see header comment).
template <typename T, int NIND, int O, bool PERM>
__global__ __launch_bounds__(TILE) void eval_tile(const Desc<T> d, const TileDesc<T> td,
                                                  const T *__restrict__ gtab, const unsigned *__restrict__ glut,
                                                  const T *__restrict__ gcoef, const Params<T> prm,
                                                  const long long N, T *__restrict__ out, const long long ostride,
                                                  const Wrt wrt, unsigned long long *bad)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const unsigned lds0 = (unsigned)(size_t)smem;          // LDS byte address of the image
    const unsigned tab_a = lds0;
    const unsigned lut_a = tab_a + td.tab_bytes;
    const unsigned coef_a = lut_a + td.lut_bytes;
    char *stage = smem + td.tab_bytes + td.lut_bytes + td.coef_bytes;
    // staging (PERM only): parameters and span indices by original lane, permutation tables,
    // results by original lane
    T *s_u = reinterpret_cast<T *>(stage);                                  // [NIND][TILE]
    unsigned short *s_ix = reinterpret_cast<unsigned short *>(s_u + NIND * TILE);   // [NIND][TILE]
    unsigned short *s_perm = s_ix + NIND * TILE;                            // [TILE]
    unsigned short *s_ovf = s_perm + TILE;                                  // [TILE]
    int *s_cnt = reinterpret_cast<int *>(s_ovf + TILE);                     // [NCLASS] class counts
    int *s_hole = s_cnt + NCLASS;                                           // [NCLASS + 1] exclusive prefix of holes
    int *s_over = s_hole + NCLASS + 1;                                      // [NCLASS + 1] exclusive prefix of overflow
    T *s_res = reinterpret_cast<T *>(s_over + NCLASS + 1 + 2);              // [nDep][TILE]  (8-byte aligned: 32+33+33+2 ints)

    {   // stage the tables
        T *stab = reinterpret_cast<T *>(smem);
        unsigned *slut = reinterpret_cast<unsigned *>(smem + td.tab_bytes);
        T *scoef = reinterpret_cast<T *>(smem + td.tab_bytes + td.lut_bytes);
        for (int i = threadIdx.x; i < d.tab_len; i += TILE) stab[i] = gtab[i];
        for (int i = threadIdx.x; i < td.lut_len; i += TILE) slut[i] = glut[i];
        for (int i = threadIdx.x; i < d.coef_len; i += TILE) scoef[i] = gcoef[i];
    }
    __syncthreads();

    const int tid = threadIdx.x;
    const long long ntiles = (N + TILE - 1) / TILE;
    for (long long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const long long n0 = tile * TILE;
        const long long n = n0 + tid;
        const bool valid = n < N;

        T u[NIND];
        int ix[NIND];
        bool outside = false;
#pragma unroll
        for (int iv = 0; iv < NIND; ++iv) {
            u[iv] = d.lo[iv];
            if (valid) u[iv] = prm.p[iv][n];
            outside |= (u[iv] < d.lo[iv]) | (u[iv] > d.hi[iv]);
            ix[iv] = find_span_lut<T>(tab_a + (unsigned)d.off[iv] * (unsigned)sizeof(T),
                                      lut_a + 4u * (unsigned)td.lut_off[iv], td.lut_m[iv], td.lut_steps[iv], d.lo[iv],
                                      td.lut_scale[iv], d.ncoef[iv], u[iv]);
        }
        if (valid && outside) record_bad(bad, n);

        int src = tid;           // original lane whose point this lane evaluates
        bool have = valid;
        if constexpr (PERM) {
            int base = 0;
#pragma unroll
            for (int iv = 0; iv < NIND; ++iv) base += (ix[iv] - O) * d.cstride[iv + 1];
            const int cls = base & (NCLASS - 1);
            if (tid < NCLASS) s_cnt[tid] = 0;
            __syncthreads();
            int rank = 0;
            if (valid) {
                rank = atomicAdd(&s_cnt[cls], 1);
#pragma unroll
                for (int iv = 0; iv < NIND; ++iv) {
                    s_u[iv * TILE + tid] = u[iv];
                    s_ix[iv * TILE + tid] = (unsigned short)ix[iv];
                }
            }
            __syncthreads();
            if (tid < WAVE) {
                // exclusive prefix sums of holes / overflow per class (one wave)
                const int c = tid & (NCLASS - 1);
                const int cnt = s_cnt[c];
                int hole = (tid < NCLASS && cnt < ROWS) ? ROWS - cnt : 0;
                int over = (tid < NCLASS && cnt > ROWS) ? cnt - ROWS : 0;
                int hs = hole, os = over;
#pragma unroll
                for (int off = 1; off < NCLASS; off <<= 1) {
                    const int h2 = __shfl_up(hs, off, WAVE);
                    const int o2 = __shfl_up(os, off, WAVE);
                    if ((tid & (WAVE - 1)) >= off) { hs += h2; os += o2; }
                }
                if (tid < NCLASS) {
                    s_hole[tid] = hs - hole;
                    s_over[tid] = os - over;
                    if (tid == NCLASS - 1) { s_hole[NCLASS] = hs; s_over[NCLASS] = os; }
                }
            }
            __syncthreads();
            if (valid) {
                if (rank < ROWS) s_perm[rank * NCLASS + cls] = (unsigned short)tid;
                else s_ovf[s_over[cls] + rank - ROWS] = (unsigned short)tid;
            }
            __syncthreads();
            {
                const int r = tid >> 5, c = tid & (NCLASS - 1);
                const int cnt = s_cnt[c];
                if (r < cnt) {
                    src = s_perm[tid];
                    have = true;
                } else {
                    const int k = s_hole[c] + (r - cnt);
                    have = k < s_over[NCLASS];
                    src = have ? (int)s_ovf[k] : tid;
                }
            }
            if (have) {
#pragma unroll
                for (int iv = 0; iv < NIND; ++iv) {
                    u[iv] = s_u[iv * TILE + src];
                    ix[iv] = s_ix[iv * TILE + src];
                }
            }
        }

        if (have) {
            T b[NIND][O];
            unsigned caddr = coef_a;
#pragma unroll
            for (int iv = 0; iv < NIND; ++iv) {
                basis_lds<T, O>(tab_a + (unsigned)d.off[iv] * (unsigned)sizeof(T), d.nk[iv], ix[iv], u[iv], wrt.w[iv],
                                b[iv]);
                caddr += (unsigned)((ix[iv] - O) * d.cstride[iv + 1]) * (unsigned)sizeof(T);
            }
            const unsigned dstride = (unsigned)d.cstride[0] * (unsigned)sizeof(T);
            for (int dep = 0; dep < d.nDep; ++dep) {
                T r;
                if constexpr (NIND == 1) {
                    r = row_contract<T, O>(caddr, b[0]);
                } else if constexpr (NIND == 2) {
                    r = slab_contract<T, O>(caddr, (unsigned)d.cstride[1] * (unsigned)sizeof(T), b[0], b[1]);
                } else {
                    r = T(0);
#pragma unroll
                    for (int a = 0; a < O; ++a)
                        r += b[0][a] * slab_contract<T, O>(caddr + (unsigned)(a * d.cstride[1]) * (unsigned)sizeof(T),
                                                           (unsigned)d.cstride[2] * (unsigned)sizeof(T), b[1], b[2]);
                }
                caddr += dstride;
                if constexpr (PERM) s_res[dep * TILE + src] = r;
                else out[dep * ostride + n] = r;
            }
        }
        if constexpr (PERM) {
            __syncthreads();
            if (valid)
                for (int dep = 0; dep < d.nDep; ++dep) out[dep * ostride + n] = s_res[dep * TILE + tid];
            // the next tile's first barrier (after zeroing s_cnt) orders these reads before
            // any lane overwrites s_res / s_u again
        }
    }
}

}  // namespace bsk
