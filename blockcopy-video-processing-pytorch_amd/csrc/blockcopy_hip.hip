// libblockcopy_hip.so -- gfx950 (MI355X / CDNA4) kernels + C ABI for the block-copy path.
// ABI and reference citations: include/blockcopy_hip.h.  Design notes: DESIGN.md.
//
// All ops are index-remapped copies (0 FLOP/byte) => HBM-bound.  Common rules used below:
//   * the unit of work is a VB-byte vector (VB = 16 where the tile row bs*E and the pointers allow it) so
//     a wavefront moves 1 KiB per memory instruction, fully coalesced on the dense side and in
//     bs*E-byte contiguous runs on the packed side;
//   * flat 1-D decomposition over the *destination* vectors (stores always contiguous), grid capped at
//     256 CUs x 8 workgroups with a grid-stride loop, several independent loads in flight per lane;
//   * integer divisions by runtime tile geometry use host-built multiply-shift constants (FastDiv) so the
//     index arithmetic stays ~30 VALU ops per 16 B and well under the memory time;
//   * 64-wide wavefronts, 256-thread workgroups, no LDS needed except the 3x3 neighbour table of the halo
//     gather.
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <hip/hip_bfloat16.h>
#include <stdint.h>
#include <string.h>

#include <mutex>
#include <vector>

#include "../../include/blockcopy_hip.h"

#define BC_EXPORT extern "C" __attribute__((visibility("default")))

namespace {

// ------------------------------------------------------------------------------------------ fast division
struct FastDiv {
    uint32_t d, m, s1, s2;
};

FastDiv make_fd(uint32_t d)
{
    FastDiv f;
    f.d = d;
    if (d <= 1) { f.m = 0; f.s1 = 0; f.s2 = 0; return f; }
    uint32_t l = 0;
    while ((1ull << l) < d) ++l;
    f.m = (uint32_t)(((1ull << 32) * ((1ull << l) - d)) / d + 1);
    f.s1 = 1;
    f.s2 = l - 1;
    return f;
}

__device__ __forceinline__ uint32_t fd_div(uint32_t n, const FastDiv &f)
{
    const uint32_t t = __umulhi(n, f.m);
    return (t + ((n - t) >> f.s1)) >> f.s2;
}

__device__ __forceinline__ void fd_divmod(uint32_t n, const FastDiv &f, uint32_t &q, uint32_t &r)
{
    q = fd_div(n, f);
    r = n - q * f.d;
}

template <int VB> struct VecOf;
template <> struct VecOf<16> { typedef uint4 type; };
template <> struct VecOf<8> { typedef uint2 type; };
template <> struct VecOf<4> { typedef uint32_t type; };
template <> struct VecOf<2> { typedef uint16_t type; };
template <> struct VecOf<1> { typedef uint8_t type; };

constexpr int WG = 256;          // 4 wavefronts
constexpr int MAX_WG = 256 * 8;  // 8 resident workgroups per CU on 256 CUs
constexpr int UNROLL = 4;

// ------------------------------------------------------------------------------------------ gather / scatter
struct TileGeom {
    FastDiv vpr, bs, C, GW, GH;  // vectors per tile row, tile size, channels, grid width/height
    uint32_t H, bsz, vprW;       // dense height, tile size, vectors per dense row
    uint32_t total;              // number of packed vectors = n_exec*C*bs*vpr
};

// dense vector index of packed vector v
__device__ __forceinline__ uint32_t dense_index(uint32_t v, const TileGeom &g, const int32_t *__restrict__ mapping_exec)
{
    uint32_t r, xv, r2, h, b, c;
    fd_divmod(v, g.vpr, r, xv);
    fd_divmod(r, g.bs, r2, h);
    fd_divmod(r2, g.C, b, c);
    const uint32_t ig = (uint32_t)mapping_exec[b];
    uint32_t t, gw, n, gh;
    fd_divmod(ig, g.GW, t, gw);
    fd_divmod(t, g.GH, n, gh);
    return ((n * g.C.d + c) * g.H + gh * g.bsz + h) * g.vprW + gw * g.vpr.d + xv;
}

// TO_PACKED: packed[v] = dense[dense_index(v)]   (split);   else dense[dense_index(v)] = packed[v]   (combine)
template <int VB, bool TO_PACKED>
__global__ __launch_bounds__(WG) void k_tiles(typename VecOf<VB>::type *__restrict__ packed_w,
                                              const typename VecOf<VB>::type *__restrict__ packed_r,
                                              typename VecOf<VB>::type *__restrict__ dense_w,
                                              const typename VecOf<VB>::type *__restrict__ dense_r,
                                              const int32_t *__restrict__ mapping_exec, TileGeom g)
{
    typedef typename VecOf<VB>::type V;
    const uint32_t stride = gridDim.x * WG;
    for (uint32_t v0 = blockIdx.x * WG + threadIdx.x; v0 < g.total; v0 += stride * UNROLL) {
        V val[UNROLL];
        uint32_t di[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            const uint32_t v = v0 + u * stride;
            if (v < g.total) {
                di[u] = dense_index(v, g, mapping_exec);
                val[u] = TO_PACKED ? dense_r[di[u]] : packed_r[v];
            }
        }
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            const uint32_t v = v0 + u * stride;
            if (v < g.total) {
                if (TO_PACKED) packed_w[v] = val[u];
                else dense_w[di[u]] = val[u];
            }
        }
    }
}

// ------------------------------------------------------------------------------------------ fused scatter + copy
struct DenseGeom {
    FastDiv vprW, H, C, vpr, bs;  // vectors per dense row, dense height, channels, vectors per tile row, tile size
    uint32_t GH, GW;
    uint32_t total;               // N*C*H*vprW
};

template <int VB>
__global__ __launch_bounds__(WG) void k_combine_copy(const typename VecOf<VB>::type *__restrict__ blocks,
                                                     const typename VecOf<VB>::type *__restrict__ prev,
                                                     typename VecOf<VB>::type *__restrict__ out,
                                                     const int32_t *__restrict__ grid_idx, DenseGeom g)
{
    typedef typename VecOf<VB>::type V;
    const uint32_t stride = gridDim.x * WG;
    for (uint32_t v0 = blockIdx.x * WG + threadIdx.x; v0 < g.total; v0 += stride * UNROLL) {
        V val[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            const uint32_t v = v0 + u * stride;
            if (v < g.total) {
                uint32_t r, xw, r2, y, n, c, gw, xv, gh, h;
                fd_divmod(v, g.vprW, r, xw);
                fd_divmod(r, g.H, r2, y);
                fd_divmod(r2, g.C, n, c);
                fd_divmod(xw, g.vpr, gw, xv);
                fd_divmod(y, g.bs, gh, h);
                const int32_t idx = grid_idx[(n * g.GH + gh) * g.GW + gw];
                const V *src = idx >= 0 ? blocks + (((uint32_t)idx * g.C.d + c) * g.bs.d + h) * g.vpr.d + xv : prev + v;
                val[u] = *src;
            }
        }
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            const uint32_t v = v0 + u * stride;
            if (v < g.total) out[v] = val[u];
        }
    }
}

// ------------------------------------------------------------------------------------------ border-ring transfer
struct TransferGeom {
    FastDiv vpr, bs, C;
    uint32_t epv;      // elements per vector
    int32_t pad;       // ring width (<0: whole tile)
    uint32_t n_total;  // N*GH*GW of the previous frame
    uint32_t total;    // n_transfer*C*bs*vpr
};

template <int VB>
__global__ __launch_bounds__(WG) void k_transfer(typename VecOf<VB>::type *__restrict__ out,
                                                 const typename VecOf<VB>::type *__restrict__ prev_computed,
                                                 const typename VecOf<VB>::type *__restrict__ prev_transfer,
                                                 const int32_t *__restrict__ transfer_idx, TransferGeom g)
{
    const uint32_t stride = gridDim.x * WG;
    const uint32_t plane = g.C.d * g.bs.d * g.vpr.d;  // vectors per tile
    for (uint32_t v = blockIdx.x * WG + threadIdx.x; v < g.total; v += stride) {
        uint32_t r, xv, r2, h, b, c;
        fd_divmod(v, g.vpr, r, xv);
        fd_divmod(r, g.bs, r2, h);
        fd_divmod(r2, g.C, b, c);
        if (g.pad >= 0) {
            // a vector is skipped only when every element of it is interior (interior is a don't-care,
            // reference utils/block_funcs.py:218-224), so partially-ring vectors are copied whole.
            const int32_t w0 = (int32_t)(xv * g.epv), w1 = w0 + (int32_t)g.epv - 1;
            const int32_t lo = g.pad, hi = (int32_t)g.bs.d - g.pad - 1;
            if (w0 >= lo && w1 <= hi && (int32_t)h >= lo && (int32_t)h <= hi) continue;
        }
        int32_t bp = transfer_idx[b];
        const typename VecOf<VB>::type *src = prev_computed;
        if (bp < 0) { bp += (int32_t)g.n_total; src = prev_transfer; }
        out[v] = src[(uint32_t)bp * plane + (v - b * plane)];
    }
}

// ------------------------------------------------------------------------------------------ halo gather
struct HaloGeom {
    FastDiv PP, BSP, GW, GH;  // padded plane (bs+2p)^2, padded row bs+2p, grid dims
    uint32_t C, bs, pad, n_total;
    uint32_t per_tile;        // C*PP output elements per executed tile
};

// One workgroup column (blockIdx.y) per executed tile; lanes run over the tile's contiguous output
// (C planes of (bs+2p)^2 elements).  The 3x3 neighbour table (which tensor, which row of it, or zero) is
// resolved once per workgroup into LDS.  RING=false: reference repad semantics (neighbour rows of the
// compacted `other` = transfer tensor).  RING=true: `other` is the persistent ring cache indexed by grid
// position, and the tile's own border ring is written back to it.
template <typename T, bool RING>
__global__ __launch_bounds__(WG) void k_halo(T *__restrict__ out, const T *__restrict__ features,
                                             const T *__restrict__ other_r, T *__restrict__ ring_w,
                                             const int32_t *__restrict__ grid_idx,
                                             const int32_t *__restrict__ mapping_exec, HaloGeom g)
{
    __shared__ int32_t nb_row[9];
    __shared__ int32_t nb_kind[9];  // 0 = features, 1 = other (transfer / ring), 2 = zero (beyond image border)
    __shared__ uint32_t own_g;
    const uint32_t b = blockIdx.y;
    if (threadIdx.x < 9) {
        const uint32_t ig = (uint32_t)mapping_exec[b];
        uint32_t t, gw, n, gh;
        fd_divmod(ig, g.GW, t, gw);
        fd_divmod(t, g.GH, n, gh);
        const int dy = (int)(threadIdx.x / 3) - 1, dx = (int)(threadIdx.x % 3) - 1;
        const int nh = (int)gh + dy, nw = (int)gw + dx;
        int kind, row;
        if (nh < 0 || nh >= (int)g.GH.d || nw < 0 || nw >= (int)g.GW.d) { kind = 2; row = 0; }
        else if (dy == 0 && dx == 0) { kind = 0; row = (int)b; }
        else {
            const uint32_t g_in = (uint32_t)((int)ig + dx + (int)g.GW.d * dy);
            const int32_t idx = grid_idx[g_in];
            if (idx >= 0) { kind = 0; row = idx; }
            else { kind = 1; row = RING ? (int32_t)g_in : idx + (int32_t)g.n_total; }
        }
        nb_row[threadIdx.x] = row;
        nb_kind[threadIdx.x] = kind;
        if (threadIdx.x == 4) own_g = ig;
    }
    __syncthreads();

    const uint32_t bs = g.bs, p = g.pad, plane = bs * bs;
    T *__restrict__ out_t = out + (size_t)b * g.per_tile;
    const uint32_t stride = gridDim.x * WG;
    for (uint32_t f = blockIdx.x * WG + threadIdx.x; f < g.per_tile; f += stride) {
        uint32_t c, e, hp, wp;
        fd_divmod(f, g.PP, c, e);
        fd_divmod(e, g.BSP, hp, wp);
        const uint32_t sy = hp < p ? 0u : (hp >= p + bs ? 2u : 1u);
        const uint32_t sx = wp < p ? 0u : (wp >= p + bs ? 2u : 1u);
        const uint32_t s = sy * 3 + sx;
        const int kind = nb_kind[s];
        const uint32_t hs = hp - p + bs - sy * bs;  // sy=0: bs-p+hp ; 1: hp-p ; 2: hp-p-bs
        const uint32_t ws = wp - p + bs - sx * bs;
        const uint32_t in_tile = c * plane + hs * bs + ws;
        T val = 0;
        if (kind != 2) {
            const T *src = kind == 0 ? features : other_r;
            val = src[(size_t)(uint32_t)nb_row[s] * (g.C * plane) + in_tile];
        }
        out_t[f] = val;
        if (RING && s == 4 && (hs < p || hs >= bs - p || ws < p || ws >= bs - p))
            ring_w[(size_t)own_g * (g.C * plane) + in_tile] = val;
    }
}

// ------------------------------------------------------------------------------------------ index tables
// One 1024-thread workgroup: exclusive scan of the grid flags in raster order (wave ballot + LDS carry).
__global__ __launch_bounds__(1024) void k_grid_tables(const uint8_t *__restrict__ grid, int n_total,
                                                      int32_t *__restrict__ grid_idx,
                                                      int32_t *__restrict__ mapping_exec,
                                                      const int32_t *__restrict__ prev_grid_idx,
                                                      int32_t *__restrict__ transfer_idx,
                                                      int32_t *__restrict__ counts)
{
    __shared__ int32_t wave_cnt[16];
    __shared__ int32_t carry;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (int base = 0; base < n_total; base += 1024) {
        const int gidx = base + (int)threadIdx.x;
        const bool on = gidx < n_total && grid[gidx] != 0;
        const unsigned long long m = __ballot(on);
        const int before = __popcll(m & ((1ull << lane) - 1ull));
        if (lane == 0) wave_cnt[wave] = __popcll(m);
        __syncthreads();
        int woff = 0;
        for (int w = 0; w < wave; ++w) woff += wave_cnt[w];
        int chunk = 0;
        for (int w = 0; w < 16; ++w) chunk += wave_cnt[w];
        const int c0 = carry;
        if (gidx < n_total) {
            const int e = c0 + woff + before;  // executed tiles before this one (raster order)
            if (on) {
                grid_idx[gidx] = e;
                mapping_exec[e] = gidx;
            } else {
                const int k = gidx - e;        // non-executed tiles before this one
                grid_idx[gidx] = -n_total + k;
                if (prev_grid_idx != nullptr) transfer_idx[k] = prev_grid_idx[gidx];
            }
        }
        __syncthreads();
        if (threadIdx.x == 0) carry = c0 + chunk;
        __syncthreads();
    }
    if (threadIdx.x == 0) { counts[0] = carry; counts[1] = n_total - carry; }
}

// ------------------------------------------------------------------------------------------ per-tile bilinear resampling
template <typename T> struct Cvt;
template <> struct Cvt<float> {
    static __device__ __forceinline__ float ld(const float *p) { return *p; }
    static __device__ __forceinline__ float st(float v) { return v; }
};
template <> struct Cvt<__half> {
    static __device__ __forceinline__ float ld(const __half *p) { return __half2float(*p); }
    static __device__ __forceinline__ __half st(float v) { return __float2half(v); }
};
template <> struct Cvt<hip_bfloat16> {
    static __device__ __forceinline__ float ld(const hip_bfloat16 *p) { return (float)(*p); }
    static __device__ __forceinline__ hip_bfloat16 st(float v) { return hip_bfloat16(v); }
};

struct InterpGeom {
    FastDiv Wq, H;        // output quads per row, output height
    uint32_t h, w, W;
    uint32_t total;       // planes*H*Wq work items
    float rh, rw;
    int align;
};

__device__ __forceinline__ void src_index(float scale, uint32_t dst, int align, uint32_t size, uint32_t &i0, uint32_t &ip, float &l1)
{
    float s = align ? scale * (float)dst : fmaxf(scale * ((float)dst + 0.5f) - 0.5f, 0.0f);
    i0 = (uint32_t)s;
    if (i0 > size - 1) i0 = size - 1;
    ip = (i0 < size - 1) ? 1u : 0u;
    l1 = s - (float)i0;
}

// One work item = Q horizontally adjacent outputs of one plane row (Q*sizeof(T) = 16 B store when W % Q == 0);
// all planes/rows/columns in parallel (the stock kernel loops over planes inside each thread).
template <typename T, int Q>
__global__ __launch_bounds__(WG) void k_interp_bilinear(T *__restrict__ out, const T *__restrict__ in, InterpGeom g)
{
    const uint32_t stride = gridDim.x * WG;
    for (uint32_t i = blockIdx.x * WG + threadIdx.x; i < g.total; i += stride) {
        uint32_t r, q, plane, oy;
        fd_divmod(i, g.Wq, r, q);
        fd_divmod(r, g.H, plane, oy);
        uint32_t y0, yp; float ly1;
        src_index(g.rh, oy, g.align, g.h, y0, yp, ly1);
        const float ly0 = 1.0f - ly1;
        const T *__restrict__ row0 = in + ((size_t)plane * g.h + y0) * g.w;
        const T *__restrict__ row1 = row0 + (size_t)yp * g.w;
        T res[Q];
#pragma unroll
        for (int k = 0; k < Q; ++k) {
            const uint32_t ox = q * Q + k;
            uint32_t x0, xp; float lx1;
            src_index(g.rw, ox < g.W ? ox : g.W - 1, g.align, g.w, x0, xp, lx1);
            const float lx0 = 1.0f - lx1;
            const float v = ly0 * (lx0 * Cvt<T>::ld(row0 + x0) + lx1 * Cvt<T>::ld(row0 + x0 + xp)) +
                            ly1 * (lx0 * Cvt<T>::ld(row1 + x0) + lx1 * Cvt<T>::ld(row1 + x0 + xp));
            res[k] = Cvt<T>::st(v);
        }
        T *__restrict__ dst = out + ((size_t)plane * g.H.d + oy) * g.W + (size_t)q * Q;
        if (Q > 1 && (g.W % Q) == 0) {
            typedef typename VecOf<sizeof(T) * Q>::type V;
            *reinterpret_cast<V *>(dst) = *reinterpret_cast<const V *>(res);
        } else {
#pragma unroll
            for (int k = 0; k < Q; ++k)
                if (q * Q + k < g.W) dst[k] = res[k];
        }
    }
}

// ------------------------------------------------------------------------------------------ host helpers
int pick_vb(size_t row_bytes, std::initializer_list<const void *> ptrs)
{
    int vb = 16;
    while (vb > 1 && (row_bytes % vb) != 0) vb >>= 1;
    for (const void *p : ptrs) {
        if (p == nullptr) continue;
        while (vb > 1 && (reinterpret_cast<uintptr_t>(p) % vb) != 0) vb >>= 1;
    }
    return vb;
}

bool elem_ok(int e) { return e == 1 || e == 2 || e == 4 || e == 8; }

bool aligned(const void *p, int e) { return (reinterpret_cast<uintptr_t>(p) % (uintptr_t)e) == 0; }

int grid_for(uint64_t items, int per_thread)
{
    const uint64_t wgs = (items + (uint64_t)WG * per_thread - 1) / ((uint64_t)WG * per_thread);
    return (int)(wgs < 1 ? 1 : (wgs > MAX_WG ? MAX_WG : wgs));
}

// ---- per-op event timing (bench.py roofline: device time of exactly these launches, on their stream)
struct ProfState {
    std::mutex mu;
    unsigned mask = 0;
    struct Rec { hipEvent_t a, b; };
    std::vector<Rec> pending[BC_OP_COUNT];
    std::vector<Rec> pool;
    long long launches[BC_OP_COUNT] = {0};
    double ms[BC_OP_COUNT] = {0};
    double bytes[BC_OP_COUNT] = {0};
} g_prof;

struct ProfScope {
    int op;
    hipStream_t st;
    bool on;
    ProfState::Rec rec;
    ProfScope(int op_, hipStream_t st_, double bytes) : op(op_), st(st_), on(false)
    {
        if (!(g_prof.mask & (1u << op))) return;
        std::lock_guard<std::mutex> lk(g_prof.mu);
        if (!g_prof.pool.empty()) { rec = g_prof.pool.back(); g_prof.pool.pop_back(); }
        else {
            if (hipEventCreate(&rec.a) != hipSuccess) return;
            if (hipEventCreate(&rec.b) != hipSuccess) { (void)hipEventDestroy(rec.a); return; }
        }
        g_prof.bytes[op] += bytes;
        on = true;
        (void)hipEventRecord(rec.a, st);
    }
    ~ProfScope()
    {
        if (!on) return;
        (void)hipEventRecord(rec.b, st);
        std::lock_guard<std::mutex> lk(g_prof.mu);
        g_prof.pending[op].push_back(rec);
    }
};

int launch_status()
{
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? BC_OK : (int)e;
}

int check_dense(int N, int C, int H, int W, int bs, int E)
{
    if (!elem_ok(E)) return BC_ERR_ELEM;
    if (N <= 0 || C <= 0 || H <= 0 || W <= 0 || bs <= 0 || H % bs != 0 || W % bs != 0) return BC_ERR_SHAPE;
    if ((uint64_t)N * C * H * W >= (1ull << 31)) return BC_ERR_RANGE;
    return BC_OK;
}

template <bool TO_PACKED>
int launch_tiles(void *packed, void *dense, const int32_t *mapping_exec, int n_exec,
                 int N, int C, int H, int W, int bs, int E, hipStream_t st)
{
    const int vb = pick_vb((size_t)bs * E, {packed, dense});
    TileGeom g;
    const uint32_t vpr = (uint32_t)((size_t)bs * E / vb);
    g.vpr = make_fd(vpr); g.bs = make_fd(bs); g.C = make_fd(C); g.GW = make_fd(W / bs); g.GH = make_fd(H / bs);
    g.H = H; g.bsz = bs; g.vprW = (uint32_t)((size_t)W * E / vb);
    g.total = (uint32_t)((uint64_t)n_exec * C * bs * vpr);
    const int grid = grid_for(g.total, UNROLL);
#define BC_TILES(VB_)                                                                                          \
    case VB_:                                                                                                  \
        hipLaunchKernelGGL((k_tiles<VB_, TO_PACKED>), dim3(grid), dim3(WG), 0, st,                             \
                           (VecOf<VB_>::type *)packed, (const VecOf<VB_>::type *)packed,                       \
                           (VecOf<VB_>::type *)dense, (const VecOf<VB_>::type *)dense, mapping_exec, g);       \
        break;
    switch (vb) { BC_TILES(16) BC_TILES(8) BC_TILES(4) BC_TILES(2) BC_TILES(1) }
#undef BC_TILES
    return launch_status();
}

template <bool RING>
int launch_halo(void *out, const void *features, const void *other_r, void *ring_w, const int32_t *grid_idx,
                const int32_t *mapping_exec, int n_exec, int N, int C, int GH, int GW, int bs, int pad, int E,
                hipStream_t st)
{
    HaloGeom g;
    const uint32_t bsp = bs + 2 * pad;
    g.PP = make_fd(bsp * bsp); g.BSP = make_fd(bsp); g.GW = make_fd(GW); g.GH = make_fd(GH);
    g.C = C; g.bs = bs; g.pad = pad; g.n_total = (uint32_t)N * GH * GW;
    g.per_tile = (uint32_t)C * bsp * bsp;
    uint64_t gx = ((uint64_t)g.per_tile + WG * 4 - 1) / (WG * 4);
    // keep the launch at >= ~2048 workgroups when there are few tiles, <= 64 column chunks per tile
    if (gx > 64) gx = 64;
    if (gx < 1) gx = 1;
    const dim3 grid((unsigned)gx, (unsigned)n_exec);
#define BC_HALO(T_)                                                                                            \
    hipLaunchKernelGGL((k_halo<T_, RING>), grid, dim3(WG), 0, st, (T_ *)out, (const T_ *)features,             \
                       (const T_ *)other_r, (T_ *)ring_w, grid_idx, mapping_exec, g)
    switch (E) {
    case 1: BC_HALO(uint8_t); break;
    case 2: BC_HALO(uint16_t); break;
    case 4: BC_HALO(uint32_t); break;
    default: BC_HALO(uint64_t); break;
    }
#undef BC_HALO
    return launch_status();
}

int check_halo(const void *out, const void *features, const int32_t *grid_idx, const int32_t *mapping_exec,
               int n_exec, int N, int C, int GH, int GW, int bs, int pad, int E)
{
    if (!elem_ok(E)) return BC_ERR_ELEM;
    if (n_exec < 0 || N <= 0 || C <= 0 || GH <= 0 || GW <= 0 || bs <= 0 || pad < 1 || pad > bs) return BC_ERR_SHAPE;
    if (n_exec == 0) return BC_OK;
    if (!out || !features || !grid_idx || !mapping_exec) return BC_ERR_NULL;
    const uint64_t bsp = (uint64_t)bs + 2 * pad;
    if ((uint64_t)n_exec * C * bsp * bsp >= (1ull << 31)) return BC_ERR_RANGE;
    if ((uint64_t)N * GH * GW * C * bs * bs >= (1ull << 31)) return BC_ERR_RANGE;
    if (!aligned(out, E) || !aligned(features, E)) return BC_ERR_ALIGN;
    return BC_OK;
}

}  // namespace

// ================================================================================================ C ABI
BC_EXPORT int bc_abi_version(void) { return BC_ABI_VERSION; }

BC_EXPORT const char *bc_error_string(int code)
{
    switch (code) {
    case BC_OK: return "ok";
    case BC_ERR_NULL: return "required pointer is NULL";
    case BC_ERR_SHAPE: return "bad shape (non-positive dim, H/W not a multiple of the block size, or bad padding)";
    case BC_ERR_ELEM: return "elem_size must be 1, 2, 4 or 8";
    case BC_ERR_RANGE: return "tensor too large for 31-bit element indexing";
    case BC_ERR_ALIGN: return "pointer not aligned to elem_size";
    default: return code > 0 ? hipGetErrorString((hipError_t)code) : "unknown blockcopy error";
    }
}

BC_EXPORT const char *bc_op_name(int op)
{
    static const char *names[BC_OP_COUNT] = {"split", "combine", "transfer", "pad", "combine_copy", "pad_ring", "grid_tables", "interp"};
    return (op >= 0 && op < BC_OP_COUNT) ? names[op] : "?";
}

BC_EXPORT int bc_split(void *blocks, const void *image, const int32_t *mapping_exec, int n_exec,
                       int N, int C, int H, int W, int bs, int E, void *stream)
{
    int rc = check_dense(N, C, H, W, bs, E);
    if (rc != BC_OK) return rc;
    if (n_exec < 0) return BC_ERR_SHAPE;
    if (n_exec == 0) return BC_OK;
    if (!blocks || !image || !mapping_exec) return BC_ERR_NULL;
    if (!aligned(blocks, E) || !aligned(image, E)) return BC_ERR_ALIGN;
    ProfScope ps(BC_OP_SPLIT, (hipStream_t)stream, 2.0 * n_exec * C * bs * bs * E);
    return launch_tiles<true>(blocks, const_cast<void *>(image), mapping_exec, n_exec, N, C, H, W, bs, E, (hipStream_t)stream);
}

BC_EXPORT int bc_combine(const void *blocks, void *out, const int32_t *mapping_exec, int n_exec,
                         int N, int C, int H, int W, int bs, int E, void *stream)
{
    int rc = check_dense(N, C, H, W, bs, E);
    if (rc != BC_OK) return rc;
    if (n_exec < 0) return BC_ERR_SHAPE;
    if (n_exec == 0) return BC_OK;
    if (!blocks || !out || !mapping_exec) return BC_ERR_NULL;
    if (!aligned(blocks, E) || !aligned(out, E)) return BC_ERR_ALIGN;
    ProfScope ps(BC_OP_COMBINE, (hipStream_t)stream, 2.0 * n_exec * C * bs * bs * E);
    return launch_tiles<false>(const_cast<void *>(blocks), out, mapping_exec, n_exec, N, C, H, W, bs, E, (hipStream_t)stream);
}

BC_EXPORT int bc_combine_copy(const void *blocks, const void *prev, void *out, const int32_t *grid_idx,
                              int N, int C, int H, int W, int bs, int E, void *stream)
{
    int rc = check_dense(N, C, H, W, bs, E);
    if (rc != BC_OK) return rc;
    if (!out || !grid_idx) return BC_ERR_NULL;
    if (!blocks && !prev) return BC_ERR_NULL;
    if (!aligned(out, E) || !aligned(blocks, E) || !aligned(prev, E)) return BC_ERR_ALIGN;
    hipStream_t st = (hipStream_t)stream;
    // NULL sides are never dereferenced when the grid is all-executed / all-skipped; give the kernel a valid base
    const void *bl = blocks ? blocks : prev, *pv = prev ? prev : blocks;
    const int vb = pick_vb((size_t)bs * E, {bl, pv, out});
    DenseGeom g;
    const uint32_t vpr = (uint32_t)((size_t)bs * E / vb), vprW = (uint32_t)((size_t)W * E / vb);
    g.vprW = make_fd(vprW); g.H = make_fd(H); g.C = make_fd(C); g.vpr = make_fd(vpr); g.bs = make_fd(bs);
    g.GH = H / bs; g.GW = W / bs;
    g.total = (uint32_t)((uint64_t)N * C * H * vprW);
    const int grid = grid_for(g.total, UNROLL);
    ProfScope ps(BC_OP_COMBINE_COPY, st, 2.0 * N * C * H * W * E);
#define BC_CC(VB_)                                                                                             \
    case VB_:                                                                                                  \
        hipLaunchKernelGGL((k_combine_copy<VB_>), dim3(grid), dim3(WG), 0, st, (const VecOf<VB_>::type *)bl,   \
                           (const VecOf<VB_>::type *)pv, (VecOf<VB_>::type *)out, grid_idx, g);                \
        break;
    switch (vb) { BC_CC(16) BC_CC(8) BC_CC(4) BC_CC(2) BC_CC(1) }
#undef BC_CC
    return launch_status();
}

BC_EXPORT int bc_transfer(void *out, const void *prev_computed, const void *prev_transfer,
                          const int32_t *transfer_idx, int n_transfer,
                          int N, int C, int GH, int GW, int bs, int padding, int E, void *stream)
{
    if (!elem_ok(E)) return BC_ERR_ELEM;
    if (n_transfer < 0 || N <= 0 || C <= 0 || GH <= 0 || GW <= 0 || bs <= 0) return BC_ERR_SHAPE;
    if (n_transfer == 0) return BC_OK;
    if (!out || !transfer_idx) return BC_ERR_NULL;
    if (!prev_computed && !prev_transfer) return BC_ERR_NULL;
    if ((uint64_t)N * GH * GW * C * bs * bs >= (1ull << 31)) return BC_ERR_RANGE;
    if (!aligned(out, E) || !aligned(prev_computed, E) || !aligned(prev_transfer, E)) return BC_ERR_ALIGN;
    hipStream_t st = (hipStream_t)stream;
    const void *pc = prev_computed ? prev_computed : prev_transfer, *pt = prev_transfer ? prev_transfer : prev_computed;
    const int vb = pick_vb((size_t)bs * E, {out, pc, pt});
    TransferGeom g;
    const uint32_t vpr = (uint32_t)((size_t)bs * E / vb);
    g.vpr = make_fd(vpr); g.bs = make_fd(bs); g.C = make_fd(C);
    g.epv = vb / E; g.pad = padding; g.n_total = (uint32_t)N * GH * GW;
    g.total = (uint32_t)((uint64_t)n_transfer * C * bs * vpr);
    const int grid = grid_for(g.total, 1);
    const double ring = padding < 0 || bs <= 2 * padding ? (double)bs * bs : (double)bs * bs - (double)(bs - 2 * padding) * (bs - 2 * padding);
    ProfScope ps(BC_OP_TRANSFER, st, 2.0 * n_transfer * C * ring * E);
#define BC_TR(VB_)                                                                                             \
    case VB_:                                                                                                  \
        hipLaunchKernelGGL((k_transfer<VB_>), dim3(grid), dim3(WG), 0, st, (VecOf<VB_>::type *)out,            \
                           (const VecOf<VB_>::type *)pc, (const VecOf<VB_>::type *)pt, transfer_idx, g);       \
        break;
    switch (vb) { BC_TR(16) BC_TR(8) BC_TR(4) BC_TR(2) BC_TR(1) }
#undef BC_TR
    return launch_status();
}

static double halo_bytes(int n_exec, int C, int bs, int pad, int E)
{
    const double bsp = bs + 2.0 * pad;
    return 2.0 * n_exec * C * bsp * bsp * E;  // upper bound: zero-filled border halo is written but not read
}

BC_EXPORT int bc_pad(void *out, const void *features, const void *transfer, const int32_t *grid_idx,
                     const int32_t *mapping_exec, int n_exec,
                     int N, int C, int GH, int GW, int bs, int pad, int E, void *stream)
{
    int rc = check_halo(out, features, grid_idx, mapping_exec, n_exec, N, C, GH, GW, bs, pad, E);
    if (rc != BC_OK || n_exec == 0) return rc;
    if (!aligned(transfer, E)) return BC_ERR_ALIGN;
    ProfScope ps(BC_OP_PAD, (hipStream_t)stream, halo_bytes(n_exec, C, bs, pad, E));
    // transfer may be NULL/empty when every tile is executed (first frame): it is then never dereferenced
    return launch_halo<false>(out, features, transfer ? transfer : features, nullptr, grid_idx, mapping_exec, n_exec,
                              N, C, GH, GW, bs, pad, E, (hipStream_t)stream);
}

BC_EXPORT int bc_pad_ring(void *out, const void *features, void *ring, const int32_t *grid_idx,
                          const int32_t *mapping_exec, int n_exec,
                          int N, int C, int GH, int GW, int bs, int pad, int E, void *stream)
{
    int rc = check_halo(out, features, grid_idx, mapping_exec, n_exec, N, C, GH, GW, bs, pad, E);
    if (rc != BC_OK || n_exec == 0) return rc;
    if (!ring) return BC_ERR_NULL;
    if (!aligned(ring, E)) return BC_ERR_ALIGN;
    ProfScope ps(BC_OP_PAD_RING, (hipStream_t)stream, halo_bytes(n_exec, C, bs, pad, E));
    return launch_halo<true>(out, features, ring, ring, grid_idx, mapping_exec, n_exec, N, C, GH, GW, bs, pad, E,
                             (hipStream_t)stream);
}

BC_EXPORT int bc_interp_bilinear(void *out, const void *in, long long planes, int h, int w, int H, int W,
                                 int align_corners, float rh, float rw, int dtype, void *stream)
{
    if (dtype < BC_F32 || dtype > BC_BF16) return BC_ERR_ELEM;
    if (planes < 0 || h <= 0 || w <= 0 || H <= 0 || W <= 0) return BC_ERR_SHAPE;
    if (planes == 0) return BC_OK;
    if (!out || !in) return BC_ERR_NULL;
    const int E = dtype == BC_F32 ? 4 : 2;
    if ((uint64_t)planes * H * W >= (1ull << 31) || (uint64_t)planes * h * w >= (1ull << 31)) return BC_ERR_RANGE;
    if (!aligned(out, E) || !aligned(in, E)) return BC_ERR_ALIGN;
    hipStream_t st = (hipStream_t)stream;
    const int Q = 16 / E;
    // vector stores need a 16-byte aligned base and W % Q == 0 (otherwise the kernel stores element-wise)
    const bool vec = (W % Q) == 0 && aligned(out, 16);
    InterpGeom g;
    const uint32_t q = vec ? Q : 1;
    g.Wq = make_fd((W + q - 1) / q); g.H = make_fd(H);
    g.h = h; g.w = w; g.W = W; g.rh = rh; g.rw = rw; g.align = align_corners;
    g.total = (uint32_t)((uint64_t)planes * H * g.Wq.d);
    const int grid = grid_for(g.total, 1);
    ProfScope ps(BC_OP_INTERP, st, ((double)planes * h * w + (double)planes * H * W) * E);
#define BC_IP(T_, Q_) hipLaunchKernelGGL((k_interp_bilinear<T_, Q_>), dim3(grid), dim3(WG), 0, st, (T_ *)out, (const T_ *)in, g)
    if (dtype == BC_F32) { if (vec) BC_IP(float, 4); else BC_IP(float, 1); }
    else if (dtype == BC_F16) { if (vec) BC_IP(__half, 8); else BC_IP(__half, 1); }
    else { if (vec) BC_IP(hip_bfloat16, 8); else BC_IP(hip_bfloat16, 1); }
#undef BC_IP
    return launch_status();
}

BC_EXPORT int bc_grid_tables(const uint8_t *grid, int n_total, int32_t *grid_idx, int32_t *mapping_exec,
                             const int32_t *prev_grid_idx, int32_t *transfer_idx, int32_t *counts, void *stream)
{
    if (n_total <= 0) return BC_ERR_SHAPE;
    if (!grid || !grid_idx || !mapping_exec || !counts) return BC_ERR_NULL;
    if (prev_grid_idx && !transfer_idx) return BC_ERR_NULL;
    ProfScope ps(BC_OP_GRID_TABLES, (hipStream_t)stream, 9.0 * n_total);
    hipLaunchKernelGGL(k_grid_tables, dim3(1), dim3(1024), 0, (hipStream_t)stream, grid, n_total, grid_idx,
                       mapping_exec, prev_grid_idx, transfer_idx, counts);
    return launch_status();
}

BC_EXPORT int bc_grid_tables_host(const uint8_t *grid, int n_total, int32_t *grid_idx, int32_t *mapping_exec,
                                  const int32_t *prev_grid_idx, int32_t *transfer_idx)
{
    if (n_total <= 0) return BC_ERR_SHAPE;
    if (!grid || !grid_idx || !mapping_exec) return BC_ERR_NULL;
    if (prev_grid_idx && !transfer_idx) return BC_ERR_NULL;
    int n_exec = 0, n_tr = 0;
    for (int g = 0; g < n_total; ++g) {
        if (grid[g]) { grid_idx[g] = n_exec; mapping_exec[n_exec++] = g; }
        else {
            if (prev_grid_idx) transfer_idx[n_tr] = prev_grid_idx[g];
            grid_idx[g] = -n_total + n_tr++;
        }
    }
    return n_exec;
}

BC_EXPORT int bc_prof_enable(unsigned op_mask)
{
    std::lock_guard<std::mutex> lk(g_prof.mu);
    g_prof.mask = op_mask & ((1u << BC_OP_COUNT) - 1u);
    return BC_OK;
}

static void prof_drain_locked(int op)
{
    for (auto &r : g_prof.pending[op]) {
        float ms = 0.f;
        if (hipEventSynchronize(r.b) == hipSuccess && hipEventElapsedTime(&ms, r.a, r.b) == hipSuccess) {
            g_prof.ms[op] += ms;
            g_prof.launches[op] += 1;
        }
        g_prof.pool.push_back(r);
    }
    g_prof.pending[op].clear();
}

BC_EXPORT int bc_prof_reset(void)
{
    std::lock_guard<std::mutex> lk(g_prof.mu);
    for (int op = 0; op < BC_OP_COUNT; ++op) {
        prof_drain_locked(op);
        g_prof.launches[op] = 0;
        g_prof.ms[op] = 0;
        g_prof.bytes[op] = 0;
    }
    return BC_OK;
}

BC_EXPORT int bc_prof_read(int op, long long *launches, double *total_ms, double *total_bytes)
{
    if (op < 0 || op >= BC_OP_COUNT) return BC_ERR_SHAPE;
    std::lock_guard<std::mutex> lk(g_prof.mu);
    prof_drain_locked(op);
    if (launches) *launches = g_prof.launches[op];
    if (total_ms) *total_ms = g_prof.ms[op];
    if (total_bytes) *total_bytes = g_prof.bytes[op];
    return BC_OK;
}
