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
// Translation-unit slicing (see ConvV2Args below): no -DBC_PART = the whole library in one unit
#ifndef BC_PART
#define BC_MONO 1
#define BC_PART 0
#endif
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <hip/hip_fp16.h>
#include <hip/hip_bfloat16.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <mutex>
#include <type_traits>
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

// streaming (nontemporal) access on the native vector type of the same width
template <typename V> struct NativeOf { typedef V type; };
template <> struct NativeOf<uint4> { typedef uint32_t type __attribute__((ext_vector_type(4))); };
template <> struct NativeOf<uint2> { typedef uint32_t type __attribute__((ext_vector_type(2))); };
template <typename V> __device__ __forceinline__ V nt_load(const V *p)
{
    typedef typename NativeOf<V>::type T;
    const T t = __builtin_nontemporal_load(reinterpret_cast<const T *>(p));
    V v;
    __builtin_memcpy(&v, &t, sizeof(V));
    return v;
}
template <typename V> __device__ __forceinline__ void nt_store(const V &v, V *p)
{
    typedef typename NativeOf<V>::type T;
    T t;
    __builtin_memcpy(&t, &v, sizeof(V));
    __builtin_nontemporal_store(t, reinterpret_cast<T *>(p));
}

constexpr int WG = 256;          // 4 wavefronts
constexpr int MAX_WG = 256 * 8;  // 8 resident workgroups per CU on 256 CUs
constexpr int UNROLL = 4;
constexpr uint32_t SMALL_LAUNCH_VECTORS = 256u * 8u * 256u * 2u;   // < 2 vectors per resident lane of the chip

// ------------------------------------------------------------------------------------------ executed-tile count on the device
// A frame whose executed-tile count the HOST does not know when it launches (the policy decided on the device and nobody waited for
// it: bc_dyn_set) runs launches that are sized for a ceiling -- every tile executed -- and read the actual count from device memory:
// units of this launch = ceil(*ptr * num / den), never more than the host-side count.  Tile-indexed kernels (num / den = units per
// executed tile) let the surplus workgroups exit before they touch anything; pointwise kernels over packed rows may round up into a
// few garbage rows of the last tile, which nothing ever reads.  ptr == NULL: the host count is exact (every other frame).
struct DynCount {
    const int32_t *ptr = nullptr;
    uint32_t num = 1, den = 1;
};

__device__ __forceinline__ uint32_t dyn_units(const DynCount &d, uint32_t host_units)
{
    if (!d.ptr) return host_units;
    const unsigned long long n = ((unsigned long long)(uint32_t)*d.ptr * d.num + (d.den - 1)) / d.den;
    return n < host_units ? (uint32_t)n : host_units;
}

// ------------------------------------------------------------------------------------------ gather / scatter
struct TileGeom {
    FastDiv vpr, bs, C, GW, GH;  // vectors per tile row, tile size, channels, grid width/height
    uint32_t H, bsz, vprW;       // dense height, tile size, vectors per dense row
    uint32_t total;              // number of packed vectors = n_exec*C*bs*vpr
};

// TO_PACKED: packed[v] = dense[dense_index(v)]   (split);   else dense[dense_index(v)] = packed[v]   (combine)
// Each lane moves UNROLL vectors that are gridDim*WG apart (every memory instruction of a wave stays contiguous).
// All index loads are issued first, then all payload loads, then the stores: no branch sits between a load and
// its use, so the UNROLL requests of a lane are in flight together (tail lanes clamp their index).
template <int VB, bool TO_PACKED, int U>
__global__ __launch_bounds__(WG) void k_tiles(typename VecOf<VB>::type *__restrict__ packed_w,
                                              const typename VecOf<VB>::type *__restrict__ packed_r,
                                              typename VecOf<VB>::type *__restrict__ dense_w,
                                              const typename VecOf<VB>::type *__restrict__ dense_r,
                                              const int32_t *__restrict__ mapping_exec, TileGeom g, DynCount dyn)
{
    typedef typename VecOf<VB>::type V;
    const uint32_t stride = gridDim.x * WG;
    const uint32_t v0 = blockIdx.x * WG + threadIdx.x;
    const uint32_t total = dyn_units(dyn, g.total);
    if (blockIdx.x * WG >= total) return;            // (surplus workgroups of a ceiling-sized launch; also total == 0)
    const uint32_t last = total - 1;
    uint32_t vc[U], rem[U], b[U], ig[U], di[U];
    V val[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const uint32_t v = v0 + u * stride;
        vc[u] = v < last ? v : last;
        uint32_t r, xv, r2, h, c;
        fd_divmod(vc[u], g.vpr, r, xv);
        fd_divmod(r, g.bs, r2, h);
        fd_divmod(r2, g.C, b[u], c);
        rem[u] = (c * g.H + h) * g.vprW + xv;   // part of the dense index that does not depend on the tile position
    }
#pragma unroll
    for (int u = 0; u < U; ++u) ig[u] = (uint32_t)mapping_exec[b[u]];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        uint32_t t, gw, n, gh;
        fd_divmod(ig[u], g.GW, t, gw);
        fd_divmod(t, g.GH, n, gh);
        di[u] = rem[u] + (n * g.C.d * g.H + gh * g.bsz) * g.vprW + gw * g.vpr.d;
    }
#pragma unroll
    for (int u = 0; u < U; ++u) val[u] = TO_PACKED ? dense_r[di[u]] : packed_r[vc[u]];
    // tail lanes carry the clamped (last) vector and store it again: same value to the same address, so the
    // stores need no predicate (a predicate here makes the compiler sink each load into its store's branch).
#pragma unroll
    for (int u = 0; u < U; ++u) {
        if (TO_PACKED) packed_w[vc[u]] = val[u];
        else dense_w[di[u]] = val[u];
    }
}

// ------------------------------------------------------------------------------------------ fused scatter + copy
struct DenseGeom {
    FastDiv vprW, H, C, vpr, bs;  // vectors per dense row, dense height, channels, vectors per tile row, tile size
    uint32_t GH, GW;
    uint32_t total;               // N*C*H*vprW
};

// One output vector per lane, every lane live (exact grid, clamped tail), streaming (nontemporal) loads and stores:
// both inputs are read exactly once and the output is not re-read by this pipeline, so none of it should displace
// L2 / Infinity Cache lines.  Measured on MI355X with caches evicted before every launch (tools/kbench_cold.py,
// profiles/r01/kbench_cold_*.txt): 268 MB 73 -> 43 us, 80 MB 24 -> 15 us, 20 MB 7.7 -> 6.0 us against the 4-per-lane
// cached variant; with warm caches it is level or faster up to 2 GB.  (Deeper per-lane unrolling only helps the warm case.)
template <int VB>
__global__ __launch_bounds__(WG) void k_combine_copy(const typename VecOf<VB>::type *__restrict__ blocks,
                                                     long long prev_delta,   // (prev - blocks) in vectors
                                                     typename VecOf<VB>::type *__restrict__ out,
                                                     const int32_t *__restrict__ grid_idx, DenseGeom g)
{
    // Both sources are addressed from ONE base (blocks + signed offset): with two base pointers selected per lane
    // the compiler loses the no-alias information against `out` and serialises load -> store -> load.
    const uint32_t v = min(blockIdx.x * WG + threadIdx.x, g.total - 1);   // tail lanes redo the last vector (same value)
    uint32_t r, xw, r2, y, n, c, gw, xv, gh, h;
    fd_divmod(v, g.vprW, r, xw);
    fd_divmod(r, g.H, r2, y);
    fd_divmod(r2, g.C, n, c);
    fd_divmod(xw, g.vpr, gw, xv);
    fd_divmod(y, g.bs, gh, h);
    const int32_t idx = grid_idx[(n * g.GH + gh) * g.GW + gw];
    const uint32_t inner = (c * g.bs.d + h) * g.vpr.d + xv;   // vector offset inside a packed tile
    const long long off = idx >= 0 ? (long long)((uint32_t)idx * g.C.d * g.bs.d * g.vpr.d + inner) : prev_delta + (long long)v;
    nt_store(nt_load(blocks + off), out + v);
}

// The same pass as a hipGraph NODE.  A captured kernel node has frozen arguments, but the output of this op must be a fresh tensor
// every frame (callers may keep every frame's result, reference core/tensorwrapper.py:421-433) and `prev` is last frame's output, so
// both addresses change per replay.  This form reads them from a small device buffer the host refreshes with the frame's index
// tables (the one H->D copy the frame has anyway): slots[0] = prev, slots[1] = out, slots[2] = optional timing record (below).
// The launch then sits inside the frame's graph, directly behind the kernel that produced `blocks`, instead of starting cold behind
// the graph's end-of-launch release fence as the one eager launch of the frame.
//
// Timing record (measurement only; NULL in production): hipGraph kernel nodes cannot carry start/stop events, so when slots[2] is
// set, wave 0 of every workgroup stores {entry time, time its store was acknowledged} of the constant 100 MHz clock
// (s_memrealtime) into its own 16-byte cell stamp[blockIdx.x] -- plain stores, 78 KB per 19.9 MB launch.  (A first version kept
// min / max with two atomics per workgroup: ~10 k same-address device-scope atomics stretched the 6 us launch to 55 us.)
// bench.py hands every frame of the timed region its own record and reduces min(entry) / max(exit) afterwards.
template <int VB>
__global__ __launch_bounds__(WG) void k_combine_copy_ind(const typename VecOf<VB>::type *__restrict__ blocks,
                                                         const unsigned long long *__restrict__ slots,
                                                         const int32_t *__restrict__ grid_idx, DenseGeom g)
{
    typedef typename VecOf<VB>::type V;
    // (addresses formed as `blocks + offset`: a pointer cast from an integer would be a generic pointer and the accesses flat_*)
    const unsigned long long b_addr = reinterpret_cast<unsigned long long>(blocks);
    V *__restrict__ out = const_cast<V *>(blocks) + (long long)(slots[1] - b_addr) / (long long)sizeof(V);
    ulonglong2 *stamp = reinterpret_cast<ulonglong2 *>(slots[2]);
    unsigned long long t0 = 0;
    if (stamp) t0 = __builtin_amdgcn_s_memrealtime();
    const long long prev_delta = (long long)(slots[0] - b_addr) / (long long)sizeof(V);
    const uint32_t v = min(blockIdx.x * WG + threadIdx.x, g.total - 1);
    uint32_t r, xw, r2, y, n, c, gw, xv, gh, h;
    fd_divmod(v, g.vprW, r, xw);
    fd_divmod(r, g.H, r2, y);
    fd_divmod(r2, g.C, n, c);
    fd_divmod(xw, g.vpr, gw, xv);
    fd_divmod(y, g.bs, gh, h);
    const int32_t idx = grid_idx[(n * g.GH + gh) * g.GW + gw];
    const uint32_t inner = (c * g.bs.d + h) * g.vpr.d + xv;
    const long long off = idx >= 0 ? (long long)((uint32_t)idx * g.C.d * g.bs.d * g.vpr.d + inner) : prev_delta + (long long)v;
    nt_store(nt_load(blocks + off), out + v);
    if (stamp && threadIdx.x < 64) {
        __builtin_amdgcn_s_waitcnt(0);      // (the record wants the time the data was accepted by memory, not the issue time)
        if (threadIdx.x == 0) stamp[blockIdx.x] = make_ulonglong2(t0, (unsigned long long)__builtin_amdgcn_s_memrealtime());
    }
}

// The network-input stage of a graph-replayed frame: dst[tile] = src[tile] for every executed tile of two dense maps of ONE geometry
// (src = the caller's frame, dst = the persistent frame-state map), i.e. the reference's split + combine_ of the input
// (core/blockcopy.py:62-68: to_blocks -> combine_ -> frame_state) without the packed tensor in between and without a staging copy of
// the frame: the source ADDRESS is read from a device word at run time (`src_slot`, refreshed by the host with the frame's index
// tables), so the captured node serves whatever tensor the caller passes each frame.  2 * n_exec*C*bs^2*E bytes instead of the
// 2 * (N*C*H*W + 2 * n_exec*C*bs^2) * E of copy-in + gather + in-place scatter.  `n_exec_dev` (optional): executed-tile count read from
// device memory (<= the count the grid was sized for), for frames whose count the host does not know at launch time.
template <int VB>
__global__ __launch_bounds__(WG) void k_tile_copy_ind(typename VecOf<VB>::type *__restrict__ dst, const unsigned long long *__restrict__ src_slot,
                                                      const int32_t *__restrict__ mapping_exec, const int32_t *__restrict__ n_exec_dev,
                                                      uint32_t per_tile /* vectors per packed tile: C*bs*vpr */, TileGeom g)
{
    typedef typename VecOf<VB>::type V;
    uint32_t total = g.total;
    if (n_exec_dev) total = min(total, (uint32_t)*n_exec_dev * per_tile);
    if (total == 0) return;
    // (address formed as `dst + offset`: a pointer cast from an integer would be a generic pointer and the loads flat_*)
    const V *__restrict__ src = dst + (long long)(src_slot[0] - reinterpret_cast<unsigned long long>(dst)) / (long long)sizeof(V);
    const uint32_t v = min(blockIdx.x * WG + threadIdx.x, total - 1);      // tail lanes redo the last vector (same value)
    uint32_t r, xv, r2, h, c, b, t, gw, n, gh;
    fd_divmod(v, g.vpr, r, xv);
    fd_divmod(r, g.bs, r2, h);
    fd_divmod(r2, g.C, b, c);
    const uint32_t ig = (uint32_t)mapping_exec[b];
    fd_divmod(ig, g.GW, t, gw);
    fd_divmod(t, g.GH, n, gh);
    const uint32_t di = (c * g.H + h) * g.vprW + xv + (n * g.C.d * g.H + gh * g.bsz) * g.vprW + gw * g.vpr.d;
    dst[di] = nt_load(src + di);        // (the frame is read once; the state map is re-read by the stem conv right behind)
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

// ------------------------------------------------------------------------------------------ fused activation prologue
// Optional per-channel affine + ReLU applied to every gathered REAL value inside the halo gather
// (y = relu?(x*scale[c] + shift[c]), fp32 arithmetic): the BN->ReLU prologue of a pre-activation conv or the
// bias+ReLU epilogue of the producing conv, fused into the copy.  Zeros written beyond the image border stay zero
// (the padded op pads the ACTIVATED tensor).  The ring cache keeps ACTIVATED values -- what the padded op sees -- so a
// record is valid whichever route produced it (prologue here, or a producer that had materialised the activation
// already): values read from a ring record are never transformed again, values read from packed tiles always are.
// DT: 0 = none (pure copy, bit-exact), 1 = f32, 2 = f16, 3 = bf16.
struct Prologue {
    const float *scale;   // may be null (= 1)
    const float *shift;   // may be null (= 0)
    int relu;
};

template <int DT, typename T> struct ActCvt;
template <typename T> struct ActCvt<0, T> {
    static __device__ __forceinline__ T apply(T v, float, float, int) { return v; }
};
template <> struct ActCvt<1, uint32_t> {
    static __device__ __forceinline__ uint32_t apply(uint32_t v, float s, float t, int relu)
    {
        float x = __uint_as_float(v) * s + t;
        if (relu) x = fmaxf(x, 0.0f);
        return __float_as_uint(x);
    }
};
template <> struct ActCvt<2, uint16_t> {
    static __device__ __forceinline__ uint16_t apply(uint16_t v, float s, float t, int relu)
    {
        float x = __half2float(__ushort_as_half(v)) * s + t;
        if (relu) x = fmaxf(x, 0.0f);
        return __half_as_ushort(__float2half(x));
    }
};
template <> struct ActCvt<3, uint16_t> {
    static __device__ __forceinline__ uint16_t apply(uint16_t v, float s, float t, int relu)
    {
        float x = __uint_as_float((uint32_t)v << 16) * s + t;
        if (relu) x = fmaxf(x, 0.0f);
        hip_bfloat16 b(x);
        return *reinterpret_cast<uint16_t *>(&b);
    }
};

template <int DT> __device__ __forceinline__ void pro_coeffs(const Prologue &pr, uint32_t c, float &s, float &t)
{
    if (DT == 0) { s = 1.0f; t = 0.0f; return; }
    s = pr.scale ? pr.scale[c] : 1.0f;
    t = pr.shift ? pr.shift[c] : 0.0f;
}

// ------------------------------------------------------------------------------------------ halo gather
// ------------------------------------------------------------------------------------------ compact ring cache layout
// Per (grid position, channel) the ring cache stores only what a neighbour can ever need, as four contiguous segments
// (RS = 4*p*bs elements):   T = rows [0,p)        at 0            (row-major, bs wide)
//                           B = rows [bs-p,bs)    at p*bs
//                           L = cols [0,p)        at 2*p*bs       (row-major, p wide)
//                           R = cols [bs-p,bs)    at 2*p*bs+bs*p
// so an executed tile refreshes its ring with full-line contiguous stores (a dense (bs,bs) ring layout made every
// left/right column element its own partial cache line: +47 % write traffic by PMC), and a reader's halo column
// becomes a contiguous run.  Source element (hs, ws) of a ring-resident neighbour in direction (sy, sx)
// (0 = above/left, 1 = same row/col, 2 = below/right of the reader):
__device__ __forceinline__ uint32_t ring_elem(uint32_t sy, uint32_t sx, uint32_t hs, uint32_t ws, uint32_t bs, uint32_t p)
{
    if (sy == 0) return p * bs + (hs - (bs - p)) * bs + ws;      // reader's top halo  <- neighbour's B rows
    if (sy == 2) return hs * bs + ws;                            // reader's bottom halo <- neighbour's T rows
    if (sx == 0) return 2 * p * bs + bs * p + hs * p + (ws - (bs - p));   // left halo  <- neighbour's R cols
    return 2 * p * bs + hs * p + ws;                             // right halo <- neighbour's L cols
}

// refresh of the executed tile's own ring record from one of its elements (hs, ws); rec = ring + (g*C + c)*RS
template <typename T>
__device__ __forceinline__ void ring_store_elem(T *__restrict__ rec, uint32_t hs, uint32_t ws, uint32_t bs, uint32_t p, T v)
{
    if (hs < p) rec[hs * bs + ws] = v;
    if (hs >= bs - p) rec[p * bs + (hs - (bs - p)) * bs + ws] = v;
    if (ws < p) rec[2 * p * bs + hs * p + ws] = v;
    if (ws >= bs - p) rec[2 * p * bs + bs * p + hs * p + (ws - (bs - p))] = v;
}

struct HaloGeom {
    FastDiv PP, BSP, GW, GH;  // padded plane (bs+2p)^2, padded row bs+2p, grid dims
    uint32_t C, bs, pad, n_total;
    uint32_t per_tile;        // C*PP output elements per executed tile
    DynCount dyn;             // executed-tile count on the device (units = tiles = gridDim.y)
};

// One workgroup column (blockIdx.y) per executed tile; lanes run over the tile's contiguous output
// (C planes of (bs+2p)^2 elements).  The 3x3 neighbour table (which tensor, which row of it, or zero) is
// resolved once per workgroup into LDS.  RING=false: reference repad semantics (neighbour rows of the
// compacted `other` = transfer tensor).  RING=true: `other` is the persistent ring cache indexed by grid
// position (compact layout above), and the tile's own border ring is written back to it.
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
    if (b >= dyn_units(g.dyn, gridDim.y)) return;     // surplus workgroups of a ceiling-sized launch (bc_dyn_set)
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
        const uint32_t RS = 4 * p * bs;
        T val = 0;
        if (kind == 0) val = features[(size_t)(uint32_t)nb_row[s] * (g.C * plane) + in_tile];
        else if (kind == 1) {
            if (RING) val = other_r[((size_t)(uint32_t)nb_row[s] * g.C + c) * RS + ring_elem(sy, sx, hs, ws, bs, p)];
            else val = other_r[(size_t)(uint32_t)nb_row[s] * (g.C * plane) + in_tile];
        }
        out_t[f] = val;
        if (RING && s == 4) ring_store_elem(ring_w + ((size_t)own_g * g.C + c) * RS, hs, ws, bs, p, val);
    }
}


// ------------------------------------------------------------------------------------------ halo gather, LDS staged
// The padded row (bs+2p elements) is never 16-byte aligned with respect to its source row, so a register-only copy is
// stuck with element-wide accesses on one side.  Here a workgroup owns a CONTIGUOUS range of L output elements of one
// tile's padded output (C planes of (bs+2p)^2, back to back) and assembles it in LDS:
//   fill  : source rows are read with aligned VE-element vector loads (interior rows from the tile itself, top/bottom
//           halo rows from the vertical neighbours), the 2p edge elements per row with scalar loads; everything is
//           written to the LDS image at its final position (ds_write per element does the sub-vector shift);
//   drain : the image is read back as aligned 16-byte vectors and stored fully coalesced.  The image is placed in LDS
//           with the same 16-byte phase as the global range, so aligned LDS vectors map to aligned global vectors.
// Loads and LDS writes are unconditional (tail items are clamped, out-of-range elements go to a dummy LDS slot) so that
// each lane keeps all of its requests in flight at once.  RING: see k_halo.
struct HaloLdsGeom {
    FastDiv BSP, vpr, P2, GW, GH;   // padded row length, source vectors per row, 2*pad, grid dims
    uint32_t C, bs, pad, n_total;
    uint32_t per_tile;              // C*(bs+2p)^2
    uint32_t L;                     // output elements per workgroup
    uint32_t plane;                 // bs*bs
    DynCount dyn;                   // executed-tile count on the device (units = tiles = gridDim.y)
};

constexpr int HALO_UM = 4;          // middle-run vectors per lane
constexpr int HALO_UE = 2;          // edge elements per lane
constexpr int HALO_TBL = 128;       // bytes reserved at the start of dynamic LDS for the 3x3 neighbour table

template <typename T, int VE, bool RING, int DT>
__global__ __launch_bounds__(WG) void k_halo_lds(T *__restrict__ out, const T *__restrict__ features, long long other_delta,
                                                 T *__restrict__ ring_w, const int32_t *__restrict__ grid_idx,
                                                 const int32_t *__restrict__ mapping_exec, HaloLdsGeom g, Prologue pr)
{
    typedef typename VecOf<VE * sizeof(T)>::type SV;   // source vector
    constexpr int DE = 16 / sizeof(T);                 // elements per drain vector
    extern __shared__ uint4 smem[];
    long long *nb_base = reinterpret_cast<long long *>(smem);            // element offset from `features`, 9 entries
    int32_t *nb_zero = reinterpret_cast<int32_t *>(smem) + 18;           // 1 = beyond image border, 2 = ring-cache record
    uint32_t *own_g = reinterpret_cast<uint32_t *>(smem) + 27;
    T *img = reinterpret_cast<T *>(reinterpret_cast<char *>(smem) + HALO_TBL);

    const uint32_t b = blockIdx.y;
    if (b >= dyn_units(g.dyn, gridDim.y)) return;     // surplus workgroups of a ceiling-sized launch (bc_dyn_set)
    const uint32_t tile_elems = g.C * g.plane;
    if (threadIdx.x < 9) {
        const uint32_t ig = (uint32_t)mapping_exec[b];
        uint32_t t, gw, n, gh;
        fd_divmod(ig, g.GW, t, gw);
        fd_divmod(t, g.GH, n, gh);
        const int dy = (int)(threadIdx.x / 3) - 1, dx = (int)(threadIdx.x % 3) - 1;
        const int nh = (int)gh + dy, nw = (int)gw + dx;
        long long base = 0;
        int zero = 0;
        if (nh < 0 || nh >= (int)g.GH.d || nw < 0 || nw >= (int)g.GW.d) zero = 1;
        else if (dy == 0 && dx == 0) base = (long long)b * tile_elems;
        else {
            const uint32_t g_in = (uint32_t)((int)ig + dx + (int)g.GW.d * dy);
            const int32_t idx = grid_idx[g_in];
            if (idx >= 0) base = (long long)idx * tile_elems;
            else if (RING) { base = other_delta + (long long)g_in * g.C * (4 * g.pad * g.bs); zero = 2; }
            else base = other_delta + (long long)(uint32_t)(idx + (int32_t)g.n_total) * tile_elems;
        }
        nb_base[threadIdx.x] = base;
        nb_zero[threadIdx.x] = zero;
        if (threadIdx.x == 4) *own_g = ig;
    }
    __syncthreads();

    const uint32_t bs = g.bs, p = g.pad, BSP = g.BSP.d;
    const uint32_t f0 = blockIdx.x * g.L;
    const uint32_t f1 = min(f0 + g.L, g.per_tile);
    const uint32_t n = f1 - f0;
    T *__restrict__ out_t = out + (size_t)b * g.per_tile;
    const uint32_t ph = (uint32_t)((reinterpret_cast<uintptr_t>(out_t + f0) & 15u) / sizeof(T));
    const uint32_t dummy = ph + g.L;                 // scratch slot behind the image
    const uint32_t r_lo = fd_div(f0, g.BSP), r_hi = fd_div(f1 - 1, g.BSP);
    const uint32_t nrows = r_hi - r_lo + 1;
    const uint32_t RS = 4 * p * bs;
    const long long ring_base = RING ? (long long)(*own_g) * g.C * RS : 0;

    // ---- fill, ONE batch: the host sizes L so that a range spans at most HALO_UM*WG middle vectors and HALO_UE*WG
    // edge elements; every lane issues all its loads (HALO_UM vectors + HALO_UE scalars) before the first LDS write,
    // so the workgroup's critical path is: neighbour table -> one global round trip -> LDS -> stores.
    SV vec[HALO_UM];
    T edge[HALO_UE];
    uint32_t fm[HALO_UM], zm[HALO_UM], rsel[HALO_UM], fe[HALO_UE], ze[HALO_UE], rhs[HALO_UM], rws[HALO_UM];
    bool gm[HALO_UM], ge[HALO_UE];   // value comes from a ring record (already activated)
    float psm[HALO_UM], ptm[HALO_UM], pse[HALO_UE], pte[HALO_UE];
    long long roff[HALO_UM];
    {
        const uint32_t nitems = nrows * g.vpr.d;
#pragma unroll
        for (int u = 0; u < HALO_UM; ++u) {
            const uint32_t it = min(threadIdx.x + u * WG, nitems - 1);
            uint32_t ri, xv, c, hp;
            fd_divmod(it, g.vpr, ri, xv);
            const uint32_t r = r_lo + ri;
            fd_divmod(r, g.BSP, c, hp);
            pro_coeffs<DT>(pr, c, psm[u], ptm[u]);
            const uint32_t sy = hp < p ? 0u : (hp >= p + bs ? 2u : 1u);
            const uint32_t hs = hp - p + bs - sy * bs;
            const uint32_t s = sy * 3 + 1;
            const uint32_t in_tile = c * g.plane + hs * bs + xv * VE;
            const uint32_t z = (uint32_t)nb_zero[s];
            zm[u] = z == 1;
            gm[u] = z == 2;
            const long long src = z == 1 ? 0 : nb_base[s] + (z == 2 ? c * RS + ring_elem(sy, 1, hs, xv * VE, bs, p) : in_tile);
            vec[u] = *reinterpret_cast<const SV *>(features + src);
            fm[u] = r * BSP + p + xv * VE;
            roff[u] = ring_base + (long long)c * RS;
            rsel[u] = RING && sy == 1 && (hs < p || hs >= bs - p || xv * VE < p || xv * VE + VE > bs - p);
            rhs[u] = hs; rws[u] = xv * VE;
        }
    }
    {
        const uint32_t nitems = nrows * g.P2.d;
#pragma unroll
        for (int u = 0; u < HALO_UE; ++u) {
            const uint32_t it = min(threadIdx.x + u * WG, nitems - 1);
            uint32_t ri, e, c, hp;
            fd_divmod(it, g.P2, ri, e);
            const uint32_t r = r_lo + ri;
            fd_divmod(r, g.BSP, c, hp);
            pro_coeffs<DT>(pr, c, pse[u], pte[u]);
            const uint32_t sy = hp < p ? 0u : (hp >= p + bs ? 2u : 1u);
            const uint32_t hs = hp - p + bs - sy * bs;
            const bool right = e >= p;
            const uint32_t s = sy * 3 + (right ? 2u : 0u);
            const uint32_t ws = right ? e - p : bs - p + e;
            const uint32_t wp = right ? bs + e : e;        // p + bs + (e - p)
            const uint32_t z = (uint32_t)nb_zero[s];
            ze[u] = z == 1;
            ge[u] = z == 2;
            const long long src = z == 1 ? 0 : nb_base[s] + (z == 2 ? c * RS + ring_elem(sy, right ? 2u : 0u, hs, ws, bs, p)
                                                                       : c * g.plane + hs * bs + ws);
            edge[u] = features[src];
            fe[u] = r * BSP + wp;
        }
    }
#pragma unroll
    for (int u = 0; u < HALO_UM; ++u) {
        SV av = vec[u];
        T *e = reinterpret_cast<T *>(&av);
        if (DT != 0 && !gm[u]) {
#pragma unroll
            for (int k = 0; k < VE; ++k) e[k] = ActCvt<DT, T>::apply(e[k], psm[u], ptm[u], pr.relu);
        }
#pragma unroll
        for (int k = 0; k < VE; ++k) {
            const uint32_t f = fm[u] + k;
            const uint32_t j = (f >= f0 && f < f1) ? ph + (f - f0) : dummy;
            img[j] = zm[u] ? (T)0 : e[k];
        }
        if (RING && rsel[u]) {   // refresh the tile's own compact ring record; the ring keeps ACTIVATED values
            T *rec = ring_w + roff[u];
            if (rhs[u] < p) *reinterpret_cast<SV *>(rec + rhs[u] * bs + rws[u]) = av;
            if (rhs[u] >= bs - p) *reinterpret_cast<SV *>(rec + p * bs + (rhs[u] - (bs - p)) * bs + rws[u]) = av;
#pragma unroll
            for (int k = 0; k < VE; ++k) {
                const uint32_t w = rws[u] + k;
                if (w < p) rec[2 * p * bs + rhs[u] * p + w] = e[k];
                if (w >= bs - p) rec[2 * p * bs + bs * p + rhs[u] * p + (w - (bs - p))] = e[k];
            }
        }
    }
#pragma unroll
    for (int u = 0; u < HALO_UE; ++u) {
        const uint32_t f = fe[u];
        const uint32_t j = (f >= f0 && f < f1) ? ph + (f - f0) : dummy;
        img[j] = ze[u] ? (T)0 : (ge[u] ? edge[u] : ActCvt<DT, T>::apply(edge[u], pse[u], pte[u], pr.relu));
    }
    __syncthreads();

    // ---- drain: aligned 16-byte vectors (element-wise at the two ragged ends of the range)
    {
        const uint32_t nv = (ph + n + DE - 1) / DE;
        const uint4 *__restrict__ img_v = reinterpret_cast<const uint4 *>(img);
        uint4 *__restrict__ dst_v = reinterpret_cast<uint4 *>(reinterpret_cast<uintptr_t>(out_t + f0) & ~(uintptr_t)15);
        for (uint32_t i = threadIdx.x; i < nv; i += WG) {
            const uint32_t lo = i * DE;
            if (lo >= ph && lo + DE <= ph + n) {
                dst_v[i] = img_v[i];
            } else {
#pragma unroll
                for (int k = 0; k < DE; ++k) {
                    const uint32_t j = lo + k;
                    if (j >= ph && j < ph + n) out_t[f0 + (j - ph)] = img[j];
                }
            }
        }
    }
}


// ------------------------------------------------------------------------------------------ halo gather, row vectors
// Register-only form.  A padded row = [p edge | bs middle | p edge].  The middle run is a whole row of some tile
// (the tile itself, or its upper/lower neighbour for the 2p halo rows), so it is read with ALIGNED VE-element vector
// loads; its destination is shifted by p elements, so it is written with UNDER-ALIGNED vector stores of the same
// width (gfx950 global memory runs in unaligned-access mode: a dwordx4 store only needs element alignment; one
// wavefront still covers one contiguous run of addresses).  The 2p edge elements per row move as scalars.
// The 3x3 neighbour table is wave-uniform (it depends on blockIdx.y only): mapping_exec[b] and the eight grid_idx
// entries are scalar loads into SGPRs -- no LDS, no barrier, and nothing between a lane and its HALO_UM + HALO_UE
// independent requests.
struct HaloRowsGeom {
    FastDiv BSP, vpr, P2, GW, GH;
    uint32_t C, bs, pad, n_total, plane, per_tile;
    uint32_t mid_items;     // C*BSP*vpr   source vectors per tile
    uint32_t edge_items;    // C*BSP*2p    edge elements per tile
    uint32_t edge_per_wg;
    DynCount dyn;           // executed-tile count on the device (units = tiles = gridDim.y)
};

template <typename T, int VE> struct RowVec {
    typedef T aligned_t __attribute__((ext_vector_type(VE)));
    typedef T packed_t __attribute__((ext_vector_type(VE), aligned(sizeof(T))));
};
template <typename T> struct RowVec<T, 1> {
    typedef T aligned_t;
    typedef T packed_t;
};

template <typename T, int VE, bool RING, int DT>
__global__ __launch_bounds__(WG) void k_halo_rows(T *__restrict__ out, const T *__restrict__ features, long long other_delta,
                                                  T *__restrict__ ring_w, const int32_t *__restrict__ grid_idx,
                                                  const int32_t *__restrict__ mapping_exec, HaloRowsGeom g, Prologue pr)
{
    typedef typename RowVec<T, VE>::aligned_t SV;
    typedef typename RowVec<T, VE>::packed_t SVU;
    const uint32_t b = blockIdx.y;
    if (b >= dyn_units(g.dyn, gridDim.y)) return;     // surplus workgroups of a ceiling-sized launch (bc_dyn_set)
    const uint32_t tile_elems = g.C * g.plane;
    const uint32_t bs = g.bs, p = g.pad, BSP = g.BSP.d;

    // ---- wave-uniform 3x3 neighbour table (scalar loads)
    const uint32_t ig = (uint32_t)mapping_exec[b];
    uint32_t t0, gw, n0, gh;
    fd_divmod(ig, g.GW, t0, gw);
    fd_divmod(t0, g.GH, n0, gh);
    const uint32_t RS = 4 * p * bs;
    long long nbb[9];
    bool nbz[9], nbr[9];     // beyond the image border / ring-cache record (compact layout)
#pragma unroll
    for (int k = 0; k < 9; ++k) {
        const int dy = k / 3 - 1, dx = k % 3 - 1;
        const int nh = (int)gh + dy, nw = (int)gw + dx;
        nbz[k] = nh < 0 || nh >= (int)g.GH.d || nw < 0 || nw >= (int)g.GW.d;
        nbr[k] = false;
        nbb[k] = 0;
        if (k == 4) nbb[k] = (long long)b * tile_elems;
        else if (!nbz[k]) {
            const uint32_t g_in = (uint32_t)((int)ig + dx + (int)g.GW.d * dy);
            const int32_t idx = grid_idx[g_in];
            if (idx >= 0) nbb[k] = (long long)idx * tile_elems;
            else if (RING) { nbb[k] = other_delta + (long long)g_in * g.C * RS; nbr[k] = true; }
            else nbb[k] = other_delta + (long long)(uint32_t)(idx + (int32_t)g.n_total) * tile_elems;
        }
    }
    const long long ring_base = (long long)ig * g.C * RS;
    T *__restrict__ out_t = out + (size_t)b * g.per_tile;

    // ---- middle runs
    {
        SV vec[HALO_UM];
        uint32_t dst[HALO_UM], rc[HALO_UM], rhs[HALO_UM], rws[HALO_UM];
        bool zero[HALO_UM], rsel[HALO_UM], fr[HALO_UM];
        float ps[HALO_UM], pt[HALO_UM];
#pragma unroll
        for (int u = 0; u < HALO_UM; ++u) {
            const uint32_t it = min((blockIdx.x * HALO_UM + u) * WG + threadIdx.x, g.mid_items - 1);
            uint32_t rr, xv, c, hp;
            fd_divmod(it, g.vpr, rr, xv);
            fd_divmod(rr, g.BSP, c, hp);
            pro_coeffs<DT>(pr, c, ps[u], pt[u]);
            const bool top = hp < p, bot = hp >= p + bs;
            const uint32_t hs = top ? hp + bs - p : (bot ? hp - p - bs : hp - p);
            const uint32_t in_tile = c * g.plane + hs * bs + xv * VE;
            zero[u] = top ? nbz[1] : (bot ? nbz[7] : false);
            const bool from_ring = top ? nbr[1] : (bot ? nbr[7] : false);
            const long long base = top ? nbb[1] : (bot ? nbb[7] : nbb[4]);
            const uint32_t off = from_ring ? c * RS + ring_elem(top ? 0u : 2u, 1, hs, xv * VE, bs, p) : in_tile;
            vec[u] = *reinterpret_cast<const SV *>(features + (zero[u] ? 0 : base + off));
            fr[u] = from_ring;
            dst[u] = rr * BSP + p + xv * VE;
            rsel[u] = RING && !top && !bot && (hs < p || hs >= bs - p || xv * VE < p || xv * VE + VE > bs - p);
            rc[u] = c; rhs[u] = hs; rws[u] = xv * VE;
        }
#pragma unroll
        for (int u = 0; u < HALO_UM; ++u) {
            SV v = vec[u];
            if (DT != 0 && !fr[u]) {   // ring records are activated already
                T *e = reinterpret_cast<T *>(&v);
#pragma unroll
                for (int k = 0; k < VE; ++k) e[k] = ActCvt<DT, T>::apply(e[k], ps[u], pt[u], pr.relu);
            }
            if (RING && rsel[u]) {   // refresh the tile's own compact ring record; the ring keeps ACTIVATED values
                T *rec = ring_w + ring_base + (long long)rc[u] * RS;
                if (rhs[u] < p) *reinterpret_cast<SV *>(rec + rhs[u] * bs + rws[u]) = v;
                if (rhs[u] >= bs - p) *reinterpret_cast<SV *>(rec + p * bs + (rhs[u] - (bs - p)) * bs + rws[u]) = v;
                const T *e = reinterpret_cast<const T *>(&v);
#pragma unroll
                for (int k = 0; k < VE; ++k) {
                    const uint32_t w = rws[u] + k;
                    if (w < p) rec[2 * p * bs + rhs[u] * p + w] = e[k];
                    if (w >= bs - p) rec[2 * p * bs + bs * p + rhs[u] * p + (w - (bs - p))] = e[k];
                }
            }
            if (zero[u]) v = (SV)0;
            *reinterpret_cast<SVU *>(out_t + dst[u]) = v;
        }
    }
    // ---- edge elements (left/right neighbours and the four corners)
    {
        const uint32_t e0 = blockIdx.x * g.edge_per_wg;
        const uint32_t e1 = min(e0 + g.edge_per_wg, g.edge_items);
        for (uint32_t base_it = e0; base_it < e1; base_it += WG * HALO_UE) {
            T val[HALO_UE];
            uint32_t dst[HALO_UE];
            bool zero[HALO_UE], fr[HALO_UE];
            float ps[HALO_UE], pt[HALO_UE];
#pragma unroll
            for (int u = 0; u < HALO_UE; ++u) {
                const uint32_t it = min(base_it + u * WG + threadIdx.x, e1 - 1);
                uint32_t rr, e, c, hp;
                fd_divmod(it, g.P2, rr, e);
                fd_divmod(rr, g.BSP, c, hp);
                pro_coeffs<DT>(pr, c, ps[u], pt[u]);
                const bool top = hp < p, bot = hp >= p + bs, right = e >= p;
                const uint32_t hs = top ? hp + bs - p : (bot ? hp - p - bs : hp - p);
                const uint32_t ws = right ? e - p : bs - p + e;
                const long long bl = top ? nbb[0] : (bot ? nbb[6] : nbb[3]);
                const long long br = top ? nbb[2] : (bot ? nbb[8] : nbb[5]);
                const bool zl = top ? nbz[0] : (bot ? nbz[6] : nbz[3]);
                const bool zr = top ? nbz[2] : (bot ? nbz[8] : nbz[5]);
                const bool rl = top ? nbr[0] : (bot ? nbr[6] : nbr[3]);
                const bool rr_ = top ? nbr[2] : (bot ? nbr[8] : nbr[5]);
                zero[u] = right ? zr : zl;
                const bool from_ring = right ? rr_ : rl;
                const uint32_t off = from_ring ? c * RS + ring_elem(top ? 0u : (bot ? 2u : 1u), right ? 2u : 0u, hs, ws, bs, p)
                                               : c * g.plane + hs * bs + ws;
                const long long src = (right ? br : bl) + off;
                val[u] = features[zero[u] ? 0 : src];
                fr[u] = from_ring;
                dst[u] = rr * BSP + (right ? bs + e : e);
            }
#pragma unroll
            for (int u = 0; u < HALO_UE; ++u)
                out_t[dst[u]] = zero[u] ? (T)0 : (fr[u] ? val[u] : ActCvt<DT, T>::apply(val[u], ps[u], pt[u], pr.relu));
        }
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

// ------------------------------------------------------------------------------------------ device policy step
// Decision of the online-RL policy for one frame, entirely on the device: Bernoulli sampling of
// the tile logits, rounding the executed count UP to a multiple (switching on randomly chosen skipped tiles) and the index
// tables -- the reference does this with a D->H copy, Python `random.sample` and a CPU TorchScript function
// (policy/policy.py:124-144,283-288; core/tensorwrapper.py:108-128).  Randomness is COUNTER-BASED so that the CPU
// restatement (oracle/bc_oracle.c bc_oracle_policy_step) reproduces every decision bit for bit:
//     z(stream) = splitmix64( seed ^ counter * 0xD1342543DE82EF95 + (2 * tile + stream + 1) * 0x9E3779B97F4A7C15 )
//     stream 0: u = (z >> 40) * 2^-24, tile sampled iff u < sigmoid(logit)      (torch: Bernoulli(logits).sample() = rand < p)
//     stream 1: key = z >> 24; of the skipped tiles the `need` smallest (key, tile) are switched on
// and sigmoid() is evaluated with one fixed sequence of IEEE operations (bc_sigmoid_repro) instead of the library expf.
__host__ __device__ __forceinline__ unsigned long long bc_policy_rand(unsigned long long seed, unsigned long long counter,
                                                                       unsigned int tile, unsigned int stream)
{
    unsigned long long z = (seed ^ (counter * 0xD1342543DE82EF95ull)) + (2ull * tile + stream + 1ull) * 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

__device__ __forceinline__ float bc_sigmoid_repro(float x)
{
    float t = -x * 1.44269504f;                    // single rounding
    t = fminf(fmaxf(t, -126.0f), 126.0f);
    const float nf = rintf(t);
    const float f = t - nf;                        // exact, |f| <= 0.5
    float p = 1.54035304e-4f;                      // 2^f = sum (f ln 2)^k / k!, Horner with fused multiply-adds
    p = fmaf(p, f, 1.33335581e-3f);
    p = fmaf(p, f, 9.61812911e-3f);
    p = fmaf(p, f, 5.55041087e-2f);
    p = fmaf(p, f, 2.40226507e-1f);
    p = fmaf(p, f, 6.93147181e-1f);
    p = fmaf(p, f, 1.0f);
    const float e = __uint_as_float(__float_as_uint(p) + ((uint32_t)(int32_t)nf << 23));   // p * 2^nf
    return __fdiv_rn(1.0f, 1.0f + e);
}

constexpr int POLICY_MAX_TILES = 8192;

__global__ __launch_bounds__(1024) void k_policy_step(const float *__restrict__ logits, int n_total, unsigned long long seed,
                                                      unsigned long long counter, int multiple, int at_least_one,
                                                      uint8_t *__restrict__ grid, int32_t *__restrict__ grid_idx,
                                                      int32_t *__restrict__ mapping_exec, int32_t *__restrict__ counts,
                                                      volatile int32_t *__restrict__ mailbox)
{
    __shared__ unsigned long long key[POLICY_MAX_TILES];    // (key40 << 16 | tile) of skipped tiles, ~0 for sampled ones
    __shared__ int32_t wave_cnt[16];
    __shared__ int32_t s_n, s_nan, carry;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) { s_n = 0; s_nan = 0; carry = 0; }
    __syncthreads();
    // ---- 1. Bernoulli sample
    int my_n = 0, my_nan = 0;
    for (int i = tid; i < n_total; i += 1024) {
        const float x = logits[i];
        const bool nan = x != x;
        const float u = (float)(bc_policy_rand(seed, counter, (unsigned)i, 0u) >> 40) * 5.9604644775390625e-8f;
        const bool on = !nan && u < bc_sigmoid_repro(x);
        my_n += on;
        my_nan |= nan;
        key[i] = on ? ~0ull : (((bc_policy_rand(seed, counter, (unsigned)i, 1u) >> 24) << 16) | (unsigned long long)i);
    }
    if (my_n) atomicAdd(&s_n, my_n);
    if (my_nan) atomicOr(&s_nan, 1);
    __syncthreads();
    int n = s_n;
    if (at_least_one && n == 0) {
        if (tid == 0) key[0] = ~0ull;
        n = 1;
    }
    __syncthreads();
    // ---- 2. round the executed count up to a multiple: the `need` skipped tiles with the smallest keys are switched on
    int rounded = 0;
    if (n > 0) {
        rounded = multiple * (1 + (n - 1) / multiple);
        if (rounded > n_total) rounded = n_total;
    }
    const int need = rounded - n;
    if (need > 0) {
        bool add[POLICY_MAX_TILES / 1024];
#pragma unroll
        for (int r = 0; r < POLICY_MAX_TILES / 1024; ++r) {
            const int i = tid + r * 1024;
            add[r] = false;
            if (i < n_total && key[i] != ~0ull) {
                const unsigned long long mine = key[i];
                int rank = 0;
                for (int j = 0; j < n_total; ++j) rank += key[j] < mine;     // sampled tiles hold ~0: never smaller
                add[r] = rank < need;
            }
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < POLICY_MAX_TILES / 1024; ++r)
            if (add[r]) key[tid + r * 1024] = ~0ull;
        __syncthreads();
    }
    // ---- 3. index tables (as k_grid_tables) + the bool grid
    for (int base = 0; base < n_total; base += 1024) {
        const int gidx = base + tid;
        const bool on = gidx < n_total && key[gidx] == ~0ull;
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
            const int e = c0 + woff + before;
            grid[gidx] = on ? 1 : 0;
            if (on) { grid_idx[gidx] = e; mapping_exec[e] = gidx; }
            else grid_idx[gidx] = -n_total + (gidx - e);
        }
        __syncthreads();
        if (tid == 0) carry = c0 + chunk;
        __syncthreads();
    }
    if (tid == 0) {
        counts[0] = carry; counts[1] = n; counts[2] = s_nan; counts[3] = (int32_t)counter;
        if (mailbox) { mailbox[0] = carry; mailbox[1] = n; mailbox[2] = s_nan; mailbox[3] = (int32_t)counter; }
    }
}

// ------------------------------------------------------------------------------------------ policy-net input features
// The policy net's input (reference policy/net.py:82-113): frame, frame state, previous output scores and previous grid, each
// brought to ONE low resolution by nearest-neighbour resampling and stacked along channels (scores and grid centred by -0.5).
// The reference (and the first version here) runs four F.interpolate launches, casts, two subtractions and a concat; this is
// one gather.  Source index exactly as ATen's legacy 'nearest': min((int)floorf(dst * scale), in - 1) with a float32 scale.
struct FeatSrc {
    const void *ptr;
    long long sn, sc, sh, sw;   // element strides
    int C, H, W, dtype;         // dtype: BC_F32 / BC_F16 / BC_BF16, 3 = uint8 / bool
    float scale_h, scale_w, offset;
};

struct FeatGeom {
    FeatSrc src[4];
    int N, h, w, Ctot;
};

__device__ __forceinline__ float feat_load(const FeatSrc &s, long long off)
{
    switch (s.dtype) {
    case 0: return reinterpret_cast<const float *>(s.ptr)[off];
    case 1: return __half2float(reinterpret_cast<const __half *>(s.ptr)[off]);
    case 2: return __uint_as_float((uint32_t)reinterpret_cast<const uint16_t *>(s.ptr)[off] << 16);
    default: return reinterpret_cast<const uint8_t *>(s.ptr)[off] ? 1.0f : 0.0f;
    }
}

__global__ __launch_bounds__(WG) void k_policy_features(float *__restrict__ out, FeatGeom g)
{
    const uint32_t total = (uint32_t)g.N * g.Ctot * g.h * g.w;
    for (uint32_t i = blockIdx.x * WG + threadIdx.x; i < total; i += gridDim.x * WG) {
        const uint32_t x = i % g.w, y = (i / g.w) % g.h, c = (i / (g.w * g.h)) % g.Ctot, n = i / (g.w * g.h * g.Ctot);
        uint32_t cc = c;
        int k = 0;
        while (k < 3 && cc >= (uint32_t)g.src[k].C) { cc -= g.src[k].C; ++k; }
        const FeatSrc &s = g.src[k];
        int sy = (int)floorf((float)y * s.scale_h), sx = (int)floorf((float)x * s.scale_w);
        sy = sy < s.H - 1 ? sy : s.H - 1;
        sx = sx < s.W - 1 ? sx : s.W - 1;
        out[i] = feat_load(s, n * s.sn + cc * s.sc + sy * s.sh + sx * s.sw) + s.offset;
    }
}

// ------------------------------------------------------------------------------------------ per-tile bilinear resampling
template <typename T> struct Cvt;
template <> struct Cvt<float> {
    static __device__ __forceinline__ float ld(const float *p) { return *p; }
    static __device__ __forceinline__ float st(float v) { return v; }
    static __device__ __forceinline__ float ld_round(float v) { return v; }
};
template <> struct Cvt<__half> {
    static __device__ __forceinline__ float ld(const __half *p) { return __half2float(*p); }
    static __device__ __forceinline__ __half st(float v) { return __float2half(v); }
    static __device__ __forceinline__ float ld_round(float v) { return __half2float(__float2half(v)); }
};
template <> struct Cvt<hip_bfloat16> {
    static __device__ __forceinline__ float ld(const hip_bfloat16 *p) { return (float)(*p); }
    static __device__ __forceinline__ hip_bfloat16 st(float v) { return hip_bfloat16(v); }
    static __device__ __forceinline__ float ld_round(float v) { return (float)hip_bfloat16(v); }
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


// ------------------------------------------------------------------------------------------ fused elementwise epilogue
// out = relu?(in * scale[c] + shift[c] + add) over a packed (B, C, h, w) tensor, fp32 arithmetic: replaces the separate
// bias-add / batch-norm / residual-add / ReLU launches between two convs with ONE pass.
struct AffineGeom {
    FastDiv hwq, C;     // vectors per plane, channels
    uint32_t total;     // B*C*hwq
};

template <typename T, int Q>
__global__ __launch_bounds__(WG) void k_affine_act(T *__restrict__ out, const T *__restrict__ in, const T *__restrict__ add,
                                                   const float *__restrict__ scale, const float *__restrict__ shift,
                                                   int relu, AffineGeom g)
{
    typedef typename VecOf<sizeof(T) * Q>::type V;
    const uint32_t i = blockIdx.x * WG + threadIdx.x;
    if (i >= g.total) return;
    uint32_t pl, q, b, c;
    fd_divmod(i, g.hwq, pl, q);
    fd_divmod(pl, g.C, b, c);
    const float s = scale ? scale[c] : 1.0f, t = shift ? shift[c] : 0.0f;
    V vi = reinterpret_cast<const V *>(in)[i];
    V va = vi;
    if (add) va = reinterpret_cast<const V *>(add)[i];
    const T *xi = reinterpret_cast<const T *>(&vi), *xa = reinterpret_cast<const T *>(&va);
    T res[Q];
#pragma unroll
    for (int k = 0; k < Q; ++k) {
        float x = Cvt<T>::ld(xi + k) * s + t;
        if (add) x += Cvt<T>::ld(xa + k);
        if (relu) x = fmaxf(x, 0.0f);
        res[k] = Cvt<T>::st(x);
    }
    reinterpret_cast<V *>(out)[i] = *reinterpret_cast<const V *>(res);
}

// VE consecutive per-channel coefficients starting at channel c0 (c0 % VE == 0): float4 loads when VE is a multiple of 4
template <int VE>
__device__ __forceinline__ void load_coeffs(const float *__restrict__ p, uint32_t c0, float fill, float (&v)[VE])
{
    if (p == nullptr) {
#pragma unroll
        for (int j = 0; j < VE; ++j) v[j] = fill;
    } else if (VE % 4 == 0) {
#pragma unroll
        for (int j = 0; j < VE / 4; ++j) {
            const float4 q = reinterpret_cast<const float4 *>(p + c0)[j];
            v[4 * j] = q.x; v[4 * j + 1] = q.y; v[4 * j + 2] = q.z; v[4 * j + 3] = q.w;
        }
    } else {
#pragma unroll
        for (int j = 0; j < VE; ++j) v[j] = p[c0 + j];
    }
}

// ========================================================================================== channels-last (NHWC) forms
// In channels-last memory a pixel is one contiguous run of C*E bytes ("fat pixel" = K vectors of VB bytes), a tile row is
// bs fat pixels, a packed tile is fully contiguous, and the halo shift p*C*E is a whole number of vectors: every access of
// the halo gather is an ALIGNED vector and there is no middle/edge distinction.  (MIOpen's fastest fp16 conv kernels are
// NHWC too: the packed 3x3 conv runs in 14 us instead of 39 us.)  Gather / scatter / scatter+copy need no new kernel: a
// channels-last (N,C,H,W) map is an NCHW (N,1,H,W) map of fat elements, which the kernels above already handle.
struct HaloNhwcGeom {
    FastDiv K, BSP, GW, GH;   // vectors per fat pixel, padded row length (pixels), grid dims
    uint32_t C, bs, pad, n_total;
    uint32_t per_tile;        // BSP*BSP*K output vectors per executed tile
    uint32_t epv;             // elements per vector
    DynCount dyn;             // executed-tile count on the device (units = tiles = gridDim.y)
};

template <int VB, typename T, bool RING, int DT>
__global__ __launch_bounds__(WG) void k_halo_nhwc(typename VecOf<VB>::type *__restrict__ out,
                                                  const typename VecOf<VB>::type *__restrict__ features, long long other_delta,
                                                  typename VecOf<VB>::type *__restrict__ ring_w,
                                                  const int32_t *__restrict__ grid_idx, const int32_t *__restrict__ mapping_exec,
                                                  HaloNhwcGeom g, Prologue pr)
{
    typedef typename VecOf<VB>::type V;
    constexpr int VE = VB / (int)sizeof(T);
    const uint32_t b = blockIdx.y;
    if (b >= dyn_units(g.dyn, gridDim.y)) return;
    const uint32_t bs = g.bs, p = g.pad, K = g.K.d;
    const uint32_t tile_vecs = bs * bs * K;          // vectors per packed tile
    const uint32_t RSV = 4 * p * bs * K;             // vectors per compact ring record (fat pixels x K)

    // wave-uniform 3x3 neighbour table (scalar loads), as in k_halo_rows
    const uint32_t ig = (uint32_t)mapping_exec[b];
    uint32_t t0, gw, n0, gh;
    fd_divmod(ig, g.GW, t0, gw);
    fd_divmod(t0, g.GH, n0, gh);
    long long nbb[9];
    bool nbz[9], nbr[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) {
        const int dy = k / 3 - 1, dx = k % 3 - 1;
        const int nh = (int)gh + dy, nw = (int)gw + dx;
        nbz[k] = nh < 0 || nh >= (int)g.GH.d || nw < 0 || nw >= (int)g.GW.d;
        nbr[k] = false;
        nbb[k] = 0;
        if (k == 4) nbb[k] = (long long)b * tile_vecs;
        else if (!nbz[k]) {
            const uint32_t g_in = (uint32_t)((int)ig + dx + (int)g.GW.d * dy);
            const int32_t idx = grid_idx[g_in];
            if (idx >= 0) nbb[k] = (long long)idx * tile_vecs;
            else if (RING) { nbb[k] = other_delta + (long long)g_in * RSV; nbr[k] = true; }
            else nbb[k] = other_delta + (long long)(uint32_t)(idx + (int32_t)g.n_total) * tile_vecs;
        }
    }
    V *__restrict__ out_t = out + (size_t)b * g.per_tile;
    V *__restrict__ rec = RING ? ring_w + (long long)ig * RSV : nullptr;

    V vec[UNROLL];
    uint32_t fo[UNROLL], kk[UNROLL], hs_[UNROLL], ws_[UNROLL];
    bool zero[UNROLL], own[UNROLL], ring[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
        const uint32_t f = min((blockIdx.x * UNROLL + u) * WG + threadIdx.x, g.per_tile - 1);
        uint32_t pix, k, hp, wp;
        fd_divmod(f, g.K, pix, k);
        fd_divmod(pix, g.BSP, hp, wp);
        const uint32_t sy = hp < p ? 0u : (hp >= p + bs ? 2u : 1u);
        const uint32_t sx = wp < p ? 0u : (wp >= p + bs ? 2u : 1u);
        const uint32_t hs = hp - p + bs - sy * bs, ws = wp - p + bs - sx * bs;
        // 3x3 select of wave-uniform values
        const long long b0 = sx == 0 ? nbb[0] : (sx == 1 ? nbb[1] : nbb[2]);
        const long long b1 = sx == 0 ? nbb[3] : (sx == 1 ? nbb[4] : nbb[5]);
        const long long b2 = sx == 0 ? nbb[6] : (sx == 1 ? nbb[7] : nbb[8]);
        const bool z0 = sx == 0 ? nbz[0] : (sx == 1 ? nbz[1] : nbz[2]);
        const bool z1 = sx == 0 ? nbz[3] : (sx == 1 ? false : nbz[5]);
        const bool z2 = sx == 0 ? nbz[6] : (sx == 1 ? nbz[7] : nbz[8]);
        const bool r0 = sx == 0 ? nbr[0] : (sx == 1 ? nbr[1] : nbr[2]);
        const bool r1 = sx == 0 ? nbr[3] : (sx == 1 ? false : nbr[5]);
        const bool r2 = sx == 0 ? nbr[6] : (sx == 1 ? nbr[7] : nbr[8]);
        const long long base = sy == 0 ? b0 : (sy == 1 ? b1 : b2);
        zero[u] = sy == 0 ? z0 : (sy == 1 ? z1 : z2);
        const bool from_ring = sy == 0 ? r0 : (sy == 1 ? r1 : r2);
        const uint32_t off = (from_ring ? ring_elem(sy, sx, hs, ws, bs, p) : hs * bs + ws) * K + k;
        vec[u] = features[zero[u] ? 0 : base + off];   // inputs are fresh from the producing kernel: streaming loads measured slower here
        fo[u] = f; kk[u] = k; hs_[u] = hs; ws_[u] = ws;
        own[u] = sy == 1 && sx == 1;
        ring[u] = from_ring;
    }
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
        V v = vec[u];
        if (DT != 0 && !ring[u]) {   // ring records hold activated values already
            T *e = reinterpret_cast<T *>(&v);
            float sc[VE], sh[VE];
            load_coeffs<VE>(pr.scale, kk[u] * VE, 1.0f, sc);
            load_coeffs<VE>(pr.shift, kk[u] * VE, 0.0f, sh);
#pragma unroll
            for (int j = 0; j < VE; ++j) e[j] = ActCvt<DT, T>::apply(e[j], sc[j], sh[j], pr.relu);
        }
        if (RING && own[u]) {   // refresh the tile's own compact ring record (activated values), aligned vector stores
            const uint32_t hs = hs_[u], ws = ws_[u], k = kk[u];
            if (hs < p) rec[(hs * bs + ws) * K + k] = v;
            if (hs >= bs - p) rec[(p * bs + (hs - (bs - p)) * bs + ws) * K + k] = v;
            if (ws < p) rec[(2 * p * bs + hs * p + ws) * K + k] = v;
            if (ws >= bs - p) rec[(2 * p * bs + bs * p + hs * p + (ws - (bs - p))) * K + k] = v;
        }
        if (zero[u]) v = V{};
        out_t[fo[u]] = v;
    }
}

// ------------------------------------------------------------------------------------------ halo gather with residual add (NHWC)
// End of a residual block:  v = relu?(raw*scale[c] + shift[c] + identity)  feeds (a) the next padded conv and (b) the next
// block's shortcut.  Instead of one bc_affine_act pass (write v) followed by a halo gather of v (read it again), this
// kernel computes v while gathering: it writes the padded batch AND the plain activated tiles (`act_out`, own interior
// only) in one launch.  Values of executed neighbours are activated on the fly from their raw + identity tiles (same
// packed row in both tensors); non-executed neighbours come from the ring cache, which for this op holds ACTIVATED
// values (= what a gather of the materialised v would have stored).  Arithmetic identical to k_affine_act_nhwc.
template <int DT, typename T> struct ActAdd;
template <> struct ActAdd<1, uint32_t> {
    static __device__ __forceinline__ uint32_t apply(uint32_t v, uint32_t a, float s, float t, int relu)
    {
        float x = __uint_as_float(v) * s + t;
        x += __uint_as_float(a);
        if (relu) x = fmaxf(x, 0.0f);
        return __float_as_uint(x);
    }
};
template <> struct ActAdd<2, uint16_t> {
    static __device__ __forceinline__ uint16_t apply(uint16_t v, uint16_t a, float s, float t, int relu)
    {
        float x = __half2float(__ushort_as_half(v)) * s + t;
        x += __half2float(__ushort_as_half(a));
        if (relu) x = fmaxf(x, 0.0f);
        return __half_as_ushort(__float2half(x));
    }
};
template <> struct ActAdd<3, uint16_t> {
    static __device__ __forceinline__ uint16_t apply(uint16_t v, uint16_t a, float s, float t, int relu)
    {
        float x = __uint_as_float((uint32_t)v << 16) * s + t;
        x += __uint_as_float((uint32_t)a << 16);
        if (relu) x = fmaxf(x, 0.0f);
        hip_bfloat16 b(x);
        return *reinterpret_cast<uint16_t *>(&b);
    }
};

template <int VB, typename T, int DT>
__global__ __launch_bounds__(WG) void k_halo_add_nhwc(typename VecOf<VB>::type *__restrict__ out,
                                                      typename VecOf<VB>::type *__restrict__ act_out,
                                                      const typename VecOf<VB>::type *__restrict__ features, long long add_delta,
                                                      long long other_delta, typename VecOf<VB>::type *__restrict__ ring_w,
                                                      const int32_t *__restrict__ grid_idx, const int32_t *__restrict__ mapping_exec,
                                                      HaloNhwcGeom g, Prologue pr)
{
    typedef typename VecOf<VB>::type V;
    constexpr int VE = VB / (int)sizeof(T);
    const uint32_t b = blockIdx.y;
    if (b >= dyn_units(g.dyn, gridDim.y)) return;
    const uint32_t bs = g.bs, p = g.pad, K = g.K.d;
    const uint32_t tile_vecs = bs * bs * K, RSV = 4 * p * bs * K;
    const uint32_t ig = (uint32_t)mapping_exec[b];
    uint32_t t0, gw, n0, gh;
    fd_divmod(ig, g.GW, t0, gw);
    fd_divmod(t0, g.GH, n0, gh);
    long long nbb[9];
    bool nbz[9], nbr[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) {
        const int dy = k / 3 - 1, dx = k % 3 - 1;
        const int nh = (int)gh + dy, nw = (int)gw + dx;
        nbz[k] = nh < 0 || nh >= (int)g.GH.d || nw < 0 || nw >= (int)g.GW.d;
        nbr[k] = false;
        nbb[k] = 0;
        if (k == 4) nbb[k] = (long long)b * tile_vecs;
        else if (!nbz[k]) {
            const uint32_t g_in = (uint32_t)((int)ig + dx + (int)g.GW.d * dy);
            const int32_t idx = grid_idx[g_in];
            if (idx >= 0) nbb[k] = (long long)idx * tile_vecs;
            else { nbb[k] = other_delta + (long long)g_in * RSV; nbr[k] = true; }
        }
    }
    V *__restrict__ out_t = out + (size_t)b * g.per_tile;
    V *__restrict__ act_t = act_out + (size_t)b * tile_vecs;
    V *__restrict__ rec = ring_w + (long long)ig * RSV;

    V vec[UNROLL], avec[UNROLL];
    uint32_t fo[UNROLL], kk[UNROLL], hs_[UNROLL], ws_[UNROLL];
    bool zero[UNROLL], own[UNROLL], ring[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
        const uint32_t f = min((blockIdx.x * UNROLL + u) * WG + threadIdx.x, g.per_tile - 1);
        uint32_t pix, k, hp, wp;
        fd_divmod(f, g.K, pix, k);
        fd_divmod(pix, g.BSP, hp, wp);
        const uint32_t sy = hp < p ? 0u : (hp >= p + bs ? 2u : 1u);
        const uint32_t sx = wp < p ? 0u : (wp >= p + bs ? 2u : 1u);
        const uint32_t hs = hp - p + bs - sy * bs, ws = wp - p + bs - sx * bs;
        const long long b0 = sx == 0 ? nbb[0] : (sx == 1 ? nbb[1] : nbb[2]);
        const long long b1 = sx == 0 ? nbb[3] : (sx == 1 ? nbb[4] : nbb[5]);
        const long long b2 = sx == 0 ? nbb[6] : (sx == 1 ? nbb[7] : nbb[8]);
        const bool z0 = sx == 0 ? nbz[0] : (sx == 1 ? nbz[1] : nbz[2]);
        const bool z1 = sx == 0 ? nbz[3] : (sx == 1 ? false : nbz[5]);
        const bool z2 = sx == 0 ? nbz[6] : (sx == 1 ? nbz[7] : nbz[8]);
        const bool r0 = sx == 0 ? nbr[0] : (sx == 1 ? nbr[1] : nbr[2]);
        const bool r1 = sx == 0 ? nbr[3] : (sx == 1 ? false : nbr[5]);
        const bool r2 = sx == 0 ? nbr[6] : (sx == 1 ? nbr[7] : nbr[8]);
        const long long base = sy == 0 ? b0 : (sy == 1 ? b1 : b2);
        zero[u] = sy == 0 ? z0 : (sy == 1 ? z1 : z2);
        ring[u] = sy == 0 ? r0 : (sy == 1 ? r1 : r2);
        const uint32_t off = (ring[u] ? ring_elem(sy, sx, hs, ws, bs, p) : hs * bs + ws) * K + k;
        const long long src = zero[u] ? 0 : base + off;
        vec[u] = features[src];
        avec[u] = features[(zero[u] || ring[u]) ? add_delta : add_delta + src];   // identity tile: same packed row / offset
        fo[u] = f; kk[u] = k; hs_[u] = hs; ws_[u] = ws;
        own[u] = sy == 1 && sx == 1;
    }
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
        V v = vec[u];
        if (!ring[u]) {   // ring records already hold activated values
            T *e = reinterpret_cast<T *>(&v);
            const T *a = reinterpret_cast<const T *>(&avec[u]);
            float sc[VE], sh[VE];
            load_coeffs<VE>(pr.scale, kk[u] * VE, 1.0f, sc);
            load_coeffs<VE>(pr.shift, kk[u] * VE, 0.0f, sh);
#pragma unroll
            for (int j = 0; j < VE; ++j) e[j] = ActAdd<DT, T>::apply(e[j], a[j], sc[j], sh[j], pr.relu);
        }
        if (zero[u]) v = V{};
        if (own[u]) {
            const uint32_t hs = hs_[u], ws = ws_[u], k = kk[u];
            act_t[(hs * bs + ws) * K + k] = v;
            if (hs < p) rec[(hs * bs + ws) * K + k] = v;
            if (hs >= bs - p) rec[(p * bs + (hs - (bs - p)) * bs + ws) * K + k] = v;
            if (ws < p) rec[(2 * p * bs + hs * p + ws) * K + k] = v;
            if (ws >= bs - p) rec[(2 * p * bs + bs * p + hs * p + (ws - (bs - p))) * K + k] = v;
        }
        out_t[fo[u]] = v;
    }
}

// ------------------------------------------------------------------------------------------ fused halo + 3x3/s2 max-pool (NHWC)
// The one padded op of the path that is not a conv (ResNet stem): max_pool2d(k=3, s=2, p=1) on the packed batch.  The
// reference materialises the halo-padded tiles and pools them with padding 0 (core/tensorwrapper.py:478-527); here one
// kernel reads tile + halo (top / left / top-left only: with an even tile the windows never reach past the bottom or
// right edge), applies the pending activation, takes the max and refreshes the ring record -- the padded tensor
// (4.4x the output) is never written or re-read.  Image-border halo = zeros that take part in the max, as in the
// reference.  One lane = one 16-byte vector of one output pixel; 3x3 neighbour table wave-uniform as in k_halo_nhwc.
struct PoolGeom {
    FastDiv K, OB, GW, GH;    // vectors per fat pixel, output tile edge bs/2, grid dims
    FastDiv OBP;              // OB / 8 when the lanes walk the tile in 8x8-pixel patches (OB % 8 == 0), else d = 0
    uint32_t bs, n_total, per_tile;   // per_tile = OB*OB*K output vectors per executed tile
    DynCount dyn;                     // executed-tile count on the device (units = tiles = gridDim.y)
};

constexpr int MP_U = 2;
template <typename T, int VE, int DT>
__global__ __launch_bounds__(WG) void k_maxpool3x3s2_nhwc(typename VecOf<sizeof(T) * VE>::type *__restrict__ out,
                                                          const typename VecOf<sizeof(T) * VE>::type *__restrict__ features,
                                                          long long other_delta, typename VecOf<sizeof(T) * VE>::type *__restrict__ ring_w,
                                                          const int32_t *__restrict__ grid_idx, const int32_t *__restrict__ mapping_exec,
                                                          PoolGeom g, Prologue pr)
{
    typedef typename VecOf<sizeof(T) * VE>::type V;
    const uint32_t b = blockIdx.y, bs = g.bs, K = g.K.d;
    if (b >= dyn_units(g.dyn, gridDim.y)) return;
    const uint32_t tile_vecs = bs * bs * K, RSV = 4 * bs * K;
    const uint32_t ig = (uint32_t)mapping_exec[b];
    uint32_t t0, gw, n0, gh;
    fd_divmod(ig, g.GW, t0, gw);
    fd_divmod(t0, g.GH, n0, gh);
    // sources of the own tile (k = 3), the tile above (1), to the left (2) and above-left (0)
    long long nbb[4];
    bool nbz[4], nbr[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int dy = (k >> 1) - 1, dx = (k & 1) - 1;
        const int nh = (int)gh + dy, nw = (int)gw + dx;
        nbz[k] = nh < 0 || nw < 0;
        nbr[k] = false;
        nbb[k] = 0;
        if (k == 3) nbb[k] = (long long)b * tile_vecs;
        else if (!nbz[k]) {
            const uint32_t g_in = (uint32_t)((int)ig + dx + (int)g.GW.d * dy);
            const int32_t idx = grid_idx[g_in];
            if (idx >= 0) nbb[k] = (long long)idx * tile_vecs;
            else { nbb[k] = other_delta + (long long)g_in * RSV; nbr[k] = true; }
        }
    }
    V *__restrict__ rec = ring_w + (long long)ig * RSV;
    // MP_U output vectors per lane behind ONE pair of dependent table lookups (the lookups are two memory round trips: with one
    // vector per lane they were as long as the lane's whole life)
#pragma unroll
    for (int u = 0; u < MP_U; ++u) {
    const uint32_t f = (blockIdx.x * MP_U + u) * WG + threadIdx.x;
    if (f >= g.per_tile) continue;
    uint32_t pix, kq, oy, ox;
    fd_divmod(f, g.K, pix, kq);
    if (g.OBP.d) {
        // 8x8-pixel patches instead of whole output rows: the three input rows of an output row are shared with the rows above and
        // below, which a row-by-row walk hands to other workgroups (other XCDs: re-fetched from memory, 1.38x the algorithmic bytes)
        uint32_t py, pxx;
        const uint32_t q = pix & 63u;
        fd_divmod(pix >> 6, g.OBP, py, pxx);
        oy = py * 8 + (q >> 3); ox = pxx * 8 + (q & 7u);
    } else fd_divmod(pix, g.OB, oy, ox);
    float sc[VE], sh[VE];
    load_coeffs<VE>(pr.scale, kq * VE, 1.0f, sc);
    load_coeffs<VE>(pr.shift, kq * VE, 0.0f, sh);

    V raw[9];
    bool zero[9], ring[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        const int y = 2 * (int)oy + t / 3 - 1, x = 2 * (int)ox + t % 3 - 1;   // >= -1, <= bs - 1
        const uint32_t sy = y < 0 ? 0u : 1u, sx = x < 0 ? 0u : 1u;
        const uint32_t k = sy * 2 + sx;
        const uint32_t hs = y < 0 ? bs - 1 : (uint32_t)y, ws = x < 0 ? bs - 1 : (uint32_t)x;
        const long long base = k == 0 ? nbb[0] : (k == 1 ? nbb[1] : (k == 2 ? nbb[2] : nbb[3]));
        const bool from_ring = k == 0 ? nbr[0] : (k == 1 ? nbr[1] : (k == 2 ? nbr[2] : false));
        zero[t] = k == 0 ? nbz[0] : (k == 1 ? nbz[1] : (k == 2 ? nbz[2] : false));
        const uint32_t pos = from_ring ? ring_elem(sy, sx, hs, ws, bs, 1u) : hs * bs + ws;
        raw[t] = features[zero[t] ? 0 : base + (long long)pos * K + kq];
        ring[t] = from_ring;
    }
    // activation of everything that comes from packed tiles (ring records hold activated values), rounded to T as the
    // reference pools the ROUNDED activations
    if (DT != 0) {
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            if (ring[t]) continue;
            T *e = reinterpret_cast<T *>(&raw[t]);
#pragma unroll
            for (int j = 0; j < VE; ++j) {
                float x = Cvt<T>::ld(e + j) * sc[j] + sh[j];
                if (pr.relu) x = fmaxf(x, 0.0f);
                e[j] = Cvt<T>::st(x);
            }
        }
    }
    // ring refresh: this lane owns the 2x2 input block (2oy..2oy+1, 2ox..2ox+1) = taps 4, 5, 7, 8
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        if (t != 4 && t != 5 && t != 7 && t != 8) continue;
        const uint32_t hs = 2 * oy + t / 3 - 1, ws = 2 * ox + t % 3 - 1;
        if (hs == 0) rec[(ws)*K + kq] = raw[t];
        if (hs == bs - 1) rec[(bs + ws) * K + kq] = raw[t];
        if (ws == 0) rec[(2 * bs + hs) * K + kq] = raw[t];
        if (ws == bs - 1) rec[(3 * bs + hs) * K + kq] = raw[t];
    }
    float best[VE];
#pragma unroll
    for (int j = 0; j < VE; ++j) best[j] = -INFINITY;
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        const T *e = reinterpret_cast<const T *>(&raw[t]);
#pragma unroll
        for (int j = 0; j < VE; ++j) {
            float x = Cvt<T>::ld(e + j);
            if (zero[t]) x = 0.0f;
            best[j] = fmaxf(best[j], x);
        }
    }
    V res;
    T *r = reinterpret_cast<T *>(&res);
#pragma unroll
    for (int j = 0; j < VE; ++j) r[j] = Cvt<T>::st(best[j]);
    out[(size_t)b * g.per_tile + ((size_t)oy * g.OB.d + ox) * K + kq] = res;
    }
}

// ------------------------------------------------------------------------------------------ GroupNorm over all executed tiles (NHWC)
// The reference runs group_norm on packed tiles by folding the tile axis into the spatial axis, so that the statistics of a group
// run over ALL executed tiles of the frame (core/tensorwrapper.py:600-633: (B,C,h,w) -> (1,C,B*h*w,1)).  On channels-last packed
// tiles that is a plain (n_pix, C) matrix, and the result of the op is a per-CHANNEL affine map y = x*a[c] + b[c] with
// a = gamma*rstd[g(c)], b = beta - mean[g(c)]*a -- exactly what the engine's lazy fusion records and the next kernel applies as
// its prologue.  So the op costs ONE read of the tensor: k_group_stats (per-workgroup per-channel partial sums, fixed ranges) and
// k_group_finalize (one workgroup per group sums the partials in double precision in a fixed order -- deterministic, no atomics --
// and writes a, b).  The stock route on this layout: two layout copies + RowwiseMoments + the elementwise normalisation + a third
// copy back (5 passes over the tensor).
template <typename T, int VE>
__global__ __launch_bounds__(WG) void k_group_stats(const typename VecOf<sizeof(T) * VE>::type *__restrict__ x, uint32_t n_pix, uint32_t K /* vectors per pixel */,
                                                    uint32_t rows_per_wg, float *__restrict__ partial /* [wg][C][2] */)
{
    typedef typename VecOf<sizeof(T) * VE>::type V;
    __shared__ float red[WG][2 * VE + 1];
    const uint32_t lanes_per_row = K;                    // one lane per vector of a pixel; WG / K pixel rows in flight
    const uint32_t rpw = WG / lanes_per_row;             // (host guarantees K <= WG and WG % K == 0)
    const uint32_t k = threadIdx.x % lanes_per_row, r = threadIdx.x / lanes_per_row;
    const uint32_t p0 = blockIdx.x * rows_per_wg, p1 = min(p0 + rows_per_wg, n_pix);
    float s1[VE], s2[VE];
#pragma unroll
    for (int j = 0; j < VE; ++j) { s1[j] = 0.0f; s2[j] = 0.0f; }
    for (uint32_t p = p0 + r; p < p1; p += rpw) {
        const V v = x[(size_t)p * K + k];
        const T *e = reinterpret_cast<const T *>(&v);
#pragma unroll
        for (int j = 0; j < VE; ++j) {
            const float f = Cvt<T>::ld(e + j);
            s1[j] += f;
            s2[j] = fmaf(f, f, s2[j]);
        }
    }
#pragma unroll
    for (int j = 0; j < VE; ++j) { red[threadIdx.x][2 * j] = s1[j]; red[threadIdx.x][2 * j + 1] = s2[j]; }
    __syncthreads();
    if (threadIdx.x < lanes_per_row) {     // fixed-order sum over the pixel rows of this workgroup
#pragma unroll
        for (int j = 0; j < VE; ++j) {
            float a = 0.0f, b = 0.0f;
            for (uint32_t q = 0; q < rpw; ++q) { a += red[q * lanes_per_row + threadIdx.x][2 * j]; b += red[q * lanes_per_row + threadIdx.x][2 * j + 1]; }
            const size_t o = ((size_t)blockIdx.x * K * VE + (size_t)threadIdx.x * VE + j) * 2;
            partial[o] = a;
            partial[o + 1] = b;
        }
    }
}

struct BnExtra {          // batch-norm bookkeeping of the training forward (cpg == 1: a "group" is a channel); all optional
    float *save_mean, *save_invstd, *running_mean, *running_var;
    long long *batches;
    float momentum;
};

__global__ __launch_bounds__(WG) void k_group_finalize(const float *__restrict__ partial, uint32_t n_wg, uint32_t C, uint32_t cpg, double count,
                                                       float eps, const float *__restrict__ gamma, const float *__restrict__ beta,
                                                       float *__restrict__ scale, float *__restrict__ shift, BnExtra bn)
{
    __shared__ double r1[WG], r2[WG];
    const uint32_t g = blockIdx.x;
    double a = 0.0, b = 0.0;
    // thread t owns the (workgroup, channel-of-group) items t, t + WG, ...: a fixed assignment, then a fixed-shape tree
    for (uint32_t it = threadIdx.x; it < n_wg * cpg; it += WG) {
        const uint32_t w = it / cpg, c = g * cpg + it % cpg;
        a += (double)partial[((size_t)w * C + c) * 2];
        b += (double)partial[((size_t)w * C + c) * 2 + 1];
    }
    r1[threadIdx.x] = a; r2[threadIdx.x] = b;
    __syncthreads();
    for (uint32_t st = WG / 2; st > 0; st >>= 1) {
        if (threadIdx.x < st) { r1[threadIdx.x] += r1[threadIdx.x + st]; r2[threadIdx.x] += r2[threadIdx.x + st]; }
        __syncthreads();
    }
    const double mean = r1[0] / count;
    double var = r2[0] / count - mean * mean;
    if (var < 0.0) var = 0.0;
    const float rstd = (float)(1.0 / sqrt(var + (double)eps));
    if (threadIdx.x == 0) {
        if (bn.save_mean) bn.save_mean[g] = (float)mean;
        if (bn.save_invstd) bn.save_invstd[g] = rstd;
        if (bn.running_mean) bn.running_mean[g] = (1.0f - bn.momentum) * bn.running_mean[g] + bn.momentum * (float)mean;
        if (bn.running_var) bn.running_var[g] = (1.0f - bn.momentum) * bn.running_var[g] + bn.momentum * (float)(count > 1.0 ? var * count / (count - 1.0) : var);
        if (bn.batches && g == 0) *bn.batches += 1;
    }
    for (uint32_t c = threadIdx.x; c < cpg; c += WG) {
        const uint32_t ch = g * cpg + c;
        const float sc = (gamma ? gamma[ch] : 1.0f) * rstd;
        scale[ch] = sc;
        shift[ch] = (beta ? beta[ch] : 0.0f) - (float)mean * sc;
    }
}

// ------------------------------------------------------------------------------------------ training-mode BatchNorm forward (policy net)
// The online-RL policy keeps its small CNN in TRAINING mode on every frame (reference policy/policy.py: the net is
// trained online, batch statistics of the single frame), i.e. ten batch-statistics BatchNorms per frame on 4-17 MB maps.  The stock
// route is three library kernels + a momentum update + a counter increment per layer (32 us on average).  Here: k_bn_stats (per-chunk
// partial sums of a channel plane) + k_bn_apply (every workgroup re-reduces the <= 64 partials of its channel in double -- fixed
// order, deterministic -- then normalises its chunk with optional ReLU; chunk 0 also writes save_mean / save_invstd for the backward
// pass and the running statistics, workgroup (0,0) the batch counter).  NCHW fp32, contiguous.
struct BnGeom {
    uint32_t N, C, HW;
    uint32_t L;          // elements (of the N*HW of a channel) per chunk, a multiple of 4
    uint32_t chunks;
};

__global__ __launch_bounds__(WG) void k_bn_stats(const float *__restrict__ x, float2 *__restrict__ partial, BnGeom g)
{
    __shared__ float r1[WG], r2[WG];
    const uint32_t c = blockIdx.y, j = blockIdx.x;
    const uint32_t total = g.N * g.HW;
    const uint32_t e0 = j * g.L, e1 = min(e0 + g.L, total);
    float s1 = 0.0f, s2 = 0.0f;
    if (g.HW % 4 == 0) {
        for (uint32_t e = e0 + threadIdx.x * 4; e < e1; e += WG * 4) {
            const uint32_t n = e / g.HW, hw = e - n * g.HW;
            const float4 v = *reinterpret_cast<const float4 *>(x + ((size_t)n * g.C + c) * g.HW + hw);
            s1 += (v.x + v.y) + (v.z + v.w);
            s2 = fmaf(v.x, v.x, fmaf(v.y, v.y, fmaf(v.z, v.z, fmaf(v.w, v.w, s2))));
        }
    } else {
        for (uint32_t e = e0 + threadIdx.x; e < e1; e += WG) {
            const uint32_t n = e / g.HW, hw = e - n * g.HW;
            const float v = x[((size_t)n * g.C + c) * g.HW + hw];
            s1 += v;
            s2 = fmaf(v, v, s2);
        }
    }
    r1[threadIdx.x] = s1; r2[threadIdx.x] = s2;
    __syncthreads();
    for (uint32_t st = WG / 2; st > 0; st >>= 1) {
        if (threadIdx.x < st) { r1[threadIdx.x] += r1[threadIdx.x + st]; r2[threadIdx.x] += r2[threadIdx.x + st]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) partial[(size_t)c * g.chunks + j] = make_float2(r1[0], r2[0]);
}

__global__ __launch_bounds__(WG) void k_bn_apply(float *__restrict__ y, const float *__restrict__ x, const float2 *__restrict__ partial,
                                                 const float *__restrict__ gamma, const float *__restrict__ beta, float *__restrict__ running_mean,
                                                 float *__restrict__ running_var, long long *__restrict__ batches, float *__restrict__ save_mean,
                                                 float *__restrict__ save_invstd, float momentum, float eps, int relu, BnGeom g)
{
    __shared__ float coef[2];
    const uint32_t c = blockIdx.y, j = blockIdx.x;
    const uint32_t total = g.N * g.HW;
    if (threadIdx.x == 0) {
        double a = 0.0, b = 0.0;
        for (uint32_t q = 0; q < g.chunks; ++q) { const float2 p = partial[(size_t)c * g.chunks + q]; a += (double)p.x; b += (double)p.y; }
        const double mean = a / total;
        double var = b / total - mean * mean;
        if (var < 0.0) var = 0.0;
        const float invstd = (float)(1.0 / sqrt(var + (double)eps));
        const float sc = (gamma ? gamma[c] : 1.0f) * invstd;
        coef[0] = sc;
        coef[1] = (beta ? beta[c] : 0.0f) - (float)mean * sc;
        if (j == 0) {
            if (save_mean) save_mean[c] = (float)mean;
            if (save_invstd) save_invstd[c] = invstd;
            if (running_mean) running_mean[c] = (1.0f - momentum) * running_mean[c] + momentum * (float)mean;
            if (running_var) running_var[c] = (1.0f - momentum) * running_var[c] + momentum * (float)(total > 1 ? var * total / (total - 1.0) : var);
            if (batches && c == 0) *batches += 1;
        }
    }
    __syncthreads();
    const float sc = coef[0], sh = coef[1];
    const uint32_t e0 = j * g.L, e1 = min(e0 + g.L, total);
    if (g.HW % 4 == 0) {
        for (uint32_t e = e0 + threadIdx.x * 4; e < e1; e += WG * 4) {
            const uint32_t n = e / g.HW, hw = e - n * g.HW;
            const size_t o = ((size_t)n * g.C + c) * g.HW + hw;
            float4 v = *reinterpret_cast<const float4 *>(x + o);
            v.x = fmaf(v.x, sc, sh); v.y = fmaf(v.y, sc, sh); v.z = fmaf(v.z, sc, sh); v.w = fmaf(v.w, sc, sh);
            if (relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
            *reinterpret_cast<float4 *>(y + o) = v;
        }
    } else {
        for (uint32_t e = e0 + threadIdx.x; e < e1; e += WG) {
            const uint32_t n = e / g.HW, hw = e - n * g.HW;
            const size_t o = ((size_t)n * g.C + c) * g.HW + hw;
            float v = fmaf(x[o], sc, sh);
            if (relu) v = fmaxf(v, 0.f);
            y[o] = v;
        }
    }
}

// ------------------------------------------------------------------------------------------ adaptive average pooling (NHWC, dense maps)
// SwiftNet's pyramid pooling (inside blockcopy_noblocks: a dense stride-32 map) calls F.adaptive_avg_pool2d three times per frame;
// the stock kernel spends 11.5 us on each 1 MB map (a handful of output bins, little parallelism).  One workgroup per output bin:
// lanes over the channel vectors, wave rows over the bin's pixels, a fixed-order LDS reduction.  Bin limits as ATen's:
// start = floor(i*H/OH), end = ceil((i+1)*H/OH).
template <typename T, int VE>
__global__ __launch_bounds__(WG) void k_adaptive_avg_pool_nhwc(typename VecOf<sizeof(T) * VE>::type *__restrict__ out,
                                                               const typename VecOf<sizeof(T) * VE>::type *__restrict__ in, uint32_t H, uint32_t W,
                                                               uint32_t K /* vectors per pixel */, uint32_t OH, uint32_t OW)
{
    typedef typename VecOf<sizeof(T) * VE>::type V;
    __shared__ float red[WG][VE + 1];
    const uint32_t bin = blockIdx.x, n = bin / (OH * OW), r = bin - n * (OH * OW);
    const uint32_t oy = r / OW, ox = r - oy * OW;
    const uint32_t y0 = (oy * H) / OH, y1 = ((oy + 1) * H + OH - 1) / OH;
    const uint32_t x0 = (ox * W) / OW, x1 = ((ox + 1) * W + OW - 1) / OW;
    const uint32_t bw = x1 - x0, npx = (y1 - y0) * bw;
    const uint32_t rows = WG / K;                        // pixels in flight (host: K <= WG, WG % K == 0)
    const uint32_t k = threadIdx.x % K, pr = threadIdx.x / K;
    float acc[VE];
#pragma unroll
    for (int j = 0; j < VE; ++j) acc[j] = 0.0f;
    for (uint32_t p = pr; p < npx; p += rows) {
        const uint32_t y = y0 + p / bw, x = x0 + p % bw;
        const V v = in[((size_t)(n * H + y) * W + x) * K + k];
        const T *e = reinterpret_cast<const T *>(&v);
#pragma unroll
        for (int j = 0; j < VE; ++j) acc[j] += Cvt<T>::ld(e + j);
    }
#pragma unroll
    for (int j = 0; j < VE; ++j) red[threadIdx.x][j] = acc[j];
    __syncthreads();
    if (threadIdx.x < K) {
        const float inv = 1.0f / (float)npx;
        V res;
        T *o = reinterpret_cast<T *>(&res);
#pragma unroll
        for (int j = 0; j < VE; ++j) {
            float s = 0.0f;
            for (uint32_t q = 0; q < rows; ++q) s += red[q * K + threadIdx.x][j];
            o[j] = Cvt<T>::st(s * inv);
        }
        out[(size_t)bin * K + threadIdx.x] = res;
    }
}

// channels-last fused epilogue: channel index runs fastest
template <typename T, int Q>
__global__ __launch_bounds__(WG) void k_affine_act_nhwc(T *__restrict__ out, const T *__restrict__ in, const T *__restrict__ add,
                                                        const float *__restrict__ scale, const float *__restrict__ shift,
                                                        int relu, FastDiv Cq, uint32_t total_host, DynCount dyn)
{
    typedef typename VecOf<sizeof(T) * Q>::type V;
    const uint32_t i = blockIdx.x * WG + threadIdx.x;
    const uint32_t total = dyn_units(dyn, total_host);
    if (i >= total) return;
    uint32_t pix, q;
    fd_divmod(i, Cq, pix, q);
    V vi = reinterpret_cast<const V *>(in)[i];
    V va = vi;
    if (add) va = reinterpret_cast<const V *>(add)[i];
    const T *xi = reinterpret_cast<const T *>(&vi), *xa = reinterpret_cast<const T *>(&va);
    T res[Q];
    float sc[Q], sh[Q];
    load_coeffs<Q>(scale, q * Q, 1.0f, sc);
    load_coeffs<Q>(shift, q * Q, 0.0f, sh);
#pragma unroll
    for (int k = 0; k < Q; ++k) {
        float x = Cvt<T>::ld(xi + k) * sc[k] + sh[k];
        if (add) x += Cvt<T>::ld(xa + k);
        if (relu) x = fmaxf(x, 0.0f);
        res[k] = Cvt<T>::st(x);
    }
    reinterpret_cast<V *>(out)[i] = *reinterpret_cast<const V *>(res);
}

// channels-last per-tile bilinear resampling: one lane = Q channels of one output pixel
struct InterpNhwcGeom {
    FastDiv Cq, W, H;
    uint32_t h, w, total;
    float rh, rw;
    int align;
    DynCount dyn;     // executed-tile count on the device (units = output vectors)
};

// one fixed evaluation order for the 4-tap blend, so that the plain and the epilogue form of the kernel agree bit for bit
__device__ __forceinline__ float bilerp(float a, float b, float c, float d, float lx0, float lx1, float ly0, float ly1)
{
    const float top = fmaf(lx1, b, lx0 * a), bot = fmaf(lx1, d, lx0 * c);
    return fmaf(ly1, bot, ly0 * top);
}

// optional epilogue relu?(y*scale[c] + shift[c] + add) on the interpolated (and, as in the two-kernel route, rounded) value:
// the decoder's "upsample, then += skip" costs one launch instead of two
struct InterpEpi {
    const float *scale, *shift;
    const void *add;     // (planes, H, W, C), may be null
    int relu, on;
};

template <typename T, int Q>
__global__ __launch_bounds__(WG) void k_interp_bilinear_nhwc(T *__restrict__ out, const T *__restrict__ in, InterpNhwcGeom g, InterpEpi ep)
{
    typedef typename VecOf<sizeof(T) * Q>::type V;
    const uint32_t i = blockIdx.x * WG + threadIdx.x;
    if (i >= dyn_units(g.dyn, g.total)) return;
    uint32_t r, q, r2, ox, plane, oy;
    fd_divmod(i, g.Cq, r, q);
    fd_divmod(r, g.W, r2, ox);
    fd_divmod(r2, g.H, plane, oy);
    uint32_t y0, yp, x0, xp; float ly1, lx1;
    src_index(g.rh, oy, g.align, g.h, y0, yp, ly1);
    src_index(g.rw, ox, g.align, g.w, x0, xp, lx1);
    const float ly0 = 1.0f - ly1, lx0 = 1.0f - lx1;
    const V *base = reinterpret_cast<const V *>(in) + (size_t)plane * g.h * g.w * g.Cq.d + q;
    const V v00 = base[((size_t)y0 * g.w + x0) * g.Cq.d], v01 = base[((size_t)y0 * g.w + x0 + xp) * g.Cq.d];
    const V v10 = base[((size_t)(y0 + yp) * g.w + x0) * g.Cq.d], v11 = base[((size_t)(y0 + yp) * g.w + x0 + xp) * g.Cq.d];
    const T *a = reinterpret_cast<const T *>(&v00), *bq = reinterpret_cast<const T *>(&v01);
    const T *c = reinterpret_cast<const T *>(&v10), *d = reinterpret_cast<const T *>(&v11);
    T res[Q];
    if (!ep.on) {
#pragma unroll
        for (int k = 0; k < Q; ++k)
            res[k] = Cvt<T>::st(bilerp(Cvt<T>::ld(a + k), Cvt<T>::ld(bq + k), Cvt<T>::ld(c + k), Cvt<T>::ld(d + k), lx0, lx1, ly0, ly1));
    } else {
        float sc[Q], sh[Q];
        load_coeffs<Q>(ep.scale, q * Q, 1.0f, sc);
        load_coeffs<Q>(ep.shift, q * Q, 0.0f, sh);
        V av = V{};
        if (ep.add) av = reinterpret_cast<const V *>(ep.add)[i];
        const T *e = reinterpret_cast<const T *>(&av);
#pragma unroll
        for (int k = 0; k < Q; ++k) {
            const float y = Cvt<T>::ld_round(bilerp(Cvt<T>::ld(a + k), Cvt<T>::ld(bq + k), Cvt<T>::ld(c + k), Cvt<T>::ld(d + k), lx0, lx1, ly0, ly1));
            float x = y * sc[k] + sh[k];
            if (ep.add) x += Cvt<T>::ld(e + k);
            if (ep.relu) x = fmaxf(x, 0.0f);
            res[k] = Cvt<T>::st(x);
        }
    }
    reinterpret_cast<V *>(out)[i] = *reinterpret_cast<const V *>(res);
}

// ------------------------------------------------------------------------------------------ prediction map of a segmentation clip
// Bilinear upsampling of a logits map to the input resolution followed by the per-pixel arg-max over the classes, in one pass: what the
// reference's driver does with the last frame of a clip (semantic_segmentation/test_swiftnet.py:190-194: F.interpolate(out, size,
// mode='bilinear') then out.max(dim=1)[1]) as two library passes over a 160 MB intermediate (19 classes at 1024 x 2048) that nothing
// else reads.  One thread per output pixel: the four neighbours of every class are read through the caches (a 4x upsampling re-reads
// each source pixel 16 times), the interpolated value is formed and rounded exactly like k_interp_bilinear_nhwc does (= ATen's
// upsample_bilinear2d), and the running maximum keeps the FIRST maximal class; a NaN wins over everything, as in torch.max.
struct UpArgGeom {
    uint32_t C, h, w, H, W;
    long long sn, sc, sy, sx;      // element strides of the logits map (any layout)
    float rh, rw;
    int align;
    uint32_t total;                // N * H * W
};

template <typename T>
__global__ __launch_bounds__(WG) void k_upsample_argmax(long long *__restrict__ out, const T *__restrict__ in, UpArgGeom g)
{
    const uint32_t i = blockIdx.x * WG + threadIdx.x;
    if (i >= g.total) return;
    const uint32_t ox = i % g.W, r = i / g.W, oy = r % g.H, n = r / g.H;
    uint32_t y0, yp, x0, xp; float ly1, lx1;
    src_index(g.rh, oy, g.align, g.h, y0, yp, ly1);
    src_index(g.rw, ox, g.align, g.w, x0, xp, lx1);
    const float ly0 = 1.0f - ly1, lx0 = 1.0f - lx1;
    const T *p00 = in + (long long)n * g.sn + (long long)y0 * g.sy + (long long)x0 * g.sx;
    const long long dx = (long long)xp * g.sx, dy = (long long)yp * g.sy;
    float best = 0.0f;
    uint32_t arg = 0;
    for (uint32_t c = 0; c < g.C; ++c) {
        const T *q = p00 + (long long)c * g.sc;
        const float v = Cvt<T>::ld_round(bilerp(Cvt<T>::ld(q), Cvt<T>::ld(q + dx), Cvt<T>::ld(q + dy), Cvt<T>::ld(q + dy + dx), lx0, lx1, ly0, ly1));
        if (c == 0 || v > best || (v != v && best == best)) { best = v; arg = c; }
    }
    out[i] = (long long)arg;
}

// ------------------------------------------------------------------------------------------ NMS (detector post-processing)
// Greedy non-maximum suppression of score-sorted boxes in ONE launch (round 5; replaces the mask kernel + host sweep of
// Pedestron/mmdet/ops/nms/src/nms_kernel.cu:23-130 -- the reference copies an (n x n/64)-word mask to the host and sweeps it there).
// Rule (the reference's, bit for bit): box j is suppressed by an earlier KEPT box i when IoU(i, j) > thr, IoU with the +1 pixel convention.
//   phase 1, many workgroups: only the UPPER-triangular 64 x 64 tiles of the pair matrix exist (a box is only ever suppressed by an earlier
//            one).  A workgroup of sixteen waves owns one tile: lane = row, each wave four of the columns (the column box is
//            wave-uniform: scalar operands); the row's 64-bit word is put together in LDS and stored whole.  Diagonal tiles also leave
//            the transposed word of every box (its possible suppressors INSIDE its own block: the IoU is symmetric) behind the matrix.
//   phase 2, the workgroup that finishes LAST (device-scope ticket; nobody spins): pulls the words into LDS (n <= 1024: n * ((ceil(n / 64)
//            | 1) + 1) words <= 150 KB; larger inputs sweep out of L2) and walks the column blocks with ONE wave, lane j = box j of the block:
//              a removed bitmap with word w in lane w; after a block is resolved its kept lanes OR their words for the LATER blocks
//                                          into it (lane-parallel reads, DPP reduction per word);
//              inside the block the greedy rule is resolved as a fixed point instead of a walk: a box whose possible suppressors are all
//                                          decided is decided (kept iff none of them was kept); each trip settles at least the first
//                                          undecided box, typically all of them in two or three trips (64-bit ballots).
//            The kept positions leave in ascending order (block by block, through popcount prefixes).
struct NmsGeom { int n, words, tiles, lds_rows, row_shift; };

__device__ __forceinline__ bool nms_suppresses(float ax1, float ay1, float ax2, float ay2, float a_area, float bx1, float by1, float bx2, float by2,
                                               float b_area, float thr)
{
    const float iw = fmaxf(fminf(ax2, bx2) - fmaxf(ax1, bx1) + 1.f, 0.f), ih = fmaxf(fminf(ay2, by2) - fmaxf(ay1, by1) + 1.f, 0.f);
    const float inter = iw * ih;
    return inter / (a_area + b_area - inter) > thr;
}

// 64-bit value of lane `src` (wave-uniform index) as a scalar: two v_readlane_b32, no LDS round trip
__device__ __forceinline__ unsigned long long nms_readlane64(unsigned long long v, int src)
{
    const unsigned int lo = (unsigned int)__builtin_amdgcn_readlane((int)(unsigned int)v, src);
    const unsigned int hi = (unsigned int)__builtin_amdgcn_readlane((int)(unsigned int)(v >> 32), src);
    return ((unsigned long long)hi << 32) | lo;
}

// OR over the 64 lanes of a wave (result valid in every lane): row-wise prefix through DPP, row totals broadcast, lane 63 read back
__device__ __forceinline__ unsigned int nms_wave_or32(unsigned int v)
{
    v |= (unsigned int)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false);     // row_shr:1
    v |= (unsigned int)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false);     // row_shr:2
    v |= (unsigned int)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false);     // row_shr:4
    v |= (unsigned int)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false);     // row_shr:8
    v |= (unsigned int)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);     // row_bcast:15 -> rows 1, 3
    v |= (unsigned int)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);     // row_bcast:31 -> rows 2, 3
    return (unsigned int)__builtin_amdgcn_readlane((int)v, 63);
}

__global__ __launch_bounds__(1024) void k_nms(NmsGeom g, float thr, const float *__restrict__ boxes, unsigned long long *__restrict__ mask,
                                              int32_t *__restrict__ keep, int32_t *__restrict__ count, unsigned int *__restrict__ ticket,
                                              unsigned long long *__restrict__ dbg, const int32_t *__restrict__ n_dev)
{
    // dbg (bc_tune_set_ptr("conv_stamps", ...), measurement only): 100 MHz stamps -- [2 b], [2 b + 1] = entry / ticket of workgroup b; the
    // sweeping workgroup adds [2 tiles ..]: entry, ticket, words in LDS, sweep done
    const unsigned long long t_in = dbg ? __builtin_amdgcn_s_memrealtime() : 0ull;
    extern __shared__ unsigned long long nms_lds[];          // phase 2: the words (n rows of W | 1, then n transposed diagonal words) if they fit
    __shared__ float colbox[64 * 5];
    __shared__ __attribute__((aligned(16))) unsigned char pieces[2][64][16];
    __shared__ int last_flag;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // box count known only on the device (bc_nms_sorted_dev: the launch is sized for g.n boxes, *n_dev <= g.n of them exist): the
    // geometry is rederived here, surplus workgroups leave before they touch anything (and take no ticket)
    if (n_dev) {
        const int nd = min(max(*n_dev, 0), g.n);
        g.n = nd; g.words = (nd + 63) / 64; g.tiles = g.words * (g.words + 1) / 2;
        g.row_shift = 0;
        while ((1 << g.row_shift) < g.words) ++g.row_shift;
        if ((int)blockIdx.x >= g.tiles) {
            if (nd == 0 && blockIdx.x == 0 && tid == 0) *count = 0;
            return;
        }
    }
    const int n = g.n, W = g.words;
    unsigned long long *lowm = mask + (size_t)n * W;         // [n]: suppressors of a box inside its own block (bit i: box 64 * block + i, i < own position)
    // ---- phase 1: tile (rb, cb), cb >= rb, of the upper triangle; tiles are numbered row by row.  Sixteen waves: lane = row, a wave takes
    //      four columns (the column box is wave-uniform: scalar operands) and leaves its four bits of the row's word in LDS
    {
        int rb = 0, t = (int)blockIdx.x;
        while (t >= W - rb) { t -= W - rb; ++rb; }
        const int cb = rb + t;
        const int ncol = min(64, n - 64 * cb);
        const int row = 64 * rb + lane;
        const float *rbx = boxes + (size_t)min(row, n - 1) * 5;
        const float x1 = rbx[0], y1 = rbx[1], x2 = rbx[2], y2 = rbx[3];
        if (tid < ncol * 5) colbox[tid] = boxes[(size_t)64 * cb * 5 + tid];
        __syncthreads();
        if (row < n) {
            const float area = (x2 - x1 + 1.f) * (y2 - y1 + 1.f);
            unsigned int up = 0, low = 0;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int c = 4 * wave + k;                // column inside the tile (wave-uniform)
                if (c >= ncol) break;
                const float cx1 = colbox[c * 5], cy1 = colbox[c * 5 + 1], cx2 = colbox[c * 5 + 2], cy2 = colbox[c * 5 + 3];
                const float c_area = (cx2 - cx1 + 1.f) * (cy2 - cy1 + 1.f);
                // (the arguments in the order the greedy rule tests them: earlier box first)
                const bool later = cb > rb || c > lane;
                const bool hit = later ? nms_suppresses(x1, y1, x2, y2, area, cx1, cy1, cx2, cy2, c_area, thr)
                                       : (c != lane && nms_suppresses(cx1, cy1, cx2, cy2, c_area, x1, y1, x2, y2, area, thr));
                if (hit) { if (later) up |= 1u << k; else low |= 1u << k; }
            }
            pieces[0][lane][wave] = (unsigned char)up;
            pieces[1][lane][wave] = (unsigned char)low;
        }
        __syncthreads();
        // whole 64-bit words, write-through (sc1): the workgroup that sweeps reads them with sc1 loads, no cache maintenance in between
        if (wave < 2 && row < n && (wave == 0 || cb == rb)) {
            const uint4 pc = *reinterpret_cast<const uint4 *>(pieces[wave][lane]);
            auto squeeze = [](unsigned int x) { return (x & 0xfu) | ((x >> 4) & 0xf0u) | ((x >> 8) & 0xf00u) | ((x >> 12) & 0xf000u); };
            const unsigned long long word = (unsigned long long)(squeeze(pc.x) | (squeeze(pc.y) << 16)) |
                                            ((unsigned long long)(squeeze(pc.z) | (squeeze(pc.w) << 16)) << 32);
            __hip_atomic_store(wave == 0 ? mask + (size_t)row * W + cb : lowm + row, word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    // ---- who is last?  The storing waves drain their stores, the workgroup meets, ONE lane takes the ticket (MI355X_MICROARCH.md, inter-
    //      workgroup visibility: sc1 stores + drained + agent-scope counter; inputs too large for the LDS sweep add the release / acquire pair)
    const bool in_lds = g.lds_rows >= n;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
        if (!in_lds) { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent"); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
        const unsigned int old = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        last_flag = old == (unsigned int)(g.tiles - 1);
        if (dbg) { dbg[2 * blockIdx.x] = t_in; dbg[2 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime(); }
        if (last_flag) {
            __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // (ready for the next launch)
            if (!in_lds) { __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
        }
    }
    __syncthreads();
    if (!last_flag) return;
    // ---- phase 2
    if (dbg && tid == 0) { dbg[2 * g.tiles] = t_in; dbg[2 * g.tiles + 1] = __builtin_amdgcn_s_memrealtime(); }
    const int Ws = in_lds ? (W | 1) : W;                     // row stride of the words the sweep reads: odd in LDS (a lane per row: 64 rows, 64 banks)
    if (in_lds) {
        // 8-byte sc1 loads, all of a thread's loads in flight at once (1024 threads: the whole matrix is one batch).  The loads carry no
        // condition (a load under a condition is a branch with its own wait: one round trip to memory each) and the addresses no
        // division (one workgroup does all of this: its instruction count is the time): rows are walked as if they had 2^k words.
        // Words left of a row's diagonal block were never written: whatever the workspace held; the sweep never looks at them
        constexpr int NL = 17;
        const int sh = g.row_shift, wmask = (1 << sh) - 1, total = n << sh;
        unsigned long long lo_v = 0;
        if (tid < n) lo_v = __hip_atomic_load(lowm + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);    // (n <= 1024 in here)
        for (int i0 = 0; i0 < total; i0 += 1024 * NL) {
            unsigned long long v[NL];
#pragma unroll
            for (int u = 0; u < NL; ++u) {
                const int i = i0 + u * 1024 + tid, r = i >> sh, w = i & wmask;
                v[u] = __hip_atomic_load(mask + min(r * W + w, n * W - 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
#pragma unroll
            for (int u = 0; u < NL; ++u) {
                const int i = i0 + u * 1024 + tid, r = i >> sh, w = i & wmask;
                if (w < W && r < n) nms_lds[r * Ws + w] = v[u];
            }
        }
        if (tid < n) nms_lds[n * Ws + tid] = lo_v;
    }
    __syncthreads();
    if (dbg && tid == 0) dbg[2 * g.tiles + 2] = __builtin_amdgcn_s_memrealtime();
    if (wave != 0) return;
    // (two instances: LDS or global addresses known at compile time -- a generic pointer would make every read a flat load)
    auto sweep = [&](auto lds_tag) {
        constexpr bool L = decltype(lds_tag)::value;
        auto low_word = [&](int r) -> unsigned long long {
            if constexpr (L) return nms_lds[n * Ws + r]; else return lowm[r];
        };
        auto row_word = [&](int r, int w) -> unsigned long long {
            if constexpr (L) return nms_lds[r * Ws + w]; else return mask[(size_t)r * W + w];
        };
        int n_kept = 0;                                       // wave-uniform
        unsigned long long remv = 0;                          // lane w: word w of the removed bitmap (W <= 64)
        unsigned long long part[17];                          // (W <= 17) this lane's share of the words of the later blocks
#pragma unroll
        for (int w = 0; w < 17; ++w) part[w] = 0ull;
        unsigned long long mine = lane < n ? low_word(lane) : 0ull;             // my possible suppressors inside the block
        for (int b = 0; b < W; ++b) {
            const int nrow = min(64, n - 64 * b);
            const unsigned long long rem_b = nms_readlane64(remv, b);      // removed by the kept boxes of the earlier blocks
            const unsigned long long valid = nrow == 64 ? ~0ull : ((1ull << nrow) - 1ull);
            const int my_r = 64 * b + (lane < nrow ? lane : 0);
            // (off the dependency chain: the next block's own words and this row's word for the next block are on their way meanwhile)
            const unsigned long long mine_next = 64 * (b + 1) + lane < n ? low_word(64 * (b + 1) + lane) : 0ull;
            const unsigned long long row_next = b + 1 < W ? row_word(my_r, b + 1) : 0ull;
            unsigned long long und = ~rem_b & valid, kw = 0;  // undecided / kept boxes of the block (wave-uniform)
            while (und) {
                const bool me = (und >> lane) & 1ull;
                const bool by_kept = (mine & kw) != 0ull, pending = (mine & und) != 0ull;
                const unsigned long long now_kept = __ballot(me && !by_kept && !pending);
                const unsigned long long now_gone = __ballot(me && by_kept);
                kw |= now_kept;
                und &= ~(now_kept | now_gone);
            }
            const bool kept_me = (kw >> lane) & 1ull;
            // the kept boxes of this block suppress into the later blocks.  Up to 17 words (the LDS sweep): every lane ORs its own row's
            // later words into per-lane partial words (registers, no cross-lane traffic); only the NEXT block's word is reduced over the
            // lanes now (one DPP reduction per block).  Wider inputs: every later word is reduced right away, four per trip.
            if (W <= 17) {
                if (b + 1 < W) {
                    unsigned long long nxt = kept_me ? row_next : 0ull;
#pragma unroll
                    for (int w = 1; w < 17; ++w)
                        nxt |= w == b + 1 ? part[w] : 0ull;
                    const unsigned long long red = ((unsigned long long)nms_wave_or32((unsigned int)(nxt >> 32)) << 32) | nms_wave_or32((unsigned int)nxt);
                    if (lane == b + 1) remv |= red;
                }
                // (words at or left of the diagonal are zeros or never looked at again: no test on b.  No test on W either -- a test would
                //  put every read and its wait behind a branch of its own: the reads past the row's end fetch words of the next row (the
                //  LDS image has 17 spare words behind it) into partial words nobody looks at; all 15 reads are in flight together)
                unsigned long long x[15];
#pragma unroll
                for (int w = 2; w < 17; ++w) x[w - 2] = row_word(my_r, L ? w : min(w, W - 1));
#pragma unroll
                for (int w = 2; w < 17; ++w) part[w] |= kept_me ? x[w - 2] : 0ull;
            } else {
                for (int w0 = b + 1; w0 < W; w0 += 4) {
                    unsigned long long v[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) v[u] = (kept_me && w0 + u < W) ? row_word(my_r, w0 + u) : 0ull;
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const unsigned long long red = ((unsigned long long)nms_wave_or32((unsigned int)(v[u] >> 32)) << 32) | nms_wave_or32((unsigned int)v[u]);
                        if (lane == w0 + u) remv |= red;
                    }
                }
            }
            if (kept_me) keep[n_kept + __builtin_popcountll(kw & ((1ull << lane) - 1ull))] = 64 * b + lane;
            n_kept += __builtin_popcountll(kw);
            mine = mine_next;
        }
        if (lane == 0) *count = n_kept;
    };
    if (in_lds) sweep(std::true_type{}); else sweep(std::false_type{});
    if (dbg && lane == 0) dbg[2 * g.tiles + 3] = __builtin_amdgcn_s_memrealtime();
}

// ------------------------------------------------------------------------------------------ detector decode (centre / scale / offset head)
// The arithmetic between the head's top-k and its NMS (Pedestron/mmdet/models/anchor_heads/csp_head.py:229-284 get_bboxes_single +
// csp_height2bbox): position of candidate k from its flat index, centre = cell centre + offset * stride, height = exp(scale) * stride
// (the exp is the caller's: heights[]), width = wh_ratio * height, box clamped to the image, [x1, y1, x2, y2, score] rows in the
// candidates' order -- and the NUMBER of candidates whose score exceeds the threshold (scores come sorted descending from the top-k,
// so those are the first n_sel rows: what the reference selects with a boolean mask and a host round trip).  One launch instead of
// ~25 elementwise ones; every operation is the single IEEE operation the reference's tensor expression performs, in its order (no
// contraction), so the boxes are the reference's bit for bit.
struct DecodeGeom { int k, W, stride; float wh_ratio, x_max, y_max, thr; };

__global__ __launch_bounds__(1024) void k_csp_decode(DecodeGeom g, const float *__restrict__ scores, const long long *__restrict__ top,
                                                     const float *__restrict__ heights, const float *__restrict__ off_y,
                                                     const float *__restrict__ off_x, float *__restrict__ dets, int32_t *__restrict__ n_sel)
{
    __shared__ int wave_cnt[16];
    const float sf = (float)g.stride, half = (float)(g.stride / 2);
    int mine = 0;
    for (int k = threadIdx.x; k < g.k; k += 1024) {
        const long long i = top[k];
        const int row = (int)(i / g.W), col = (int)(i - (long long)row * g.W);
        const float px = __fadd_rn((float)(col * g.stride), half), py = __fadd_rn((float)(row * g.stride), half);
        const float x = __fadd_rn(px, __fmul_rn(off_x[k], sf)), y = __fadd_rn(py, __fmul_rn(off_y[k], sf));
        const float hh = __fmul_rn(heights[k], sf);
        const float a = __fmul_rn(__fmul_rn(g.wh_ratio, hh), 0.5f), b = __fmul_rn(hh, 0.5f);      // (x / 2 == x * 0.5 exactly)
        float *d = dets + (size_t)k * 5;
        d[0] = fminf(fmaxf(__fsub_rn(x, a), 0.0f), g.x_max);
        d[1] = fminf(fmaxf(__fsub_rn(y, b), 0.0f), g.y_max);
        d[2] = fminf(fmaxf(__fadd_rn(x, a), 0.0f), g.x_max);
        d[3] = fminf(fmaxf(__fadd_rn(y, b), 0.0f), g.y_max);
        const float sc = scores[k];
        d[4] = sc;
        mine += sc > g.thr ? 1 : 0;
    }
    // count: wave sums (DPP-free: ballots would need one pass per trip), then 16 partials
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mine += __shfl_xor(mine, o);
    if ((threadIdx.x & 63) == 0) wave_cnt[threadIdx.x >> 6] = mine;
    __syncthreads();
    if (threadIdx.x == 0) {
        int t = 0;
        for (int w = 0; w < 16; ++w) t += wave_cnt[w];
        *n_sel = t;
    }
}

// The head's top-k IN the decode launch (csp_head.py:262-267: cls.sigmoid().topk(nms_pre), then the gathers of scale and offset): one workgroup,
// no sort of the map, and no sigmoid outside the k selected rows.  The score is a function of the logit, and the fp32 expression
// 1 / (1 + exp(-x)) as this device evaluates it is monotone non-decreasing over the whole float line (k_csp_score_monotone checks every pair
// of neighbouring floats; tests/test_gpu_ops.py) -- so the k largest scores are the k largest LOGITS, and rows ordered by logit are ordered
// by score.  Order among equal scores (torch.topk leaves it unspecified): the larger logit first, equal logits: lowest position first.
// An exact radix select on the logits (as order-preserving integers), whatever their distribution -- hot regions, saturated scores, flat maps:
// (1) a sweep over the map (16-byte loads, eight in flight) histograms the top 11 bits -- one histogram per wave, and every lane
//     accumulates runs of equal digits in a register, so the common digit of a map costs no LDS atomics; the digit that holds the k-th largest
//     logit follows from a scan.  While the elements at or above that digit number more than 2048, the next 11 (10) bits inside it.
// (2) the elements at or above the final digit go to an LDS list as (logit, ~position) words, one LDS atomic per wave and 32 elements;
// (3) the list is sorted and its first k rows are scored and decoded.
// If 32 bits leave more than TOPK_CAP elements (a long run of EQUAL logits at the k-th: flat maps), the run is counted per block of 16-byte
// vectors in position order and a scan finds the position of its last selected element.
constexpr int TOPK_CAP = 8192, TOPK_RUN = 32768;
struct TopkGeom { int n, k, W, stride; float wh_ratio, x_max, y_max, thr; long long off_cs, off_ps; int cls_dtype, vec; };

// descending bitonic sort of P (a power of two) keys in LDS by 1024 threads.  Compare distances <= 64 stay inside the 128 keys a wave owns:
// those stages need no workgroup barrier (LDS operations of one wave complete in order)
template <typename K>
__device__ __forceinline__ void lds_bitonic_desc(K *s, int P)
{
    for (int size = 2; size <= P; size <<= 1)
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            for (int t = threadIdx.x; t < P / 2; t += 1024) {
                const int i = 2 * t - (t & (stride - 1)), j = i + stride;
                const K a = s[i], b = s[j];
                if ((a < b) == ((i & size) == 0)) { s[i] = b; s[j] = a; }
            }
            if (stride > 64 || stride == 1) __syncthreads();          // (stride 1 ends a merge: the next one starts across waves)
            else { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); }
        }
}

__device__ __forceinline__ float csp_score(float x) { return __fdiv_rn(1.0f, __fadd_rn(1.0f, expf(-x))); }
// floats in the order of the real line as unsigned integers (-inf = 0x007fffff ... +inf = 0xff800000; NaNs at either end)
__device__ __forceinline__ uint32_t float_ord(float f) { const uint32_t u = __builtin_bit_cast(uint32_t, f); return (u & 0x80000000u) ? ~u : (u | 0x80000000u); }
__device__ __forceinline__ float ord_float(uint32_t o) { return __builtin_bit_cast(float, (o & 0x80000000u) ? (o ^ 0x80000000u) : ~o); }

// self-test of the property the selection rests on: counts the neighbouring float pairs (x, next x) in [-inf, +inf] with score(x) > score(next x)
__global__ __launch_bounds__(256) void k_csp_score_monotone(unsigned long long *__restrict__ violations)
{
    unsigned long long bad = 0;
    for (unsigned long long o = 0x007fffffull + blockIdx.x * 256ull + threadIdx.x; o < 0xff800000ull; o += (unsigned long long)gridDim.x * 256ull)
        bad += csp_score(ord_float((uint32_t)o)) > csp_score(ord_float((uint32_t)o + 1u)) ? 1ull : 0ull;
#pragma unroll
    for (int sh = 32; sh > 0; sh >>= 1) bad += __shfl_xor(bad, sh);
    if ((threadIdx.x & 63) == 0 && bad) atomicAdd(violations, bad);
}

__global__ __launch_bounds__(1024) void k_csp_topk_decode(TopkGeom g, const void *__restrict__ cls, const float *__restrict__ reg,
                                                          const float *__restrict__ off, float *__restrict__ dets, int32_t *__restrict__ n_sel,
                                                          int32_t *__restrict__ top_out)
{
    extern __shared__ __align__(16) unsigned char topk_lds[];
    // one region, three lives: 16 histograms of 2048 bins (pitch 2049: the replicas of a bin sit in different banks) -> the run table ->
    // the candidate words; the summed histogram behind it
    uint32_t *rep = reinterpret_cast<uint32_t *>(topk_lds), *tot = rep + 16 * 2049;
    uint32_t *run = reinterpret_cast<uint32_t *>(topk_lds);
    unsigned long long *cand = reinterpret_cast<unsigned long long *>(topk_lds);
    __shared__ uint32_t s_cnt, s_digit, s_above, s_bin;
    __shared__ int wave_cnt[16];
    constexpr int U = 8;
    const int tid = threadIdx.x, lane = tid & 63;
    auto logit = [&](int i) -> float {
        if (g.cls_dtype == BC_F32) return static_cast<const float *>(cls)[i];
        if (g.cls_dtype == BC_F16) return __half2float(static_cast<const __half *>(cls)[i]);
        return Cvt<hip_bfloat16>::ld(static_cast<const hip_bfloat16 *>(cls) + i);
    };
    auto load4 = [&](int v, uint32_t *o) {        // float_ord of elements 4 v .. 4 v + 3
        if (g.vec) {
            if (4 * v < g.n) {
                const float4 q = static_cast<const float4 *>(cls)[v];
                o[0] = float_ord(q.x), o[1] = float_ord(q.y), o[2] = float_ord(q.z), o[3] = float_ord(q.w);
            } else o[0] = o[1] = o[2] = o[3] = 0u;
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = 4 * v + e < g.n ? float_ord(logit(4 * v + e)) : 0u;
        }
    };
    auto word_of = [&](uint32_t o, int i) -> unsigned long long { return ((unsigned long long)o << 32) | (uint32_t)(0xffffffffu - (uint32_t)i); };
    const int nvec = (g.n + 3) / 4, trips = (nvec + 1023) / 1024;
    // visit(o, i) for every element of the map, in batches of U vectors per thread
    auto sweep = [&](auto &&visit, auto &&batch_done) {
        for (int j0 = 0; j0 < trips; j0 += U) {
            uint32_t o[U][4];
            int vidx[U];
#pragma unroll
            for (int jj = 0; jj < U; ++jj) { vidx[jj] = (j0 + jj) * 1024 + tid; load4(vidx[jj], o[jj]); }
            uint32_t mask = 0;
#pragma unroll
            for (int jj = 0; jj < U; ++jj)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int i = 4 * vidx[jj] + e;
                    if (i < g.n && visit(o[jj][e], i)) mask |= 1u << (4 * jj + e);
                }
            batch_done(mask, o, vidx);
        }
    };
    auto nothing = [](uint32_t, const uint32_t (&)[U][4], const int (&)[U]) {};
    // (1) digits from the top until the elements at or above the k-th's digit fit the list
    uint32_t prefix = 0u, r = (uint32_t)g.k, total = 0u;      // r: rank of the k-th inside the current digit's bin; total: elements at or above the bin
    int consumed = 0;
    for (int pass = 0; pass < 3; ++pass) {
        const int bits = pass < 2 ? 11 : 10, shift = 32 - consumed - bits;
        for (int b = tid; b < 16 * 2049; b += 1024) rep[b] = 0u;
        __syncthreads();
        uint32_t *mine = rep + (tid >> 6) * 2049;
        uint32_t cd = 0u, cc = 0u;                  // a run of equal digits
        sweep([&](uint32_t o, int) {
            if (consumed == 0 || (o >> (32 - consumed)) == prefix) {
                const uint32_t d = (o >> shift) & ((1u << bits) - 1u);
                if (d == cd) ++cc;
                else { if (cc) atomicAdd(&mine[cd], cc); cd = d; cc = 1u; }
            }
            return false; }, nothing);
        if (cc) atomicAdd(&mine[cd], cc);
        __syncthreads();
        for (int b = tid; b < 2048; b += 1024) {
            uint32_t t = 0;
#pragma unroll
            for (int w = 0; w < 16; ++w) t += rep[w * 2049 + b];
            tot[b] = t;
        }
        __syncthreads();
        if (tid < 64) {                 // bins from the top: lane l owns bins 2047 - 32 l ... 2016 - 32 l
            uint32_t sum = 0;
            for (int q = 0; q < 32; ++q) sum += tot[2047 - (32 * lane + q)];
            uint32_t incl = sum;
#pragma unroll
            for (int sh = 1; sh < 64; sh <<= 1) {
                const uint32_t v = __shfl_up(incl, sh);
                if (lane >= sh) incl += v;
            }
            const uint32_t excl = incl - sum;
            if (excl < r && r <= incl) {
                uint32_t above = excl;
                for (int q = 0; q < 32; ++q) {
                    const uint32_t c = tot[2047 - (32 * lane + q)];
                    if (r <= above + c) { s_digit = 2047u - (uint32_t)(32 * lane + q); s_above = above; s_bin = c; break; }
                    above += c;
                }
            }
        }
        __syncthreads();
        total = ((uint32_t)g.k - r) + s_above + s_bin;
        r -= s_above;
        prefix = (prefix << bits) | s_digit;
        consumed += bits;
        __syncthreads();
        if (total <= 2048u) break;      // (a list of 2048 sorts in the time of one more sweep; anything up to TOPK_CAP is sorted if 32 bits leave it)
    }
    const uint32_t T = consumed == 32 ? prefix : (prefix << (32 - consumed));       // the lowest logit of the final bin
    // (2) candidates
    auto append = [&](uint32_t mask, const uint32_t (&o)[U][4], const int (&vidx)[U]) {        // mask: bit 4 jj + e = take element e of vector jj
        const uint32_t cnt = (uint32_t)__builtin_popcount(mask);
        uint32_t incl = cnt;
#pragma unroll
        for (int sh = 1; sh < 64; sh <<= 1) {
            const uint32_t v = __shfl_up(incl, sh);
            if (lane >= sh) incl += v;
        }
        const uint32_t all = __shfl(incl, 63);
        uint32_t base = 0;
        if (lane == 0 && all) base = atomicAdd(&s_cnt, all);
        base = __shfl(base, 0);
        uint32_t slot = base + incl - cnt;
#pragma unroll
        for (int jj = 0; jj < U; ++jj)
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (mask & (1u << (4 * jj + e))) {
                    if (slot < (uint32_t)TOPK_CAP) cand[slot] = word_of(o[jj][e], 4 * vidx[jj] + e);
                    ++slot;
                }
    };
    int last = g.n;                     // of the elements EQUAL to T, positions <= last are candidates
    if (total > (uint32_t)TOPK_CAP) {
        // 32 bits consumed: more than TOPK_CAP - k elements EQUAL the k-th largest logit, r of them are selected: the first r in position order.
        // Count the run per block of B vectors (position order), scan, walk the block that holds the r-th.
        const int B = (nvec + TOPK_RUN - 1) / TOPK_RUN, n_blk = (nvec + B - 1) / B, bpt = (n_blk + 1023) / 1024;
        for (int b = tid; b < n_blk; b += 1024) run[b] = 0u;
        __syncthreads();
        sweep([&](uint32_t o, int i) { if (o == T) atomicAdd(&run[(i >> 2) / B], 1u); return false; }, nothing);
        __syncthreads();
        uint32_t mine = 0;
        for (int q = 0; q < bpt; ++q) mine += (tid * bpt + q) < n_blk ? run[tid * bpt + q] : 0u;
        uint32_t incl = mine;
#pragma unroll
        for (int sh = 1; sh < 64; sh <<= 1) {
            const uint32_t v = __shfl_up(incl, sh);
            if (lane >= sh) incl += v;
        }
        if (lane == 63) wave_cnt[tid >> 6] = (int)incl;
        __syncthreads();
        uint32_t before = incl - mine;
        for (int w = 0; w < (tid >> 6); ++w) before += (uint32_t)wave_cnt[w];
        if (before < r && r <= before + mine) {
            for (int q = 0; q < bpt; ++q) {
                const int blk = tid * bpt + q;
                const uint32_t c = blk < n_blk ? run[blk] : 0u;
                if (r <= before + c) {
                    bool found = false;
                    for (int i = 4 * blk * B; i < g.n && !found; ++i)
                        if (float_ord(logit(i)) == T && ++before == r) { s_bin = (uint32_t)i; found = true; }
                    break;
                }
                before += c;
            }
        }
        __syncthreads();
        last = (int)s_bin;
        __syncthreads();
    }
    if (tid == 0) s_cnt = 0u;
    __syncthreads();
    sweep([&](uint32_t o, int i) { return o > T || (o == T && i <= last); }, append);
    __syncthreads();
    const uint32_t n_cand = s_cnt;      // k <= n_cand <= TOPK_CAP
    // (3) sort the candidates (padding: 0 < every word), score and decode the first k
    int P = 128;
    while (P < (int)n_cand) P <<= 1;
    for (int t = (int)n_cand + tid; t < P; t += 1024) cand[t] = 0ull;
    __syncthreads();
    lds_bitonic_desc(cand, P);
    const float sf = (float)g.stride, half = (float)(g.stride / 2);
    int mine = 0;
    for (int k = tid; k < g.k; k += 1024) {
        const unsigned long long wd = cand[k];
        const int i = (int)(0xffffffffu - (uint32_t)wd);
        const float sc = csp_score(ord_float((uint32_t)(wd >> 32)));
        const int row = i / g.W, col = i - row * g.W;
        const float px = __fadd_rn((float)(col * g.stride), half), py = __fadd_rn((float)(row * g.stride), half);
        const float oy = off[(size_t)i * g.off_ps], ox = off[(size_t)g.off_cs + (size_t)i * g.off_ps];
        const float x = __fadd_rn(px, __fmul_rn(ox, sf)), y = __fadd_rn(py, __fmul_rn(oy, sf));
        const float hh = __fmul_rn(expf(reg[i]), sf);
        const float a = __fmul_rn(__fmul_rn(g.wh_ratio, hh), 0.5f), b = __fmul_rn(hh, 0.5f);
        float *d = dets + (size_t)k * 5;
        d[0] = fminf(fmaxf(__fsub_rn(x, a), 0.0f), g.x_max);
        d[1] = fminf(fmaxf(__fsub_rn(y, b), 0.0f), g.y_max);
        d[2] = fminf(fmaxf(__fadd_rn(x, a), 0.0f), g.x_max);
        d[3] = fminf(fmaxf(__fadd_rn(y, b), 0.0f), g.y_max);
        d[4] = sc;
        if (top_out) top_out[k] = i;
        mine += sc > g.thr ? 1 : 0;
    }
#pragma unroll
    for (int sh = 32; sh > 0; sh >>= 1) mine += __shfl_xor(mine, sh);
    if (lane == 0) wave_cnt[tid >> 6] = mine;
    __syncthreads();
    if (tid == 0) {
        int t = 0;
        for (int w = 0; w < 16; ++w) t += wave_cnt[w];
        *n_sel = t;
    }
}

// ------------------------------------------------------------------------------------------ L2 normalisation into a channel slice
// out[p][c_off + c] = weight[c] * (x[p][c] / (sqrt(sum_c x[p][c]^2) + eps)) for the pixels p of a channels-last tensor: the detector
// neck's L2Norm (Pedestron/mmdet/models/necks/csp_neck.py:85 -- pow, sum over channels, sqrt, + eps, div, scale: six elementwise /
// reduction passes) AND its channel concatenation (csp_neck.py:83) in one read and one write.  One wave per pixel and trip: a lane holds
// C / 64 / 4 float4 (C <= 1024), the sum of squares goes through a fixed-order DPP reduction.
template <typename T, int NV>
__global__ __launch_bounds__(256) void k_l2norm_cat(T *__restrict__ out, const T *__restrict__ x, const float *__restrict__ weight, long long n_pix,
                                                    uint32_t C, uint32_t C_total, uint32_t c_off, float eps)
{
    constexpr int EPV = 16 / sizeof(T);
    const uint32_t lane = threadIdx.x & 63, vecs = C / EPV;
    const long long wave0 = (long long)blockIdx.x * 4 + (threadIdx.x >> 6), n_waves = (long long)gridDim.x * 4;
    for (long long p = wave0; p < n_pix; p += n_waves) {
        uint4 v[NV];
        float sq = 0.0f;
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            const uint32_t q = lane + 64 * k;
            v[k] = q < vecs ? reinterpret_cast<const uint4 *>(x + (size_t)p * C)[q] : make_uint4(0u, 0u, 0u, 0u);
            const T *e = reinterpret_cast<const T *>(&v[k]);
#pragma unroll
            for (int j = 0; j < EPV; ++j) { const float f = Cvt<T>::ld(e + j); sq = fmaf(f, f, sq); }
        }
        // wave sum, fixed order: row prefix (DPP), row totals to the last row, lane 63 read back
        sq += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, sq), 0x111, 0xf, 0xf, false));
        sq += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, sq), 0x112, 0xf, 0xf, false));
        sq += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, sq), 0x114, 0xf, 0xf, false));
        sq += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, sq), 0x118, 0xf, 0xf, false));
        sq += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, sq), 0x142, 0xa, 0xf, false));
        sq += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, sq), 0x143, 0xc, 0xf, false));
        const float norm = sqrtf(__builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, sq), 63))) + eps;
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            const uint32_t q = lane + 64 * k;
            if (q >= vecs) continue;
            uint4 o;
            T *eo = reinterpret_cast<T *>(&o);
            const T *e = reinterpret_cast<const T *>(&v[k]);
#pragma unroll
            for (int j = 0; j < EPV; ++j) eo[j] = Cvt<T>::st(weight[q * EPV + j] * (Cvt<T>::ld(e + j) / norm));
            reinterpret_cast<uint4 *>(out + (size_t)p * C_total + c_off)[q] = o;
        }
    }
}

// The detector neck's TRANSPOSED convs (Pedestron/mmdet/models/necks/csp_neck.py:37-39: k4 s2 p1 on the stride-8 stage, k4 s4 p0 on the stride-16
// stages) as a pointwise GEMM + this pass.  conv_transpose2d(x, W)[Y][X][co] = sum over the taps (ky, kx) that reach the pixel of
// sum_ci x[y][x][ci] W[ci][co][ky][kx]: the inner sum for ALL 16 taps of an input pixel is ONE 1x1 conv to 16 C channels (t[y][x][(4 ky + kx) C + co],
// bc_conv1x1_nhwc on the matrix cores, exactly the transposed conv's multiplications -- no zeros inserted, no im2col), and what is left is a
// gather: stride 4 -- every output pixel has one tap, (y, ky) = (Y / 4, Y % 4): depth-to-space; stride 2, pad 1 -- up to 2 x 2 taps,
// Y = 2 y + ky - 1, contributions from outside the TILE are absent (the reference applies the layer to packed tiles without a halo).  The gather
// rides in the L2Norm + concat pass that reads the result anyway (k_l2norm_cat): the up-sampled map is never written.  + bias, one wave per
// output pixel, C = 64 lanes x 4 channels (C <= 256 per 16-byte lane vector of fp32; 16-bit: 8 per lane).
template <typename T, int STRIDE>
__global__ __launch_bounds__(256) void k_l2norm_cat_deconv(T *__restrict__ out, const T *__restrict__ t, const float *__restrict__ bias,
                                                           const float *__restrict__ weight, uint32_t n_img, uint32_t h, uint32_t w, uint32_t C,
                                                           uint32_t C_total, uint32_t c_off, float eps)
{
    constexpr int EPV = 16 / sizeof(T);
    const uint32_t lane = threadIdx.x & 63, vecs = C / EPV;
    const uint32_t H = h * STRIDE, W = w * STRIDE;
    const long long n_pix = (long long)n_img * H * W;
    const long long wave0 = (long long)blockIdx.x * 4 + (threadIdx.x >> 6), n_waves = (long long)gridDim.x * 4;
    for (long long p = wave0; p < n_pix; p += n_waves) {
        const uint32_t X = (uint32_t)(p % W), Y = (uint32_t)((p / W) % H), img = (uint32_t)(p / ((long long)W * H));
        float f[EPV];
#pragma unroll
        for (int j = 0; j < EPV; ++j) f[j] = 0.0f;
        if (lane < vecs) {
            auto add_tap = [&](uint32_t y, uint32_t x, uint32_t ky, uint32_t kx) {
                const uint4 v = reinterpret_cast<const uint4 *>(t + ((((size_t)img * h + y) * w + x) * 16 + (ky * 4 + kx)) * C)[lane];
                const T *e = reinterpret_cast<const T *>(&v);
#pragma unroll
                for (int j = 0; j < EPV; ++j) f[j] += Cvt<T>::ld(e + j);
            };
            if (STRIDE == 4) {
                add_tap(Y >> 2, X >> 2, Y & 3, X & 3);
            } else {
                // Y = 2 y + ky - 1: ky has the parity of Y + 1; (ky, y) = (p, (Y + 1 - p) / 2), (p + 2, (Y - 1 - p) / 2)
                const uint32_t py = (Y + 1) & 1, px = (X + 1) & 1;
#pragma unroll
                for (int a = 0; a < 2; ++a) {
                    const int y = ((int)Y + 1 - (int)py - 2 * a) / 2;
                    if ((int)Y + 1 - (int)py - 2 * a < 0 || y >= (int)h) continue;
#pragma unroll
                    for (int b = 0; b < 2; ++b) {
                        const int x = ((int)X + 1 - (int)px - 2 * b) / 2;
                        if ((int)X + 1 - (int)px - 2 * b < 0 || x >= (int)w) continue;
                        add_tap((uint32_t)y, (uint32_t)x, py + 2 * a, px + 2 * b);
                    }
                }
            }
#pragma unroll
            for (int j = 0; j < EPV; ++j) {
                if (bias) f[j] += bias[lane * EPV + j];
                f[j] = Cvt<T>::ld_round(f[j]);      // (the transposed conv's result in the tensor's type, as the stock op stores it)
            }
        }
        float sq = 0.0f;
#pragma unroll
        for (int j = 0; j < EPV; ++j) sq = fmaf(f[j], f[j], sq);
        sq += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, sq), 0x111, 0xf, 0xf, false));
        sq += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, sq), 0x112, 0xf, 0xf, false));
        sq += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, sq), 0x114, 0xf, 0xf, false));
        sq += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, sq), 0x118, 0xf, 0xf, false));
        sq += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, sq), 0x142, 0xa, 0xf, false));
        sq += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, sq), 0x143, 0xc, 0xf, false));
        const float norm = sqrtf(__builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, sq), 63))) + eps;
        if (lane < vecs) {
            uint4 o;
            T *eo = reinterpret_cast<T *>(&o);
#pragma unroll
            for (int j = 0; j < EPV; ++j) eo[j] = Cvt<T>::st(weight[lane * EPV + j] * (f[j] / norm));
            reinterpret_cast<uint4 *>(out + (size_t)p * C_total + c_off)[lane] = o;
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

// copy kernels: any unit size (1..2^20 bytes): a channels-last map is an NCHW map of C*E-byte "fat" elements
bool elem_ok(int e) { return e >= 1 && e <= (1 << 20); }

// alignment demanded of a pointer for unit size e: the largest power of two dividing e, at most 16
bool aligned(const void *p, int e)
{
    int a = 1;
    while (a < 16 && (e % (a * 2)) == 0) a *= 2;
    return (reinterpret_cast<uintptr_t>(p) % (uintptr_t)a) == 0;
}

int grid_for(uint64_t items, int per_thread)
{
    const uint64_t wgs = (items + (uint64_t)WG * per_thread - 1) / ((uint64_t)WG * per_thread);
    return (int)(wgs < 1 ? 1 : (wgs > MAX_WG ? MAX_WG : wgs));
}

// exact cover: every lane handles `per_thread` items, no grid-stride loop (items < 2^31 => grid < 2^23)
int grid_exact(uint64_t items, int per_thread)
{
    const uint64_t wgs = (items + (uint64_t)WG * per_thread - 1) / ((uint64_t)WG * per_thread);
    return (int)(wgs < 1 ? 1 : wgs);
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
    double aux[BC_OP_COUNT] = {0};     // second per-op total (conv3x3: matrix FLOPs actually ISSUED -- the Winograd form issues 16/36 of the direct count)
} g_prof;

// When an op is being profiled its kernel is launched with hipExtLaunchKernelGGL, which attaches the start/stop events
// to the dispatch packet itself: the elapsed time is the kernel's own execution time (what rocprofv3 reports), not the
// launch bracket of two separately recorded events (which adds ~4 us of dispatch latency to a 5 us kernel).
struct ProfScope {
    int op;
    bool on;
    ProfState::Rec rec;
    ProfScope(int op_, double bytes) : op(op_), on(false)
    {
        rec.a = rec.b = nullptr;
        if (!(g_prof.mask & (1u << op))) return;
        std::lock_guard<std::mutex> lk(g_prof.mu);
        if (!g_prof.pool.empty()) { rec = g_prof.pool.back(); g_prof.pool.pop_back(); }
        else {
            // timing-only events: no system-scope fence (its cache write-back would be charged to the timed kernel;
            // measured +0.8-1.1 us on a 6 us launch).  Readers synchronise the stream before bc_prof_read.
            if (hipEventCreateWithFlags(&rec.a, hipEventDisableSystemFence) != hipSuccess) return;
            if (hipEventCreateWithFlags(&rec.b, hipEventDisableSystemFence) != hipSuccess) { (void)hipEventDestroy(rec.a); return; }
        }
        g_prof.bytes[op] += bytes;
        on = true;
    }
    void add_aux(double v) const
    {
        if (!on) return;
        std::lock_guard<std::mutex> lk(g_prof.mu);
        g_prof.aux[op] += v;
    }
    ~ProfScope()
    {
        if (!on) return;
        std::lock_guard<std::mutex> lk(g_prof.mu);
        g_prof.pending[op].push_back(rec);
    }
};

#define BC_LAUNCH(ps_, kernel_, grid_, block_, lds_, st_, ...)                                                     \
    do {                                                                                                           \
        if ((ps_).on) hipExtLaunchKernelGGL(kernel_, grid_, block_, lds_, st_, (ps_).rec.a, (ps_).rec.b, 0, __VA_ARGS__); \
        else hipLaunchKernelGGL(kernel_, grid_, block_, lds_, st_, __VA_ARGS__);                                   \
    } while (0)

// ---- executed-tile count on the device: bc_dyn_set ARMS the next launch only (the capable launchers take it at their first line,
// armed or not, so an armed state can never leak into a later, unrelated launch of theirs)
struct DynArm { const int32_t *ptr = nullptr; int ceiling = 0; };
thread_local DynArm g_dyn_arm;      // per host thread: an arm set by one thread is only ever consumed by that thread's next launch

DynArm dyn_take()
{
    const DynArm d = g_dyn_arm;
    g_dyn_arm = DynArm{};
    return d;
}

// ---- bc_conv_upsample_arm: "+ bilinear(src)" term for the epilogue of the NEXT bc_conv1x1_nhwc (taken and cleared at its first lines)
struct UpsampleArm { const void *src = nullptr; int src_bs = 0, out_bs = 0, align = 0; float rh = 0.0f, rw = 0.0f; };
thread_local UpsampleArm g_up_arm;

// tile-indexed launch (units = executed tiles x per_tile): the launch must have been sized for the ceiling
bool dyn_tiles(const DynArm &a, int n_exec, uint32_t per_tile, DynCount &d)
{
    d = DynCount{};
    if (!a.ptr) return true;
    if (n_exec != a.ceiling) return false;
    d.ptr = a.ptr; d.num = per_tile; d.den = 1;
    return true;
}

// pointwise launch over `units` items that belong to the ceiling's tiles in equal shares (rounds up into the last tile's garbage rows)
DynCount dyn_flat(const DynArm &a, unsigned long long units)
{
    DynCount d;
    if (!a.ptr || a.ceiling <= 0) return d;
    unsigned long long x = units, y = (unsigned long long)a.ceiling;
    while (y) { const unsigned long long t = x % y; x = y; y = t; }
    d.ptr = a.ptr; d.num = (uint32_t)(units / x); d.den = (uint32_t)((unsigned long long)a.ceiling / x);
    return d;
}

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
int launch_tiles(ProfScope &ps, void *packed, void *dense, const int32_t *mapping_exec, int n_exec,
                 int N, int C, int H, int W, int bs, int E, hipStream_t st, const DynArm &arm = DynArm{})
{
    const int vb = pick_vb((size_t)bs * E, {packed, dense});
    TileGeom g;
    const uint32_t vpr = (uint32_t)((size_t)bs * E / vb);
    g.vpr = make_fd(vpr); g.bs = make_fd(bs); g.C = make_fd(C); g.GW = make_fd(W / bs); g.GH = make_fd(H / bs);
    g.H = H; g.bsz = bs; g.vprW = (uint32_t)((size_t)W * E / vb);
    g.total = (uint32_t)((uint64_t)n_exec * C * bs * vpr);
    // small launches are one wave of workgroups: one vector per lane gives the shortest critical path;
    // large ones amortise index math and keep UNROLL requests per lane in flight
    const bool small = g.total < SMALL_LAUNCH_VECTORS;
    const int grid = grid_exact(g.total, small ? 1 : UNROLL);
    DynCount dyn;
    if (!dyn_tiles(arm, n_exec, (uint32_t)C * bs * vpr, dyn)) return BC_ERR_SHAPE;
#define BC_TILES(VB_)                                                                                          \
    case VB_:                                                                                                  \
        if (small)                                                                                             \
            BC_LAUNCH(ps, (k_tiles<VB_, TO_PACKED, 1>), dim3(grid), dim3(WG), 0, st,                           \
                      (VecOf<VB_>::type *)packed, (const VecOf<VB_>::type *)packed,                            \
                      (VecOf<VB_>::type *)dense, (const VecOf<VB_>::type *)dense, mapping_exec, g, dyn);       \
        else                                                                                                   \
        BC_LAUNCH(ps, (k_tiles<VB_, TO_PACKED, UNROLL>), dim3(grid), dim3(WG), 0, st,                     \
                           (VecOf<VB_>::type *)packed, (const VecOf<VB_>::type *)packed,                       \
                           (VecOf<VB_>::type *)dense, (const VecOf<VB_>::type *)dense, mapping_exec, g, dyn);  \
        break;
    switch (vb) { BC_TILES(16) BC_TILES(8) BC_TILES(4) BC_TILES(2) BC_TILES(1) }
#undef BC_TILES
    return launch_status();
}

template <bool RING>
int launch_halo_simple(ProfScope &ps, void *out, const void *features, const void *other_r, void *ring_w, const int32_t *grid_idx,
                       const int32_t *mapping_exec, int n_exec, int N, int C, int GH, int GW, int bs, int pad, int E,
                       hipStream_t st, const DynCount &dyn = DynCount{})
{
    HaloGeom g;
    g.dyn = dyn;
    const uint32_t bsp = bs + 2 * pad;
    g.PP = make_fd(bsp * bsp); g.BSP = make_fd(bsp); g.GW = make_fd(GW); g.GH = make_fd(GH);
    g.C = C; g.bs = bs; g.pad = pad; g.n_total = (uint32_t)N * GH * GW;
    g.per_tile = (uint32_t)C * bsp * bsp;
    uint64_t gx = ((uint64_t)g.per_tile + WG * 4 - 1) / (WG * 4);
    if (gx > 64) gx = 64;
    if (gx < 1) gx = 1;
    const dim3 grid((unsigned)gx, (unsigned)n_exec);
#define BC_HALO(T_)                                                                                            \
    BC_LAUNCH(ps, (k_halo<T_, RING>), grid, dim3(WG), 0, st, (T_ *)out, (const T_ *)features,             \
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

// Which halo-gather kernel a launch uses.  Measured on MI355X (profiles/r01/kbench_*.txt): below ~30 MB of traffic
// the launch is a single wave of workgroups and the register-only row kernel has the shortest critical path
// (6-8 us); above it the LDS-staged kernel's aligned stores win (3.7-4.4 TB/s vs 2.4-2.8 TB/s).
// BC_HALO_KERNEL = auto (default) | rows | lds | simple overrides the choice for A/B measurements.
enum { HALO_AUTO = -1, HALO_ROWS = 0, HALO_LDS = 1, HALO_SIMPLE = 2 };
int halo_kernel_override()
{
    static const int v = [] {
        const char *e = getenv("BC_HALO_KERNEL");
        if (!e) return (int)HALO_AUTO;
        return e[0] == 'r' ? (int)HALO_ROWS : (e[0] == 'l' ? (int)HALO_LDS : (e[0] == 's' ? (int)HALO_SIMPLE : (int)HALO_AUTO));
    }();
    return v;
}
int halo_kernel_choice(double bytes)
{
    const int o = halo_kernel_override();
    return o != HALO_AUTO ? o : (bytes < 30e6 ? HALO_ROWS : HALO_LDS);
}

// dt: 0 = pure copy, BC_F32+1 / BC_F16+1 / BC_BF16+1 = fused activation prologue in that dtype (ring form only)
template <bool RING>
int launch_halo(ProfScope &ps, void *out, const void *features, const void *other_r, void *ring_w, const int32_t *grid_idx,
                const int32_t *mapping_exec, int n_exec, int N, int C, int GH, int GW, int bs, int pad, int E,
                hipStream_t st, int dt = 0, Prologue pr = Prologue{nullptr, nullptr, 0}, const DynCount &dyn = DynCount{})
{
    int choice = halo_kernel_choice(2.0 * n_exec * C * (double)(bs + 2 * pad) * (bs + 2 * pad) * E);
    if (dt != 0 && choice == HALO_SIMPLE) choice = HALO_ROWS;   // the element-wise fallback has no prologue
    if ((E != 2 && E != 4) || choice == HALO_SIMPLE)
        return launch_halo_simple<RING>(ps, out, features, other_r, ring_w, grid_idx, mapping_exec, n_exec, N, C, GH, GW, bs, pad, E, st, dyn);
    const int vb = pick_vb((size_t)bs * E, {features, other_r, ring_w});
    if (choice == HALO_ROWS) {
        HaloRowsGeom g;
        g.dyn = dyn;
        const uint32_t bsp = bs + 2 * pad, vpr = (uint32_t)((size_t)bs * E / vb);
        g.BSP = make_fd(bsp); g.vpr = make_fd(vpr); g.P2 = make_fd(2 * pad); g.GW = make_fd(GW); g.GH = make_fd(GH);
        g.C = C; g.bs = bs; g.pad = pad; g.n_total = (uint32_t)N * GH * GW; g.plane = (uint32_t)bs * bs;
        g.per_tile = (uint32_t)C * bsp * bsp;
        g.mid_items = (uint32_t)C * bsp * vpr;
        g.edge_items = (uint32_t)C * bsp * 2 * pad;
        const uint32_t gx = (g.mid_items + WG * HALO_UM - 1) / (WG * HALO_UM);
        g.edge_per_wg = (g.edge_items + gx - 1) / gx;
        const dim3 grid(gx, (unsigned)n_exec);
        const long long delta = ((const char *)other_r - (const char *)features) / E;
#define BC_HR(T_, VE_, DT_)                                                                                    \
        BC_LAUNCH(ps, (k_halo_rows<T_, VE_, RING, DT_>), grid, dim3(WG), 0, st, (T_ *)out, (const T_ *)features, \
                  delta, (T_ *)ring_w, grid_idx, mapping_exec, g, pr)
#define BC_HR4(DT_) do { if (ve >= 4) BC_HR(uint32_t, 4, DT_); else if (ve == 2) BC_HR(uint32_t, 2, DT_); else BC_HR(uint32_t, 1, DT_); } while (0)
#define BC_HR2(DT_) do { if (ve >= 8) BC_HR(uint16_t, 8, DT_); else if (ve == 4) BC_HR(uint16_t, 4, DT_);       \
                         else if (ve == 2) BC_HR(uint16_t, 2, DT_); else BC_HR(uint16_t, 1, DT_); } while (0)
        const int ve = vb / E;
        if (E == 4) { if (RING && dt == 1) BC_HR4(1); else BC_HR4(0); }
        else { if (RING && dt == 2) BC_HR2(2); else if (RING && dt == 3) BC_HR2(3); else BC_HR2(0); }
#undef BC_HR2
#undef BC_HR4
#undef BC_HR
        return launch_status();
    }
    HaloLdsGeom g;
    g.dyn = dyn;
    const uint32_t bsp = bs + 2 * pad;
    g.BSP = make_fd(bsp); g.vpr = make_fd((uint32_t)((size_t)bs * E / vb)); g.P2 = make_fd(2 * pad);
    g.GW = make_fd(GW); g.GH = make_fd(GH);
    g.C = C; g.bs = bs; g.pad = pad; g.n_total = (uint32_t)N * GH * GW;
    g.per_tile = (uint32_t)C * bsp * bsp;
    g.plane = (uint32_t)bs * bs;
    // range length L: (1) a range of L elements spans at most L/BSP + 1 padded rows, and one batch covers
    // HALO_UM*WG middle vectors and HALO_UE*WG edge elements; (2) aim at >= ~2048 workgroups per launch.
    const uint64_t rows_mid = (uint64_t)HALO_UM * WG / g.vpr.d, rows_edge = (uint64_t)HALO_UE * WG / (2 * pad);
    const uint64_t rows_max = rows_mid < rows_edge ? rows_mid : rows_edge;
    if (rows_max < 2 && dt != 0) return BC_ERR_SHAPE;
    if (rows_max < 2)
        return launch_halo_simple<RING>(ps, out, features, other_r, ring_w, grid_idx, mapping_exec, n_exec, N, C, GH, GW, bs, pad, E, st, dyn);
    uint64_t L = (rows_max - 1) * bsp;
    const uint64_t total = (uint64_t)g.per_tile * n_exec;
    const uint64_t want = (total + 2047) / 2048, Lmin = 2048 / E;
    if (L > want) L = want < Lmin ? (Lmin < L ? Lmin : L) : want;
    if (L > 32768 / (uint64_t)E) L = 32768 / E;
    if (L > g.per_tile) L = g.per_tile;
    g.L = (uint32_t)L;
    const uint32_t chunks = (g.per_tile + g.L - 1) / g.L;
    const size_t lds = HALO_TBL + ((size_t)(g.L + 16 / E + 1) * E + 15) / 16 * 16;
    const dim3 grid(chunks, (unsigned)n_exec);
    const long long delta = ((const char *)other_r - (const char *)features) / E;
#define BC_HL(T_, VE_, DT_)                                                                                    \
    BC_LAUNCH(ps, (k_halo_lds<T_, VE_, RING, DT_>), grid, dim3(WG), lds, st, (T_ *)out, (const T_ *)features,  \
              delta, (T_ *)ring_w, grid_idx, mapping_exec, g, pr)
#define BC_HL4(DT_) do { if (ve >= 4) BC_HL(uint32_t, 4, DT_); else if (ve == 2) BC_HL(uint32_t, 2, DT_); else BC_HL(uint32_t, 1, DT_); } while (0)
#define BC_HL2(DT_) do { if (ve >= 8) BC_HL(uint16_t, 8, DT_); else if (ve == 4) BC_HL(uint16_t, 4, DT_);       \
                         else if (ve == 2) BC_HL(uint16_t, 2, DT_); else BC_HL(uint16_t, 1, DT_); } while (0)
    const int ve = vb / E;
    if (E == 4) { if (RING && dt == 1) BC_HL4(1); else BC_HL4(0); }
    else { if (RING && dt == 2) BC_HL2(2); else if (RING && dt == 3) BC_HL2(3); else BC_HL2(0); }
#undef BC_HL2
#undef BC_HL4
#undef BC_HL
    return launch_status();
}

int check_halo(const void *out, const void *features, const int32_t *grid_idx, const int32_t *mapping_exec,
               int n_exec, int N, int C, int GH, int GW, int bs, int pad, int E)
{
    if (E != 1 && E != 2 && E != 4 && E != 8) return BC_ERR_ELEM;   // NCHW halo kernels are element-typed
    if (n_exec < 0 || N <= 0 || C <= 0 || GH <= 0 || GW <= 0 || bs <= 0 || pad < 1 || pad > bs) return BC_ERR_SHAPE;
    if (n_exec == 0) return BC_OK;
    if (!out || !features || !grid_idx || !mapping_exec) return BC_ERR_NULL;
    const uint64_t bsp = (uint64_t)bs + 2 * pad;
    if ((uint64_t)n_exec * C * bsp * bsp >= (1ull << 31)) return BC_ERR_RANGE;
    if ((uint64_t)N * GH * GW * C * bs * bs >= (1ull << 31)) return BC_ERR_RANGE;
    if (!aligned(out, E) || !aligned(features, E)) return BC_ERR_ALIGN;
    return BC_OK;
}

#ifndef BC_NBR_KO
#define BC_NBR_KO 0
#endif
#include "conv3x3_mfma.inc"
#include "conv3x3_v2.inc"
#include "conv3x3_wino.inc"
#include "conv3x3_wino32.inc"
#include "conv3x3_wino4.inc"
#include "stem7x7.inc"
#include "head1x1.inc"
#include "pred3x3.inc"
#include "gemm1x1.inc"
#include "spp.inc"

}  // namespace

// ---- host side of conv3x3_v2.inc: choose the decomposition whose workgroup count best fills whole rounds of one
// 8-wave workgroup per CU with equal MFMA counts per wave, then launch it.
struct Conv2Cfg { int RM, RN, WMW, WNW, WKW; };

static int device_cu_count()
{
    static const int n = [] {
        int dev = 0, cus = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0)
            cus = 256;
        return cus;
    }();
    return n;
}

static const Conv2Cfg CONV2_CFGS[] = {{2, 2, 4, 2, 1}, {2, 1, 4, 2, 1}, {1, 2, 4, 2, 1}, {1, 1, 4, 2, 1},     // 0-3: many pixels
                                      {1, 1, 2, 4, 1}, {1, 1, 2, 2, 2}, {1, 1, 1, 4, 2}, {1, 1, 1, 2, 4},     // 4-7: 1x1 wave tiles
                                      {2, 2, 2, 2, 2}, {2, 2, 1, 2, 4}, {2, 1, 2, 2, 2}, {2, 1, 1, 4, 2},     // 8-11: 2-block tiles + K groups
                                      {2, 1, 1, 2, 4}, {2, 1, 1, 1, 8}, {1, 2, 1, 1, 8}, {1, 1, 1, 1, 8},     // 12-15
                                      {2, 2, 4, 1, 2}, {2, 2, 2, 1, 4}, {1, 2, 4, 1, 2}, {1, 2, 2, 1, 4}};    // 16-19: one 64-channel column (2x2 / 1x2 wave tiles)

// geometry of one decomposition for a layer (bs = OUTPUT tile size); false if the decomposition does not cover the layer
struct Conv2Plan { long long wgs; size_t lds_bytes; uint32_t n_rows, patches_per_tile, patches_x; };

// split = the fp32 tensor on the 16-bit matrix pipe (BC_F32S, code | 0x2000): steps are consumed in pairs, so a wave must own an even number of
// steps per tap -- K-group decompositions stage more 32-channel units per iteration (WKW 4: 2, WKW 8: 4)
static bool conv2_plan(const Conv2Cfg &k, int E, int S, int n_exec, int Cin, int Cout, int bs, Conv2Plan &p, int KS = 3, int D = 1, bool split = false)
{
    const int pw = bs == 4 ? 4 : (bs == 2 ? 2 : 8);
    const int uv = E == 4 ? 8 : 4, sc_lo = E == 4 ? 1 : 2;
    const int sc = split ? (k.WKW == 8 ? 4 : (k.WKW == 4 ? 2 : 1)) : (k.WKW == 8 ? 2 * sc_lo : sc_lo);
    if (pw == 8 && bs % 8 != 0) return false;
    if (pw == 4 && k.RM != 1 && k.WMW != 1) return false;       // 4x4 tiles: 4 tile slots per wave row only in single-row workgroups (LDS)
    if (pw == 2 && k.RM != 1) return false;                     // 2x2 tiles: eight tiles per wave row, one 32-pixel block
    if (pw == 8 && bs % (4 * k.RM) != 0) return false;
    if (Cout % (32 * k.RN * k.WNW) != 0 || Cin % (32 * sc) != 0) return false;
    const uint32_t ph = pw == 8 ? 4u * k.RM : (uint32_t)pw, tpr = pw == 8 ? 1u : (32u / (pw * pw)) * k.RM;
    const int ss = KS == 1 ? 1 : S;                              // (a pointwise stride-2 conv stages only the pixels it reads)
    const uint32_t slot_px = (uint32_t)(ss * (pw - 1) + D * (KS - 1) + 1) * (ss * (ph - 1) + D * (KS - 1) + 1);
    const size_t img = (size_t)k.WMW * tpr * slot_px * (uv * sc + 1) * 16;
    const size_t red = (size_t)k.WMW * k.WNW * (k.WKW - 1) * k.RM * k.RN * 16 * 64 * sizeof(float);
    p.lds_bytes = 2 * img > red ? 2 * img : red;
    if (p.lds_bytes < 8 * 4096) p.lds_bytes = 8 * 4096;         // the epilogue transposes through 4 KB per wave
    if (p.lds_bytes > 160 * 1024 - 2048 - 1024 - (pw == 2 ? 6144 : 0)) return false;   // (static tables of the kernel take < 1 KB; 2x2 tiles < 5 KB)
    p.patches_x = pw == 8 ? bs / 8 : 1;
    p.patches_per_tile = pw == 8 ? (bs / 8) * (bs / ph) : 1;
    p.n_rows = pw == 8 ? (uint32_t)n_exec * p.patches_per_tile : ((uint32_t)n_exec + tpr - 1) / tpr;
    p.wgs = (long long)((p.n_rows + k.WMW - 1) / k.WMW) * (Cout / (32 * k.RN * k.WNW));
    return true;
}

// ---- how the main translation unit reaches the decompositions of conv3x3_v2.inc.  The ~600 kernel instantiations (3 dtypes x
// 2 strides x 2 kernel sizes x 20 decompositions x 2 patch widths, plus the Winograd form) take minutes in one translation unit, so build.py compiles
// this same source several times in parallel: -DBC_PART=0 is everything but them, -DBC_PART=1..6 is one (dtype, kernel size)
// slice each, exported to part 0 as bc_part_conv_v2_<n>(ConvV2Args *), -DBC_PART=7 the Winograd form (bc_part_conv_wino).
// Without -DBC_PART the file is the whole library.
struct ConvV2Args {
    void *out; const void *features; void *ring; const void *wpk;
    const int32_t *grid_idx, *mapping_exec;
    int n_exec, Cin, Cout, GH, GW, bs, stride;
    Prologue pr; EpilogueT ep;
    hipStream_t st;
    int prof_on; hipEvent_t ev_a, ev_b;           // the caller's ProfScope (events attached to the dispatch when profiling)
    int force_cfg, min_lds;                        // tuning knobs of the caller (g_tune)
    unsigned long long *stamps;
    int chosen;                                    // out: the decomposition that was launched
    int xcd_remap;                                 // 1: XCD-aware workgroup order (xcd_remap() in conv3x3_v2.inc)
    int dilation;                                  // 1, or 2 (bc_conv3x3_dil_ring_nhwc: part 9)
    DynCount dyn;                                  // executed-tile count on the device, units = the launch's tiles (bc_dyn_set)
};

// Workgroup order of a conv launch (bit 0 of ConvGeom2::xcd): the tuning knob if it is set, else XCD-aware exactly where the eight L2s
// together would otherwise stream more weight bytes (one copy each) than the launch has activation bytes -- the stages with few, small
// tiles: layer3 / layer4 of the segmentation backbones, the dilated stage of the detector.
static inline uint32_t conv_xcd_order(const ConvV2Args &a)
{
    if (a.xcd_remap >= 0) return (uint32_t)a.xcd_remap & 1u;
    const double weights = 9.0 * a.Cin * a.Cout, acts = (double)a.n_exec * a.bs * a.bs * (a.Cin + a.Cout);
    return 8.0 * weights > acts ? 1u : 0u;
}
struct LaunchProf { bool on; struct { hipEvent_t a, b; } rec; };   // what BC_LAUNCH needs of a ProfScope

#if defined(BC_MONO) || (BC_PART != 0 && BC_PART != 7 && BC_PART != 8 && BC_PART != 10)
template <int DT, int RM, int RN, int WMW, int WNW, int WKW, int SC, int PW, int S, int KS, int D = 1>
static void launch_conv3x3_v2_cfg(LaunchProf &ps, dim3 grid, size_t lds_bytes, const ConvV2Args &a, const ConvGeom2 &g)
{
    static bool attr_set = false;   // > 64 KB of dynamic LDS needs the opt-in once per kernel
    if (!attr_set) {
        // (the kernel's static tables: < 1 KB, 2x2-pixel tiles -- up to 32 tile slots per workgroup -- < 5 KB)
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_conv3x3_v2<DT, RM, RN, WMW, WNW, WKW, SC, PW, S, KS, D>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - (PW == 2 ? 8192 : 2048));
        attr_set = true;
    }
    BC_LAUNCH(ps, (k_conv3x3_v2<DT, RM, RN, WMW, WNW, WKW, SC, PW, S, KS, D>), grid, dim3(512), lds_bytes, a.st, (typename CvType<DT>::T *)a.out,
              (const uint4 *)a.features, (long long)(((const char *)a.ring - (const char *)a.features) / 16), (uint4 *)a.ring, (const uint4 *)a.wpk,
              a.grid_idx, a.mapping_exec, g, a.pr, a.ep, a.stamps);
}

template <int DT, int S, int KS>
static int conv_v2_run(ConvV2Args &a)
{
    constexpr int E = CvType<DT>::E;
    const int n_exec = a.n_exec, Cin = a.Cin, Cout = a.Cout, GH = a.GH, GW = a.GW, bs = a.bs;
    LaunchProf ps{a.prof_on != 0, {a.ev_a, a.ev_b}};
    constexpr int SC_LO = (DT == BC_F32 || DT == BC_F32S) ? 1 : 2, SC_HI = 2 * SC_LO;      // 32-channel units staged per iteration (SC_HI with 8 K groups)
    constexpr bool SPLIT = DT == BC_F32S;
    // forced decomposition: bits 0-7 = index into CONV2_CFGS, bit 8 = no LDS floor (two workgroups may share a CU: the prologue /
    // epilogue bursts of one overlap the matrix phase of the other; pays when the launch is several rounds of workgroups)
    const int force = a.force_cfg < 0 ? -1 : (a.force_cfg & 0xff);
    const int min_lds = (a.force_cfg >= 0 && (a.force_cfg & 0x100)) ? 0 : a.min_lds;
    const int pw = bs == 4 ? 4 : (bs == 2 ? 2 : 8);
    const int cus = device_cu_count();
    static const bool dbg_model = getenv("BC_CONV2_DEBUG") != nullptr;
    int best = -1;
    double best_t = 0;
    Conv2Plan plan, best_plan{};
    for (int c = 0; c < (int)(sizeof(CONV2_CFGS) / sizeof(CONV2_CFGS[0])); ++c) {
        const Conv2Cfg &k = CONV2_CFGS[c];
        if (force >= 0 && c != force) continue;
        if (!conv2_plan(k, E, S, n_exec, Cin, Cout, bs, plan, KS, 1, SPLIT)) continue;
        if (KS == 1 && pw != 8 && S == 1) continue;            // (pointwise stride-1 launches are re-tiled to 8x8 by the caller)
        const long long rounds = (plan.wgs + cus - 1) / cus;
        const double mf = (double)k.RM * k.RN * (double)(KS * KS) * (Cin / 8) * 4.0 / k.WKW;   // fp32 MFMAs per wave (16-bit: the same ranking)
        // per-MFMA slowdown from operand delivery (tools/probes/mfma_probe2: 1x1 tiles ~0.72, 2x1 ~0.79, 2x2 ~0.82 of peak)
        const double eff = k.RM * k.RN >= 4 ? 0.82 : (k.RM * k.RN == 2 ? (k.RM == 2 ? 0.79 : 0.74) : 0.70);
        const double t = rounds * (mf / eff + 70.0 + 12.0 * (k.WKW - 1) * k.RM * k.RN);
        if (dbg_model) fprintf(stderr, "[conv2 model] dt %d S %d n %d %d->%d bs %d: c%d wgs %lld rounds %lld mf %.0f t %.0f (cus %d)\n", DT, S, n_exec, Cin, Cout, bs, c, plan.wgs, rounds, mf, t, cus);
        if (best < 0 || t < best_t) { best = c; best_t = t; best_plan = plan; }
    }
    if (best < 0) return BC_ERR_SHAPE;
    a.chosen = best;
    const Conv2Cfg &k = CONV2_CFGS[best];
    ConvGeom2 g;
    g.Cin = Cin; g.Cout = Cout; g.bs = bs; g.GH = GH; g.GW = GW; g.n_exec = n_exec;
    g.patches_x = best_plan.patches_x;
    g.patches_per_tile = best_plan.patches_per_tile;
    g.n_rows = best_plan.n_rows;
    g.cin_chunks = Cin / CV_CH;
    g.xcd = conv_xcd_order(a);
    g.dyn = a.dyn;
    size_t lds_bytes = best_plan.lds_bytes;
    if (lds_bytes < (size_t)min_lds) lds_bytes = min_lds;      // one workgroup per CU: two waves on every SIMD, no more
    dim3 grid((g.n_rows + k.WMW - 1) / k.WMW, (unsigned)Cout / (32 * k.RN * k.WNW));
    if (g.dyn.ptr && grid.x <= 65535u) { grid = dim3(grid.y, grid.x); g.xcd |= 2u; }     // device-side count: live rows first in dispatch order (xcd_remap, dyn_order)
    // (decompositions whose double-buffered patch images cannot fit the LDS are never chosen by conv2_plan and are not compiled)
#define BC_CV2(RM_, RN_, WMW_, WNW_, WKW_)                                                                                   \
    do {                                                                                                                 \
        constexpr int SC_ = SPLIT ? (WKW_ == 8 ? 4 : (WKW_ == 4 ? 2 : 1)) : (WKW_ == 8 ? SC_HI : SC_LO);                  \
        constexpr int RM4_ = WMW_ == 1 ? RM_ : 1;                                                                        \
        constexpr int SS_ = KS == 1 ? 1 : S;                                                                             \
        constexpr size_t img8_ = (size_t)WMW_ * (SS_ * 7 + KS) * (SS_ * (4 * RM_ - 1) + KS) * (CvType<DT>::UV * SC_ + 1) * 32; \
        constexpr size_t img4_ = (size_t)WMW_ * 2 * RM4_ * (SS_ * 3 + KS) * (SS_ * 3 + KS) * (CvType<DT>::UV * SC_ + 1) * 32; \
        if (pw == 8) {                                                                                                   \
            if constexpr (img8_ <= 160 * 1024 - 3072)                                                                    \
                launch_conv3x3_v2_cfg<DT, RM_, RN_, WMW_, WNW_, WKW_, SC_, 8, S, KS>(ps, grid, lds_bytes, a, g); \
        } else if (pw == 4) {                                                                                            \
            if constexpr (img4_ <= 160 * 1024 - 3072 && !(KS == 1 && S == 1))                                            \
                launch_conv3x3_v2_cfg<DT, RM4_, RN_, WMW_, WNW_, WKW_, SC_, 4, S, KS>(ps, grid, lds_bytes, a, g); \
        } else {                                                                                                         \
            if constexpr (RM_ == 1 && !(KS == 1 && S == 1) && (DT == BC_F32S || DT == BC_F16))                           \
                launch_conv3x3_v2_cfg<DT, 1, RN_, WMW_, WNW_, WKW_, SC_, 2, S, KS>(ps, grid, lds_bytes, a, g);  \
            else return BC_ERR_SHAPE;                                                                                    \
        }                                                                                                                \
    } while (0)
    switch (best) {
    case 0: BC_CV2(2, 2, 4, 2, 1); break;
    case 1: BC_CV2(2, 1, 4, 2, 1); break;
    case 2: BC_CV2(1, 2, 4, 2, 1); break;
    case 3: BC_CV2(1, 1, 4, 2, 1); break;
    case 4: BC_CV2(1, 1, 2, 4, 1); break;
    case 5: BC_CV2(1, 1, 2, 2, 2); break;
    case 6: BC_CV2(1, 1, 1, 4, 2); break;
    case 7: BC_CV2(1, 1, 1, 2, 4); break;
    case 8: BC_CV2(2, 2, 2, 2, 2); break;
    case 9: BC_CV2(2, 2, 1, 2, 4); break;
    case 10: BC_CV2(2, 1, 2, 2, 2); break;
    case 11: BC_CV2(2, 1, 1, 4, 2); break;
    case 12: BC_CV2(2, 1, 1, 2, 4); break;
    case 13: BC_CV2(2, 1, 1, 1, 8); break;
    case 14: BC_CV2(1, 2, 1, 1, 8); break;
    case 15: BC_CV2(1, 1, 1, 1, 8); break;
    case 16: BC_CV2(2, 2, 4, 1, 2); break;
    case 17: BC_CV2(2, 2, 2, 1, 4); break;
    case 18: BC_CV2(1, 2, 4, 1, 2); break;
    default: BC_CV2(1, 2, 2, 1, 4); break;
    }
#undef BC_CV2
    return launch_status();
}

#endif

// ---- dilation 2 (3x3, stride 1, tiles of a multiple of 8 pixels): the decompositions of CONV2_CFGS with one 32-channel column per wave
static const int CONV2_DIL_CFGS[] = {3, 5, 6, 7, 11, 15, 8, 9};      // (8, 9: 2x2 wave tiles, split form only)
constexpr int CONV2_DIL_N = (int)(sizeof(CONV2_DIL_CFGS) / sizeof(CONV2_DIL_CFGS[0]));

#if defined(BC_MONO) || BC_PART == 9
template <int DT>
static int conv_v2_dil_run(ConvV2Args &a)
{
    constexpr int E = CvType<DT>::E;
    constexpr bool SPLIT = DT == BC_F32S;
    LaunchProf ps{a.prof_on != 0, {a.ev_a, a.ev_b}};
    constexpr int SC_LO = (DT == BC_F32 || SPLIT) ? 1 : 2, SC_HI = SPLIT ? 4 : 2 * SC_LO, SC_K4 = SPLIT ? 2 : SC_LO;    // (split: WKW 4 stages 2 units, WKW 8 stages 4)
    const int force = a.force_cfg < 0 ? -1 : (a.force_cfg & 0xff);
    const int min_lds = (a.force_cfg >= 0 && (a.force_cfg & 0x100)) ? 0 : a.min_lds;
    if (a.bs % 8 != 0) return BC_ERR_SHAPE;
    const int cus = device_cu_count();
    int best = -1;
    double best_t = 0;
    Conv2Plan plan, best_plan{};
    for (int i = 0; i < CONV2_DIL_N; ++i) {
        const int c = CONV2_DIL_CFGS[i];
        const Conv2Cfg &k = CONV2_CFGS[c];
        if (force >= 0 && c != force) continue;
        if (!SPLIT && k.RM * k.RN >= 4) continue;
        if (!conv2_plan(k, E, 1, a.n_exec, a.Cin, a.Cout, a.bs, plan, 3, 2, SPLIT)) continue;
        const long long rounds = (plan.wgs + cus - 1) / cus;
        const double mf = (double)k.RM * k.RN * 9.0 * (a.Cin / 8) * 4.0 / k.WKW;
        const double t = rounds * (mf / (k.RM == 2 ? 0.79 : 0.70) + 70.0 + 12.0 * (k.WKW - 1) * k.RM * k.RN);
        if (best < 0 || t < best_t) { best = c; best_t = t; best_plan = plan; }
    }
    if (best < 0) return BC_ERR_SHAPE;
    a.chosen = best;
    const Conv2Cfg &k = CONV2_CFGS[best];
    ConvGeom2 g;
    g.Cin = a.Cin; g.Cout = a.Cout; g.bs = a.bs; g.GH = a.GH; g.GW = a.GW; g.n_exec = a.n_exec;
    g.patches_x = best_plan.patches_x;
    g.patches_per_tile = best_plan.patches_per_tile;
    g.n_rows = best_plan.n_rows;
    g.cin_chunks = a.Cin / CV_CH;
    g.xcd = conv_xcd_order(a);
    g.dyn = a.dyn;
    size_t lds_bytes = best_plan.lds_bytes;
    if (lds_bytes < (size_t)min_lds) lds_bytes = min_lds;
    dim3 grid((g.n_rows + k.WMW - 1) / k.WMW, (unsigned)a.Cout / (32 * k.RN * k.WNW));
    if (g.dyn.ptr && grid.x <= 65535u) { grid = dim3(grid.y, grid.x); g.xcd |= 2u; }     // device-side count: live rows first in dispatch order (xcd_remap, dyn_order)
    switch (best) {
    case 3: launch_conv3x3_v2_cfg<DT, 1, 1, 4, 2, 1, SC_LO, 8, 1, 3, 2>(ps, grid, lds_bytes, a, g); break;
    case 5: launch_conv3x3_v2_cfg<DT, 1, 1, 2, 2, 2, SC_LO, 8, 1, 3, 2>(ps, grid, lds_bytes, a, g); break;
    case 6: launch_conv3x3_v2_cfg<DT, 1, 1, 1, 4, 2, SC_LO, 8, 1, 3, 2>(ps, grid, lds_bytes, a, g); break;
    case 7: launch_conv3x3_v2_cfg<DT, 1, 1, 1, 2, 4, SC_K4, 8, 1, 3, 2>(ps, grid, lds_bytes, a, g); break;
    case 11: launch_conv3x3_v2_cfg<DT, 2, 1, 1, 4, 2, SC_LO, 8, 1, 3, 2>(ps, grid, lds_bytes, a, g); break;
    case 8: if constexpr (SPLIT) launch_conv3x3_v2_cfg<DT, 2, 2, 2, 2, 2, SC_LO, 8, 1, 3, 2>(ps, grid, lds_bytes, a, g); break;
    case 9: if constexpr (SPLIT) launch_conv3x3_v2_cfg<DT, 2, 2, 1, 2, 4, SC_K4, 8, 1, 3, 2>(ps, grid, lds_bytes, a, g); break;
    default: launch_conv3x3_v2_cfg<DT, 1, 1, 1, 1, 8, SC_HI, 8, 1, 3, 2>(ps, grid, lds_bytes, a, g); break;
    }
    return launch_status();
}
#endif

#if BC_PART == 9
extern "C" int bc_part_conv_v2_dil(void *p)
{
    ConvV2Args &a = *static_cast<ConvV2Args *>(p);
    // dtype travels in `stride` for this part (the stride of a dilated launch is always 1)
    return a.stride == BC_F32S ? conv_v2_dil_run<BC_F32S>(a)
                               : (a.stride == BC_F32 ? conv_v2_dil_run<BC_F32>(a) : (a.stride == BC_F16 ? conv_v2_dil_run<BC_F16>(a) : conv_v2_dil_run<BC_BF16>(a)));
}
#endif

// ---- host side of conv3x3_wino.inc (Winograd F(2x2,3x3), fp32 / stride 1): decompositions (MB, WMW, WNW, WKW); code 0x200 | index
struct WinoCfg { int MB, WMW, WNW, WKW, SH; };
static const WinoCfg WINO_CFGS[] = {{2, 2, 4, 1, 0}, {2, 2, 2, 2, 0}, {2, 1, 4, 2, 0}, {2, 1, 2, 4, 0}, {1, 2, 4, 1, 0}, {1, 4, 2, 1, 0}, {1, 2, 2, 2, 0}, {1, 1, 4, 2, 0}, {1, 1, 2, 4, 0},
                                    {1, 1, 8, 1, 0}, {2, 1, 8, 1, 0},     // 9, 10: 128 output channels per workgroup (the patches of a 128-channel layer are staged once)
                                    {1, 1, 8, 1, 1}, {1, 1, 4, 2, 1}, {1, 1, 2, 4, 1}};     // 11-13: input transform shared by the workgroup (SH in conv3x3_wino.inc)
constexpr int WINO_N = (int)(sizeof(WINO_CFGS) / sizeof(WINO_CFGS[0]));

struct WinoPlan { long long wgs; size_t lds_bytes; uint32_t n_rows; };
static bool wino_plan(const WinoCfg &k, int n_exec, int Cin, int Cout, int bs, WinoPlan &p)
{
    if (!(bs % 8 == 0 || bs == 4) || bs > 248 || Cin % 32 != 0 || Cout % (16 * k.WNW) != 0) return false;
    if (bs == 4 && k.MB * k.WMW > 2) return false;
    const size_t img = (size_t)k.WMW * k.MB * (bs == 4 ? 4 * 36 : 100) * 9 * 16;
    const size_t red = (size_t)k.WMW * k.WNW * (k.WKW - 1) * k.MB * 16 * 64 * sizeof(float);
    const size_t coeffs = (size_t)2 * Cin * sizeof(float);                            // the activation coefficients, behind the images
    p.lds_bytes = 2 * img + coeffs > red ? 2 * img + coeffs : red;
    if (k.SH) p.lds_bytes = 2 * img + 2 * (size_t)16 * 16 * 36 * sizeof(float) + coeffs;      // + two transformed images
    if (p.lds_bytes > 160 * 1024 - 2048 - 1024) return false;
    const long long slots = bs == 4 ? ((long long)n_exec + 3) / 4 : (long long)n_exec * (bs / 8) * (bs / 8);     // M-blocks
    p.n_rows = (uint32_t)((slots + k.MB - 1) / k.MB);
    p.wgs = (long long)((p.n_rows + k.WMW - 1) / k.WMW) * (Cout / (16 * k.WNW));
    return true;
}

#if defined(BC_MONO) || BC_PART == 7
template <int MB, int WMW, int WNW, int WKW, int TS, bool SH = false>
static void launch_wino_ts(LaunchProf &ps, dim3 grid, size_t lds_bytes, const ConvV2Args &a, const ConvGeom2 &g)
{
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_conv3x3_wino<MB, WMW, WNW, WKW, TS, SH>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 2048);
        attr_set = true;
    }
    // the Winograd weight stream follows the direct one in the packed buffer (pack_conv3x3_weights: 9 + 16 values per (cin, cout))
    const float4 *wino_w = reinterpret_cast<const float4 *>(reinterpret_cast<const float *>(a.wpk) + (size_t)9 * a.Cin * a.Cout);
    BC_LAUNCH(ps, (k_conv3x3_wino<MB, WMW, WNW, WKW, TS, SH>), grid, dim3(512), lds_bytes, a.st, (float *)a.out, (const uint4 *)a.features,
              (long long)(((const char *)a.ring - (const char *)a.features) / 16), (uint4 *)a.ring, wino_w, a.grid_idx, a.mapping_exec, g, a.pr, a.ep, a.stamps);
}

template <int MB, int WMW, int WNW, int WKW, bool SH = false>
static void launch_wino_cfg(LaunchProf &ps, dim3 grid, size_t lds_bytes, const ConvV2Args &a, const ConvGeom2 &g)
{
    if (a.bs == 4) {
        if constexpr (MB * WMW <= 2) launch_wino_ts<MB, WMW, WNW, WKW, 4, SH>(ps, grid, lds_bytes, a, g);     // (4 tile slots per M-block: LDS)
    } else {
        launch_wino_ts<MB, WMW, WNW, WKW, 8, SH>(ps, grid, lds_bytes, a, g);
    }
}

static int conv_wino_run(ConvV2Args &a)
{
    const int c = a.force_cfg & 0xff;
    if (c >= WINO_N) return BC_ERR_SHAPE;
    const WinoCfg &k = WINO_CFGS[c];
    WinoPlan plan;
    if (!wino_plan(k, a.n_exec, a.Cin, a.Cout, a.bs, plan)) return BC_ERR_SHAPE;
    LaunchProf ps{a.prof_on != 0, {a.ev_a, a.ev_b}};
    ConvGeom2 g;
    g.Cin = a.Cin; g.Cout = a.Cout; g.bs = a.bs; g.GH = a.GH; g.GW = a.GW; g.n_exec = a.n_exec;
    g.patches_x = a.bs == 4 ? 1 : a.bs / 8;
    g.patches_per_tile = a.bs == 4 ? 1 : (a.bs / 8) * (a.bs / 8);
    g.n_rows = plan.n_rows;
    g.cin_chunks = a.Cin / 32;
    g.xcd = conv_xcd_order(a);
    g.dyn = a.dyn;
    size_t lds_bytes = plan.lds_bytes;
    if (!(a.force_cfg & 0x100) && lds_bytes < (size_t)a.min_lds) lds_bytes = a.min_lds;
    dim3 grid((plan.n_rows + k.WMW - 1) / k.WMW, (unsigned)a.Cout / (16 * k.WNW));
    if (g.dyn.ptr && grid.x <= 65535u) { grid = dim3(grid.y, grid.x); g.xcd |= 2u; }     // device-side count: live rows first in dispatch order (xcd_remap, dyn_order)
    switch (c) {
    case 0: launch_wino_cfg<2, 2, 4, 1>(ps, grid, lds_bytes, a, g); break;
    case 1: launch_wino_cfg<2, 2, 2, 2>(ps, grid, lds_bytes, a, g); break;
    case 2: launch_wino_cfg<2, 1, 4, 2>(ps, grid, lds_bytes, a, g); break;
    case 3: launch_wino_cfg<2, 1, 2, 4>(ps, grid, lds_bytes, a, g); break;
    case 4: launch_wino_cfg<1, 2, 4, 1>(ps, grid, lds_bytes, a, g); break;
    case 5: launch_wino_cfg<1, 4, 2, 1>(ps, grid, lds_bytes, a, g); break;
    case 6: launch_wino_cfg<1, 2, 2, 2>(ps, grid, lds_bytes, a, g); break;
    case 7: launch_wino_cfg<1, 1, 4, 2>(ps, grid, lds_bytes, a, g); break;
    case 8: launch_wino_cfg<1, 1, 2, 4>(ps, grid, lds_bytes, a, g); break;
    case 9: launch_wino_cfg<1, 1, 8, 1>(ps, grid, lds_bytes, a, g); break;
    case 10: launch_wino_cfg<2, 1, 8, 1>(ps, grid, lds_bytes, a, g); break;
    case 11: launch_wino_cfg<1, 1, 8, 1, true>(ps, grid, lds_bytes, a, g); break;
    case 12: launch_wino_cfg<1, 1, 4, 2, true>(ps, grid, lds_bytes, a, g); break;
    default: launch_wino_cfg<1, 1, 2, 4, true>(ps, grid, lds_bytes, a, g); break;
    }
    a.chosen = a.force_cfg & 0x3ff;
    return launch_status();
}
#endif

#if BC_PART == 7
extern "C" int bc_part_conv_wino(void *p) { return conv_wino_run(*static_cast<ConvV2Args *>(p)); }
#endif

// ---- host side of conv3x3_wino32.inc (wide wave tile, one wave per SIMD): decompositions (WMW, WNW, WKW) of 4 waves; code 0x400 | index
struct Wino32Cfg { int WMW, WNW, WKW; };
static const Wino32Cfg WINO32_CFGS[] = {{2, 2, 1}, {1, 4, 1}, {1, 2, 2}, {2, 1, 2}, {1, 1, 4}};
constexpr int WINO32_N = (int)(sizeof(WINO32_CFGS) / sizeof(WINO32_CFGS[0]));

struct Wino32Plan { long long wgs; size_t lds_bytes; uint32_t n_slots; };
static bool wino32_plan(const Wino32Cfg &k, int n_exec, int Cin, int Cout, int bs, Wino32Plan &p)
{
    if (bs % 8 != 0 || bs > 248 || Cin % 32 != 0 || Cout % (32 * k.WNW) != 0) return false;
    const size_t img = (size_t)k.WMW * 2 * 100 * 9 * 16;                                   // one staged image: 2 * WMW patches
    const size_t per_wave = 64 * 64 * sizeof(float);                                        // transposition / K-group partial area of a wave
    const size_t tail = k.WKW > 1 ? (size_t)k.WMW * k.WNW * (k.WKW - 1) * per_wave : 4 * per_wave;
    p.lds_bytes = 2 * img > tail ? 2 * img : tail;
    if (p.lds_bytes > 160 * 1024 - 2048 - 1024) return false;
    p.n_slots = (uint32_t)n_exec * (uint32_t)((bs / 8) * (bs / 8));
    const long long rows = (p.n_slots + 2 * k.WMW - 1) / (2 * k.WMW);
    p.wgs = rows * (Cout / (32 * k.WNW));
    return true;
}

#if defined(BC_MONO) || BC_PART == 8
template <int WMW, int WNW, int WKW>
static void launch_wino32_cfg(LaunchProf &ps, dim3 grid, size_t lds_bytes, const ConvV2Args &a, const ConvGeom2 &g)
{
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_conv3x3_wino32<WMW, WNW, WKW>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 2048);
        attr_set = true;
    }
    // the wide-tile Winograd stream follows the direct and the 16-channel Winograd streams (pack_conv3x3_weights: 9 + 16 + 16 values per pair)
    const float4 *w32 = reinterpret_cast<const float4 *>(reinterpret_cast<const float *>(a.wpk) + (size_t)25 * a.Cin * a.Cout);
    BC_LAUNCH(ps, (k_conv3x3_wino32<WMW, WNW, WKW>), grid, dim3(256), lds_bytes, a.st, (float *)a.out, (const uint4 *)a.features,
              (long long)(((const char *)a.ring - (const char *)a.features) / 16), (uint4 *)a.ring, w32, a.grid_idx, a.mapping_exec, g, a.pr, a.ep);
}

static int conv_wino32_run(ConvV2Args &a)
{
    const int c = a.force_cfg & 0xff;
    if (c >= WINO32_N) return BC_ERR_SHAPE;
    const Wino32Cfg &k = WINO32_CFGS[c];
    Wino32Plan plan;
    if (!wino32_plan(k, a.n_exec, a.Cin, a.Cout, a.bs, plan)) return BC_ERR_SHAPE;
    LaunchProf ps{a.prof_on != 0, {a.ev_a, a.ev_b}};
    ConvGeom2 g;
    g.Cin = a.Cin; g.Cout = a.Cout; g.bs = a.bs; g.GH = a.GH; g.GW = a.GW; g.n_exec = a.n_exec;
    g.patches_x = a.bs / 8;
    g.patches_per_tile = (a.bs / 8) * (a.bs / 8);
    g.n_rows = plan.n_slots;
    g.cin_chunks = a.Cin / 32;
    g.xcd = conv_xcd_order(a);
    g.dyn = a.dyn;
    dim3 grid((plan.n_slots + 2 * k.WMW - 1) / (2 * k.WMW), (unsigned)a.Cout / (32 * k.WNW));
    if (g.dyn.ptr && grid.x <= 65535u) { grid = dim3(grid.y, grid.x); g.xcd |= 2u; }     // device-side count: live rows first in dispatch order (xcd_remap, dyn_order)
    switch (c) {
    case 0: launch_wino32_cfg<2, 2, 1>(ps, grid, plan.lds_bytes, a, g); break;
    case 1: launch_wino32_cfg<1, 4, 1>(ps, grid, plan.lds_bytes, a, g); break;
    case 2: launch_wino32_cfg<1, 2, 2>(ps, grid, plan.lds_bytes, a, g); break;
    case 3: launch_wino32_cfg<2, 1, 2>(ps, grid, plan.lds_bytes, a, g); break;
    default: launch_wino32_cfg<1, 1, 4>(ps, grid, plan.lds_bytes, a, g); break;
    }
    a.chosen = a.force_cfg & 0x7ff;
    return launch_status();
}
#endif

#if BC_PART == 8
extern "C" int bc_part_conv_wino32(void *p) { return conv_wino32_run(*static_cast<ConvV2Args *>(p)); }
#endif

// ---- host side of conv3x3_wino4.inc (Winograd F(4x4,3x3), fp32 / stride 1, tiles of a multiple of 16 pixels or 8x8 tiles):
// decompositions (WNW, WFW, NB) of 8 waves; code 0x1000 | index
struct Wino4Cfg { int WNW, WFW, NB; };
static const Wino4Cfg WINO4_CFGS[] = {{4, 2, 1}, {2, 4, 1}, {2, 4, 2}};      // ((4,2,2): 144 accumulators per lane, 327 spilled registers -- removed)
constexpr int WINO4_N = (int)(sizeof(WINO4_CFGS) / sizeof(WINO4_CFGS[0]));

struct Wino4Plan { long long wgs; size_t lds_bytes; uint32_t n_rows; };
static bool wino4_plan(const Wino4Cfg &k, int n_exec, int Cin, int Cout, int bs, Wino4Plan &p)
{
    if (!(bs % 16 == 0 || bs == 8) || bs > 240 || Cin % 16 != 0 || Cout % (16 * k.NB * k.WNW) != 0) return false;
    if (bs == 8 && !(k.WFW == 4 && k.NB == 1)) return false;      // (four 8x8 slots: one staging vector more per thread; the other forms would spill)
    const size_t raw = (size_t)(bs == 8 ? 4 * 100 : 324) * 64, vimg = (size_t)36 * 1024;
    p.lds_bytes = 2 * raw + 2 * vimg + (size_t)2 * Cin * sizeof(float) + 2 * 512 * 8;      // + the activation coefficients + the ring-refresh plan
    if (p.lds_bytes < (size_t)8 * 16384) p.lds_bytes = 8 * 16384;      // the output stage: one [256 pixels][16 channels] area per wave
    if (p.lds_bytes > 160 * 1024 - 2048 - 1024) return false;          // (very wide layers: the coefficient table no longer fits)
    const long long slots = bs == 8 ? ((long long)n_exec + 3) / 4 : (long long)n_exec * (bs / 16) * (bs / 16);     // M-blocks
    p.n_rows = (uint32_t)slots;
    p.wgs = slots * (Cout / (16 * k.NB * k.WNW));
    return true;
}

#if defined(BC_MONO) || BC_PART == 10
template <int TS, int WNW, int WFW, int NB, bool SP>
static void launch_wino4_ts(LaunchProf &ps, dim3 grid, size_t lds_bytes, const ConvV2Args &a, const ConvGeom2 &g)
{
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_conv3x3_wino4<TS, WNW, WFW, NB, SP>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 2048);
        attr_set = true;
    }
    // the F(4x4) stream follows the direct and the two F(2x2) streams (pack_conv3x3_weights: 9 + 16 + 16 + 36 values per pair); its split form
    // (SP: [hi | lo] halves of 256 U) is the sixth stream, behind the split stream of the direct form (9)
    const float4 *w4 = reinterpret_cast<const float4 *>(reinterpret_cast<const float *>(a.wpk) + (size_t)(SP ? 86 : 41) * a.Cin * a.Cout);
    BC_LAUNCH(ps, (k_conv3x3_wino4<TS, WNW, WFW, NB, SP>), grid, dim3(512), lds_bytes, a.st, (float *)a.out, (const uint4 *)a.features,
              (long long)(((const char *)a.ring - (const char *)a.features) / 16), (uint4 *)a.ring, w4, a.grid_idx, a.mapping_exec, g, a.pr, a.ep, a.stamps);
}

template <int WNW, int WFW, int NB>
static void launch_wino4_cfg(LaunchProf &ps, dim3 grid, size_t lds_bytes, const ConvV2Args &a, const ConvGeom2 &g)
{
    const bool sp = (a.force_cfg & 0x4000) != 0;
    if (a.bs == 8) {
        if constexpr (WFW == 4 && NB == 1) {
            if (sp) launch_wino4_ts<8, WNW, WFW, NB, true>(ps, grid, lds_bytes, a, g);
            else launch_wino4_ts<8, WNW, WFW, NB, false>(ps, grid, lds_bytes, a, g);
        }
    } else if (sp) launch_wino4_ts<16, WNW, WFW, NB, true>(ps, grid, lds_bytes, a, g);
    else launch_wino4_ts<16, WNW, WFW, NB, false>(ps, grid, lds_bytes, a, g);
}

static int conv_wino4_run(ConvV2Args &a)
{
    const int c = a.force_cfg & 0xff;
    if (c >= WINO4_N) return BC_ERR_SHAPE;
    const Wino4Cfg &k = WINO4_CFGS[c];
    Wino4Plan plan;
    if (!wino4_plan(k, a.n_exec, a.Cin, a.Cout, a.bs, plan)) return BC_ERR_SHAPE;
    LaunchProf ps{a.prof_on != 0, {a.ev_a, a.ev_b}};
    ConvGeom2 g;
    g.Cin = a.Cin; g.Cout = a.Cout; g.bs = a.bs; g.GH = a.GH; g.GW = a.GW; g.n_exec = a.n_exec;
    g.patches_x = a.bs == 8 ? 1 : a.bs / 16;
    g.patches_per_tile = a.bs == 8 ? 1 : (a.bs / 16) * (a.bs / 16);
    g.n_rows = plan.n_rows;
    g.cin_chunks = a.Cin / 16;
    g.xcd = conv_xcd_order(a);
    g.dyn = a.dyn;
    dim3 grid(plan.n_rows, (unsigned)a.Cout / (16 * k.NB * k.WNW));
    if (g.dyn.ptr && grid.x <= 65535u) { grid = dim3(grid.y, grid.x); g.xcd |= 2u; }     // device-side count: live rows first in dispatch order (xcd_remap, dyn_order)
    switch (c) {
    case 0: launch_wino4_cfg<4, 2, 1>(ps, grid, plan.lds_bytes, a, g); break;
    case 1: launch_wino4_cfg<2, 4, 1>(ps, grid, plan.lds_bytes, a, g); break;
    case 2: launch_wino4_cfg<2, 4, 2>(ps, grid, plan.lds_bytes, a, g); break;
    default: return BC_ERR_SHAPE;
    }
    a.chosen = a.force_cfg & 0x50ff;
    return launch_status();
}
#endif

#if BC_PART == 10
extern "C" int bc_part_conv_wino4(void *p) { return conv_wino4_run(*static_cast<ConvV2Args *>(p)); }
#endif

#if BC_PART != 0 && BC_PART != 7 && BC_PART != 8 && BC_PART != 9 && BC_PART != 10
// this slice: dtype (BC_PART - 1) / 2, kernel size 3 (odd parts) or 1 (even parts), both strides
#define BC_PART_NAME2(n_) bc_part_conv_v2_##n_
#define BC_PART_NAME(n_) BC_PART_NAME2(n_)
extern "C" int BC_PART_NAME(BC_PART)(void *p)
{
    ConvV2Args &a = *static_cast<ConvV2Args *>(p);
    // parts 1..6: dtype (BC_PART - 1) / 2; parts 12 / 13: the split form of the fp32 tensors (BC_F32S), kernel size 3 / 1
    constexpr int DT = BC_PART >= 12 ? BC_F32S : (BC_PART - 1) / 2, KS = (BC_PART >= 12 ? BC_PART - 12 : BC_PART - 1) % 2 == 0 ? 3 : 1;
    return a.stride == 1 ? conv_v2_run<DT, 1, KS>(a) : conv_v2_run<DT, 2, KS>(a);
}
#endif

#if BC_PART == 0
// ---- tuning knobs (bc_tune_set; defaults may also come from the environment, read once)
struct TuneState {
    int conv_impl = [] { const char *e = getenv("BC_CONV_IMPL"); return e ? atoi(e) : 2; }();              // 1 = first-generation conv kernel
    int conv2_cfg = [] { const char *e = getenv("BC_CONV2_CFG"); return e ? atoi(e) : -1; }();             // >= 0: force a decomposition
    unsigned long long *conv_stamps = nullptr;   // device buffer for in-kernel s_memtime stamps (bc_tune_set_ptr), measurement only
    int conv_last_cfg = -2;   // decomposition of the most recent bc_conv3x3_ring_nhwc launch (-1: first-generation kernel)
    int stem_min_lds = [] { const char *e = getenv("BC_STEM_MINLDS"); return e ? atoi(e) : 84 * 1024; }();
    int conv2_min_lds = [] { const char *e = getenv("BC_CONV2_MINLDS"); return e ? atoi(e) : 84 * 1024; }();   // bytes; > 80 KB = one workgroup per CU
    // XCD-aware workgroup order of the conv kernels (xcd_remap in conv3x3_v2.inc): 0 launch order, 1 XCD-aware, -1 (default) XCD-aware
    // where the weights outweigh the activations of the launch (conv_xcd_order).  In TIME the two orders are equal on every layer shape
    // of the configs (profiles/r03/11, r04/22: +-1 %, the re-fetched weights are served by the Infinity Cache); in TRAFFIC the XCD-aware
    // order cuts the weight-dominated launches to a third (PMC, profiles/r04/21: layer4 Winograd 7.76x -> 2.52x of the algorithmic bytes,
    // direct 4.57x -> 2.12x, dilated detector stage 5.8x -> 3.35x) -- fabric bandwidth a concurrent copy or replica does not have to share
    int xcd_remap = [] { const char *e = getenv("BC_XCD_REMAP"); return e ? atoi(e) : -1; }();
    // fp32 stem on the 16-bit matrix pipe: needs the hi / lo weight streams behind the fp32 one (blockcopy.backend packs them and switches this on)
    int stem_split = [] { const char *e = getenv("BC_STEM_SPLIT"); return e ? atoi(e) : 0; }();
    int head_split = [] { const char *e = getenv("BC_HEAD_SPLIT"); return e ? atoi(e) : 0; }();            // 1: k_head1x1_s for fp32 / Cout <= 20 (measured neutral in the frame, profiles/r05/09: stays off)
} g_tune;


#if !defined(BC_MONO)
extern "C" {
int bc_part_conv_v2_1(void *); int bc_part_conv_v2_2(void *); int bc_part_conv_v2_3(void *);
int bc_part_conv_v2_4(void *); int bc_part_conv_v2_5(void *); int bc_part_conv_v2_6(void *);
int bc_part_conv_wino(void *);
int bc_part_conv_wino32(void *);
int bc_part_conv_wino4(void *);
int bc_part_conv_v2_dil(void *);
int bc_part_conv_v2_12(void *); int bc_part_conv_v2_13(void *);
}
#endif

static thread_local DynCount g_conv_dyn;      // set by the exported conv launchers for the ONE launch_conv3x3_v2 call they make (bc_dyn_set)

template <int DT, int S, int KS = 3>
static int launch_conv3x3_v2(ProfScope &ps, void *out, const void *features, void *ring, const void *wpk, const int32_t *grid_idx,
                             const int32_t *mapping_exec, int n_exec, int Cin, int Cout, int GH, int GW, int bs,
                             const Prologue &pr, const EpilogueT &ep, hipStream_t st)
{
    ConvV2Args a{out, features, ring, wpk, grid_idx, mapping_exec, n_exec, Cin, Cout, GH, GW, bs, S, pr, ep, st,
                 ps.on ? 1 : 0, ps.rec.a, ps.rec.b, g_tune.conv2_cfg, g_tune.conv2_min_lds, g_tune.conv_stamps, -2, g_tune.xcd_remap, 1};
    a.dyn = g_conv_dyn;
    g_conv_dyn = DynCount{};
    const double direct_flops = 2.0 * n_exec * (double)(bs / S) * (bs / S) * (KS * KS) * (double)Cin * Cout;
    if (DT == BC_F32 && S == 1 && KS == 3 && a.force_cfg >= 0 && (a.force_cfg & 0x1000)) {      // Winograd F(4x4,3x3) (conv3x3_wino4.inc)
        // 36 multiplications per 4x4 outputs instead of 144; | 0x4000: on the 16-bit pipe (three MFMAs of 16 cycles where the fp32 pipe runs four of 32)
        ps.add_aux(direct_flops * 36.0 / 144.0 * ((a.force_cfg & 0x4000) ? 3.0 / 8.0 : 1.0));
#if defined(BC_MONO)
        const int rcw = conv_wino4_run(a);
#else
        const int rcw = bc_part_conv_wino4(&a);
#endif
        if (a.chosen >= 0) g_tune.conv_last_cfg = a.chosen;
        return rcw;
    }
    if (DT == BC_F32 && S == 1 && KS == 3 && a.force_cfg >= 0 && (a.force_cfg & 0x600)) {       // Winograd forms (conv3x3_wino.inc, conv3x3_wino32.inc)
        ps.add_aux(direct_flops * 16.0 / 36.0);     // F(2x2,3x3): 16 multiplications per 2x2 outputs instead of 36
#if defined(BC_MONO)
        const int rcw = (a.force_cfg & 0x400) ? conv_wino32_run(a) : conv_wino_run(a);
#else
        const int rcw = (a.force_cfg & 0x400) ? bc_part_conv_wino32(&a) : bc_part_conv_wino(&a);
#endif
        if (a.chosen >= 0) g_tune.conv_last_cfg = a.chosen;
        return rcw;
    }
    if (DT == BC_F32 && ((a.force_cfg >= 0 && (a.force_cfg & 0x2000)) || a.bs == 2)) {
        // the direct form on the 16-bit matrix pipe (operands split hi + lo, conv3x3_v2.inc BC_F32S): its own weight stream behind the others
        // (2x2 output tiles of an fp32 layer exist in this form only: any request is served by it)
        ps.add_aux(direct_flops * 3.0 / 16.0);      // three 16-bit MFMAs of 16 channels where the fp32 pipe runs eight of 2
        a.wpk = reinterpret_cast<const float *>(a.wpk) + (size_t)(KS == 3 ? 77 : 1) * a.Cin * a.Cout;
        if (a.force_cfg >= 0) a.force_cfg &= ~0x2000;
#if defined(BC_MONO)
        const int rcs = conv_v2_run<BC_F32S, S, KS>(a);
#else
        const int rcs = KS == 3 ? bc_part_conv_v2_12(&a) : bc_part_conv_v2_13(&a);
#endif
        if (a.chosen >= 0) { a.chosen |= 0x2000; g_tune.conv_last_cfg = a.chosen; }
        return rcs;
    }
    ps.add_aux(direct_flops);
#if defined(BC_MONO)
    const int rc = conv_v2_run<DT, S, KS>(a);
#else
    typedef int (*part_fn)(void *);
    static const part_fn parts[6] = {bc_part_conv_v2_1, bc_part_conv_v2_2, bc_part_conv_v2_3, bc_part_conv_v2_4, bc_part_conv_v2_5, bc_part_conv_v2_6};
    const int rc = parts[2 * DT + (KS == 3 ? 0 : 1)](&a);
#endif
    if (a.chosen >= 0) g_tune.conv_last_cfg = a.chosen;
    return rc;
}

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
    static const char *names[BC_OP_COUNT] = {"split", "combine", "transfer", "pad", "combine_copy", "pad_ring", "grid_tables", "interp", "affine", "nms", "conv3x3", "head1x1", "pred3x3"};
    return (op >= 0 && op < BC_OP_COUNT) ? names[op] : "?";
}

BC_EXPORT int bc_split(void *blocks, const void *image, const int32_t *mapping_exec, int n_exec,
                       int N, int C, int H, int W, int bs, int E, void *stream)
{
    const DynArm arm = dyn_take();
    int rc = check_dense(N, C, H, W, bs, E);
    if (rc != BC_OK) return rc;
    if (n_exec < 0) return BC_ERR_SHAPE;
    if (n_exec == 0) return BC_OK;
    if (!blocks || !image || !mapping_exec) return BC_ERR_NULL;
    if (!aligned(blocks, E) || !aligned(image, E)) return BC_ERR_ALIGN;
    ProfScope ps(BC_OP_SPLIT, 2.0 * n_exec * C * bs * bs * E);
    return launch_tiles<true>(ps, blocks, const_cast<void *>(image), mapping_exec, n_exec, N, C, H, W, bs, E, (hipStream_t)stream, arm);
}

BC_EXPORT int bc_combine(const void *blocks, void *out, const int32_t *mapping_exec, int n_exec,
                         int N, int C, int H, int W, int bs, int E, void *stream)
{
    const DynArm arm = dyn_take();
    int rc = check_dense(N, C, H, W, bs, E);
    if (rc != BC_OK) return rc;
    if (n_exec < 0) return BC_ERR_SHAPE;
    if (n_exec == 0) return BC_OK;
    if (!blocks || !out || !mapping_exec) return BC_ERR_NULL;
    if (!aligned(blocks, E) || !aligned(out, E)) return BC_ERR_ALIGN;
    ProfScope ps(BC_OP_COMBINE, 2.0 * n_exec * C * bs * bs * E);
    return launch_tiles<false>(ps, const_cast<void *>(blocks), out, mapping_exec, n_exec, N, C, H, W, bs, E, (hipStream_t)stream, arm);
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
    const int grid = grid_exact(g.total, 1);
    ProfScope ps(BC_OP_COMBINE_COPY, 2.0 * N * C * H * W * E);
#define BC_CC(VB_)                                                                                             \
    case VB_:                                                                                                  \
        BC_LAUNCH(ps, (k_combine_copy<VB_>), dim3(grid), dim3(WG), 0, st, (const VecOf<VB_>::type *)bl,        \
                  (long long)(((const char *)pv - (const char *)bl) / VB_), (VecOf<VB_>::type *)out,           \
                  grid_idx, g);                                                                                \
        break;
    switch (vb) { BC_CC(16) BC_CC(8) BC_CC(4) BC_CC(2) BC_CC(1) }
#undef BC_CC
    return launch_status();
}

BC_EXPORT int bc_combine_copy_indirect(const void *blocks, const void *slots, const int32_t *grid_idx,
                                       int N, int C, int H, int W, int bs, int E, int align, void *stream)
{
    int rc = check_dense(N, C, H, W, bs, E);
    if (rc != BC_OK) return rc;
    if (!blocks || !slots || !grid_idx) return BC_ERR_NULL;
    if (align <= 0 || (align & (align - 1))) return BC_ERR_ALIGN;
    if (!aligned(blocks, E) || !aligned(slots, 8)) return BC_ERR_ALIGN;
    hipStream_t st = (hipStream_t)stream;
    // the dense maps' addresses are not known here: `align` is the caller's promise about them (any future prev / out)
    const int vb = pick_vb((size_t)bs * E, {blocks, reinterpret_cast<const void *>((uintptr_t)(align > 16 ? 16 : align))});
    DenseGeom g;
    const uint32_t vpr = (uint32_t)((size_t)bs * E / vb), vprW = (uint32_t)((size_t)W * E / vb);
    g.vprW = make_fd(vprW); g.H = make_fd(H); g.C = make_fd(C); g.vpr = make_fd(vpr); g.bs = make_fd(bs);
    g.GH = H / bs; g.GW = W / bs;
    g.total = (uint32_t)((uint64_t)N * C * H * vprW);
    const int grid = grid_exact(g.total, 1);
#define BC_CI(VB_)                                                                                             \
    case VB_:                                                                                                  \
        hipLaunchKernelGGL((k_combine_copy_ind<VB_>), dim3(grid), dim3(WG), 0, st, (const VecOf<VB_>::type *)blocks, \
                           (const unsigned long long *)slots, grid_idx, g);                                    \
        break;
    switch (vb) { BC_CI(16) BC_CI(8) BC_CI(4) BC_CI(2) BC_CI(1) }
#undef BC_CI
    return launch_status();
}

template <int DT, int CIN>
static int launch_head1x1(ProfScope &ps, void *out, const void *features, const void *wpk, const void *prev, const void *slots,
                          const int32_t *grid_idx, const int32_t *mapping_exec, const HeadGeom &g, const Prologue &pr,
                          const float *out_shift, hipStream_t st)
{
    typedef typename CvType<DT>::T T;
    constexpr int PXV_ = CIN * CvType<DT>::E / 16;
    constexpr size_t lds_bytes = (size_t)4 * 32 * ((PXV_ < 16 ? PXV_ : 16) + 1) * 16;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_head1x1<DT, CIN>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        attr_set = true;
    }
    const dim3 grid(g.n_waves / 4);
    BC_LAUNCH(ps, (k_head1x1<DT, CIN>), grid, dim3(256), lds_bytes, st, (T *)out, (const uint4 *)features, (const uint4 *)wpk, (const T *)prev,
              (const unsigned long long *)slots, grid_idx, mapping_exec, g, pr, out_shift);
    return launch_status();
}

// fp32 / Cout <= 20: the 16 + 4 output-channel split of head1x1.inc (k_head1x1_s)
template <int CIN, int LEAN = 0>
static int launch_head1x1_s(ProfScope &ps, void *out, const void *features, const void *wpk, const void *prev, const void *slots,
                            const int32_t *grid_idx, const int32_t *mapping_exec, const HeadGeom &g, const Prologue &pr,
                            const float *out_shift, hipStream_t st)
{
    constexpr int PXV_ = CIN / 4;
    constexpr size_t lds_bytes = (size_t)4 * 32 * ((PXV_ < 16 ? PXV_ : 16) + 1) * 16 + (size_t)PXV_ * 4 * 16;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_head1x1_s<CIN, LEAN>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        attr_set = true;
    }
    const dim3 grid(g.n_waves / 4);
    BC_LAUNCH(ps, (k_head1x1_s<CIN, LEAN>), grid, dim3(256), lds_bytes, st, (float *)out, (const uint4 *)features, (const uint4 *)wpk, (const float *)prev,
              (const unsigned long long *)slots, grid_idx, mapping_exec, g, pr, out_shift);
    return launch_status();
}

BC_EXPORT int bc_tile_copy_indirect(void *dst, const void *src_slot, const int32_t *mapping_exec, const int32_t *n_exec_dev, int n_exec,
                                    int N, int C, int H, int W, int bs, int E, int align, void *stream)
{
    int rc = check_dense(N, C, H, W, bs, E);
    if (rc != BC_OK) return rc;
    if (n_exec < 0 || n_exec > N * (H / bs) * (W / bs)) return BC_ERR_SHAPE;
    if (n_exec == 0) return BC_OK;
    if (!dst || !src_slot || !mapping_exec) return BC_ERR_NULL;
    if (align <= 0 || (align & (align - 1))) return BC_ERR_ALIGN;
    if (!aligned(dst, E) || !aligned(src_slot, 8) || !aligned(n_exec_dev, 4)) return BC_ERR_ALIGN;
    // the source address is not known here: `align` is the caller's promise about it (any future frame)
    const int vb = pick_vb((size_t)bs * E, {dst, reinterpret_cast<const void *>((uintptr_t)(align > 16 ? 16 : align))});
    TileGeom g;
    const uint32_t vpr = (uint32_t)((size_t)bs * E / vb);
    g.vpr = make_fd(vpr); g.bs = make_fd(bs); g.C = make_fd(C); g.GW = make_fd(W / bs); g.GH = make_fd(H / bs);
    g.H = H; g.bsz = bs; g.vprW = (uint32_t)((size_t)W * E / vb);
    g.total = (uint32_t)((uint64_t)n_exec * C * bs * vpr);
    const int grid = grid_exact(g.total, 1);
    ProfScope ps(BC_OP_SPLIT, 2.0 * n_exec * C * bs * bs * E);
    hipStream_t st = (hipStream_t)stream;
#define BC_TC(VB_)                                                                                                          \
    case VB_:                                                                                                               \
        BC_LAUNCH(ps, (k_tile_copy_ind<VB_>), dim3(grid), dim3(WG), 0, st, (VecOf<VB_>::type *)dst,                         \
                  (const unsigned long long *)src_slot, mapping_exec, n_exec_dev, (uint32_t)C * bs * vpr, g);               \
        break;
    switch (vb) { BC_TC(16) BC_TC(8) BC_TC(4) BC_TC(2) BC_TC(1) }
#undef BC_TC
    return launch_status();
}

BC_EXPORT int bc_head1x1_scatter_nhwc(void *out, const void *features, const void *weights_packed, const void *prev, const void *slots,
                                      const int32_t *grid_idx, const int32_t *mapping_exec, int n_exec, int N, int Cin, int Cout,
                                      int GH, int GW, int bs, int dtype, const float *in_scale, const float *in_shift, int in_relu,
                                      const float *out_shift, int scatter, void *stream)
{
    const DynArm arm = dyn_take();
    if (dtype < BC_F32 || dtype > BC_BF16) return BC_ERR_ELEM;
    if (n_exec < 0 || N <= 0 || Cin <= 0 || Cout <= 0 || Cout > 32 || GH <= 0 || GW <= 0 || bs <= 0) return BC_ERR_SHAPE;
    // a 32-pixel M-block is stored as 32 / run_px row segments of run_px = min(bs, 32) pixels: bs must divide 32 or be a multiple of it
    // (bs = 24 would leave 8 of every 32 pixels unwritten and start segments mid-row)
    if (bs % 8 != 0 || !((bs >= 32 && bs % 32 == 0) || 32 % bs == 0)) return BC_ERR_SHAPE;
    const bool cin_ok = dtype == BC_F32 ? (Cin == 64 || Cin == 128) : (Cin == 64 || Cin == 128 || Cin == 256);
    if (!cin_ok) return BC_ERR_SHAPE;
    if (scatter && n_exec > N * GH * GW) return BC_ERR_SHAPE;
    if (n_exec == 0 && !scatter) return BC_OK;
    if (!features && n_exec > 0) return BC_ERR_NULL;
    if (!weights_packed || (!out && !slots)) return BC_ERR_NULL;
    if (scatter && (!grid_idx || (!mapping_exec && n_exec > 0))) return BC_ERR_NULL;
    if (scatter && !slots && !prev && (n_exec < N * GH * GW || arm.ptr)) return BC_ERR_NULL;     // skipped tiles need the previous map
    const int E = dtype == BC_F32 ? 4 : 2;
    if ((uint64_t)N * GH * GW * bs * bs * (uint64_t)(Cin > Cout ? Cin : Cout) >= (1ull << 31)) return BC_ERR_RANGE;
    if (!aligned(out, 16) || !aligned(features, 16) || !aligned(weights_packed, 16) || !aligned(prev, 16) || !aligned(slots, 8)) return BC_ERR_ALIGN;
    HeadGeom g;
    g.n_exec = n_exec; g.bs = bs; g.Cout = Cout; g.GH = GH; g.GW = GW;
    g.blocks_per_tile = bs * bs / 32;
    g.n_mblocks = (uint32_t)n_exec * g.blocks_per_tile;
    g.scatter = scatter ? 1 : 0;
    if (!dyn_tiles(arm, n_exec, 1, g.dyn)) return BC_ERR_SHAPE;
    // (a count that is only known on the device may leave any tile skipped: the copy half is always on)
    g.copy_rows = (scatter && (n_exec < N * GH * GW || arm.ptr)) ? (uint32_t)(N * GH * GW) * (uint32_t)bs : 0;
    // one round of 256 CUs x 2 workgroups x 4 waves at most; at least one wave per M-block or per two tile rows to look at
    uint32_t want = g.n_mblocks > (g.copy_rows + 1) / 2 ? g.n_mblocks : (g.copy_rows + 1) / 2;
    const bool lean = dtype == BC_F32 && Cout <= 20 && g_tune.head_split == 2 && Cin == 128;
    if (lean) want = g.n_mblocks + g.copy_rows / 4;      // (three waves per SIMD: the waves behind the M-blocks only copy)
    static const uint32_t wave_cap = [] { const char *e = getenv("BC_HEAD_WAVES"); return e ? (uint32_t)atoi(e) : 2048u; }();      // (measurement knob)
    if (want > (lean ? wave_cap + wave_cap / 2 : wave_cap)) want = lean ? wave_cap + wave_cap / 2 : wave_cap;
    if (want < 1u) want = 1u;
    g.n_waves = ((want + 3) / 4) * 4;
    g.run_px = bs < 32 ? bs : 32;
    g.runs = 32 / g.run_px;
    Prologue pr{in_scale, in_shift, in_relu};
    const double px = (double)n_exec * bs * bs;
    // algorithmic bytes: packed features read; scatter: every skipped tile read once from the previous map + the whole map written
    // (the packed logits of the executed tiles never exist in memory), else the packed result written
    const double map_px = (double)N * GH * GW * bs * bs;
    ProfScope ps(BC_OP_HEAD, px * Cin * E + (scatter ? (map_px - px) * Cout * E + map_px * Cout * E : px * Cout * E));
    ps.add_aux(2.0 * px * Cin * 32.0);
    hipStream_t st = (hipStream_t)stream;
#define BC_HD(DT_, CIN_) return launch_head1x1<DT_, CIN_>(ps, out, features, weights_packed, prev, slots, grid_idx, mapping_exec, g, pr, out_shift, st)
    if (dtype == BC_F32 && Cout <= 20 && g_tune.head_split) {      // the matrix work cut to 16 + 4 output channels (k_head1x1_s)
        ps.add_aux(2.0 * px * Cin * 20.0 - 2.0 * px * Cin * 32.0);
        if (g_tune.head_split == 2 && Cin == 128) return launch_head1x1_s<128, 1>(ps, out, features, weights_packed, prev, slots, grid_idx, mapping_exec, g, pr, out_shift, st);
        if (Cin == 64) return launch_head1x1_s<64>(ps, out, features, weights_packed, prev, slots, grid_idx, mapping_exec, g, pr, out_shift, st);
        return launch_head1x1_s<128>(ps, out, features, weights_packed, prev, slots, grid_idx, mapping_exec, g, pr, out_shift, st);
    }
    if (dtype == BC_F32) { if (Cin == 64) BC_HD(BC_F32, 64); BC_HD(BC_F32, 128); }
    if (dtype == BC_F16) { if (Cin == 64) BC_HD(BC_F16, 64); if (Cin == 128) BC_HD(BC_F16, 128); BC_HD(BC_F16, 256); }
    if (Cin == 64) BC_HD(BC_BF16, 64);
    if (Cin == 128) BC_HD(BC_BF16, 128);
    BC_HD(BC_BF16, 256);
#undef BC_HD
}

template <int DT, int COUT, int R>
static int launch_pred3x3(ProfScope &ps, void *out, const void *x, const float *wpk, const float *bias, PredGeom g, hipStream_t st)
{
    typedef typename CvType<DT>::T T;
    constexpr int NPX = PRED_PW * 8 * R;
    constexpr size_t stage = (size_t)NPX * PRED_PS * 4, sums = (size_t)9 * COUT * NPX * 4;
    constexpr size_t lds_bytes = stage > sums ? stage : sums;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_pred3x3<DT, COUT, R>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        attr_set = true;
    }
    g.tiles_x = (g.W + (PRED_PW - 2) - 1) / (PRED_PW - 2);
    g.tiles_y = (g.H + (8 * R - 2) - 1) / (8 * R - 2);
    const dim3 grid(g.tiles_x * g.tiles_y, g.N);
    BC_LAUNCH(ps, (k_pred3x3<DT, COUT, R>), grid, dim3(256), lds_bytes, st, (T *)out, (const uint4 *)x, wpk, bias, g);
    return launch_status();
}

BC_EXPORT int bc_pred3x3_nhwc(void *out, const void *x, const float *weights_packed, const float *bias, int N, int H, int W, int Cin, int Cout,
                              int dtype, void *stream)
{
    if (dtype < BC_F32 || dtype > BC_BF16) return BC_ERR_ELEM;
    if (N <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cin % PRED_CK != 0 || Cout < 1 || Cout > 4) return BC_ERR_SHAPE;
    if (!out || !x || !weights_packed) return BC_ERR_NULL;
    if ((uint64_t)N * H * W * (uint64_t)Cin >= (1ull << 31) || N > 65535) return BC_ERR_RANGE;
    const int E = dtype == BC_F32 ? 4 : 2;
    if (!aligned(out, E) || !aligned(x, 16) || !aligned(weights_packed, 4) || !aligned(bias, 4)) return BC_ERR_ALIGN;
    PredGeom g;
    g.N = N; g.H = H; g.W = W; g.C = Cin; g.tiles_x = g.tiles_y = 0;
    ProfScope ps(BC_OP_PRED, (double)N * H * W * ((double)Cin + Cout) * E);
    ps.add_aux(2.0 * N * H * W * 9.0 * Cin * Cout);
    hipStream_t st = (hipStream_t)stream;
    // (8-row patches, one pixel per lane.  16-row patches -- 14 % halo rows instead of 25 % -- measured the same or slower: the halo
    //  re-reads are L2 hits, and two pixels per lane with two chunks in flight leave one wave per SIMD)
#define BC_PR3(DT_, CO_) return launch_pred3x3<DT_, CO_, 1>(ps, out, x, weights_packed, bias, g, st)
#define BC_PR2(DT_) switch (Cout) { case 1: BC_PR3(DT_, 1); case 2: BC_PR3(DT_, 2); case 3: BC_PR3(DT_, 3); default: BC_PR3(DT_, 4); }
    if (dtype == BC_F32) BC_PR2(BC_F32)
    if (dtype == BC_F16) BC_PR2(BC_F16)
    BC_PR2(BC_BF16)
#undef BC_PR2
#undef BC_PR3
}

static int spp_geom(SppGeom &g, int H, int W, int C, int CO, int L, const int32_t *grids, int N)
{
    if (H <= 0 || W <= 0 || C <= 0 || CO <= 0 || L <= 0 || L > SPP_MAX_LEVELS || !grids) return BC_ERR_SHAPE;
    if (C % 4 != 0 || C > 1024 || 256 % (C / 4) != 0 || CO > 1024) return BC_ERR_SHAPE;
    if ((uint64_t)H * W * (uint64_t)(C + L * CO) >= (1ull << 31)) return BC_ERR_RANGE;
    g.H = H; g.W = W; g.C = C; g.CO = CO; g.L = L;
    uint32_t nb = 0;
    for (int l = 0; l < L; ++l) {
        if (grids[2 * l] <= 0 || grids[2 * l + 1] <= 0 || grids[2 * l] > 4096 || grids[2 * l + 1] > 4096) return BC_ERR_SHAPE;     // (a grid finer than the map is fine: ATen's bins overlap then)
        g.gh[l] = grids[2 * l]; g.gw[l] = grids[2 * l + 1]; g.bin0[l] = nb;
        nb += g.gh[l] * g.gw[l];
    }
    for (int l = L; l < SPP_MAX_LEVELS; ++l) { g.gh[l] = g.gw[l] = 1; g.bin0[l] = nb; }
    g.n_bins = nb; g.K = C + L * CO; g.N = N;
    g.mapping = nullptr; g.bs = 0; g.GW = 0; g.n_rows = 0;
    return BC_OK;
}

BC_EXPORT int bc_spp_levels_n_nhwc(void *lv, const void *x, const float *scale, const float *shift, const float *weights, int B, int H, int W, int C,
                                   int CO, int L, const int32_t *grids, int dtype, void *stream)
{
    if (dtype < BC_F32 || dtype > BC_BF16) return BC_ERR_ELEM;
    if (B <= 0 || B > 65535) return BC_ERR_SHAPE;
    SppGeom g;
    const int rc = spp_geom(g, H, W, C, CO, L, grids, 0);
    if (rc != BC_OK) return rc;
    if (!lv || !x || !weights) return BC_ERR_NULL;
    const int E = dtype == BC_F32 ? 4 : 2;
    if (!aligned(lv, E) || !aligned(x, 16) || !aligned(scale, 4) || !aligned(shift, 4) || !aligned(weights, 4)) return BC_ERR_ALIGN;
    ProfScope ps(BC_OP_AFFINE, (double)B * L * H * W * C * E);
    hipStream_t st = (hipStream_t)stream;
    const size_t lds_bytes = ((size_t)(256 / (C / 4) + 1) * C + (size_t)C * CO) * sizeof(float);       // partial sums, activated means, the level's weights
    if (lds_bytes > 150 * 1024) return BC_ERR_SHAPE;
    static size_t lv_attr[3] = {0, 0, 0};
#define BC_SL(DT_)                                                                                                                                   \
    if (lds_bytes > lv_attr[DT_] && lds_bytes > 48 * 1024) {                                                                                           \
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_spp_levels<DT_>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);   \
        lv_attr[DT_] = lds_bytes;                                                                                                                      \
    }                                                                                                                                                  \
    BC_LAUNCH(ps, (k_spp_levels<DT_>), dim3(g.n_bins, B), dim3(256), lds_bytes, st, (CvType<DT_>::T *)lv, (const CvType<DT_>::T *)x, scale, shift, weights, g)
    if (dtype == BC_F32) { BC_SL(BC_F32); } else if (dtype == BC_F16) { BC_SL(BC_F16); } else { BC_SL(BC_BF16); }
#undef BC_SL
    return launch_status();
}

BC_EXPORT int bc_spp_levels_nhwc(void *lv, const void *x, const float *scale, const float *shift, const float *weights, int H, int W, int C, int CO,
                                 int L, const int32_t *grids, int dtype, void *stream)
{
    return bc_spp_levels_n_nhwc(lv, x, scale, shift, weights, 1, H, W, C, CO, L, grids, dtype, stream);
}

static int spp_fuse_impl(void *out, const void *x, const void *lv, const float *scale, const float *shift, const void *weights_packed, int B,
                         int H, int W, int C, int CO, int L, const int32_t *grids, int N, int dtype, const int32_t *mapping_exec, int n_exec, int bs,
                         void *stream)
{
    if (dtype < BC_F32 || dtype > BC_BF16) return BC_ERR_ELEM;
    if (B <= 0 || B > 65535) return BC_ERR_SHAPE;
    SppGeom g;
    const int rc = spp_geom(g, H, W, C, CO, L, grids, N);
    if (rc != BC_OK) return rc;
    if (mapping_exec) {
        // packed result: rows = the pixels of the executed tiles of ONE map
        if (B != 1 || bs <= 0 || H % bs != 0 || W % bs != 0 || n_exec < 0 || n_exec > (H / bs) * (W / bs)) return BC_ERR_SHAPE;
        if (n_exec == 0) return BC_OK;
        g.mapping = mapping_exec; g.bs = bs; g.GW = W / bs; g.n_rows = (uint32_t)n_exec * bs * bs;
    }
    const uint32_t rows = mapping_exec ? g.n_rows : (uint32_t)(H * W);
    if (N <= 0 || N % 64 != 0) return BC_ERR_SHAPE;
    if (!out || !x || !lv || !weights_packed) return BC_ERR_NULL;
    const int E = dtype == BC_F32 ? 4 : 2;
    if (!aligned(out, E) || !aligned(x, E) || !aligned(lv, E) || !aligned(weights_packed, 16) || !aligned(scale, 4) || !aligned(shift, 4)) return BC_ERR_ALIGN;
    ProfScope ps(BC_OP_CONV3X3, 2.0 * B * rows * (double)g.K * N);
    ps.add_aux(2.0 * B * rows * (double)(((g.K + 31) / 32) * 32) * N);
    hipStream_t st = (hipStream_t)stream;
    const size_t kp = (size_t)((g.K + 31) / 32) * 32;
    // two stages per K group + the level maps (fp32) + the block's folded BN + per (row, level) bilinear taps and weights
    const size_t stage_bytes = (size_t)2 * (64 * 9 + 2 * 256) * 16, table_bytes = ((size_t)g.n_bins * CO + 2 * kp + 64 * SPP_MAX_LEVELS * 6) * 4;
    // two K groups (512 threads, csrc/spp.inc) where the kernel's preload path covers the shape and the second pair of stages fits
    static const int ks_knob = [] { const char *e = getenv("BC_SPP_KS"); return e ? atoi(e) : 2; }();
    const bool pre = kp / 32 <= 8 && C % 32 == 0 && C / 32 <= 4;
    const int KS = (ks_knob >= 2 && pre && kp / 32 >= 2 && 2 * stage_bytes + table_bytes <= 150 * 1024) ? 2 : 1;
    const size_t lds_bytes = KS * stage_bytes + table_bytes;
    if (lds_bytes > 150 * 1024) return BC_ERR_SHAPE;
    static size_t attr_set[3][2] = {{0, 0}, {0, 0}, {0, 0}};
    const dim3 grid((rows + 63) / 64, N / 64, B);
#define BC_SF2(DT_, KS_)                                                                                                                             \
    do {                                                                                                                                               \
        if (lds_bytes > attr_set[DT_][KS_ - 1] && lds_bytes > 48 * 1024) {                                                                             \
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_spp_fuse<DT_, KS_>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes); \
            attr_set[DT_][KS_ - 1] = lds_bytes;                                                                                                        \
        }                                                                                                                                              \
        BC_LAUNCH(ps, (k_spp_fuse<DT_, KS_>), grid, dim3(256 * KS_), lds_bytes, st, (CvType<DT_>::T *)out, (const CvType<DT_>::T *)x,               \
                  (const CvType<DT_>::T *)lv, scale, shift, (const uint4 *)weights_packed, g);                                                        \
    } while (0)
#define BC_SF(DT_) do { if (KS == 2) BC_SF2(DT_, 2); else BC_SF2(DT_, 1); } while (0)
    if (dtype == BC_F32) { BC_SF(BC_F32); } else if (dtype == BC_F16) { BC_SF(BC_F16); } else { BC_SF(BC_BF16); }
#undef BC_SF
#undef BC_SF2
    return launch_status();
}

BC_EXPORT int bc_spp_fuse_n_nhwc(void *out, const void *x, const void *lv, const float *scale, const float *shift, const void *weights_packed, int B,
                                 int H, int W, int C, int CO, int L, const int32_t *grids, int N, int dtype, void *stream)
{
    return spp_fuse_impl(out, x, lv, scale, shift, weights_packed, B, H, W, C, CO, L, grids, N, dtype, nullptr, 0, 0, stream);
}

BC_EXPORT int bc_spp_fuse_nhwc(void *out, const void *x, const void *lv, const float *scale, const float *shift, const void *weights_packed, int H, int W,
                               int C, int CO, int L, const int32_t *grids, int N, int dtype, void *stream)
{
    return spp_fuse_impl(out, x, lv, scale, shift, weights_packed, 1, H, W, C, CO, L, grids, N, dtype, nullptr, 0, 0, stream);
}

BC_EXPORT int bc_spp_fuse_packed_nhwc(void *out, const void *x, const void *lv, const float *scale, const float *shift, const void *weights_packed,
                                      const int32_t *mapping_exec, int n_exec, int bs, int H, int W, int C, int CO, int L, const int32_t *grids, int N,
                                      int dtype, void *stream)
{
    if (!mapping_exec) return BC_ERR_NULL;
    return spp_fuse_impl(out, x, lv, scale, shift, weights_packed, 1, H, W, C, CO, L, grids, N, dtype, mapping_exec, n_exec, bs, stream);
}

BC_EXPORT int bc_combine_copy_cells(const void *blocks, int N, int C, int H, int W, int bs, int E, int align)
{
    int rc = check_dense(N, C, H, W, bs, E);
    if (rc != BC_OK) return rc;
    if (align <= 0 || (align & (align - 1))) return BC_ERR_ALIGN;
    const int vb = pick_vb((size_t)bs * E, {blocks, reinterpret_cast<const void *>((uintptr_t)(align > 16 ? 16 : align))});
    return grid_exact((uint64_t)N * C * H * ((size_t)W * E / vb), 1);
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
    ProfScope ps(BC_OP_TRANSFER, 2.0 * n_transfer * C * ring * E);
#define BC_TR(VB_)                                                                                             \
    case VB_:                                                                                                  \
        BC_LAUNCH(ps, (k_transfer<VB_>), dim3(grid), dim3(WG), 0, st, (VecOf<VB_>::type *)out,            \
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
    ProfScope ps(BC_OP_PAD, halo_bytes(n_exec, C, bs, pad, E));
    // transfer may be NULL/empty when every tile is executed (first frame): it is then never dereferenced
    return launch_halo<false>(ps, out, features, transfer ? transfer : features, nullptr, grid_idx, mapping_exec, n_exec,
                              N, C, GH, GW, bs, pad, E, (hipStream_t)stream);
}

BC_EXPORT int bc_pad_ring(void *out, const void *features, void *ring, const int32_t *grid_idx,
                          const int32_t *mapping_exec, int n_exec,
                          int N, int C, int GH, int GW, int bs, int pad, int E, void *stream)
{
    const DynArm arm = dyn_take();
    int rc = check_halo(out, features, grid_idx, mapping_exec, n_exec, N, C, GH, GW, bs, pad, E);
    if (rc != BC_OK || n_exec == 0) return rc;
    if (!ring) return BC_ERR_NULL;
    if (!aligned(ring, E)) return BC_ERR_ALIGN;
    DynCount dyn;
    if (!dyn_tiles(arm, n_exec, 1, dyn)) return BC_ERR_SHAPE;
    ProfScope ps(BC_OP_PAD_RING, halo_bytes(n_exec, C, bs, pad, E));
    return launch_halo<true>(ps, out, features, ring, ring, grid_idx, mapping_exec, n_exec, N, C, GH, GW, bs, pad, E,
                             (hipStream_t)stream, 0, Prologue{nullptr, nullptr, 0}, dyn);
}

BC_EXPORT int bc_pad_ring_act(void *out, const void *features, void *ring, const int32_t *grid_idx,
                              const int32_t *mapping_exec, int n_exec, int N, int C, int GH, int GW, int bs, int pad,
                              int dtype, const float *scale, const float *shift, int relu, void *stream)
{
    const DynArm arm = dyn_take();
    if (dtype < BC_F32 || dtype > BC_BF16) return BC_ERR_ELEM;
    const int E = dtype == BC_F32 ? 4 : 2;
    int rc = check_halo(out, features, grid_idx, mapping_exec, n_exec, N, C, GH, GW, bs, pad, E);
    if (rc != BC_OK || n_exec == 0) return rc;
    if (!ring) return BC_ERR_NULL;
    if (!aligned(ring, E)) return BC_ERR_ALIGN;
    DynCount dyn;
    if (!dyn_tiles(arm, n_exec, 1, dyn)) return BC_ERR_SHAPE;
    ProfScope ps(BC_OP_PAD_RING, halo_bytes(n_exec, C, bs, pad, E));
    Prologue pr{scale, shift, relu};
    const int dt = (scale || shift || relu) ? dtype + 1 : 0;
    return launch_halo<true>(ps, out, features, ring, ring, grid_idx, mapping_exec, n_exec, N, C, GH, GW, bs, pad, E,
                             (hipStream_t)stream, dt, pr, dyn);
}

BC_EXPORT int bc_affine_act(void *out, const void *in, const void *add, const float *scale, const float *shift, int relu,
                            long long B, int C, long long hw, int dtype, void *stream)
{
    if (dtype < BC_F32 || dtype > BC_BF16) return BC_ERR_ELEM;
    if (B < 0 || C <= 0 || hw <= 0) return BC_ERR_SHAPE;
    if (B == 0) return BC_OK;
    if (!out || !in) return BC_ERR_NULL;
    const int E = dtype == BC_F32 ? 4 : 2;
    if ((uint64_t)B * C * hw >= (1ull << 31)) return BC_ERR_RANGE;
    if (!aligned(out, E) || !aligned(in, E) || !aligned(add, E)) return BC_ERR_ALIGN;
    hipStream_t st = (hipStream_t)stream;
    int q = 16 / E;
    while (q > 1 && ((hw % q) != 0 || !aligned(out, q * E) || !aligned(in, q * E) || !aligned(add, q * E))) q >>= 1;
    AffineGeom g;
    g.hwq = make_fd((uint32_t)(hw / q)); g.C = make_fd(C);
    g.total = (uint32_t)((uint64_t)B * C * (hw / q));
    const int grid = grid_exact(g.total, 1);
    ProfScope ps(BC_OP_AFFINE, (add ? 3.0 : 2.0) * B * C * hw * E);
#define BC_AF(T_, Q_) BC_LAUNCH(ps, (k_affine_act<T_, Q_>), dim3(grid), dim3(WG), 0, st, (T_ *)out, (const T_ *)in, (const T_ *)add, scale, shift, relu, g)
#define BC_AFQ(T_, QMAX_) do { if (q == QMAX_) BC_AF(T_, QMAX_); else if (q == QMAX_ / 2) BC_AF(T_, QMAX_ / 2); \
                               else if (QMAX_ >= 8 && q == 2) BC_AF(T_, 2); else BC_AF(T_, 1); } while (0)
    if (dtype == BC_F32) BC_AFQ(float, 4);
    else if (dtype == BC_F16) BC_AFQ(__half, 8);
    else BC_AFQ(hip_bfloat16, 8);
#undef BC_AFQ
#undef BC_AF
    return launch_status();
}

// one device-scope ticket per stream that has run an NMS (concurrent launches on different streams must not share one)
__device__ unsigned int g_nms_tickets[64];

static int nms_launch(const float *boxes, int n, const int32_t *n_dev, float iou_thr, unsigned long long *mask_ws, int32_t *keep,
                      int32_t *count, void *stream)
{
    if (n < 0 || n > 4096) return BC_ERR_SHAPE;
    if (!count || (n > 0 && (!boxes || !mask_ws || !keep))) return BC_ERR_NULL;
    hipStream_t st = (hipStream_t)stream;
    if (n == 0) { (void)hipMemsetAsync(count, 0, sizeof(int32_t), st); return launch_status(); }
    // ticket slot = (device, stream): the symbol lives once per device, and two devices' null streams must not share a word
    static std::mutex mu;
    struct Slot { int dev; void *stream; };
    static Slot slot_of[8][64];
    static int n_slots[8];
    static unsigned int *tickets_of[8];
    static bool attr_set[8];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return launch_status();
    if (dev < 0 || dev >= 8) return BC_ERR_SHAPE;
    unsigned int *tickets = nullptr;
    int slot = -1;
    {
        std::lock_guard<std::mutex> lk(mu);
        if (!tickets_of[dev] && hipGetSymbolAddress((void **)&tickets_of[dev], HIP_SYMBOL(g_nms_tickets)) != hipSuccess) return launch_status();
        tickets = tickets_of[dev];
        for (int i = 0; i < n_slots[dev]; ++i)
            if (slot_of[dev][i].stream == stream) slot = i;
        if (slot < 0) {
            if (n_slots[dev] == 64) return BC_ERR_SHAPE;      // (more than 64 streams of one device running detectors in one process)
            slot_of[dev][n_slots[dev]] = Slot{dev, stream};
            slot = n_slots[dev]++;
        }
        if (!attr_set[dev]) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_nms), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
            attr_set[dev] = true;
        }
    }
    NmsGeom g;
    g.n = n; g.words = (n + 63) / 64; g.tiles = g.words * (g.words + 1) / 2;
    g.row_shift = 0;
    while ((1 << g.row_shift) < g.words) ++g.row_shift;
    // phase 2 in LDS: the n x words matrix and the n transposed diagonal words
    const size_t all = ((size_t)n * (g.words | 1) + n + 17) * sizeof(unsigned long long);     // (odd row stride, 17 spare words: see the kernel)
    g.lds_rows = (all <= (size_t)150 * 1024 && n <= 1024) ? n : 0;
    const size_t lds_bytes = g.lds_rows ? all : 0;
    ProfScope ps(BC_OP_NMS, 20.0 * n);
    BC_LAUNCH(ps, k_nms, dim3(g.tiles), dim3(1024), lds_bytes, st, g, iou_thr, boxes, mask_ws, keep, count, tickets + slot, g_tune.conv_stamps, n_dev);
    return launch_status();
}

BC_EXPORT int bc_nms_sorted(const float *boxes, int n, float iou_thr, unsigned long long *mask_ws, int32_t *keep,
                            int32_t *count, void *stream)
{
    return nms_launch(boxes, n, nullptr, iou_thr, mask_ws, keep, count, stream);
}

BC_EXPORT int bc_nms_sorted_dev(const float *boxes, int n_max, const int32_t *n_dev, float iou_thr, unsigned long long *mask_ws,
                                int32_t *keep, int32_t *count, void *stream)
{
    if (!n_dev) return BC_ERR_NULL;
    if (n_max <= 0) return BC_ERR_SHAPE;
    return nms_launch(boxes, n_max, n_dev, iou_thr, mask_ws, keep, count, stream);
}

BC_EXPORT int bc_csp_decode(const float *scores, const long long *top, const float *heights, const float *off_y, const float *off_x,
                            int k, int map_w, int stride, float wh_ratio, int img_h, int img_w, float score_thr, float *dets,
                            int32_t *n_sel, void *stream)
{
    if (k < 0 || map_w <= 0 || stride <= 0 || img_h <= 0 || img_w <= 0) return BC_ERR_SHAPE;
    if (!n_sel || (k > 0 && (!scores || !top || !heights || !off_y || !off_x || !dets))) return BC_ERR_NULL;
    DecodeGeom g{k, map_w, stride, wh_ratio, (float)(img_w - 1), (float)(img_h - 1), score_thr};
    ProfScope ps(BC_OP_NMS, 44.0 * k);
    BC_LAUNCH(ps, k_csp_decode, dim3(1), dim3(1024), 0, (hipStream_t)stream, g, scores, top, heights, off_y, off_x, dets, n_sel);
    return launch_status();
}

BC_EXPORT int bc_csp_topk_decode(const void *cls, int cls_dtype, const float *reg, const float *off, long long off_channel_stride,
                                 long long off_pixel_stride, int n, int k, int map_w, int stride, float wh_ratio, int img_h, int img_w,
                                 float score_thr, float *dets, int32_t *n_sel, int32_t *top_out, void *stream)
{
    if (n <= 0 || k <= 0 || k > n || k > 4096 || map_w <= 0 || stride <= 0 || img_h <= 0 || img_w <= 0 || n > (1 << 30)) return BC_ERR_SHAPE;
    if (cls_dtype != BC_F32 && cls_dtype != BC_F16 && cls_dtype != BC_BF16) return BC_ERR_ELEM;
    if (!cls || !reg || !off || !dets || !n_sel) return BC_ERR_NULL;
    const int vec = (cls_dtype == BC_F32 && (reinterpret_cast<uintptr_t>(cls) & 15u) == 0 && n % 4 == 0) ? 1 : 0;    // 16-byte loads
    TopkGeom g{n, k, map_w, stride, wh_ratio, (float)(img_w - 1), (float)(img_h - 1), score_thr, off_channel_stride, off_pixel_stride, cls_dtype, vec};
    const size_t lds = (size_t)16 * 2049 * 4 + 2048 * 4;            // (>= TOPK_CAP words, >= TOPK_RUN counters)
    static thread_local bool attr_set = false;                      // (the attribute is per function; setting it again is harmless)
    ProfScope ps(BC_OP_NMS, 12.0 * n + 44.0 * k);
    if (!attr_set) { (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_csp_topk_decode), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); attr_set = true; }
    BC_LAUNCH(ps, k_csp_topk_decode, dim3(1), dim3(1024), lds, (hipStream_t)stream, g, cls, reg, off, dets, n_sel, top_out);
    return launch_status();
}

BC_EXPORT int bc_csp_score_monotone(unsigned long long *violations, void *stream)
{
    if (!violations) return BC_ERR_NULL;
    (void)hipMemsetAsync(violations, 0, sizeof(unsigned long long), (hipStream_t)stream);      // (a failure surfaces in launch_status below)
    ProfScope ps(BC_OP_NMS, 0.0);
    BC_LAUNCH(ps, k_csp_score_monotone, dim3(65536), dim3(256), 0, (hipStream_t)stream, violations);
    return launch_status();
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
    ProfScope ps(BC_OP_INTERP, ((double)planes * h * w + (double)planes * H * W) * E);
#define BC_IP(T_, Q_) BC_LAUNCH(ps, (k_interp_bilinear<T_, Q_>), dim3(grid), dim3(WG), 0, st, (T_ *)out, (const T_ *)in, g)
    if (dtype == BC_F32) { if (vec) BC_IP(float, 4); else BC_IP(float, 1); }
    else if (dtype == BC_F16) { if (vec) BC_IP(__half, 8); else BC_IP(__half, 1); }
    else { if (vec) BC_IP(hip_bfloat16, 8); else BC_IP(hip_bfloat16, 1); }
#undef BC_IP
    return launch_status();
}

// ---------------------------------------------------------------------------------------------- channels-last entry points
BC_EXPORT int bc_pad_ring_nhwc(void *out, const void *features, void *ring, const int32_t *grid_idx,
                               const int32_t *mapping_exec, int n_exec, int N, int C, int GH, int GW, int bs, int pad,
                               int elem_size, int dtype, const float *scale, const float *shift, int relu, void *stream)
{
    const DynArm arm = dyn_take();
    const int E = elem_size;
    if (E != 1 && E != 2 && E != 4 && E != 8) return BC_ERR_ELEM;
    const bool act = scale || shift || relu;
    if (act && (dtype < BC_F32 || dtype > BC_BF16 || E != (dtype == BC_F32 ? 4 : 2))) return BC_ERR_ELEM;
    if (n_exec < 0 || N <= 0 || C <= 0 || GH <= 0 || GW <= 0 || bs <= 0 || pad < 1 || pad > bs) return BC_ERR_SHAPE;
    if (n_exec == 0) return BC_OK;
    if (!out || !features || !ring || !grid_idx || !mapping_exec) return BC_ERR_NULL;
    const uint64_t bsp = (uint64_t)bs + 2 * pad;
    if ((uint64_t)n_exec * C * bsp * bsp >= (1ull << 31) || (uint64_t)N * GH * GW * C * bs * bs >= (1ull << 31)) return BC_ERR_RANGE;
    const int vb = pick_vb((size_t)C * E, {out, features, ring});
    if (vb < 2 || (act && vb < E)) return BC_ERR_ALIGN;
    hipStream_t st = (hipStream_t)stream;
    HaloNhwcGeom g;
    const uint32_t K = (uint32_t)((size_t)C * E / vb);
    g.K = make_fd(K); g.BSP = make_fd((uint32_t)bsp); g.GW = make_fd(GW); g.GH = make_fd(GH);
    g.C = C; g.bs = bs; g.pad = pad; g.n_total = (uint32_t)N * GH * GW;
    g.per_tile = (uint32_t)(bsp * bsp * K);
    g.epv = vb / E;
    if (!dyn_tiles(arm, n_exec, 1, g.dyn)) return BC_ERR_SHAPE;
    const dim3 grid((g.per_tile + WG * UNROLL - 1) / (WG * UNROLL), (unsigned)n_exec);
    const long long delta = ((const char *)ring - (const char *)features) / vb;
    Prologue pr{scale, shift, relu};
    ProfScope ps(BC_OP_PAD_RING, halo_bytes(n_exec, C, bs, pad, E));
#define BC_HN(VB_, T_, DT_)                                                                                        \
    BC_LAUNCH(ps, (k_halo_nhwc<VB_, T_, true, DT_>), grid, dim3(WG), 0, st, (VecOf<VB_>::type *)out,                \
              (const VecOf<VB_>::type *)features, delta, (VecOf<VB_>::type *)ring, grid_idx, mapping_exec, g, pr)
#define BC_HNV(T_, DT_) do { if (vb == 16) BC_HN(16, T_, DT_); else if (vb == 8) BC_HN(8, T_, DT_);               \
                             else if (vb == 4) BC_HN(4, T_, DT_); else BC_HN(2, uint16_t, 0); } while (0)
    if (!act) BC_HNV(uint16_t, 0);
    else if (dtype == BC_F32) { if (vb == 16) BC_HN(16, uint32_t, 1); else if (vb == 8) BC_HN(8, uint32_t, 1); else BC_HN(4, uint32_t, 1); }
    else if (dtype == BC_F16) { if (vb == 16) BC_HN(16, uint16_t, 2); else if (vb == 8) BC_HN(8, uint16_t, 2); else if (vb == 4) BC_HN(4, uint16_t, 2); else BC_HN(2, uint16_t, 2); }
    else { if (vb == 16) BC_HN(16, uint16_t, 3); else if (vb == 8) BC_HN(8, uint16_t, 3); else if (vb == 4) BC_HN(4, uint16_t, 3); else BC_HN(2, uint16_t, 3); }
#undef BC_HNV
#undef BC_HN
    return launch_status();
}

// which decompositions cover a layer: out[0..n) = their indices, returns n (the engine times exactly these)
static int conv_candidates(int dtype, int stride, int ks, int n_exec, int Cin, int Cout, int bs_in, int *out, int max_out)
{
    if (dtype < BC_F32 || dtype > BC_BF16 || (stride != 1 && stride != 2) || !out || bs_in % stride) return BC_ERR_SHAPE;
    const int E = dtype == BC_F32 ? 4 : 2, bs = bs_in / stride;
    if (!(bs == 4 || bs % 8 == 0 || bs == 2) || bs > 248 / stride) return 0;
    if (ks == 1 && stride == 1 && bs <= 4) return 0;
    const bool direct_ok = bs != 2 || dtype == BC_F16;          // 2x2 tiles: compiled for fp16 and for the split form of fp32 only
    int n = 0;
    Conv2Plan plan;
    const int n_cfg = (int)(sizeof(CONV2_CFGS) / sizeof(CONV2_CFGS[0]));
    for (int c = 0; c < n_cfg && n < max_out && direct_ok; ++c)
        if (conv2_plan(CONV2_CFGS[c], E, stride, n_exec, Cin, Cout, bs, plan, ks)) out[n++] = c;
    // the same decompositions without the one-workgroup-per-CU LDS floor, where two workgroups fit a CU at all
    for (int c = 0; c < n_cfg && n < max_out && direct_ok; ++c)
        if (conv2_plan(CONV2_CFGS[c], E, stride, n_exec, Cin, Cout, bs, plan, ks) && plan.lds_bytes <= (size_t)78 * 1024 && plan.wgs > device_cu_count())
            out[n++] = c | 0x100;
    // the Winograd form (fp32, 3x3, stride 1, tiles of a multiple of 8 pixels)
    if (dtype == BC_F32 && stride == 1 && ks == 3) {
        WinoPlan wp;
        for (int c = 0; c < WINO_N && n < max_out; ++c)
            if (wino_plan(WINO_CFGS[c], n_exec, Cin, Cout, bs, wp)) {
                out[n++] = c | 0x200;
                // (no 0x100 variant: at 208-255 VGPRs per lane a second 8-wave workgroup cannot join the CU whatever the LDS size.
                //  Four-wave workgroups -- two independent ones per CU -- were tried and lost 5-20 %: profiles/r02 README)
            }
        // the wide-tile Winograd form (one wave per SIMD, conv3x3_wino32.inc)
        Wino32Plan wp32;
        for (int c = 0; c < WINO32_N && n < max_out; ++c)
            if (wino32_plan(WINO32_CFGS[c], n_exec, Cin, Cout, bs, wp32)) out[n++] = c | 0x400;
        // the F(4x4,3x3) form (conv3x3_wino4.inc)
        Wino4Plan wp4;
        for (int c = 0; c < WINO4_N && n < max_out; ++c)
            if (wino4_plan(WINO4_CFGS[c], n_exec, Cin, Cout, bs, wp4)) out[n++] = c | 0x1000;
        // ... and with its products on the 16-bit matrix pipe (split operands, conv3x3_wino4.inc SP)
        for (int c = 0; c < WINO4_N && n < max_out; ++c)
            if (wino4_plan(WINO4_CFGS[c], n_exec, Cin, Cout, bs, wp4)) out[n++] = c | 0x5000;
    }
    // the direct form on the 16-bit matrix pipe (fp32 tensors, operands split hi + lo; BC_F32S): with and without the LDS floor
    if (dtype == BC_F32) {
        for (int c = 0; c < n_cfg && n < max_out; ++c)
            if (conv2_plan(CONV2_CFGS[c], E, stride, n_exec, Cin, Cout, bs, plan, ks, 1, true)) out[n++] = c | 0x2000;
        for (int c = 0; c < n_cfg && n < max_out; ++c)
            if (conv2_plan(CONV2_CFGS[c], E, stride, n_exec, Cin, Cout, bs, plan, ks, 1, true) && plan.lds_bytes <= (size_t)78 * 1024 && plan.wgs > device_cu_count())
                out[n++] = c | 0x2100;
    }
    // the plain-GEMM form of a pointwise conv (fp32, stride 1; gemm1x1.inc): workgroup tiles 128x128, 128x64, 64x128, 64x64
    if (dtype == BC_F32 && stride == 1 && ks == 1 && Cin % 32 == 0) {
        for (int c = 0; c < 4 && n < max_out; ++c)
            if (Cout % (64 * (2 - (c & 1))) == 0) out[n++] = c | 0x800;
    }
    return n;
}

// host side of gemm1x1.inc: code 0x800 | c, c = 0..3 -> (TM, TN) = (2,2), (2,1), (1,2), (1,1)
template <int TM, int TN>
static int launch_gemm1x1(ProfScope &ps, void *out, const void *x, const void *wpk, const GemmGeom &g, const Prologue &pr, const EpilogueT &ep, hipStream_t st)
{
    constexpr size_t lds_bytes = (size_t)2 * (64 * TM * 9 + 2 * TN * 256) * 16;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_gemm1x1<TM, TN>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        attr_set = true;
    }
    const dim3 grid((g.M + 64 * TM - 1) / (64 * TM), g.N / (64 * TN));
    BC_LAUNCH(ps, (k_gemm1x1<TM, TN>), grid, dim3(256), lds_bytes, st, (float *)out, (const uint4 *)x, (const uint4 *)wpk, g, pr, ep);
    return launch_status();
}

BC_EXPORT int bc_conv3x3_candidates(int dtype, int stride, int n_exec, int Cin, int Cout, int bs_in, int *out, int max_out)
{
    return conv_candidates(dtype, stride, 3, n_exec, Cin, Cout, bs_in, out, max_out);
}

BC_EXPORT int bc_conv1x1_candidates(int dtype, int stride, int n_tiles, int Cin, int Cout, int bs_in, int *out, int max_out)
{
    return conv_candidates(dtype, stride, 1, n_tiles, Cin, Cout, bs_in, out, max_out);
}

// pointwise (1x1) conv of a channels-last tensor viewed as n_tiles tiles of bs x bs pixels (any view with n_tiles*bs*bs pixels does
// for stride 1; stride 2 needs the real tiles): the GEMM of the fused kernel with ONE tap -- no halo, no ring cache, no grid tables
BC_EXPORT int bc_conv1x1_nhwc(void *out, const void *features, const void *weights_packed, int n_tiles, int Cin, int Cout, int bs, int stride,
                              int dtype, const float *in_scale, const float *in_shift, int in_relu, const float *out_scale,
                              const float *out_shift, const void *out_add, int out_relu, void *stream)
{
    const DynArm arm = dyn_take();
    const UpsampleArm up = g_up_arm;                // (one shot: bc_conv_upsample_arm)
    g_up_arm = UpsampleArm{};
    if (dtype < BC_F32 || dtype > BC_BF16) return BC_ERR_ELEM;
    if (n_tiles < 0 || Cin <= 0 || Cout <= 0 || bs <= 0 || (stride != 1 && stride != 2) || bs % stride) return BC_ERR_SHAPE;
    const int bso = bs / stride;
    if (Cin % CV_CH != 0 || Cout % 64 != 0 || !(bso % 8 == 0 || ((bso == 4 || (bso == 2 && dtype != BC_BF16)) && stride == 2)) || bs > 248) return BC_ERR_SHAPE;
    if (n_tiles == 0) return BC_OK;
    if (!out || !features || !weights_packed) return BC_ERR_NULL;
    if ((uint64_t)n_tiles * bs * bs * (uint64_t)(Cin > Cout ? Cin : Cout) >= (1ull << 31)) return BC_ERR_RANGE;
    if (!aligned(out, 16) || !aligned(features, 16) || !aligned(weights_packed, 16) || !aligned(out_add, 8)) return BC_ERR_ALIGN;
    Prologue pr{in_scale, in_shift, in_relu};
    EpilogueT ept{out_scale, out_shift, out_add, out_relu};
    if (up.src) {
        // "+ bilinear(src)" in the epilogue: the direct one-tap form on 8x8 re-tiles of whole power-of-two tiles
        if (stride != 1 || bs != 8 || (g_tune.conv2_cfg >= 0 && (g_tune.conv2_cfg & 0x800))) return BC_ERR_SHAPE;
        if (!aligned(up.src, dtype == BC_F32 ? 16 : 8)) return BC_ERR_ALIGN;
        int lg = 0;
        while ((1 << lg) < up.out_bs) ++lg;
        if ((1 << lg) != up.out_bs || up.out_bs < 2 || ((long long)n_tiles * 64) % ((long long)up.out_bs * up.out_bs) != 0) return BC_ERR_SHAPE;
        ept.up_src = up.src; ept.up_bs = (uint32_t)up.src_bs; ept.up_obs_log2 = (uint32_t)lg;
        ept.up_rh = up.rh; ept.up_rw = up.rw; ept.up_align = up.align;
    }
    ProfScope ps(BC_OP_CONV3X3, 2.0 * n_tiles * bso * bso * (double)Cin * Cout);
    void *ring = const_cast<void *>(features);      // (unused by the one-tap form)
    hipStream_t st = (hipStream_t)stream;
    if (g_tune.conv2_cfg >= 0 && (g_tune.conv2_cfg & 0x800)) {
        const int c = g_tune.conv2_cfg & 3;
        if (dtype != BC_F32 || stride != 1 || Cout % (64 * (2 - (c & 1))) != 0) return BC_ERR_SHAPE;
        if (!aligned(out_add, 16)) return BC_ERR_ALIGN;
        GemmGeom g{(uint32_t)n_tiles * (uint32_t)bs * (uint32_t)bs, (uint32_t)Cin, (uint32_t)Cout, (uint32_t)Cin / 32};
        g.dyn = dyn_flat(arm, g.M);
        ps.add_aux(2.0 * g.M * (double)Cin * Cout);
        g_tune.conv_last_cfg = 0x800 | c;
        switch (c) {
        case 0: return launch_gemm1x1<2, 2>(ps, out, features, weights_packed, g, pr, ept, st);
        case 1: return launch_gemm1x1<2, 1>(ps, out, features, weights_packed, g, pr, ept, st);
        case 2: return launch_gemm1x1<1, 2>(ps, out, features, weights_packed, g, pr, ept, st);
        default: return launch_gemm1x1<1, 1>(ps, out, features, weights_packed, g, pr, ept, st);
        }
    }
    g_conv_dyn = dyn_flat(arm, (unsigned long long)n_tiles);      // (units = the launch's tiles: 8x8 re-tiles of the packed pixels, or the real tiles)
#define BC_C1(DT_)                                                                                                         \
    (stride == 1 ? launch_conv3x3_v2<DT_, 1, 1>(ps, out, features, ring, weights_packed, nullptr, nullptr, n_tiles, Cin, Cout, 1, 1, bso, pr, ept, st) \
                 : launch_conv3x3_v2<DT_, 2, 1>(ps, out, features, ring, weights_packed, nullptr, nullptr, n_tiles, Cin, Cout, 1, 1, bso, pr, ept, st))
    if (dtype == BC_F32) return BC_C1(BC_F32);
    if (dtype == BC_F16) return BC_C1(BC_F16);
    return BC_C1(BC_BF16);
#undef BC_C1
}

BC_EXPORT int bc_conv_upsample_arm(const void *src, int src_bs, int out_bs, int align_corners, float rh, float rw)
{
    if (src && (src_bs <= 0 || out_bs <= 0)) return BC_ERR_SHAPE;
    g_up_arm = UpsampleArm{src, src_bs, out_bs, align_corners, rh, rw};
    return BC_OK;
}

BC_EXPORT int bc_dyn_set(const void *n_exec_dev, int ceiling)
{
    if (n_exec_dev && (ceiling <= 0 || !aligned(n_exec_dev, 4))) return BC_ERR_SHAPE;
    g_dyn_arm.ptr = static_cast<const int32_t *>(n_exec_dev);
    g_dyn_arm.ceiling = n_exec_dev ? ceiling : 0;
    return BC_OK;
}

BC_EXPORT int bc_tune_set(const char *key, int value)
{
    if (!key) return BC_ERR_NULL;
    if (!strcmp(key, "conv_impl")) g_tune.conv_impl = value;
    else if (!strcmp(key, "conv2_cfg")) g_tune.conv2_cfg = value;
    else if (!strcmp(key, "stem_split")) g_tune.stem_split = value;
    else if (!strcmp(key, "conv2_min_lds")) g_tune.conv2_min_lds = value;
    else if (!strcmp(key, "xcd_remap")) g_tune.xcd_remap = value;
    else if (!strcmp(key, "head_split")) g_tune.head_split = value;
    else if (!strcmp(key, "stem_min_lds")) g_tune.stem_min_lds = value;
    else return BC_ERR_SHAPE;
    return BC_OK;
}

BC_EXPORT int bc_tune_set_ptr(const char *key, void *ptr)
{
    if (!key) return BC_ERR_NULL;
    if (!strcmp(key, "conv_stamps")) g_tune.conv_stamps = (unsigned long long *)ptr;
    else return BC_ERR_SHAPE;
    return BC_OK;
}

BC_EXPORT int bc_tune_get(const char *key, int *value)
{
    if (!key || !value) return BC_ERR_NULL;
    if (!strcmp(key, "conv_impl")) *value = g_tune.conv_impl;
    else if (!strcmp(key, "conv2_cfg")) *value = g_tune.conv2_cfg;
    else if (!strcmp(key, "stem_split")) *value = g_tune.stem_split;
    else if (!strcmp(key, "conv2_min_lds")) *value = g_tune.conv2_min_lds;
    else if (!strcmp(key, "xcd_remap")) *value = g_tune.xcd_remap;
    else if (!strcmp(key, "head_split")) *value = g_tune.head_split;
    else if (!strcmp(key, "conv_last_cfg")) *value = g_tune.conv_last_cfg;
    else return BC_ERR_SHAPE;
    return BC_OK;
}

BC_EXPORT int bc_conv3x3_ring_nhwc(void *out, const void *features, void *ring, const void *weights_packed,
                                   const int32_t *grid_idx, const int32_t *mapping_exec, int n_exec, int N, int Cin, int Cout,
                                   int GH, int GW, int bs, int dtype, const float *in_scale, const float *in_shift, int in_relu,
                                   const float *out_scale, const float *out_shift, const void *out_add, int out_relu, void *stream)
{
    const DynArm arm = dyn_take();
    if (dtype < BC_F32 || dtype > BC_BF16) return BC_ERR_ELEM;
    if (n_exec < 0 || N <= 0 || Cin <= 0 || Cout <= 0 || GH <= 0 || GW <= 0 || bs <= 0) return BC_ERR_SHAPE;
    if (Cin % CV_CH != 0 || Cout % 64 != 0) return BC_ERR_SHAPE;
    if (!(bs == 4 || bs % 8 == 0 || (bs == 2 && dtype != BC_BF16)) || bs > 248) return BC_ERR_SHAPE;      // (2x2 tiles: fp16, and fp32 in the split form only)
    if (n_exec == 0) return BC_OK;
    if (!out || !features || !ring || !weights_packed || !grid_idx || !mapping_exec) return BC_ERR_NULL;
    if ((uint64_t)n_exec * bs * bs * (uint64_t)(Cin > Cout ? Cin : Cout) >= (1ull << 31) ||
        (uint64_t)N * GH * GW * 4 * bs * Cin >= (1ull << 31)) return BC_ERR_RANGE;
    if (!aligned(out, 16) || !aligned(features, 16) || !aligned(ring, 16) || !aligned(weights_packed, 16) || !aligned(out_add, 2))
        return BC_ERR_ALIGN;
    Prologue pr{in_scale, in_shift, in_relu};
    ProfScope ps(BC_OP_CONV3X3, 2.0 * n_exec * bs * bs * 9.0 * Cin * Cout);   // FLOPs, not bytes: the op is MFMA-bound
    // conv_impl = 1 selects the first-generation kernel (fp32 only; A/B runs); default: the CU-balanced kernel (conv3x3_v2.inc)
    DynCount dyn;
    if (!dyn_tiles(arm, n_exec, 1, dyn)) return BC_ERR_SHAPE;
    if (g_tune.conv_impl != 1 || dtype != BC_F32) {
        EpilogueT ept{out_scale, out_shift, out_add, out_relu};
        int rc;
        g_conv_dyn = dyn;
        if (dtype == BC_F32)
            rc = launch_conv3x3_v2<BC_F32, 1>(ps, out, features, ring, weights_packed, grid_idx, mapping_exec, n_exec, Cin, Cout, GH, GW, bs, pr, ept, (hipStream_t)stream);
        else if (dtype == BC_F16)
            rc = launch_conv3x3_v2<BC_F16, 1>(ps, out, features, ring, weights_packed, grid_idx, mapping_exec, n_exec, Cin, Cout, GH, GW, bs, pr, ept, (hipStream_t)stream);
        else
            rc = launch_conv3x3_v2<BC_BF16, 1>(ps, out, features, ring, weights_packed, grid_idx, mapping_exec, n_exec, Cin, Cout, GH, GW, bs, pr, ept, (hipStream_t)stream);
        if (rc != BC_ERR_SHAPE || dtype != BC_F32) return rc;   // fp32 shapes the balanced kernel does not cover fall through
    }
    if (dyn.ptr) return BC_ERR_SHAPE;       // (the first-generation kernel has no device-side count)
    g_tune.conv_last_cfg = -1;
    ps.add_aux(2.0 * n_exec * bs * bs * 9.0 * Cin * Cout);
    Epilogue ep{out_scale, out_shift, (const float *)out_add, out_relu};
    // workgroup = 32*WM pixels x 64 output channels.  64-pixel items halve the weight traffic and the halo overhead; 32-pixel
    // items balance better over the 256 CUs when there are few of them (measured, profiles/r01/kbench_conv_*.txt).
    // BC_CONV_WM=1|2 overrides for A/B runs.
    static const int wm_env = [] { const char *e = getenv("BC_CONV_WM"); return e ? atoi(e) : 0; }();
    const uint64_t items64 = (uint64_t)n_exec * bs * bs / 64 * (Cout / 64);
    const int WM_ = wm_env == 1 || wm_env == 2 ? wm_env : ((bs >= 16 && items64 < 768) ? 1 : 2);
    ConvGeom g;
    g.Cin = Cin; g.Cout = Cout; g.bs = bs; g.GH = GH; g.GW = GW; g.n_exec = n_exec;
    g.pw = bs < 8 ? 4 : 8;
    const uint32_t ph = g.pw == 8 ? 4 * WM_ : 4;
    g.patches_x = bs / g.pw;
    g.patches_per_tile = g.patches_x * (bs / ph);
    g.tiles_per_wg = 32 * WM_ / (g.pw * ph);
    g.PW2 = g.pw + 2; g.PP = g.PW2 * (ph + 2);
    g.n_vec = g.tiles_per_wg * g.PP * CV_VPP;
    g.cin_chunks = Cin / CV_CH; g.CG = Cin / 8; g.NB = Cout / 32;
    const unsigned gx = g.pw == 8 ? (unsigned)n_exec * g.patches_per_tile : ((unsigned)n_exec + g.tiles_per_wg - 1) / g.tiles_per_wg;
    const dim3 grid(gx, (unsigned)Cout / 64);
    const size_t lds_bytes = 2 * (size_t)g.tiles_per_wg * g.PP * CV_CHP * sizeof(float);   // two images (double buffer)
#define BC_CV(PW_, WMT_)                                                                                                \
    BC_LAUNCH(ps, (k_conv3x3_f32<PW_, WMT_>), grid, dim3(128 * WMT_), lds_bytes, (hipStream_t)stream, (float *)out,     \
              (const float *)features, (long long)((const float *)ring - (const float *)features), (float *)ring,     \
              (const float4 *)weights_packed, grid_idx, mapping_exec, g, pr, ep)
    if (g.pw == 8) { if (WM_ == 2) BC_CV(8, 2); else BC_CV(8, 1); }
    else { if (WM_ == 2) BC_CV(4, 2); else BC_CV(4, 1); }
#undef BC_CV
    return launch_status();
}

BC_EXPORT int bc_conv3x3_dil_ring_nhwc(void *out, const void *features, void *ring, const void *weights_packed,
                                       const int32_t *grid_idx, const int32_t *mapping_exec, int n_exec, int N, int Cin, int Cout,
                                       int GH, int GW, int bs, int dilation, int dtype, const float *in_scale, const float *in_shift, int in_relu,
                                       const float *out_scale, const float *out_shift, const void *out_add, int out_relu, void *stream)
{
    if (dilation == 1)
        return bc_conv3x3_ring_nhwc(out, features, ring, weights_packed, grid_idx, mapping_exec, n_exec, N, Cin, Cout, GH, GW, bs, dtype,
                                    in_scale, in_shift, in_relu, out_scale, out_shift, out_add, out_relu, stream);
    const DynArm arm = dyn_take();
    if (dtype < BC_F32 || dtype > BC_BF16) return BC_ERR_ELEM;
    if (dilation != 2 || n_exec < 0 || N <= 0 || Cin <= 0 || Cout <= 0 || GH <= 0 || GW <= 0 || bs <= 0) return BC_ERR_SHAPE;
    if (Cin % CV_CH != 0 || Cout % 32 != 0 || bs % 8 != 0 || bs > 248) return BC_ERR_SHAPE;
    if (n_exec == 0) return BC_OK;
    if (!out || !features || !ring || !weights_packed || !grid_idx || !mapping_exec) return BC_ERR_NULL;
    if ((uint64_t)n_exec * bs * bs * (uint64_t)(Cin > Cout ? Cin : Cout) >= (1ull << 31) ||
        (uint64_t)N * GH * GW * 8 * bs * Cin >= (1ull << 31)) return BC_ERR_RANGE;
    if (!aligned(out, 16) || !aligned(features, 16) || !aligned(ring, 16) || !aligned(weights_packed, 16) || !aligned(out_add, 2))
        return BC_ERR_ALIGN;
    Prologue pr{in_scale, in_shift, in_relu};
    EpilogueT ept{out_scale, out_shift, out_add, out_relu};
    const double flops = 2.0 * n_exec * bs * bs * 9.0 * Cin * Cout;
    ProfScope ps(BC_OP_CONV3X3, flops);
    // code | 0x2000 (fp32 only): the products on the 16-bit matrix pipe, operands split hi + lo (conv3x3_v2.inc BC_F32S; its weight stream sits behind the others)
    const bool split = dtype == BC_F32 && g_tune.conv2_cfg >= 0 && (g_tune.conv2_cfg & 0x2000);
    ps.add_aux(split ? flops * 3.0 / 16.0 : flops);
    // (this part's dispatcher reads the dtype from the `stride` field: the stride of a dilated launch is always 1)
    ConvV2Args a{out, features, ring, split ? static_cast<const void *>(static_cast<const float *>(weights_packed) + (size_t)77 * Cin * Cout) : weights_packed,
                 grid_idx, mapping_exec, n_exec, Cin, Cout, GH, GW, bs, split ? BC_F32S : dtype, pr, ept, (hipStream_t)stream,
                 ps.on ? 1 : 0, ps.rec.a, ps.rec.b, split ? (g_tune.conv2_cfg & ~0x2000) : g_tune.conv2_cfg, g_tune.conv2_min_lds, g_tune.conv_stamps, -2, g_tune.xcd_remap, 2};
    if (!dyn_tiles(arm, n_exec, 1, a.dyn)) return BC_ERR_SHAPE;
#if defined(BC_MONO)
    const int rc = split ? conv_v2_dil_run<BC_F32S>(a) : (dtype == BC_F32 ? conv_v2_dil_run<BC_F32>(a) : (dtype == BC_F16 ? conv_v2_dil_run<BC_F16>(a) : conv_v2_dil_run<BC_BF16>(a)));
#else
    const int rc = bc_part_conv_v2_dil(&a);
#endif
    if (a.chosen >= 0) g_tune.conv_last_cfg = split ? (a.chosen | 0x2000) : a.chosen;
    return rc;
}

BC_EXPORT int bc_conv3x3_dil_candidates(int dtype, int dilation, int n_exec, int Cin, int Cout, int bs, int *out, int max_out)
{
    if (dilation == 1) return bc_conv3x3_candidates(dtype, 1, n_exec, Cin, Cout, bs, out, max_out);
    if (dtype < BC_F32 || dtype > BC_BF16 || dilation != 2 || !out) return BC_ERR_SHAPE;
    if (bs % 8 != 0 || bs > 248) return 0;
    const int E = dtype == BC_F32 ? 4 : 2;
    int n = 0;
    Conv2Plan plan;
    for (int i = 0; i < CONV2_DIL_N && n < max_out; ++i) {
        const Conv2Cfg &k = CONV2_CFGS[CONV2_DIL_CFGS[i]];
        if (k.RM * k.RN < 4 && conv2_plan(k, E, 1, n_exec, Cin, Cout, bs, plan, 3, 2)) out[n++] = CONV2_DIL_CFGS[i];
    }
    if (dtype == BC_F32)        // the same decompositions (and the 2x2 wave tiles) on the 16-bit matrix pipe, operands split hi + lo
        for (int i = 0; i < CONV2_DIL_N && n < max_out; ++i)
            if (conv2_plan(CONV2_CFGS[CONV2_DIL_CFGS[i]], E, 1, n_exec, Cin, Cout, bs, plan, 3, 2, true)) out[n++] = CONV2_DIL_CFGS[i] | 0x2000;
    return n;
}

BC_EXPORT int bc_stem7x7s2_nhwc(void *out, const void *frame_state, const void *weights_packed, const int32_t *mapping_exec, int n_exec,
                                int N, int H, int W, int bs, int Cout, int dtype, const float *out_scale, const float *out_shift,
                                const void *out_add, int out_relu, void *stream)
{
    const DynArm arm = dyn_take();
    if (dtype < BC_F32 || dtype > BC_BF16) return BC_ERR_ELEM;
    if (n_exec < 0 || N <= 0 || H <= 0 || W <= 0 || bs <= 0 || H % bs || W % bs) return BC_ERR_SHAPE;
    if (Cout != 64 || bs % 2 || (bs / 2) % ST_OW != 0 || (bs / 2) % ST_OH != 0) return BC_ERR_SHAPE;
    if (n_exec == 0) return BC_OK;
    if (!out || !frame_state || !weights_packed || !mapping_exec) return BC_ERR_NULL;
    if ((uint64_t)N * 3 * H * W >= (1ull << 31) || (uint64_t)n_exec * (bs / 2) * (bs / 2) * 64 >= (1ull << 31)) return BC_ERR_RANGE;
    if (!aligned(out, 4) || !aligned(frame_state, 4) || !aligned(weights_packed, 16)) return BC_ERR_ALIGN;
    StemGeom g;
    g.H = H; g.W = W; g.bs = bs; g.GH = H / bs; g.GW = W / bs; g.n_exec = n_exec;
    g.patches_x = (bs / 2) / ST_OW;
    g.patches_per_tile = g.patches_x * ((bs / 2) / ST_OH);
    if (!dyn_tiles(arm, n_exec, 1, g.dyn)) return BC_ERR_SHAPE;
    EpilogueT ep{out_scale, out_shift, out_add, out_relu};
    const int E = dtype == BC_F32 ? 4 : 2;
    size_t lds = ((size_t)3 * ST_WH * ST_WS + 64) * E + 16;
    // one workgroup per CU at a time (LDS request > half a CU's): the 67 MB output of a C2 launch is then written by one round
    // of workgroups while the next round computes, instead of by all of them at the end (stem_min_lds = 0: natural occupancy)
    // (measured, tools/kbench_stem.py: fp32 67 -> 61 us; 16-bit launches are too short to gain: 19.8 -> 23.3 us, so fp32 only)
    // (the split form is as short as the 16-bit launches: natural occupancy 31.6 us against 36.4 with the floor)
    if (dtype == BC_F32 && !g_tune.stem_split && lds < (size_t)g_tune.stem_min_lds) lds = g_tune.stem_min_lds;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_stem7x7<BC_F32>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 2048);
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_stem7x7<BC_F16>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 2048);
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_stem7x7<BC_BF16>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 2048);
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_stem7x7<BC_F32S>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 2048);
        attr_set = true;
    }
    ProfScope ps(BC_OP_CONV3X3, 2.0 * n_exec * (bs / 2) * (bs / 2) * 147.0 * 64);
    // issued: 7 row taps x a K segment of 24 (fp32, 11 of its 12 k pairs: the all-padding pair is skipped) / 32 (16-bit) instead of 21
    ps.add_aux(2.0 * n_exec * (bs / 2) * (bs / 2) * (dtype == BC_F32 ? 154.0 : 224.0) * 64);
    const dim3 grid((unsigned)n_exec * g.patches_per_tile);
    if (dtype == BC_F32 && g_tune.stem_split)
        // fp32 frame on the 16-bit matrix pipe (operands split hi + lo, stem7x7.inc BC_F32S): the hi / lo 16-bit weight streams follow the fp32 one
        BC_LAUNCH(ps, (k_stem7x7<BC_F32S>), grid, dim3(512), lds, (hipStream_t)stream, (float *)out, (const float *)frame_state,
                  (const uint4 *)weights_packed + 2 * 21 * 64, mapping_exec, g, ep);
    else if (dtype == BC_F32)
        BC_LAUNCH(ps, (k_stem7x7<BC_F32>), grid, dim3(512), lds, (hipStream_t)stream, (float *)out, (const float *)frame_state,
                  (const uint4 *)weights_packed, mapping_exec, g, ep);
    else if (dtype == BC_F16)
        BC_LAUNCH(ps, (k_stem7x7<BC_F16>), grid, dim3(512), lds, (hipStream_t)stream, (__half *)out, (const __half *)frame_state,
                  (const uint4 *)weights_packed, mapping_exec, g, ep);
    else
        BC_LAUNCH(ps, (k_stem7x7<BC_BF16>), grid, dim3(512), lds, (hipStream_t)stream, (hip_bfloat16 *)out, (const hip_bfloat16 *)frame_state,
                  (const uint4 *)weights_packed, mapping_exec, g, ep);
    return launch_status();
}

BC_EXPORT int bc_conv3x3s2_ring_nhwc(void *out, const void *features, void *ring, const void *weights_packed,
                                     const int32_t *grid_idx, const int32_t *mapping_exec, int n_exec, int N, int Cin, int Cout,
                                     int GH, int GW, int bs, int dtype, const float *in_scale, const float *in_shift, int in_relu,
                                     const float *out_scale, const float *out_shift, const void *out_add, int out_relu, void *stream)
{
    const DynArm arm = dyn_take();
    if (dtype < BC_F32 || dtype > BC_BF16) return BC_ERR_ELEM;
    if (n_exec < 0 || N <= 0 || Cin <= 0 || Cout <= 0 || GH <= 0 || GW <= 0 || bs <= 0 || bs % 2) return BC_ERR_SHAPE;
    const int bso = bs / 2;
    if (Cin % CV_CH != 0 || Cout % 64 != 0 || !(bso == 4 || bso % 8 == 0 || (bso == 2 && dtype != BC_BF16)) || bs > 248) return BC_ERR_SHAPE;
    if (n_exec == 0) return BC_OK;
    if (!out || !features || !ring || !weights_packed || !grid_idx || !mapping_exec) return BC_ERR_NULL;
    if ((uint64_t)n_exec * bs * bs * (uint64_t)(Cin > Cout ? Cin : Cout) >= (1ull << 31) ||
        (uint64_t)N * GH * GW * 4 * bs * Cin >= (1ull << 31)) return BC_ERR_RANGE;
    if (!aligned(out, 16) || !aligned(features, 16) || !aligned(ring, 16) || !aligned(weights_packed, 16) || !aligned(out_add, 2))
        return BC_ERR_ALIGN;
    Prologue pr{in_scale, in_shift, in_relu};
    EpilogueT ept{out_scale, out_shift, out_add, out_relu};
    ProfScope ps(BC_OP_CONV3X3, 2.0 * n_exec * bso * bso * 9.0 * Cin * Cout);
    if (!dyn_tiles(arm, n_exec, 1, g_conv_dyn)) return BC_ERR_SHAPE;
    if (dtype == BC_F32)
        return launch_conv3x3_v2<BC_F32, 2>(ps, out, features, ring, weights_packed, grid_idx, mapping_exec, n_exec, Cin, Cout, GH, GW, bso, pr, ept, (hipStream_t)stream);
    if (dtype == BC_F16)
        return launch_conv3x3_v2<BC_F16, 2>(ps, out, features, ring, weights_packed, grid_idx, mapping_exec, n_exec, Cin, Cout, GH, GW, bso, pr, ept, (hipStream_t)stream);
    return launch_conv3x3_v2<BC_BF16, 2>(ps, out, features, ring, weights_packed, grid_idx, mapping_exec, n_exec, Cin, Cout, GH, GW, bso, pr, ept, (hipStream_t)stream);
}

BC_EXPORT int bc_pad_ring_add_nhwc(void *out, void *act_out, const void *features, const void *add, void *ring,
                                   const int32_t *grid_idx, const int32_t *mapping_exec, int n_exec, int N, int C, int GH, int GW,
                                   int bs, int pad, int dtype, const float *scale, const float *shift, int relu, void *stream)
{
    const DynArm arm = dyn_take();
    if (dtype < BC_F32 || dtype > BC_BF16) return BC_ERR_ELEM;
    const int E = dtype == BC_F32 ? 4 : 2;
    if (n_exec < 0 || N <= 0 || C <= 0 || GH <= 0 || GW <= 0 || bs <= 0 || pad < 1 || pad > bs) return BC_ERR_SHAPE;
    if (((size_t)C * E) % 16 != 0) return BC_ERR_SHAPE;
    if (n_exec == 0) return BC_OK;
    if (!out || !act_out || !features || !add || !ring || !grid_idx || !mapping_exec) return BC_ERR_NULL;
    const uint64_t bsp = (uint64_t)bs + 2 * pad;
    if ((uint64_t)n_exec * C * bsp * bsp >= (1ull << 31) || (uint64_t)N * GH * GW * C * bs * bs >= (1ull << 31)) return BC_ERR_RANGE;
    if (!aligned(out, 16) || !aligned(act_out, 16) || !aligned(features, 16) || !aligned(add, 16) || !aligned(ring, 16)) return BC_ERR_ALIGN;
    HaloNhwcGeom g;
    const uint32_t K = (uint32_t)((size_t)C * E / 16);
    g.K = make_fd(K); g.BSP = make_fd((uint32_t)bsp); g.GW = make_fd(GW); g.GH = make_fd(GH);
    g.C = C; g.bs = bs; g.pad = pad; g.n_total = (uint32_t)N * GH * GW;
    g.per_tile = (uint32_t)(bsp * bsp * K);
    g.epv = 16 / E;
    if (!dyn_tiles(arm, n_exec, 1, g.dyn)) return BC_ERR_SHAPE;
    const dim3 grid((g.per_tile + WG * UNROLL - 1) / (WG * UNROLL), (unsigned)n_exec);
    const long long ring_delta = ((const char *)ring - (const char *)features) / 16;
    const long long add_delta = ((const char *)add - (const char *)features) / 16;
    Prologue pr{scale, shift, relu};
    // bytes: raw + identity tiles in, padded batch + activated tiles out
    ProfScope ps(BC_OP_PAD_RING, halo_bytes(n_exec, C, bs, pad, E) + 2.0 * n_exec * C * bs * bs * E);
#define BC_HA(T_, DT_)                                                                                               \
    BC_LAUNCH(ps, (k_halo_add_nhwc<16, T_, DT_>), grid, dim3(WG), 0, (hipStream_t)stream, (VecOf<16>::type *)out,      \
              (VecOf<16>::type *)act_out, (const VecOf<16>::type *)features, add_delta, ring_delta, (VecOf<16>::type *)ring, \
              grid_idx, mapping_exec, g, pr)
    if (dtype == BC_F32) BC_HA(uint32_t, 1); else if (dtype == BC_F16) BC_HA(uint16_t, 2); else BC_HA(uint16_t, 3);
#undef BC_HA
    return launch_status();
}

BC_EXPORT int bc_maxpool3x3s2_ring_nhwc(void *out, const void *features, void *ring, const int32_t *grid_idx,
                                        const int32_t *mapping_exec, int n_exec, int N, int C, int GH, int GW, int bs, int dtype,
                                        const float *scale, const float *shift, int relu, void *stream)
{
    const DynArm arm = dyn_take();
    if (dtype < BC_F32 || dtype > BC_BF16) return BC_ERR_ELEM;
    const int E = dtype == BC_F32 ? 4 : 2;
    if (n_exec < 0 || N <= 0 || C <= 0 || GH <= 0 || GW <= 0 || bs < 2 || (bs & 1)) return BC_ERR_SHAPE;
    if (((size_t)C * E) % 16 != 0) return BC_ERR_SHAPE;
    if (n_exec == 0) return BC_OK;
    if (!out || !features || !ring || !grid_idx || !mapping_exec) return BC_ERR_NULL;
    if ((uint64_t)n_exec * C * bs * bs >= (1ull << 31) || (uint64_t)N * GH * GW * C * 4 * bs >= (1ull << 31)) return BC_ERR_RANGE;
    if (!aligned(out, 16) || !aligned(features, 16) || !aligned(ring, 16)) return BC_ERR_ALIGN;
    PoolGeom g;
    const uint32_t K = (uint32_t)((size_t)C * E / 16), OB = bs / 2;
    g.K = make_fd(K); g.OB = make_fd(OB); g.GW = make_fd(GW); g.GH = make_fd(GH);
    g.bs = bs; g.n_total = (uint32_t)N * GH * GW; g.per_tile = OB * OB * K;
    g.OBP = make_fd(OB / 8);
    if (OB % 8 != 0) g.OBP.d = 0;
    if (!dyn_tiles(arm, n_exec, 1, g.dyn)) return BC_ERR_SHAPE;
    const dim3 grid((g.per_tile + WG * MP_U - 1) / (WG * MP_U), (unsigned)n_exec);
    const long long delta = ((const char *)ring - (const char *)features) / 16;
    const bool act = scale || shift || relu;
    Prologue pr{scale, shift, relu};
    // algorithmic bytes: tile + top/left halo in, pooled tile out (+ ring refresh)
    ProfScope ps(BC_OP_PAD_RING, (double)n_exec * C * E * ((double)(bs + 1) * (bs + 1) + (double)OB * OB + 4.0 * bs));
#define BC_MP(T_, VE_, DT_)                                                                                           \
    BC_LAUNCH(ps, (k_maxpool3x3s2_nhwc<T_, VE_, DT_>), grid, dim3(WG), 0, (hipStream_t)stream, (VecOf<16>::type *)out, \
              (const VecOf<16>::type *)features, delta, (VecOf<16>::type *)ring, grid_idx, mapping_exec, g, pr)
    if (dtype == BC_F32) { if (act) BC_MP(float, 4, 1); else BC_MP(float, 4, 0); }
    else if (dtype == BC_F16) { if (act) BC_MP(__half, 8, 1); else BC_MP(__half, 8, 0); }
    else { if (act) BC_MP(hip_bfloat16, 8, 1); else BC_MP(hip_bfloat16, 8, 0); }
#undef BC_MP
    return launch_status();
}

BC_EXPORT int bc_affine_act_nhwc(void *out, const void *in, const void *add, const float *scale, const float *shift, int relu,
                                 long long pixels, int C, int dtype, void *stream)
{
    const DynArm arm = dyn_take();
    if (dtype < BC_F32 || dtype > BC_BF16) return BC_ERR_ELEM;
    if (pixels < 0 || C <= 0) return BC_ERR_SHAPE;
    if (pixels == 0) return BC_OK;
    if (!out || !in) return BC_ERR_NULL;
    const int E = dtype == BC_F32 ? 4 : 2;
    if ((uint64_t)pixels * C >= (1ull << 31)) return BC_ERR_RANGE;
    hipStream_t st = (hipStream_t)stream;
    int q = 16 / E;
    while (q > 1 && ((C % q) != 0 || !aligned(out, q * E) || !aligned(in, q * E) || !aligned(add, q * E))) q >>= 1;
    const FastDiv Cq = make_fd((uint32_t)(C / q));
    const uint32_t total = (uint32_t)((uint64_t)pixels * (C / q));
    const int grid = grid_exact(total, 1);
    const DynCount dyn = dyn_flat(arm, total);
    ProfScope ps(BC_OP_AFFINE, (add ? 3.0 : 2.0) * pixels * C * E);
#define BC_AN(T_, Q_) BC_LAUNCH(ps, (k_affine_act_nhwc<T_, Q_>), dim3(grid), dim3(WG), 0, st, (T_ *)out, (const T_ *)in, (const T_ *)add, scale, shift, relu, Cq, total, dyn)
#define BC_ANQ(T_, QMAX_) do { if (q == QMAX_) BC_AN(T_, QMAX_); else if (q == QMAX_ / 2) BC_AN(T_, QMAX_ / 2);     \
                               else if (QMAX_ >= 8 && q == 2) BC_AN(T_, 2); else BC_AN(T_, 1); } while (0)
    if (dtype == BC_F32) BC_ANQ(float, 4);
    else if (dtype == BC_F16) BC_ANQ(__half, 8);
    else BC_ANQ(hip_bfloat16, 8);
#undef BC_ANQ
#undef BC_AN
    return launch_status();
}

static int launch_group_stats(const void *features, long long n_pix, int C, int groups, int dtype, float eps, const float *gamma, const float *beta,
                              float *scale, float *shift, float *workspace, long long workspace_floats, const BnExtra &bn, hipStream_t st)
{
    if (dtype < BC_F32 || dtype > BC_BF16) return BC_ERR_ELEM;
    if (n_pix <= 0 || C <= 0 || groups <= 0 || C % groups != 0 || n_pix >= (1ll << 31)) return BC_ERR_SHAPE;
    const int E = dtype == BC_F32 ? 4 : 2, VE = 16 / E;
    if (C % VE != 0) return BC_ERR_SHAPE;
    const uint32_t K = (uint32_t)(C / VE);
    if (K > WG || WG % K != 0) return BC_ERR_SHAPE;
    if (!features || !scale || !shift || !workspace) return BC_ERR_NULL;
    if (!aligned(features, 16)) return BC_ERR_ALIGN;
    const uint32_t rpw = WG / K;
    // ~2 workgroups per CU, each at least 8 steps of its pixel rows
    uint32_t n_wg = 512;
    while (n_wg > 1 && (uint64_t)n_wg * rpw * 8 > (uint64_t)n_pix) n_wg >>= 1;
    const uint32_t rows_per_wg = (uint32_t)((n_pix + n_wg - 1) / n_wg);
    n_wg = (uint32_t)((n_pix + rows_per_wg - 1) / rows_per_wg);
    if ((long long)n_wg * C * 2 > workspace_floats) return BC_ERR_RANGE;
    ProfScope ps(BC_OP_AFFINE, (double)n_pix * C * E);
    if (dtype == BC_F32)
        BC_LAUNCH(ps, (k_group_stats<float, 4>), dim3(n_wg), dim3(WG), 0, st, (const VecOf<16>::type *)features, (uint32_t)n_pix, K, rows_per_wg, workspace);
    else if (dtype == BC_F16)
        BC_LAUNCH(ps, (k_group_stats<__half, 8>), dim3(n_wg), dim3(WG), 0, st, (const VecOf<16>::type *)features, (uint32_t)n_pix, K, rows_per_wg, workspace);
    else
        BC_LAUNCH(ps, (k_group_stats<hip_bfloat16, 8>), dim3(n_wg), dim3(WG), 0, st, (const VecOf<16>::type *)features, (uint32_t)n_pix, K, rows_per_wg, workspace);
    hipLaunchKernelGGL(k_group_finalize, dim3(groups), dim3(WG), 0, st, (const float *)workspace, n_wg, (uint32_t)C, (uint32_t)(C / groups),
                       (double)n_pix * (C / groups), eps, gamma, beta, scale, shift, bn);
    return launch_status();
}

/* L2 normalisation over the channels of every pixel, scaled per channel, written into channels [c_off, c_off + C) of a wider
 * channels-last tensor (see k_l2norm_cat) */
BC_EXPORT int bc_l2norm_cat_nhwc(void *out, const void *x, const float *weight, long long n_pix, int C, int C_total, int c_off, float eps, int dtype,
                                 void *stream)
{
    if (dtype < BC_F32 || dtype > BC_BF16) return BC_ERR_ELEM;
    const int E = dtype == BC_F32 ? 4 : 2, epv = 16 / E;
    if (n_pix < 0 || C <= 0 || C_total < C || c_off < 0 || c_off + C > C_total || C % epv || c_off % epv || C_total % epv || C / epv > 256) return BC_ERR_SHAPE;
    if (n_pix == 0) return BC_OK;
    if (!out || !x || !weight) return BC_ERR_NULL;
    if ((uint64_t)n_pix * (uint64_t)C_total >= (1ull << 40)) return BC_ERR_RANGE;
    if (!aligned(out, 16) || !aligned(x, 16)) return BC_ERR_ALIGN;
    hipStream_t st = (hipStream_t)stream;
    long long wgs = (n_pix + 3) / 4;
    if (wgs > 256 * 8) wgs = 256 * 8;
    ProfScope ps(BC_OP_AFFINE, 2.0 * n_pix * C * E);
    const int nv = (C / epv + 63) / 64;
#define BC_L2(T_, NV_) BC_LAUNCH(ps, (k_l2norm_cat<T_, NV_>), dim3((unsigned)wgs), dim3(256), 0, st, (T_ *)out, (const T_ *)x, weight, n_pix, (uint32_t)C, \
                                 (uint32_t)C_total, (uint32_t)c_off, eps)
#define BC_L2T(T_) do { if (nv == 1) BC_L2(T_, 1); else if (nv == 2) BC_L2(T_, 2); else BC_L2(T_, 4); } while (0)
    if (dtype == BC_F32) BC_L2T(float);
    else if (dtype == BC_F16) BC_L2T(__half);
    else BC_L2T(hip_bfloat16);
#undef BC_L2T
#undef BC_L2
    return launch_status();
}

/* transposed conv (k4 s4 p0 or k4 s2 p1, per image = per packed tile, no halo) + bias + L2Norm + concat from the 16-tap patches
 * t (n_img, h, w, 16 C) a pointwise conv produced (see k_l2norm_cat_deconv): out (n_img, stride h, stride w, C_total)[..., c_off : c_off + C] */
BC_EXPORT int bc_l2norm_cat_deconv_nhwc(void *out, const void *t, const float *bias, const float *weight, int n_img, int h, int w, int C, int C_total,
                                        int c_off, int stride, float eps, int dtype, void *stream)
{
    if (dtype < BC_F32 || dtype > BC_BF16) return BC_ERR_ELEM;
    const int E = dtype == BC_F32 ? 4 : 2, epv = 16 / E;
    if (n_img < 0 || h <= 0 || w <= 0 || C <= 0 || C_total < C || c_off < 0 || c_off + C > C_total || C % epv || c_off % epv || C_total % epv ||
        C / epv > 64 || !(stride == 4 || stride == 2))
        return BC_ERR_SHAPE;
    if (n_img == 0) return BC_OK;
    if (!out || !t || !weight) return BC_ERR_NULL;
    const long long n_pix = (long long)n_img * h * w * stride * stride;
    if ((uint64_t)n_pix * (uint64_t)C_total >= (1ull << 40) || (uint64_t)n_img * h * w * 16 * C >= (1ull << 40)) return BC_ERR_RANGE;
    if (!aligned(out, 16) || !aligned(t, 16)) return BC_ERR_ALIGN;
    hipStream_t st = (hipStream_t)stream;
    long long wgs = (n_pix + 3) / 4;
    if (wgs > 256 * 16) wgs = 256 * 16;
    ProfScope ps(BC_OP_AFFINE, (double)n_pix * C * E * (stride == 4 ? 2.0 : 5.0));
#define BC_LD(T_)                                                                                                                                   \
    do {                                                                                                                                             \
        if (stride == 4) BC_LAUNCH(ps, (k_l2norm_cat_deconv<T_, 4>), dim3((unsigned)wgs), dim3(256), 0, st, (T_ *)out, (const T_ *)t, bias, weight, \
                                   (uint32_t)n_img, (uint32_t)h, (uint32_t)w, (uint32_t)C, (uint32_t)C_total, (uint32_t)c_off, eps);                  \
        else BC_LAUNCH(ps, (k_l2norm_cat_deconv<T_, 2>), dim3((unsigned)wgs), dim3(256), 0, st, (T_ *)out, (const T_ *)t, bias, weight,             \
                       (uint32_t)n_img, (uint32_t)h, (uint32_t)w, (uint32_t)C, (uint32_t)C_total, (uint32_t)c_off, eps);                             \
    } while (0)
    if (dtype == BC_F32) BC_LD(float);
    else if (dtype == BC_F16) BC_LD(__half);
    else BC_LD(hip_bfloat16);
#undef BC_LD
    return launch_status();
}

/* GroupNorm statistics over ALL pixels of a channels-last (n_pix, C) matrix (= the packed tiles of one frame), returned as the
 * per-channel affine map of the normalisation: scale[c] = gamma[c]*rstd[g], shift[c] = beta[c] - mean[g]*scale[c]. */
BC_EXPORT int bc_group_norm_affine_nhwc(const void *features, long long n_pix, int C, int groups, int dtype, float eps, const float *gamma,
                                        const float *beta, float *scale, float *shift, float *workspace, long long workspace_floats, void *stream)
{
    return launch_group_stats(features, n_pix, C, groups, dtype, eps, gamma, beta, scale, shift, workspace, workspace_floats,
                              BnExtra{nullptr, nullptr, nullptr, nullptr, nullptr, 0.0f}, (hipStream_t)stream);
}

/* channels-last form of the training-mode BatchNorm statistics: the same two kernels with one channel per group; also writes what the
 * backward pass and the module state need.  The normalisation itself is bc_affine_act_nhwc with the returned (scale, shift). */
BC_EXPORT int bc_bn_train_stats_nhwc(const void *features, long long n_pix, int C, int dtype, float eps, const float *gamma, const float *beta,
                                     float *running_mean, float *running_var, long long *num_batches_tracked, float momentum, float *save_mean,
                                     float *save_invstd, float *scale, float *shift, float *workspace, long long workspace_floats, void *stream)
{
    return launch_group_stats(features, n_pix, C, C, dtype, eps, gamma, beta, scale, shift, workspace, workspace_floats,
                              BnExtra{save_mean, save_invstd, running_mean, running_var, num_batches_tracked, momentum}, (hipStream_t)stream);
}

/* training-mode BatchNorm2d forward of an NCHW fp32 tensor (batch statistics; optional fused ReLU): see k_bn_stats / k_bn_apply */
BC_EXPORT int bc_bn_train_fwd(void *y, const void *x, int N, int C, long long HW, const float *gamma, const float *beta, float *running_mean,
                              float *running_var, long long *num_batches_tracked, float *save_mean, float *save_invstd, float momentum,
                              float eps, int relu, float *workspace, long long workspace_floats, void *stream)
{
    if (N <= 0 || C <= 0 || HW <= 0 || (uint64_t)N * HW >= (1ull << 31) || (uint64_t)N * C * HW >= (1ull << 32)) return BC_ERR_SHAPE;
    if (!y || !x || !workspace) return BC_ERR_NULL;
    if (!aligned(y, 16) || !aligned(x, 16)) return BC_ERR_ALIGN;
    BnGeom g;
    g.N = N; g.C = C; g.HW = (uint32_t)HW;
    const uint32_t total = (uint32_t)N * g.HW;
    uint32_t chunks = (total + 8191) / 8192;
    if (chunks > 64) chunks = 64;
    if (chunks < 1) chunks = 1;
    g.L = ((total + chunks - 1) / chunks + 3) / 4 * 4;
    g.chunks = (total + g.L - 1) / g.L;
    if ((long long)C * g.chunks * 2 > workspace_floats) return BC_ERR_RANGE;
    ProfScope ps(BC_OP_AFFINE, 3.0 * N * C * (double)HW * 4);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(k_bn_stats, dim3(g.chunks, C), dim3(WG), 0, st, (const float *)x, (float2 *)workspace, g);
    BC_LAUNCH(ps, k_bn_apply, dim3(g.chunks, C), dim3(WG), 0, st, (float *)y, (const float *)x, (const float2 *)workspace, gamma, beta, running_mean,
              running_var, num_batches_tracked, save_mean, save_invstd, momentum, eps, relu, g);
    return launch_status();
}

/* F.adaptive_avg_pool2d of a channels-last (N, C, H, W) tensor to (OH, OW): one workgroup per output bin */
BC_EXPORT int bc_adaptive_avg_pool_nhwc(void *out, const void *in, int N, int C, int H, int W, int OH, int OW, int dtype, void *stream)
{
    if (dtype < BC_F32 || dtype > BC_BF16) return BC_ERR_ELEM;
    if (N <= 0 || C <= 0 || H <= 0 || W <= 0 || OH <= 0 || OW <= 0) return BC_ERR_SHAPE;
    const int E = dtype == BC_F32 ? 4 : 2, VE = 16 / E;
    if (C % VE != 0 || C / VE > WG || WG % (C / VE) != 0) return BC_ERR_SHAPE;
    if ((uint64_t)N * H * W * C >= (1ull << 31)) return BC_ERR_RANGE;
    if (!out || !in) return BC_ERR_NULL;
    if (!aligned(out, 16) || !aligned(in, 16)) return BC_ERR_ALIGN;
    ProfScope ps(BC_OP_AFFINE, (double)N * H * W * C * E);
    const dim3 grid((unsigned)(N * OH * OW));
    const uint32_t K = (uint32_t)(C / VE);
    hipStream_t st = (hipStream_t)stream;
    if (dtype == BC_F32)
        BC_LAUNCH(ps, (k_adaptive_avg_pool_nhwc<float, 4>), grid, dim3(WG), 0, st, (VecOf<16>::type *)out, (const VecOf<16>::type *)in, (uint32_t)H, (uint32_t)W, K, (uint32_t)OH, (uint32_t)OW);
    else if (dtype == BC_F16)
        BC_LAUNCH(ps, (k_adaptive_avg_pool_nhwc<__half, 8>), grid, dim3(WG), 0, st, (VecOf<16>::type *)out, (const VecOf<16>::type *)in, (uint32_t)H, (uint32_t)W, K, (uint32_t)OH, (uint32_t)OW);
    else
        BC_LAUNCH(ps, (k_adaptive_avg_pool_nhwc<hip_bfloat16, 8>), grid, dim3(WG), 0, st, (VecOf<16>::type *)out, (const VecOf<16>::type *)in, (uint32_t)H, (uint32_t)W, K, (uint32_t)OH, (uint32_t)OW);
    return launch_status();
}

BC_EXPORT int bc_interp_bilinear_act_nhwc(void *out, const void *in, long long planes, int C, int h, int w, int H, int W,
                                          int align_corners, float rh, float rw, int dtype, const float *scale, const float *shift,
                                          const void *add, int relu, void *stream)
{
    const DynArm arm = dyn_take();
    if (dtype < BC_F32 || dtype > BC_BF16) return BC_ERR_ELEM;
    if (planes < 0 || C <= 0 || h <= 0 || w <= 0 || H <= 0 || W <= 0) return BC_ERR_SHAPE;
    if (planes == 0) return BC_OK;
    if (!out || !in) return BC_ERR_NULL;
    const int E = dtype == BC_F32 ? 4 : 2;
    if ((uint64_t)planes * H * W * C >= (1ull << 31) || (uint64_t)planes * h * w * C >= (1ull << 31)) return BC_ERR_RANGE;
    hipStream_t st = (hipStream_t)stream;
    int q = 16 / E;
    while (q > 1 && ((C % q) != 0 || !aligned(out, q * E) || !aligned(in, q * E) || !aligned(add, q * E))) q >>= 1;
    InterpEpi ep{scale, shift, add, relu, (scale || shift || add || relu) ? 1 : 0};
    InterpNhwcGeom g;
    g.Cq = make_fd((uint32_t)(C / q)); g.W = make_fd(W); g.H = make_fd(H);
    g.h = h; g.w = w; g.rh = rh; g.rw = rw; g.align = align_corners;
    g.total = (uint32_t)((uint64_t)planes * H * W * (C / q));
    g.dyn = dyn_flat(arm, g.total);
    const int grid = grid_exact(g.total, 1);
    ProfScope ps(BC_OP_INTERP, ((double)planes * h * w + (double)planes * H * W * (add ? 2 : 1)) * C * E);
#define BC_IN(T_, Q_) BC_LAUNCH(ps, (k_interp_bilinear_nhwc<T_, Q_>), dim3(grid), dim3(WG), 0, st, (T_ *)out, (const T_ *)in, g, ep)
#define BC_INQ(T_, QMAX_) do { if (q == QMAX_) BC_IN(T_, QMAX_); else if (q == QMAX_ / 2) BC_IN(T_, QMAX_ / 2);     \
                               else if (QMAX_ >= 8 && q == 2) BC_IN(T_, 2); else BC_IN(T_, 1); } while (0)
    if (dtype == BC_F32) BC_INQ(float, 4);
    else if (dtype == BC_F16) BC_INQ(__half, 8);
    else BC_INQ(hip_bfloat16, 8);
#undef BC_INQ
#undef BC_IN
    return launch_status();
}

BC_EXPORT int bc_interp_bilinear_nhwc(void *out, const void *in, long long planes, int C, int h, int w, int H, int W,
                                      int align_corners, float rh, float rw, int dtype, void *stream)
{
    return bc_interp_bilinear_act_nhwc(out, in, planes, C, h, w, H, W, align_corners, rh, rw, dtype, nullptr, nullptr, nullptr, 0, stream);
}

BC_EXPORT int bc_upsample_argmax(long long *out, const void *in, int N, int C, int h, int w, int H, int W, long long sn, long long sc,
                                 long long sy, long long sx, int align_corners, float rh, float rw, int dtype, void *stream)
{
    if (dtype < BC_F32 || dtype > BC_BF16) return BC_ERR_ELEM;
    if (N <= 0 || C <= 0 || h <= 0 || w <= 0 || H <= 0 || W <= 0) return BC_ERR_SHAPE;
    if (!out || !in) return BC_ERR_NULL;
    if ((uint64_t)N * H * W >= (1ull << 31) || (uint64_t)N * C * h * w >= (1ull << 31)) return BC_ERR_RANGE;
    const int E = dtype == BC_F32 ? 4 : 2;
    if (!aligned(out, 8) || !aligned(in, E)) return BC_ERR_ALIGN;
    UpArgGeom g;
    g.C = C; g.h = h; g.w = w; g.H = H; g.W = W; g.sn = sn; g.sc = sc; g.sy = sy; g.sx = sx; g.rh = rh; g.rw = rw; g.align = align_corners;
    g.total = (uint32_t)((uint64_t)N * H * W);
    const int grid = grid_exact(g.total, 1);
    hipStream_t st = (hipStream_t)stream;
    ProfScope ps(BC_OP_INTERP, (double)N * h * w * C * E + (double)N * H * W * 8.0);
    if (dtype == BC_F32) BC_LAUNCH(ps, (k_upsample_argmax<float>), dim3(grid), dim3(WG), 0, st, out, (const float *)in, g);
    else if (dtype == BC_F16) BC_LAUNCH(ps, (k_upsample_argmax<__half>), dim3(grid), dim3(WG), 0, st, out, (const __half *)in, g);
    else BC_LAUNCH(ps, (k_upsample_argmax<hip_bfloat16>), dim3(grid), dim3(WG), 0, st, out, (const hip_bfloat16 *)in, g);
    return launch_status();
}

BC_EXPORT int bc_grid_tables(const uint8_t *grid, int n_total, int32_t *grid_idx, int32_t *mapping_exec,
                             const int32_t *prev_grid_idx, int32_t *transfer_idx, int32_t *counts, void *stream)
{
    if (n_total <= 0) return BC_ERR_SHAPE;
    if (!grid || !grid_idx || !mapping_exec || !counts) return BC_ERR_NULL;
    if (prev_grid_idx && !transfer_idx) return BC_ERR_NULL;
    ProfScope ps(BC_OP_GRID_TABLES, 9.0 * n_total);
    BC_LAUNCH(ps, k_grid_tables, dim3(1), dim3(1024), 0, (hipStream_t)stream, grid, n_total, grid_idx,
                       mapping_exec, prev_grid_idx, transfer_idx, counts);
    return launch_status();
}

BC_EXPORT int bc_policy_step(const float *logits, int n_total, unsigned long long seed, unsigned long long counter, int multiple,
                              int at_least_one, uint8_t *grid, int32_t *grid_idx, int32_t *mapping_exec, int32_t *counts,
                              int32_t *host_mailbox, void *stream)
{
    if (n_total <= 0 || n_total > POLICY_MAX_TILES || multiple <= 0) return BC_ERR_SHAPE;
    if (!logits || !grid || !grid_idx || !mapping_exec || !counts) return BC_ERR_NULL;
    ProfScope ps(BC_OP_GRID_TABLES, 13.0 * n_total);
    BC_LAUNCH(ps, k_policy_step, dim3(1), dim3(1024), 0, (hipStream_t)stream, logits, n_total, seed, counter, multiple, at_least_one,
              grid, grid_idx, mapping_exec, counts, (volatile int32_t *)host_mailbox);
    return launch_status();
}

BC_EXPORT int bc_policy_features(float *out, int N, int h, int w, const void *const *ptrs, const long long *strides, const int *dims,
                                  const float *scales, void *stream)
{
    if (!out || !ptrs || !strides || !dims || !scales) return BC_ERR_NULL;
    if (N <= 0 || h <= 0 || w <= 0) return BC_ERR_SHAPE;
    FeatGeom g;
    g.N = N; g.h = h; g.w = w; g.Ctot = 0;
    for (int k = 0; k < 4; ++k) {
        FeatSrc &s = g.src[k];
        s.ptr = ptrs[k];
        if (!s.ptr) return BC_ERR_NULL;
        s.sn = strides[4 * k]; s.sc = strides[4 * k + 1]; s.sh = strides[4 * k + 2]; s.sw = strides[4 * k + 3];
        s.C = dims[4 * k]; s.H = dims[4 * k + 1]; s.W = dims[4 * k + 2]; s.dtype = dims[4 * k + 3];
        if (s.C <= 0 || s.H <= 0 || s.W <= 0 || s.dtype < 0 || s.dtype > 3) return BC_ERR_SHAPE;
        s.scale_h = scales[3 * k]; s.scale_w = scales[3 * k + 1]; s.offset = scales[3 * k + 2];
        g.Ctot += s.C;
    }
    const uint64_t total = (uint64_t)N * g.Ctot * h * w;
    if (total >= (1ull << 31)) return BC_ERR_RANGE;
    ProfScope ps(BC_OP_AFFINE, 2.0 * total * 4);
    const unsigned grid = (unsigned)((total + WG - 1) / WG < (uint64_t)MAX_WG * 4 ? (total + WG - 1) / WG : (uint64_t)MAX_WG * 4);
    BC_LAUNCH(ps, k_policy_features, dim3(grid), dim3(WG), 0, (hipStream_t)stream, out, g);
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
        g_prof.aux[op] = 0;
    }
    return BC_OK;
}

BC_EXPORT int bc_prof_read_aux(int op, double *total_aux)
{
    if (op < 0 || op >= BC_OP_COUNT) return BC_ERR_SHAPE;
    if (!total_aux) return BC_ERR_NULL;
    std::lock_guard<std::mutex> lk(g_prof.mu);
    *total_aux = g_prof.aux[op];
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
#endif  // BC_PART == 0
