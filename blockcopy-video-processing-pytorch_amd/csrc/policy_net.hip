// policy_net.hip -- the online-RL policy's CNN (forward, backward, optimizer) on own gfx950 kernels.  Part of libblockcopy_hip.so.
//
// Reference: blockcopy/blockcopy/policy/net.py:17-125 (input recipe, resnet8 trunk + three stride-2 head stages),
// policy/resnet.py:60-115 (BasicBlock, training-mode BatchNorm with momentum 0.02), policy/policy.py:265-306 (forward) and :319-370
// (REINFORCE loss, backward, RMSprop step), policy/information_gain.py:22-41 (KL reward).  The reference runs all of it through
// PyTorch/cuDNN: ~45 launches per forward, ~120 per training step.  Here every tensor of the net is a dense channels-last fp32 map
// [N][H][W][C] and the step is a fixed sequence of a few kernel kinds:
//   k_pn_conv      implicit-GEMM conv on the fp32 matrix cores (v_mfma_f32_32x32x2_f32): forward (stride 1 / 2, 3x3 / 1x1), data gradient
//                  (stride 1: the same conv on transposed weights; stride 2: four output-parity classes with 1 / 2 / 2 / 4 taps).  The
//                  producer's BatchNorm + ReLU is applied while the input patch is staged (prologue), the training-mode batch statistics
//                  of the OUTPUT are left as per-workgroup partial sums (epilogue), the residual gradient rides in the store.
//   k_pn_wgrad     weight gradient as a GEMM over pixels (M = input channel, N = output channel, K = pixel), nine taps per wave in
//                  registers, split over pixel-tile groups with a FIXED-ORDER second pass (no atomics: run-to-run identical).
//   k_pn_*         BatchNorm finalize / backward reduce / backward apply, residual join, information gain, REINFORCE seed, RMSprop,
//                  parameter export -- single passes over small maps.
// Layouts: weights W[tap][Cin][Cout] (tap = 3*ky + kx), transposed copy WT[tap][Cout][Cin] for the data gradient.
// Every entry point is stateless (raw pointers, sizes, stream) so the host can capture the whole step in one hipGraph.
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/blockcopy_hip.h"

#define BC_EXPORT extern "C" __attribute__((visibility("default")))

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

template <int PREC> struct PnHalf { typedef _Float16 T; typedef f16x4 V4; typedef f16x8 V8; static constexpr float scale = 16.f; };
template <> struct PnHalf<2> { typedef __bf16 T; typedef bf16x4 V4; typedef bf16x8 V8; static constexpr float scale = 1.f; };

unsigned long long *g_pn_stamps = nullptr;      // measurement only (bc_pn_set_stamps)

inline int pn_status()
{
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? BC_OK : (int)e;
}

// ------------------------------------------------------------------------------------------------------------------ conv
constexpr int PN_TH = 4, PN_TW = 32;     // output tile of a workgroup: 4 rows (one per wave) x 32 columns (the M index of a 32x32 MFMA)
constexpr int PN_MAXTAPS = 9;

struct PnTapSet {
    int ntaps;
    int dmin_y, dmin_x;      // smallest tap offset (input row of output row 0 of the tile = oy0 * S + dmin_y)
    int PH, PW;              // staged patch
    uint32_t pw_magic;       // pix / PW = (pix * pw_magic) >> 20
    int Hc, Wc;              // class grid (outputs of this parity class)
    int out_oy, out_ox;      // output pixel = (oy * out_s + out_oy, ox * out_s + out_ox)
    int8_t dy[PN_MAXTAPS], dx[PN_MAXTAPS], wt[PN_MAXTAPS];
};

// training-mode BatchNorm coefficients from fixed-point sums (k_pn_bn_finalize's arithmetic, so that every consumer and the end-of-forward
// finalize derive bit-identical values): S1 = sum x, S2 = sum x^2 in units of 2^-24
constexpr double PN_ACC_UNIT = 16777216.0;
constexpr int PN_ACC_R = 16;         // replicas of an accumulator set (workgroup w adds to replica w mod 16: 1/16 of the same-address atomics)
struct PnBnSrc {
    const unsigned long long *acc;   // [PN_ACC_R][2][C]
    const float *gamma, *beta;
    double count;
    float eps;
    int C;
};
constexpr int PN_COEF_MAX = 128;      // channels of a BatchNorm whose coefficients a kernel derives in its LDS
__device__ __forceinline__ void pn_bn_coef(const PnBnSrc &b, int c, float &scale, float &shift, float *mean_out = nullptr, float *invstd_out = nullptr,
                                           double *var_out = nullptr)
{
    long long i1 = 0, i2 = 0;
#pragma unroll
    for (int r = 0; r < PN_ACC_R; ++r) { i1 += (long long)b.acc[(size_t)r * 2 * b.C + c]; i2 += (long long)b.acc[(size_t)r * 2 * b.C + b.C + c]; }
    const double s1 = (double)i1 / PN_ACC_UNIT, s2 = (double)i2 / PN_ACC_UNIT;
    const double mean = s1 / b.count;
    double var = s2 / b.count - mean * mean;
    var = var > 0.0 ? var : 0.0;
    const float invstd = (float)(1.0 / sqrt(var + (double)b.eps));
    const float g = b.gamma ? b.gamma[c] : 1.0f, bt = b.beta ? b.beta[c] : 0.0f;
    scale = g * invstd;
    shift = bt - (float)mean * scale;
    if (mean_out) *mean_out = (float)mean;
    if (invstd_out) *invstd_out = invstd;
    if (var_out) *var_out = var;
}

struct PnConvArgs {
    const float *x, *w;
    float *out;
    const float *in_scale, *in_shift;
    const float *add, *add_mask;
    float *stats;
    // BatchNorm without a finalize launch (bc_pn_arm_bn): the producer adds its per-workgroup sums to FIXED-POINT accumulators (order-independent:
    // run-to-run identical), the consumer derives scale / shift of its input channels from them at kernel start
    PnBnSrc in_bn;           // acc != nullptr: the prologue's coefficients come from these sums instead of in_scale / in_shift
    unsigned long long *out_acc;      // [PN_ACC_R][2][Nn] sums of the output and of its squares, in units of 2^-24
    int N, Hi, Wi, K, Nn;    // input map, reduction channels, output channels
    int Hout, Wout;          // full output map
    int S, out_s;            // input stride, output pixel stride
    int in_relu, accumulate;
    int tiles_x, tiles_y;
    int npix_pad;            // padded pixel count of the staged patch (= 1 mod 8)
    int n_cls;
    unsigned long long *stamps;      // measurement only (bc_pn_set_stamps): 8 x 100 MHz stamps per workgroup
    int dbg;                 // measurement only (PN_DBG): 1 no MFMAs, 2 no global loads, 4 no output stores, 8 no LDS stores, 16 no next-tile planning
    PnTapSet cls[4];
};

// Patch image  P[c4][h][pix][2]  (channel 4*c4 + 2*h + j at element j): lane (r, h) reads the operands of two k-steps with one ds_read_b64.
// Weight image W[t][c4][h][n][2].  MFMA roles: A = weights (M = output channel), B = pixels (N = the 32 columns of tile row `wave`), so a lane
// ends up with 16 output channels of ONE pixel -- four float4 stores per lane instead of sixteen dword stores.
// PERSISTENT workgroups over the tiles wg, wg + G, ... of a class; stage = (tile, channel chunk).  Both images are DOUBLE BUFFERED: while stage s
// multiplies, stage s + 1 is written to the other buffer (behind the first tap's MFMAs) and stage s + 2 is requested from memory (behind the
// fifth tap's) -- one barrier per stage, no phase in which a wave only moves data.  A layer whose K is one chunk keeps its weights resident.
// TS (small maps: too few 4-row tiles to fill the chip): the tile is ONE row, the four waves split the TAPS of every stage (tap t -> wave t mod 4)
// and their accumulators are joined through the LDS in a fixed order before wave 0 runs the epilogue -- four times the workgroups, a third of the
// serial MFMA chain per workgroup.
// PREC 1 / 2: the same sums on the 16-bit matrix pipe.  PREC 1, fp32 accuracy -- every operand is split  x = hi + lo  into two fp16 numbers (22 bits of mantissa
// together; both operands scaled by 16 so that lo stays a normal number down to 4e-6, |x| < 4094) and a product is the three MFMAs
// hi*hi + hi*lo + lo*hi into the fp32 accumulator (lo*lo, 2^-22 of the product, is dropped): v_mfma_f32_32x32x16_f16 multiplies 16 channels in
// 32 cycles where v_mfma_f32_32x32x2_f32 needs 8 x 64, so three of them are 5.3 x faster than the fp32 pipe.  Used for the FORWARD convs
// (activations and weights are O(1) numbers behind BatchNorm).  PREC 2: hi + lo in bf16 (fp32's exponent range: no scale, nothing under- or
// overflows; 16 bits of mantissa together, products good to 2^-16) -- for the DATA GRADIENT, whose operand (1e-7 .. 1e-3) has no fixed scale
// and whose consumer is an RMSprop step.
template <int NB, int KC, int S, bool TS, int PREC>
__global__ __launch_bounds__(256, (KC == 16 && S == 1 && !TS) ? 2 : 1) void k_pn_conv(const PnConvArgs a)
{
    static_assert(NB == 1, "one 32-channel output block per workgroup");
    constexpr int TH = TS ? 1 : PN_TH;
    constexpr bool F16 = PREC != 0;      // 16-bit matrix pipe, operands split hi + lo (PREC 1: fp16 scaled by 16, 2: bf16)
    typedef typename PnHalf<PREC>::T HT;
    typedef typename PnHalf<PREC>::V4 HV4;
    typedef typename PnHalf<PREC>::V8 HV8;
    constexpr float HS = PnHalf<PREC>::scale;
    extern __shared__ __attribute__((aligned(16))) float pn_lds[];
    constexpr int BN = 32, C4 = KC / 4, TG = PN_MAXTAPS;
    const PnTapSet &cs = a.cls[blockIdx.z];
    const int a_floats = KC * a.npix_pad;
    constexpr int b_floats = TG * KC * BN;
    float *As0 = pn_lds, *As1 = pn_lds + a_floats;
    float *Bs0 = pn_lds + 2 * a_floats, *Bs1 = Bs0 + b_floats;
    float *join = Bs0 + (a.K == KC ? 1 : 2) * b_floats;      // TS: [wave][16][64] accumulators (behind the one or two weight images)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, h = lane >> 5;
    const int n0 = blockIdx.y * BN;
    const int npix = cs.PH * cs.PW;
    const int n_chunks = cs.ntaps > 0 ? a.K / KC : 0;
    const int tiles_img = a.tiles_x * a.tiles_y, n_tiles = a.N * tiles_img;

    constexpr int MAXA = (TS ? (S == 1 ? 102 : 195) : (S == 1 ? 204 : 585)) * C4 / 256 + 1;     // float4 of the patch per thread and chunk
    constexpr int NBQ = TG * KC * BN / 4;                 // float4 of the weights per stage
    constexpr int NBV = (NBQ + 255) / 256;                // ... per thread
    float4 pa[MAXA];
    float4 pb[NBV];
    int aoff[MAXA];       // element offset of pa[i] inside the image at chunk 0, -1 outside the image / the patch
    int boff[NBV];        // element offset of pb[i] inside w at chunk 0, -1 = no such tap
    f32x16 acc;
    float s1[16], s2[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) { acc[e] = 0.0f; s1[e] = 0.f; s2[e] = 0.f; }

    unsigned long long *stp = a.stamps ? a.stamps + 8 * ((size_t)(blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) : nullptr;
    auto stamp = [&](int k) { if (stp && tid == 0) stp[k] = __builtin_amdgcn_s_memrealtime(); };
    stamp(0);
    __shared__ __align__(16) float pn_coef[2 * PN_COEF_MAX];      // [scale | shift] of the input channels (in_bn): derived under the first loads' latency

    auto tile_origin = [&](int tile, int &n, int &oy0, int &ox0) {
        n = tile / tiles_img;
        const int t2 = tile - n * tiles_img;
        const int ty = t2 / a.tiles_x;
        oy0 = ty * TH; ox0 = (t2 - ty * a.tiles_x) * PN_TW;
    };
    auto tile_live = [&](int tile) {
        int n, oy0, ox0;
        tile_origin(tile, n, oy0, ox0);
        return oy0 < cs.Hc && ox0 < cs.Wc;      // (the classes of a stride-2 data gradient differ in size)
    };
    auto next_live = [&](int tile) {
        tile += gridDim.x;
        while (tile < n_tiles && !tile_live(tile)) tile += gridDim.x;
        return tile;
    };
    auto plan_tile = [&](int tile) {
        int n, oy0, ox0;
        tile_origin(tile, n, oy0, ox0);
        const int iy0 = oy0 * S + cs.dmin_y, ix0 = ox0 * S + cs.dmin_x;
#pragma unroll
        for (int i = 0; i < MAXA; ++i) {
            const int idx = tid + 256 * i;
            const int pix = idx / C4, c4 = idx % C4;
            int o = -1;
            if (pix < npix) {
                const int py = (int)(((uint32_t)pix * cs.pw_magic) >> 20), px = pix - py * cs.PW;
                const int iy = iy0 + py, ix = ix0 + px;
                if (iy >= 0 && iy < a.Hi && ix >= 0 && ix < a.Wi) o = ((n * a.Hi + iy) * a.Wi + ix) * a.K + 4 * c4;
            }
            aoff[i] = o;
        }
    };
    auto load_stage = [&](int chunk, bool with_b) {
        if (a.dbg & 2) return;
#pragma unroll
        for (int i = 0; i < MAXA; ++i) {
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (aoff[i] >= 0) v = *reinterpret_cast<const float4 *>(a.x + (size_t)aoff[i] + chunk * KC);
            pa[i] = v;
        }
        if (with_b) {
#pragma unroll
            for (int i = 0; i < NBV; ++i) {
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (boff[i] >= 0) v = *reinterpret_cast<const float4 *>(a.w + (size_t)boff[i] + (size_t)chunk * KC * a.Nn);
                pb[i] = v;
            }
        }
    };
    auto store_stage = [&](float *As, float *Bs, int chunk, bool with_b) {
        if (a.dbg & 8) return;
#pragma unroll
        for (int i = 0; i < MAXA; ++i) {
            const int idx = tid + 256 * i;
            const int pix = idx / C4, c4 = idx % C4;
            if (pix < npix) {
                float4 v = pa[i];
                if (aoff[i] >= 0) {      // (zero padding is applied AFTER the prologue)
                    const int c = chunk * KC + 4 * c4;
                    if (a.in_bn.acc) {
                        const float4 sc = *reinterpret_cast<const float4 *>(pn_coef + c), sh = *reinterpret_cast<const float4 *>(pn_coef + PN_COEF_MAX + c);
                        v.x = fmaf(v.x, sc.x, sh.x); v.y = fmaf(v.y, sc.y, sh.y); v.z = fmaf(v.z, sc.z, sh.z); v.w = fmaf(v.w, sc.w, sh.w);
                    } else if (a.in_scale) {
                        const float4 sc = *reinterpret_cast<const float4 *>(a.in_scale + c), sh = *reinterpret_cast<const float4 *>(a.in_shift + c);
                        v.x = fmaf(v.x, sc.x, sh.x); v.y = fmaf(v.y, sc.y, sh.y); v.z = fmaf(v.z, sc.z, sh.z); v.w = fmaf(v.w, sc.w, sh.w);
                    }
                    if (a.in_relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
                }
                if constexpr (F16) {
                    // halves: [part hi | lo][kg = channel / 8][pix][8]
                    HT *Ah = reinterpret_cast<HT *>(As);
                    const float x4[4] = {v.x * HS, v.y * HS, v.z * HS, v.w * HS};
                    HV4 hi, lo;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        hi[j] = (HT)x4[j];
                        lo[j] = (HT)(x4[j] - (float)hi[j]);
                    }
                    const size_t o = ((size_t)(c4 >> 1) * a.npix_pad + pix) * 8 + (c4 & 1) * 4;
                    *reinterpret_cast<HV4 *>(Ah + o) = hi;
                    *reinterpret_cast<HV4 *>(Ah + (size_t)KC * a.npix_pad + o) = lo;
                } else {
                    *reinterpret_cast<float2 *>(As + ((size_t)(c4 * 2 + 0) * a.npix_pad + pix) * 2) = make_float2(v.x, v.y);
                    *reinterpret_cast<float2 *>(As + ((size_t)(c4 * 2 + 1) * a.npix_pad + pix) * 2) = make_float2(v.z, v.w);
                }
            }
        }
        if (with_b) {
#pragma unroll
            for (int i = 0; i < NBV; ++i) {
                const int idx = tid + 256 * i;
                const int n4 = idx % (BN / 4), k = (idx / (BN / 4)) % KC, t = idx / (BN / 4 * KC);
                if (idx < NBQ) {
                    if constexpr (F16) {
                        // halves: [part hi | lo][t][kg = k / 8][n][8]
                        HT *Bh = reinterpret_cast<HT *>(Bs);
                        const float w4[4] = {pb[i].x * HS, pb[i].y * HS, pb[i].z * HS, pb[i].w * HS};
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const HT hi = (HT)w4[e];
                            const size_t o = ((size_t)(t * (KC / 8) + (k >> 3)) * BN + 4 * n4 + e) * 8 + (k & 7);
                            Bh[o] = hi;
                            Bh[(size_t)TG * KC * BN + o] = (HT)(w4[e] - (float)hi);
                        }
                    } else {
                        float *dst = Bs + ((size_t)((t * C4 + k / 4) * 2 + ((k >> 1) & 1)) * BN + 4 * n4) * 2 + (k & 1);
                        dst[0] = pb[i].x; dst[2] = pb[i].y; dst[4] = pb[i].z; dst[6] = pb[i].w;
                    }
                }
            }
        }
    };

    // the stage sequence of this workgroup: (tile, chunk) over its live tiles (uniform over the workgroup: every barrier is reached by all waves)
    const bool b_resident = n_chunks == 1;
    int tile = blockIdx.x;                    // tile of the stage being multiplied
    while (tile < n_tiles && !tile_live(tile)) tile += gridDim.x;
    int ld_tile = tile, ld_chunk = 0;         // stage whose data the registers hold / will hold next
    auto advance = [&](int &t, int &c) {
        if (++c >= n_chunks) { c = 0; t = next_live(t); }
    };
    int st_tile = tile, st_chunk = 0;         // stage that goes to the LDS next
    int s_idx = 0;
    if (tile < n_tiles && n_chunks > 0) {
        plan_tile(tile);
        // the patch goes out first, the (tile-independent) weight offsets are built under its latency
        load_stage(0, false);
#pragma unroll
        for (int i = 0; i < NBV; ++i) {
            const int idx = tid + 256 * i;
            const int n4 = idx % (BN / 4), k = (idx / (BN / 4)) % KC, t = idx / (BN / 4 * KC);
            boff[i] = (idx < NBQ && t < cs.ntaps) ? (cs.wt[t < PN_MAXTAPS ? t : 0] * a.K + k) * a.Nn + n0 + 4 * n4 : -1;
        }
        if (!(a.dbg & 2)) {
#pragma unroll
            for (int i = 0; i < NBV; ++i) {
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (boff[i] >= 0) v = *reinterpret_cast<const float4 *>(a.w + (size_t)boff[i]);
                pb[i] = v;
            }
        }
        if (a.in_bn.acc) {      // (the first patch and the weights are in flight)
            for (int c = tid; c < a.in_bn.C; c += 256) pn_bn_coef(a.in_bn, c, pn_coef[c], pn_coef[PN_COEF_MAX + c]);
            __syncthreads();
        }
        stamp(1);
        store_stage(As0, Bs0, 0, true);
        advance(st_tile, st_chunk);
        ld_tile = st_tile; ld_chunk = st_chunk;
        if (ld_tile < n_tiles) {
            if (ld_chunk == 0) plan_tile(ld_tile);
            load_stage(ld_chunk, !b_resident);
        }
        __syncthreads();
        stamp(2);
    }
    // ---- epilogue of a tile: lane = pixel column r of tile row `wave`; register e = output channel n0 + (e & 3) + 8 (e >> 2) + 4 h
    auto epilogue = [&](int tile_) {
        int n, oy0, ox0;
        tile_origin(tile_, n, oy0, ox0);
        const int oy = oy0 + (TS ? 0 : wave), ox = ox0 + r;
        if ((!TS || wave == 0) && oy < cs.Hc && ox < cs.Wc && !(a.dbg & 4)) {
            const int Y = oy * a.out_s + cs.out_oy, X = ox * a.out_s + cs.out_ox;
            const size_t o = (((size_t)n * a.Hout + Y) * a.Wout + X) * a.Nn + n0 + 4 * h;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float4 v = make_float4(acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]);
                if constexpr (PREC == 1) { v.x *= 1.0f / 256.f; v.y *= 1.0f / 256.f; v.z *= 1.0f / 256.f; v.w *= 1.0f / 256.f; }      // (both operands carried a factor 16)
                if (a.add) {
                    const float4 g = *reinterpret_cast<const float4 *>(a.add + o + 8 * q);
                    if (a.add_mask) {
                        const float4 m = *reinterpret_cast<const float4 *>(a.add_mask + o + 8 * q);
                        v.x += m.x > 0.f ? g.x : 0.f; v.y += m.y > 0.f ? g.y : 0.f; v.z += m.z > 0.f ? g.z : 0.f; v.w += m.w > 0.f ? g.w : 0.f;
                    } else { v.x += g.x; v.y += g.y; v.z += g.z; v.w += g.w; }
                }
                if (a.accumulate) {
                    const float4 p = *reinterpret_cast<const float4 *>(a.out + o + 8 * q);
                    v.x += p.x; v.y += p.y; v.z += p.z; v.w += p.w;
                }
                *reinterpret_cast<float4 *>(a.out + o + 8 * q) = v;
                s1[4 * q] += v.x; s1[4 * q + 1] += v.y; s1[4 * q + 2] += v.z; s1[4 * q + 3] += v.w;
                s2[4 * q] = fmaf(v.x, v.x, s2[4 * q]); s2[4 * q + 1] = fmaf(v.y, v.y, s2[4 * q + 1]);
                s2[4 * q + 2] = fmaf(v.z, v.z, s2[4 * q + 2]); s2[4 * q + 3] = fmaf(v.w, v.w, s2[4 * q + 3]);
            }
        }
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[e] = 0.0f;
    };
    bool first_tile = true;
    while (tile < n_tiles) {
        if (n_chunks == 0) epilogue(tile);      // (a parity class no tap reaches: zeros, plus the residual terms; TS: wave 0 stores)
        for (int chunk = 0; chunk < n_chunks; ++chunk, ++s_idx) {
            const int cur = s_idx & 1;
            const float *As = cur ? As1 : As0;
            const float *Bs = (b_resident || !cur) ? Bs0 : Bs1;
#pragma unroll
            for (int t = 0; t < TG; ++t) {
                if (t == 1 && st_tile < n_tiles) {
                    // stage s + 1 into the other buffer, behind the first tap's MFMAs (its loads were requested during stage s - 1)
                    store_stage(cur ? As0 : As1, cur ? Bs0 : Bs1, st_chunk, !b_resident);
                    advance(st_tile, st_chunk);
                }
                if (t == 4 && st_tile < n_tiles) {
                    // stage s + 2 from memory: address arithmetic and requests in the shadow of the MFMAs
                    if (st_chunk == 0 && !(a.dbg & 16)) plan_tile(st_tile);
                    load_stage(st_chunk, !b_resident);
                }
                if (t < cs.ntaps && !(a.dbg & 1) && (!TS || (t & 3) == wave)) {
                    const int pb0 = ((TS ? 0 : wave * S) + cs.dy[t] - cs.dmin_y) * cs.PW + r * S + cs.dx[t] - cs.dmin_x;
                    if constexpr (F16) {
                        const HT *Ah = reinterpret_cast<const HT *>(As), *Bh = reinterpret_cast<const HT *>(Bs);
#pragma unroll
                        for (int m = 0; m < KC / 16; ++m) {      // lane half h takes channels 8 (2 m + h) .. + 7 of the 16 an MFMA multiplies
                            const size_t po = ((size_t)(2 * m + h) * a.npix_pad + pb0) * 8;
                            const size_t wo = ((size_t)(t * (KC / 8) + 2 * m + h) * BN + r) * 8;
                            const HV8 ph = *reinterpret_cast<const HV8 *>(Ah + po), pl = *reinterpret_cast<const HV8 *>(Ah + (size_t)KC * a.npix_pad + po);
                            const HV8 wh = *reinterpret_cast<const HV8 *>(Bh + wo), wl = *reinterpret_cast<const HV8 *>(Bh + (size_t)TG * KC * BN + wo);
                            if constexpr (PREC == 2) {
                                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh, ph, acc, 0, 0, 0);      // D[i = channel][j = pixel]
                                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh, pl, acc, 0, 0, 0);
                                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wl, ph, acc, 0, 0, 0);
                            } else {
                                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, ph, acc, 0, 0, 0);
                                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, pl, acc, 0, 0, 0);
                                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl, ph, acc, 0, 0, 0);
                            }
                        }
                    } else {
                        const float *ap = As + ((size_t)h * a.npix_pad + pb0) * 2;
                        const float *bp = Bs + ((size_t)(t * C4 * 2 + h) * BN + r) * 2;
#pragma unroll
                        for (int c4 = 0; c4 < C4; ++c4) {
                            const float2 av = *reinterpret_cast<const float2 *>(ap + (size_t)c4 * 4 * a.npix_pad);
                            const float2 bv = *reinterpret_cast<const float2 *>(bp + (size_t)c4 * 4 * BN);
                            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(bv.x, av.x, acc, 0, 0, 0);      // D[i = channel][j = pixel]
                            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(bv.y, av.y, acc, 0, 0, 0);
                        }
                    }
                }
            }
            if (chunk + 1 == n_chunks) {
                if (first_tile) stamp(3);
                if (TS) {
                    // join the four waves' partial sums (fixed order 0..3) in wave 0; the region is not touched by the staging of later stages
#pragma unroll
                    for (int e = 0; e < 16; ++e) join[(wave * 16 + e) * 64 + lane] = acc[e];
                    __syncthreads();
                    if (wave == 0) {
#pragma unroll
                        for (int e = 0; e < 16; ++e)
                            acc[e] = ((join[(0 * 16 + e) * 64 + lane] + join[(1 * 16 + e) * 64 + lane]) + join[(2 * 16 + e) * 64 + lane]) + join[(3 * 16 + e) * 64 + lane];
                    }
                }
                epilogue(tile);
                if (first_tile) stamp(4);
                first_tile = false;
            }
            __syncthreads();      // stage s + 1 is complete in its buffer; every wave is done reading stage s
        }
        tile = next_live(tile);
    }
    stamp(5);
    if (a.stats || a.out_acc) {
        // per-workgroup partial sums of the output (training-mode batch statistics): a lane holds the sums of ITS pixels for 16 channels;
        // transpose through the LDS (row = (wave, h, e, which), column = lane's pixel column; row stride 33: conflict-free both ways),
        // a thread sums a row, then the four waves are joined in a fixed order
        float *red = pn_lds;      // [256 rows][33] + [256]
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            red[(((wave * 2 + h) * 16 + e) * 2 + 0) * 33 + r] = s1[e];
            red[(((wave * 2 + h) * 16 + e) * 2 + 1) * 33 + r] = s2[e];
        }
        __syncthreads();
        float tot = 0.f;
#pragma unroll 8
        for (int j = 0; j < 32; ++j) tot += red[tid * 33 + j];
        __syncthreads();
        red[tid] = tot;           // row id = ((wave * 2 + h) * 16 + e) * 2 + which
        __syncthreads();
        if (tid < 64) {
            const int which = tid >> 5, c = tid & 31;                       // channel c = (e & 3) + 8 (e >> 2) + 4 h
            const int hh = (c >> 2) & 1, e = (c & 3) + 4 * (c >> 3);
            const int row = (hh * 16 + e) * 2 + which;
            const float v = ((red[0 * 64 + row] + red[1 * 64 + row]) + red[2 * 64 + row]) + red[3 * 64 + row];
            const size_t wg = (size_t)blockIdx.z * gridDim.x + blockIdx.x;
            if (a.stats) a.stats[(wg * 2 + which) * a.Nn + n0 + c] = v;
            if (a.out_acc) atomicAdd(&a.out_acc[((wg & (PN_ACC_R - 1)) * 2 + which) * a.Nn + n0 + c], (unsigned long long)__double2ll_rn((double)v * PN_ACC_UNIT));
        }
    }
}

uint32_t pn_magic(int d, int max_n)
{
    const uint32_t m = (uint32_t)(((1u << 20) + d - 1) / d);
    for (int n = 0; n < max_n; ++n)
        if ((int)(((uint32_t)n * m) >> 20) != n / d) return 0;
    return m;
}

constexpr int PN_PERSIST_WGS = 512;      // workgroups that walk the tiles of a launch (and rows of partial statistics its last workgroup reduces)

template <int NB, int KC, int S, bool TS, int PREC>
int pn_conv_launch(const PnConvArgs &a, hipStream_t st)
{
    const int n_b = a.K == KC ? 1 : 2;      // (a layer whose K is one chunk keeps its weights resident: one weight image)
    size_t lds = (2 * (size_t)KC * a.npix_pad + n_b * (size_t)PN_MAXTAPS * KC * 32 * NB) * sizeof(float);      // images double buffered
    if (TS) lds += 4 * 16 * 64 * sizeof(float);                                                            // + the accumulator join
    if (lds < (256 * 33 + 256) * sizeof(float)) lds = (256 * 33 + 256) * sizeof(float);                    // (the statistics transpose)
    if (lds > 160 * 1024 - 2048) return BC_ERR_SHAPE;      // (the kernel's static tables: 1 KB of BatchNorm coefficients)
    static bool attr_set[16];      // per device (the first launch on a device is never inside a stream capture: the host runs a warm pass first)
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev >= 0 && dev < 16 && !attr_set[dev]) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_pn_conv<NB, KC, S, TS, PREC>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 2048);
        attr_set[dev] = true;
    }
    const long long n_tiles = (long long)a.N * a.tiles_y * a.tiles_x;
    const dim3 grid((unsigned)(n_tiles < PN_PERSIST_WGS ? n_tiles : PN_PERSIST_WGS), (unsigned)(a.Nn / (32 * NB)), (unsigned)a.n_cls);
    hipLaunchKernelGGL((k_pn_conv<NB, KC, S, TS, PREC>), grid, dim3(256), lds, st, a);
    return pn_status();
}

// one-row tiles with the taps split over the waves where four-row tiles would leave most of the chip idle
static bool pn_tap_split(int N, int Hc_max, int Wc_max, int Nn, int n_cls)
{
    const long long wgs = (long long)N * ((Hc_max + PN_TH - 1) / PN_TH) * ((Wc_max + PN_TW - 1) / PN_TW) * (Nn / 32) * n_cls;
    return wgs < 160;
}

// ------------------------------------------------------------------------------------------------------------------ wgrad
struct PnWgradArgs {
    const float *x, *gz;
    float *part;             // [group][tap][Cx][Cy]
    const float *in_scale, *in_shift;
    int in_relu;
    int N, Hx, Wx, Cx, Hy, Wy, Cy;
    int S, ntaps, pad;       // taps = 9 (3x3, pad 1) or 1 (1x1, pad 0)
    int tiles_x, tiles_y, n_tiles, tiles_per_group;
    int PH, PW, npix;
    uint32_t pw_magic;
};

// workgroup = (group of pixel tiles, 32 input channels, 32 output channels), four waves = the four rows of a tile; a wave keeps one
// 32x32 block per tap in registers across all its tiles; at the end the four waves are summed through the LDS in a fixed order.
template <int TAPS, int S>
__global__ __launch_bounds__(256) void k_pn_wgrad(const PnWgradArgs a)
{
    extern __shared__ __attribute__((aligned(16))) float pn_lds[];
    float *xs = pn_lds;                             // [pix][32]
    float *gs = pn_lds + (size_t)a.npix * 32;       // [128][32]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, h = lane >> 5;
    const int ci0 = blockIdx.y * 32, co0 = blockIdx.z * 32;
    const int t_begin = blockIdx.x * a.tiles_per_group, t_end = min(a.n_tiles, t_begin + a.tiles_per_group);
    constexpr int MAXA = S == 1 ? 7 : 19;           // 204 * 8 / 256, 585 * 8 / 256
    float4 pa[MAXA], pg[4];
    f32x16 acc[TAPS];
#pragma unroll
    for (int t = 0; t < TAPS; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[t][e] = 0.0f;

    uint32_t pa_ok = 0;
    auto load_tile = [&](int tile) {
        int b = tile;
        const int tx = b % a.tiles_x;
        b /= a.tiles_x;
        const int ty = b % a.tiles_y, n = b / a.tiles_y;
        const int iy0 = ty * PN_TH * S - a.pad, ix0 = tx * PN_TW * S - a.pad;
        pa_ok = 0;
#pragma unroll
        for (int i = 0; i < MAXA; ++i) {
            const int idx = tid + 256 * i;
            const int pix = idx >> 3, c4 = idx & 7;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (pix < a.npix) {
                const int py = (int)(((uint32_t)pix * a.pw_magic) >> 20), px = pix - py * a.PW;
                const int iy = iy0 + py, ix = ix0 + px;
                if (iy >= 0 && iy < a.Hx && ix >= 0 && ix < a.Wx) {
                    v = *reinterpret_cast<const float4 *>(a.x + (((size_t)n * a.Hx + iy) * a.Wx + ix) * a.Cx + ci0 + 4 * c4);
                    pa_ok |= 1u << i;
                }
            }
            pa[i] = v;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int idx = tid + 256 * i;           // 128 pixels x 8 quads
            const int p = idx >> 3, c4 = idx & 7;
            const int oy = ty * PN_TH + (p >> 5), ox = tx * PN_TW + (p & 31);
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (oy < a.Hy && ox < a.Wy) v = *reinterpret_cast<const float4 *>(a.gz + (((size_t)n * a.Hy + oy) * a.Wy + ox) * a.Cy + co0 + 4 * c4);
            pg[i] = v;
        }
    };
    auto store_tile = [&]() {
#pragma unroll
        for (int i = 0; i < MAXA; ++i) {
            const int idx = tid + 256 * i;
            if ((idx >> 3) < a.npix) {
                float4 v = pa[i];
                if ((pa_ok >> i) & 1u) {
                    const int c = ci0 + 4 * (idx & 7);
                    if (a.in_scale) {
                        const float4 sc = *reinterpret_cast<const float4 *>(a.in_scale + c), sh = *reinterpret_cast<const float4 *>(a.in_shift + c);
                        v.x = fmaf(v.x, sc.x, sh.x); v.y = fmaf(v.y, sc.y, sh.y); v.z = fmaf(v.z, sc.z, sh.z); v.w = fmaf(v.w, sc.w, sh.w);
                    }
                    if (a.in_relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
                }
                *reinterpret_cast<float4 *>(xs + (size_t)idx * 4) = v;
            }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) *reinterpret_cast<float4 *>(gs + (size_t)(tid + 256 * i) * 4) = pg[i];
    };

    if (t_begin < t_end) load_tile(t_begin);
    for (int tile = t_begin; tile < t_end; ++tile) {
        __syncthreads();
        store_tile();
        __syncthreads();
        if (tile + 1 < t_end) load_tile(tile + 1);
        // K = the 32 pixels of tile row `wave`, two per step (lane half h = pixel parity)
#pragma unroll 4
        for (int s = 0; s < 16; ++s) {
            const int col = 2 * s + h;
            const float bv = gs[(wave * 32 + col) * 32 + r];
#pragma unroll
            for (int t = 0; t < TAPS; ++t) {
                const int ky = TAPS == 9 ? t / 3 : 0, kx = TAPS == 9 ? t % 3 : 0;
                const float av = xs[((wave * S + ky) * a.PW + col * S + kx) * 32 + r];
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[t], 0, 0, 0);
            }
        }
    }
    // ---- sum the four waves (fixed order) and leave the group's partial: D[i = ci][j = co], j = lane & 31, i = (e & 3) + 8 (e >> 2) + 4 h
    float *red = pn_lds;       // [wave][16][64]
#pragma unroll
    for (int t = 0; t < TAPS; ++t) {
        __syncthreads();
#pragma unroll
        for (int e = 0; e < 16; ++e) red[(wave * 16 + e) * 64 + lane] = acc[t][e];
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int idx = tid + 256 * q, e = idx >> 6, l = idx & 63;
            const float v = ((red[(0 * 16 + e) * 64 + l] + red[(1 * 16 + e) * 64 + l]) + red[(2 * 16 + e) * 64 + l]) + red[(3 * 16 + e) * 64 + l];
            const int ci = ci0 + (e & 3) + 8 * (e >> 2) + 4 * (l >> 5), co = co0 + (l & 31);
            a.part[(((size_t)blockIdx.x * a.ntaps + t) * a.Cx + ci) * a.Cy + co] = v;
        }
    }
}

// G[i] = sum over groups of part[group][i] in a FIXED order: eight slices of consecutive groups per element (a thread each, eight loads in
// flight), the slices joined through the LDS
__global__ __launch_bounds__(256) void k_pn_reduce_groups(float *__restrict__ out, const float *__restrict__ part, int n, int groups)
{
    __shared__ float red[8][32];
    const int e = threadIdx.x & 31, slice = threadIdx.x >> 5;
    const int i = blockIdx.x * 32 + e;
    const int per = (groups + 7) / 8, g0 = slice * per, g1 = min(groups, g0 + per);
    float s = 0.f;
    if (i < n) {
        int g = g0;
        for (; g + 8 <= g1; g += 8) {
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = part[(size_t)(g + j) * n + i];
#pragma unroll
            for (int j = 0; j < 8; ++j) s += v[j];
        }
        for (; g < g1; ++g) s += part[(size_t)g * n + i];
    }
    red[slice][e] = s;
    __syncthreads();
    if (slice == 0 && i < n) {
        float t = red[0][e];
#pragma unroll
        for (int j = 1; j < 8; ++j) t += red[j][e];
        out[i] = t;
    }
}

// ------------------------------------------------------------------------------------------------------------------ BatchNorm
// training-mode statistics from the conv's per-workgroup partial sums [n_part][2][C]: mean, biased variance -> scale / shift for the
// consumers' prologue, saved mean / invstd for the backward, running statistics (unbiased variance, momentum) and the batch counter
__global__ __launch_bounds__(1024) void k_pn_bn_finalize(const float *__restrict__ part, int n_part, int C, double count, const float *__restrict__ gamma,
                                                         const float *__restrict__ beta, float eps, float momentum, float *__restrict__ running_mean,
                                                         float *__restrict__ running_var, long long *__restrict__ batches, float *__restrict__ scale,
                                                         float *__restrict__ shift, float *__restrict__ save_mean, float *__restrict__ save_invstd)
{
    __shared__ double red[2][1024];
    const int tid = threadIdx.x, c = tid % C, slice = tid / C, n_slices = 1024 / C;
    double s1 = 0.0, s2 = 0.0;
    {
        int p = slice;
        for (; p + 7 * n_slices < n_part; p += 8 * n_slices) {      // eight rows in flight per thread
            float v1[8], v2[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                v1[j] = part[((size_t)(p + j * n_slices) * 2 + 0) * C + c];
                v2[j] = part[((size_t)(p + j * n_slices) * 2 + 1) * C + c];
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) { s1 += (double)v1[j]; s2 += (double)v2[j]; }
        }
        for (; p < n_part; p += n_slices) {
            s1 += (double)part[((size_t)p * 2 + 0) * C + c];
            s2 += (double)part[((size_t)p * 2 + 1) * C + c];
        }
    }
    red[0][tid] = s1; red[1][tid] = s2;
    __syncthreads();
    if (tid < C) {
        for (int s = 1; s < n_slices; ++s) { s1 += red[0][s * C + c]; s2 += red[1][s * C + c]; }
        const double mean = s1 / count;
        double var = s2 / count - mean * mean;
        var = var > 0.0 ? var : 0.0;
        const float invstd = (float)(1.0 / sqrt(var + (double)eps));
        const float g = gamma ? gamma[c] : 1.0f, b = beta ? beta[c] : 0.0f;
        const float sc = g * invstd;
        scale[c] = sc;
        shift[c] = b - (float)mean * sc;
        save_mean[c] = (float)mean;
        save_invstd[c] = invstd;
        if (running_mean) {
            const double unbiased = count > 1.0 ? var * count / (count - 1.0) : var;
            running_mean[c] = (1.0f - momentum) * running_mean[c] + momentum * (float)mean;
            running_var[c] = (1.0f - momentum) * running_var[c] + momentum * (float)unbiased;
        }
        if (batches && tid == 0) *batches += 1;
    }
}

// out = relu(za * sa + ta + B):  B = zb (mode 0) | zb * sb + tb (mode 1) | relu(zb * sb + tb) (mode 2)
__global__ __launch_bounds__(256) void k_pn_join(float4 *__restrict__ out, const float4 *__restrict__ za, const float *__restrict__ sa,
                                                 const float *__restrict__ ta, const float4 *__restrict__ zb, const float *__restrict__ sb,
                                                 const float *__restrict__ tb, int mode, int C4, long long total4)
{
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total4) return;
    const int c = (int)(i % C4) * 4;
    const float4 a = za[i], b = zb[i];
    const float4 s = *reinterpret_cast<const float4 *>(sa + c), t = *reinterpret_cast<const float4 *>(ta + c);
    float4 y = make_float4(fmaf(a.x, s.x, t.x), fmaf(a.y, s.y, t.y), fmaf(a.z, s.z, t.z), fmaf(a.w, s.w, t.w));
    float4 v = b;
    if (mode >= 1) {
        const float4 s2 = *reinterpret_cast<const float4 *>(sb + c), t2 = *reinterpret_cast<const float4 *>(tb + c);
        v = make_float4(fmaf(b.x, s2.x, t2.x), fmaf(b.y, s2.y, t2.y), fmaf(b.z, s2.z, t2.z), fmaf(b.w, s2.w, t2.w));
        if (mode == 2) v = make_float4(fmaxf(v.x, 0.f), fmaxf(v.y, 0.f), fmaxf(v.z, 0.f), fmaxf(v.w, 0.f));
    }
    out[i] = make_float4(fmaxf(y.x + v.x, 0.f), fmaxf(y.y + v.y, 0.f), fmaxf(y.z + v.z, 0.f), fmaxf(y.w + v.w, 0.f));
}

// the same with the coefficients derived from the producers' fixed-point sums (no finalize launch in between; PnBnSrc above)
__global__ __launch_bounds__(256) void k_pn_join_acc(float4 *__restrict__ out, const float4 *__restrict__ za, PnBnSrc ba, const float4 *__restrict__ zb,
                                                     PnBnSrc bb, int mode, int C4, long long total4)
{
    __shared__ __align__(16) float coef[4 * PN_COEF_MAX];       // [scale a | shift a | scale b | shift b]
    for (int c = threadIdx.x; c < 4 * C4; c += 256) {
        pn_bn_coef(ba, c, coef[c], coef[PN_COEF_MAX + c]);
        if (mode >= 1) pn_bn_coef(bb, c, coef[2 * PN_COEF_MAX + c], coef[3 * PN_COEF_MAX + c]);
    }
    __syncthreads();
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total4; i += (long long)gridDim.x * 256) {
        const int c = (int)(i % C4) * 4;
        const float4 a = za[i], b = zb[i];
        const float4 s = *reinterpret_cast<const float4 *>(coef + c), t = *reinterpret_cast<const float4 *>(coef + PN_COEF_MAX + c);
        float4 y = make_float4(fmaf(a.x, s.x, t.x), fmaf(a.y, s.y, t.y), fmaf(a.z, s.z, t.z), fmaf(a.w, s.w, t.w));
        float4 v = b;
        if (mode >= 1) {
            const float4 s2 = *reinterpret_cast<const float4 *>(coef + 2 * PN_COEF_MAX + c), t2 = *reinterpret_cast<const float4 *>(coef + 3 * PN_COEF_MAX + c);
            v = make_float4(fmaf(b.x, s2.x, t2.x), fmaf(b.y, s2.y, t2.y), fmaf(b.z, s2.z, t2.z), fmaf(b.w, s2.w, t2.w));
            if (mode == 2) v = make_float4(fmaxf(v.x, 0.f), fmaxf(v.y, 0.f), fmaxf(v.z, 0.f), fmaxf(v.w, 0.f));
        }
        out[i] = make_float4(fmaxf(y.x + v.x, 0.f), fmaxf(y.y + v.y, 0.f), fmaxf(y.z + v.z, 0.f), fmaxf(y.w + v.w, 0.f));
    }
}

// end of a forward pass: every BatchNorm layer's arrays for the backward pass and for bookkeeping (scale / shift / mean / invstd, running statistics,
// batch counter) from its accumulators, which are then ZEROED for the next pass -- one launch for all layers (grid = layers)
struct PnBnLayer {
    unsigned long long *acc;      // [PN_ACC_R][2][C]
    const float *gamma, *beta;
    float *running_mean, *running_var;
    long long *batches;
    float *scale, *shift, *mean, *invstd;
    double count;
    float eps, momentum;
    int C, pad;
};
__global__ __launch_bounds__(PN_COEF_MAX) void k_pn_bn_finalize_acc(const PnBnLayer *__restrict__ layers)
{
    const PnBnLayer L = layers[blockIdx.x];
    const int c = threadIdx.x;
    if (c >= L.C) return;
    const PnBnSrc b{L.acc, L.gamma, L.beta, L.count, L.eps, L.C};
    float sc, sh, mean, invstd;
    double var;
    pn_bn_coef(b, c, sc, sh, &mean, &invstd, &var);
    L.scale[c] = sc; L.shift[c] = sh; L.mean[c] = mean; L.invstd[c] = invstd;
    if (L.running_mean) {
        const double unbiased = L.count > 1.0 ? var * L.count / (L.count - 1.0) : var;
        L.running_mean[c] = (1.0f - L.momentum) * L.running_mean[c] + L.momentum * mean;
        L.running_var[c] = (1.0f - L.momentum) * L.running_var[c] + L.momentum * (float)unbiased;
    }
    if (L.batches && c == 0) *L.batches += 1;
#pragma unroll
    for (int r = 0; r < PN_ACC_R; ++r) { L.acc[(size_t)r * 2 * L.C + c] = 0ull; L.acc[(size_t)r * 2 * L.C + L.C + c] = 0ull; }
}

// backward of training-mode BatchNorm, pass 1: per-workgroup partial sums of g_m and g_m * xhat, g_m = g * mask
//   mask_mode 0: none; 1: own output  z * scale + shift > 0  (BN -> ReLU); 2: external map m > 0 (the block output after the residual join)
struct PnBnBwdArgs {
    const float *g, *z, *mask;
    const float *scale, *shift, *mean, *invstd;
    float *part;        // [n_part][2][C]
    long long pixels;
    int C, mask_mode, pix_per_wg;
};

__device__ __forceinline__ float4 pn_masked(const PnBnBwdArgs &a, long long o, int c, const float4 &z)
{
    float4 g = *reinterpret_cast<const float4 *>(a.g + o);
    if (a.mask_mode == 1) {
        const float4 s = *reinterpret_cast<const float4 *>(a.scale + c), t = *reinterpret_cast<const float4 *>(a.shift + c);
        g.x = fmaf(z.x, s.x, t.x) > 0.f ? g.x : 0.f; g.y = fmaf(z.y, s.y, t.y) > 0.f ? g.y : 0.f;
        g.z = fmaf(z.z, s.z, t.z) > 0.f ? g.z : 0.f; g.w = fmaf(z.w, s.w, t.w) > 0.f ? g.w : 0.f;
    } else if (a.mask_mode == 2) {
        const float4 m = *reinterpret_cast<const float4 *>(a.mask + o);
        g.x = m.x > 0.f ? g.x : 0.f; g.y = m.y > 0.f ? g.y : 0.f; g.z = m.z > 0.f ? g.z : 0.f; g.w = m.w > 0.f ? g.w : 0.f;
    }
    return g;
}

__global__ __launch_bounds__(256) void k_pn_bn_bwd_reduce(const PnBnBwdArgs a)
{
    __shared__ float red[2][256][4];
    const int C4 = a.C / 4, tid = threadIdx.x, q = tid % C4, slot = tid / C4, slots = 256 / C4, c = 4 * q;
    const long long p0 = (long long)blockIdx.x * a.pix_per_wg, p1 = min(a.pixels, p0 + a.pix_per_wg);
    const float4 mu = *reinterpret_cast<const float4 *>(a.mean + c), is = *reinterpret_cast<const float4 *>(a.invstd + c);
    float4 s1 = make_float4(0.f, 0.f, 0.f, 0.f), s2 = s1;
    for (long long p = p0 + slot; p < p1; p += slots) {
        const long long o = p * a.C + c;
        const float4 z = *reinterpret_cast<const float4 *>(a.z + o);
        const float4 g = pn_masked(a, o, c, z);
        s1.x += g.x; s1.y += g.y; s1.z += g.z; s1.w += g.w;
        s2.x = fmaf(g.x, (z.x - mu.x) * is.x, s2.x); s2.y = fmaf(g.y, (z.y - mu.y) * is.y, s2.y);
        s2.z = fmaf(g.z, (z.z - mu.z) * is.z, s2.z); s2.w = fmaf(g.w, (z.w - mu.w) * is.w, s2.w);
    }
    red[0][tid][0] = s1.x; red[0][tid][1] = s1.y; red[0][tid][2] = s1.z; red[0][tid][3] = s1.w;
    red[1][tid][0] = s2.x; red[1][tid][1] = s2.y; red[1][tid][2] = s2.z; red[1][tid][3] = s2.w;
    __syncthreads();
    if (tid < 2 * a.C) {
        const int which = tid / a.C, cc = tid % a.C;
        float v = 0.f;
        for (int s = 0; s < slots; ++s) v += red[which][s * C4 + cc / 4][cc % 4];
        a.part[((size_t)blockIdx.x * 2 + which) * a.C + cc] = v;
    }
}

// pass 2: dgamma, dbeta (into the flat gradient buffer) and the coefficients of  gz = A g_m + B z + D
__global__ __launch_bounds__(1024) void k_pn_bn_bwd_finalize(const float *__restrict__ part, int n_part, int C, double count, const float *__restrict__ gamma,
                                                             const float *__restrict__ mean, const float *__restrict__ invstd, float *__restrict__ dgamma,
                                                             float *__restrict__ dbeta, float *__restrict__ coef /* [3][C] */)
{
    __shared__ double red[2][1024];
    const int tid = threadIdx.x, c = tid % C, slice = tid / C, n_slices = 1024 / C;
    double s1 = 0.0, s2 = 0.0;
    {
        int p = slice;
        for (; p + 7 * n_slices < n_part; p += 8 * n_slices) {      // eight rows in flight per thread
            float v1[8], v2[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                v1[j] = part[((size_t)(p + j * n_slices) * 2 + 0) * C + c];
                v2[j] = part[((size_t)(p + j * n_slices) * 2 + 1) * C + c];
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) { s1 += (double)v1[j]; s2 += (double)v2[j]; }
        }
        for (; p < n_part; p += n_slices) {
            s1 += (double)part[((size_t)p * 2 + 0) * C + c];
            s2 += (double)part[((size_t)p * 2 + 1) * C + c];
        }
    }
    red[0][tid] = s1; red[1][tid] = s2;
    __syncthreads();
    if (tid < C) {
        for (int s = 1; s < n_slices; ++s) { s1 += red[0][s * C + c]; s2 += red[1][s * C + c]; }
        const double g = gamma ? (double)gamma[c] : 1.0, is = (double)invstd[c], mu = (double)mean[c];
        const double A = g * is, B = -A * is * s2 / count, D = -A * s1 / count - B * mu;
        if (dgamma) dgamma[c] = (float)s2;
        if (dbeta) dbeta[c] = (float)s1;
        coef[c] = (float)A; coef[C + c] = (float)B; coef[2 * C + c] = (float)D;
    }
}

// pass 3: gz = A g_m + B z + D
__global__ __launch_bounds__(256) void k_pn_bn_bwd_apply(const PnBnBwdArgs a, const float *__restrict__ coef, float *__restrict__ gz)
{
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x, total4 = a.pixels * (a.C / 4);
    if (i >= total4) return;
    const int c = (int)(i % (a.C / 4)) * 4;
    const long long o = i * 4;
    const float4 z = *reinterpret_cast<const float4 *>(a.z + o);
    const float4 g = pn_masked(a, o, c, z);
    const float4 A = *reinterpret_cast<const float4 *>(coef + c), B = *reinterpret_cast<const float4 *>(coef + a.C + c), D = *reinterpret_cast<const float4 *>(coef + 2 * a.C + c);
    *reinterpret_cast<float4 *>(gz + o) = make_float4(fmaf(A.x, g.x, fmaf(B.x, z.x, D.x)), fmaf(A.y, g.y, fmaf(B.y, z.y, D.y)),
                                                      fmaf(A.z, g.z, fmaf(B.z, z.z, D.z)), fmaf(A.w, g.w, fmaf(B.w, z.w, D.w)));
}

// ------------------------------------------------------------------------------------------------------------------ last head stage
// 3x3 / stride 2 / pad 1 conv to ONE channel with bias (the tile logits): input a = relu(z * scale + shift) [N][Hi][Wi][C]
__global__ __launch_bounds__(64) void k_pn_head_fwd(float *__restrict__ logits, const float *__restrict__ z, const float *__restrict__ scale,
                                                    const float *__restrict__ shift, const float *__restrict__ w /* [9][C] */, const float *__restrict__ bias,
                                                    int N, int Hi, int Wi, int C, int Ho, int Wo)
{
    const int o = blockIdx.x, ox = o % Wo, oy = (o / Wo) % Ho, n = o / (Wo * Ho), lane = threadIdx.x;
    float s = 0.f;
    for (int t = 0; t < 9; ++t) {
        const int iy = 2 * oy + t / 3 - 1, ix = 2 * ox + t % 3 - 1;
        if (iy < 0 || iy >= Hi || ix < 0 || ix >= Wi) continue;
        const float *zp = z + (((size_t)n * Hi + iy) * Wi + ix) * C;
        for (int c = lane; c < C; c += 64) s = fmaf(fmaxf(fmaf(zp[c], scale[c], shift[c]), 0.f), w[t * C + c], s);
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) s += __shfl_xor(s, d);
    if (lane == 0) logits[o] = s + (bias ? bias[0] : 0.f);
}

__global__ __launch_bounds__(64) void k_pn_head_fwd_acc(float *__restrict__ logits, const float *__restrict__ z, PnBnSrc bn, const float *__restrict__ w /* [9][C] */,
                                                        const float *__restrict__ bias, int N, int Hi, int Wi, int C, int Ho, int Wo)
{
    __shared__ float coef[2 * PN_COEF_MAX];
    const int o = blockIdx.x, ox = o % Wo, oy = (o / Wo) % Ho, n = o / (Wo * Ho), lane = threadIdx.x;
    for (int c = lane; c < C; c += 64) pn_bn_coef(bn, c, coef[c], coef[PN_COEF_MAX + c]);
    __syncthreads();
    float s = 0.f;
    for (int t = 0; t < 9; ++t) {
        const int iy = 2 * oy + t / 3 - 1, ix = 2 * ox + t % 3 - 1;
        if (iy < 0 || iy >= Hi || ix < 0 || ix >= Wi) continue;
        const float *zp = z + (((size_t)n * Hi + iy) * Wi + ix) * C;
        for (int c = lane; c < C; c += 64) s = fmaf(fmaxf(fmaf(zp[c], coef[c], coef[PN_COEF_MAX + c]), 0.f), w[t * C + c], s);
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) s += __shfl_xor(s, d);
    if (lane == 0) logits[o] = s + (bias ? bias[0] : 0.f);
}

// its backward: ga[n][iy][ix][c] = sum over (output, tap) hitting the pixel of gl * w;  dW[t][c] = sum_o gl[o] * a[o, t][c];  db = sum gl
__global__ __launch_bounds__(256) void k_pn_head_bwd_data(float *__restrict__ ga, const float *__restrict__ gl, const float *__restrict__ w, int N, int Hi,
                                                          int Wi, int C, int Ho, int Wo)
{
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x, total = (long long)N * Hi * Wi * C;
    if (i >= total) return;
    const int c = (int)(i % C);
    long long p = i / C;
    const int ix = (int)(p % Wi);
    p /= Wi;
    const int iy = (int)(p % Hi), n = (int)(p / Hi);
    float s = 0.f;
    for (int ky = 0; ky < 3; ++ky) {
        const int ty = iy + 1 - ky;
        if (ty < 0 || (ty & 1) || ty / 2 >= Ho) continue;
        for (int kx = 0; kx < 3; ++kx) {
            const int tx = ix + 1 - kx;
            if (tx < 0 || (tx & 1) || tx / 2 >= Wo) continue;
            s = fmaf(gl[((size_t)n * Ho + ty / 2) * Wo + tx / 2], w[(ky * 3 + kx) * C + c], s);
        }
    }
    ga[i] = s;
}

__global__ __launch_bounds__(1024) void k_pn_head_bwd_weight(float *__restrict__ dw /* [9][C] */, float *__restrict__ db, const float *__restrict__ gl,
                                                             const float *__restrict__ z, const float *__restrict__ scale, const float *__restrict__ shift, int N,
                                                             int Hi, int Wi, int C, int Ho, int Wo)
{
    // block t < 9: tap t, thread = (channel, slice of the outputs); block 9: the bias.  Slices joined in a fixed order through the LDS.
    __shared__ float red[1024];
    const int t = blockIdx.x, tid = threadIdx.x, n_out = N * Ho * Wo;
    if (t == 9) {
        float s = 0.f;
        for (int o = tid; o < n_out; o += 1024) s += gl[o];
        red[tid] = s;
        __syncthreads();
        if (tid == 0 && db) {
            float v = 0.f;
            for (int j = 0; j < 1024; ++j) v += red[j];
            db[0] = v;
        }
        return;
    }
    const int c = tid % C, slice = tid / C, slices = 1024 / C;
    const float sc = scale[c], sh = shift[c];
    float s = 0.f;
    for (int o = slice; o < n_out; o += slices) {
        const int ox = o % Wo, oy = (o / Wo) % Ho, n = o / (Wo * Ho);
        const int iy = 2 * oy + t / 3 - 1, ix = 2 * ox + t % 3 - 1;
        if (iy < 0 || iy >= Hi || ix < 0 || ix >= Wi) continue;
        const float av = fmaxf(fmaf(z[(((size_t)n * Hi + iy) * Wi + ix) * C + c], sc, sh), 0.f);
        s = fmaf(gl[o], av, s);
    }
    red[tid] = s;
    __syncthreads();
    if (tid < C) {
        float v = red[tid];
        for (int j = 1; j < slices; ++j) v += red[j * C + tid];
        dw[t * C + tid] = v;
    }
}

// ------------------------------------------------------------------------------------------------------------------ reward, loss seed
// information gain of a semantic-segmentation output (information_gain.py:22-41): bilinear resampling (ATen upsample_bilinear2d index
// arithmetic, align_corners = False, scale = 1 / scale_factor) of both logit maps, log-softmax over the classes, KL(prev || cur) averaged
// over the classes.  One thread per output pixel; maps are read through their element strides (NCHW or channels-last).
struct PnIgArgs {
    const void *cur, *prev;
    int dtype;          // BC_F32 / BC_F16 / BC_BF16 element type of the two maps (arithmetic in fp32)
    float *ig;          // [N][h][w]
    long long sn, sc, sh, sw;
    int N, C, H, W, h, w;
    float rh, rw;
};

__device__ __forceinline__ void pn_src(float scale, int dst, int size, int &i0, int &i1, float &l0, float &l1)
{
    const float s = fmaxf(scale * ((float)dst + 0.5f) - 0.5f, 0.0f);
    i0 = (int)s;
    i0 = i0 < size - 1 ? i0 : size - 1;
    i1 = i0 < size - 1 ? i0 + 1 : i0;
    l1 = s - (float)i0;
    l1 = l1 < 0.f ? 0.f : (l1 > 1.f ? 1.f : l1);
    l0 = 1.0f - l1;
}


// half a wave per output pixel, a class per lane: the 8 taps of a lane are requested together, maximum / normaliser / divergence are 32-lane
// reductions (C <= 32)
__global__ __launch_bounds__(256) void k_pn_infogain(const PnIgArgs a)
{
    const int lane = threadIdx.x & 31;
    const int i = (blockIdx.x * 256 + threadIdx.x) >> 5;
    const bool live = i < a.N * a.h * a.w;
    const int ii = live ? i : 0;
    const int x = ii % a.w, y = (ii / a.w) % a.h, n = ii / (a.w * a.h);
    int y0, y1, x0, x1;
    float ly0, ly1, lx0, lx1;
    pn_src(a.rh, y, a.H, y0, y1, ly0, ly1);
    pn_src(a.rw, x, a.W, x0, x1, lx0, lx1);
    const int c = lane < a.C ? lane : 0;
    const long long base = n * a.sn + c * a.sc;
    const long long b00 = base + y0 * a.sh + x0 * a.sw, b01 = base + y0 * a.sh + x1 * a.sw, b10 = base + y1 * a.sh + x0 * a.sw, b11 = base + y1 * a.sh + x1 * a.sw;
    auto ld = [&](const void *m, long long o) -> float {
        if (a.dtype == 0) return reinterpret_cast<const float *>(m)[o];
        if (a.dtype == 1) return __half2float(reinterpret_cast<const __half *>(m)[o]);
        return __uint_as_float((uint32_t)reinterpret_cast<const uint16_t *>(m)[o] << 16);
    };
    const float c00 = ld(a.cur, b00), c01 = ld(a.cur, b01), c10 = ld(a.cur, b10), c11 = ld(a.cur, b11);
    const float p00 = ld(a.prev, b00), p01 = ld(a.prev, b01), p10 = ld(a.prev, b10), p11 = ld(a.prev, b11);
    const bool on = lane < a.C;
    const float vc = ly0 * (lx0 * c00 + lx1 * c01) + ly1 * (lx0 * c10 + lx1 * c11);
    const float vp = ly0 * (lx0 * p00 + lx1 * p01) + ly1 * (lx0 * p10 + lx1 * p11);
    float mc = on ? vc : -INFINITY, mp = on ? vp : -INFINITY;
#pragma unroll
    for (int d = 16; d >= 1; d >>= 1) { mc = fmaxf(mc, __shfl_xor(mc, d)); mp = fmaxf(mp, __shfl_xor(mp, d)); }
    float sc = on ? expf(vc - mc) : 0.f, sp = on ? expf(vp - mp) : 0.f;
#pragma unroll
    for (int d = 16; d >= 1; d >>= 1) { sc += __shfl_xor(sc, d); sp += __shfl_xor(sp, d); }
    const float ls_c = vc - mc - logf(sc), ls_p = vp - mp - logf(sp);
    float kl = on ? expf(ls_p) * (ls_p - ls_c) : 0.f;
#pragma unroll
    for (int d = 16; d >= 1; d >>= 1) kl += __shfl_xor(kl, d);
    if (live && lane == 0) a.ig[i] = kl / (float)a.C;
}

// REINFORCE seed (policy.py:334-349): reward = adaptive_max_pool2d(ig + rc) with rc = -(cost - target) |cost - target| gamma, sign-flipped on
// skipped tiles; loss = mean(-log_prob * reward); d loss / d logit = (sigmoid(l) - g) * reward / n.  One workgroup.
__global__ __launch_bounds__(256) void k_pn_reward_seed(float *__restrict__ gl, float *__restrict__ loss, float *__restrict__ reward_out, const float *__restrict__ logits,
                                                        const uint8_t *__restrict__ grid, const float *__restrict__ ig, const double *__restrict__ cost_dev,
                                                        double cost_host, double target, double gamma, int N, int h, int w, int GH, int GW)
{
    __shared__ float red[256];
    const int tid = threadIdx.x, n_total = N * GH * GW;
    const double cost = cost_dev ? *cost_dev : cost_host;
    const double rr = -(cost - target);
    const float rc = (float)(rr * fabs(rr) * gamma);
    float part = 0.f;
    for (int i = tid; i < n_total; i += 256) {
        const int gx = i % GW, gy = (i / GW) % GH, n = i / (GW * GH);
        const int ys = (gy * h) / GH, ye = ((gy + 1) * h + GH - 1) / GH, xs = (gx * w) / GW, xe = ((gx + 1) * w + GW - 1) / GW;
        float m = -INFINITY;
        for (int y = ys; y < ye; ++y)
            for (int x = xs; x < xe; ++x) m = fmaxf(m, ig[((size_t)n * h + y) * w + x] + rc);
        const bool on = grid[i] != 0;
        const float rew = on ? m : -m;
        const float l = logits[i], g = on ? 1.f : 0.f;
        const float sg = 1.0f / (1.0f + expf(-l));
        const float bce = fmaxf(l, 0.f) - l * g + log1pf(expf(-fabsf(l)));
        gl[i] = (sg - g) * rew / (float)n_total;
        if (reward_out) reward_out[i] = rew;
        part += bce * rew;
    }
    red[tid] = part;
    __syncthreads();
    for (int d = 128; d >= 1; d >>= 1) {
        if (tid < d) red[tid] += red[tid + d];
        __syncthreads();
    }
    if (tid == 0 && loss) loss[0] = red[0] / (float)n_total;
}

// ------------------------------------------------------------------------------------------------------------------ optimizer, export
// torch.optim.RMSprop (centered = False) over the flat parameter buffer; same operation order as torch/optim/rmsprop.py _single_tensor_rmsprop
__global__ __launch_bounds__(256) void k_pn_rmsprop(float *__restrict__ p, const float *__restrict__ g, float *__restrict__ sq, float *__restrict__ mom, int n,
                                                    float lr, float alpha, float eps, float wd, float momentum)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float grad = g[i];
    const float w = p[i];
    if (wd != 0.f) grad = fmaf(w, wd, grad);
    const float s = sq[i] * alpha + (1.0f - alpha) * grad * grad;
    sq[i] = s;
    const float avg = sqrtf(s) + eps;
    if (momentum > 0.f) {
        const float b = mom[i] * momentum + grad / avg;
        mom[i] = b;
        p[i] = w - lr * b;
    } else {
        p[i] = w - lr * (grad / avg);
    }
}

// segment table of the flat buffer: conv weights [tap][Cin_pad][Cout] <-> a torch parameter (Cout, Cin, kh, kw) with arbitrary strides, plus the
// transposed copy WT[tap][Cout][Cin_pad] for the data gradient; vectors are plain copies.  dir 0: flat -> torch (+ transposed copy), 1: torch -> flat
struct PnSeg {
    long long off, off_t;       // offsets into the flat buffer / the transposed buffer (off_t < 0: none)
    float *param;
    long long s_co, s_ci, s_ky, s_kx;
    int taps, kw, cin, cin_pad, cout, numel;     // numel = taps * cin_pad * cout (weights) or the vector length (taps = 0)
    long long ws_off;           // bc_pn_update: the segment's gradient is the sum of `groups` partial copies at ws + ws_off (groups = 0: it is in G)
    int groups, pad_;
};

__global__ __launch_bounds__(256) void k_pn_sync_params(float *__restrict__ flat, float *__restrict__ flat_t, const PnSeg *__restrict__ segs, int dir)
{
    const PnSeg s = segs[blockIdx.y];
    for (int i = blockIdx.x * 256 + threadIdx.x; i < s.numel; i += gridDim.x * 256) {
        if (s.taps == 0) {
            if (dir == 0) s.param[i] = flat[s.off + i];
            else flat[s.off + i] = s.param[i];
            continue;
        }
        const int co = i % s.cout, ci = (i / s.cout) % s.cin_pad, t = i / (s.cout * s.cin_pad);
        const long long po = co * s.s_co + ci * s.s_ci + (t / s.kw) * s.s_ky + (t % s.kw) * s.s_kx;
        float v;
        if (dir == 0) {
            v = flat[s.off + i];
            if (ci < s.cin) s.param[po] = v;
        } else {
            v = ci < s.cin ? s.param[po] : 0.f;
            flat[s.off + i] = v;
        }
        if (s.off_t >= 0 && flat_t) flat_t[s.off_t + ((long long)t * s.cout + co) * s.cin_pad + ci] = v;
    }
}

// The tail of a training step in ONE launch: the weight gradients' group partials summed in a fixed order (the reduction of k_pn_reduce_groups:
// eight slices of consecutive groups per element, joined through the LDS), torch.optim.RMSprop on the flat parameters, and the export of the
// new values into the module's parameter tensors and the transposed copies the data gradient reads (k_pn_rmsprop + k_pn_sync_params).
__global__ __launch_bounds__(256) void k_pn_update(float *__restrict__ p, float *__restrict__ g, float *__restrict__ sq, float *__restrict__ mom,
                                                   float *__restrict__ flat_t, const float *__restrict__ ws, const PnSeg *__restrict__ segs, float lr, float alpha,
                                                   float eps, float wd, float momentum)
{
    __shared__ float red[8][32];
    const PnSeg s = segs[blockIdx.y];
    const int e = threadIdx.x & 31, slice = threadIdx.x >> 5;
    for (int base = blockIdx.x * 32; base < s.numel; base += gridDim.x * 32) {
        const int i = base + e;
        const bool live = i < s.numel;
        float acc = 0.f;
        if (s.groups > 0) {
            const int per = (s.groups + 7) / 8, g0 = slice * per, g1 = min(s.groups, g0 + per);
            if (live) {
                const float *src = ws + s.ws_off + i;
                int k = g0;
                for (; k + 8 <= g1; k += 8) {
                    float v[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) v[j] = src[(size_t)(k + j) * s.numel];
#pragma unroll
                    for (int j = 0; j < 8; ++j) acc += v[j];
                }
                for (; k < g1; ++k) acc += src[(size_t)k * s.numel];
            }
            red[slice][e] = acc;
            __syncthreads();
        }
        if (slice == 0 && live) {
            float grad;
            if (s.groups > 0) {
                grad = red[0][e];
#pragma unroll
                for (int j = 1; j < 8; ++j) grad += red[j][e];
                g[s.off + i] = grad;
            } else {
                grad = g[s.off + i];
            }
            const float w = p[s.off + i];
            if (wd != 0.f) grad = fmaf(w, wd, grad);
            const float sv = sq[s.off + i] * alpha + (1.0f - alpha) * grad * grad;
            sq[s.off + i] = sv;
            const float avg = sqrtf(sv) + eps;
            float v;
            if (momentum > 0.f) {
                const float b = mom[s.off + i] * momentum + grad / avg;
                mom[s.off + i] = b;
                v = w - lr * b;
            } else {
                v = w - lr * (grad / avg);
            }
            p[s.off + i] = v;
            if (s.taps == 0) {
                s.param[i] = v;
            } else {
                const int co = i % s.cout, ci = (i / s.cout) % s.cin_pad, t = i / (s.cout * s.cin_pad);
                if (ci < s.cin) s.param[co * s.s_co + ci * s.s_ci + (t / s.kw) * s.s_ky + (t % s.kw) * s.s_kx] = v;
                if (s.off_t >= 0 && flat_t) flat_t[s.off_t + ((long long)t * s.cout + co) * s.cin_pad + ci] = v;
            }
        }
        if (s.groups > 0) __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------------------------ policy input (NHWC)
// The policy net's input (policy/net.py:82-113) as ONE gather straight into the channels-last, channel-padded layout the conv kernel reads:
// nearest-neighbour resampling of four sources (frame, frame state, previous output, previous grid), min((int)floorf(dst * scale), in - 1)
// as ATen's legacy 'nearest', offsets folded in, channels >= sum(C) zero.
struct PnFeatSrc {
    const void *ptr;
    long long sn, sc, sh, sw;
    int C, H, W, dtype;
    float scale_h, scale_w, offset;
};
struct PnFeatGeom {
    PnFeatSrc src[4];
    int N, h, w, Cpad;
};

__device__ __forceinline__ float pn_feat_load(const PnFeatSrc &s, long long off)
{
    switch (s.dtype) {
    case 0: return reinterpret_cast<const float *>(s.ptr)[off];
    case 1: return __half2float(reinterpret_cast<const __half *>(s.ptr)[off]);
    case 2: return __uint_as_float((uint32_t)reinterpret_cast<const uint16_t *>(s.ptr)[off] << 16);
    default: return reinterpret_cast<const uint8_t *>(s.ptr)[off] ? 1.0f : 0.0f;
    }
}

// a lane per (output pixel, channel quad): the Cpad / 4 quads of a pixel sit in neighbouring lanes, so the stores of a wave are whole 16-byte
// runs of consecutive pixels and every lane has at most four independent loads in flight (a lane per pixel looping over 26 channels was
// latency-bound: 37 us for the 8 MB policy input of a batch of two frames; the sources it samples are 25 MB)
__global__ __launch_bounds__(256) void k_pn_features(float *__restrict__ out, const PnFeatGeom g)
{
    const int Cq = g.Cpad >> 2;
    const long long total = (long long)g.N * g.h * g.w * Cq;
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int q = (int)(i % Cq);
    const long long pix = i / Cq;
    const int x = (int)(pix % g.w);
    const long long p = pix / g.w;
    const int y = (int)(p % g.h), n = (int)(p / g.h);
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    int c0 = 0;                                       // first channel of source k in the concatenation
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const PnFeatSrc &s = g.src[k];
        if (4 * q + 3 >= c0 && 4 * q < c0 + s.C) {    // the quad holds channels of this source
            int sy = (int)floorf((float)y * s.scale_h), sx = (int)floorf((float)x * s.scale_w);
            sy = sy < s.H - 1 ? sy : s.H - 1;
            sx = sx < s.W - 1 ? sx : s.W - 1;
            const long long base = n * s.sn + sy * s.sh + sx * s.sw;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int cc = 4 * q + j - c0;
                if (cc >= 0 && cc < s.C) v[j] = pn_feat_load(s, base + cc * s.sc) + s.offset;
            }
        }
        c0 += s.C;
    }
    reinterpret_cast<float4 *>(out + pix * g.Cpad)[q] = make_float4(v[0], v[1], v[2], v[3]);
}

// decision bookkeeping of a frame (policy.py:283-288 via torch.distributions.Bernoulli): probs = sigmoid(l), log_prob(grid) = -BCE-with-logits
__global__ __launch_bounds__(256) void k_pn_probs(float *__restrict__ probs, float *__restrict__ log_probs, const float *__restrict__ logits,
                                                  const uint8_t *__restrict__ grid, int n)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float l = logits[i], g = grid[i] ? 1.f : 0.f;
    probs[i] = 1.0f / (1.0f + expf(-l));
    log_probs[i] = -(fmaxf(l, 0.f) - l * g + log1pf(expf(-fabsf(l))));
}

bool pn_aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// one-shot, per host thread: BatchNorm sources / sink of the NEXT bc_pn_conv_nhwc (slot 0: its input's BatchNorm, slot 2: accumulators for its
// output), bc_pn_join_acc / bc_pn_head_fwd_acc take theirs as arguments
thread_local PnBnSrc g_pn_arm_in = {nullptr, nullptr, nullptr, 0.0, 0.f, 0};
thread_local unsigned long long *g_pn_arm_out = nullptr;

}  // namespace

// ====================================================================================================================== C ABI
// direction 0: y = conv(x) forward.  x (N,Hx,Wx,Cx) -> out (N,Hy,Wy,Cy), w = W[tap][Cx][Cy]
// direction 1: data gradient.        x = gz (N,Hy,Wy,Cy) -> out = gx (N,Hx,Wx,Cx), w = WT[tap][Cy][Cx]
BC_EXPORT int bc_pn_conv_nhwc(float *out, const float *x, const float *w, int N, int Hx, int Wx, int Cx, int Hy, int Wy, int Cy, int ks, int stride,
                              int direction, const float *in_scale, const float *in_shift, int in_relu, const float *add, const float *add_mask,
                              int accumulate, float *stats, long long stats_capacity, int precision, void *stream)
{
    // (one shot: whatever happens below, an armed BatchNorm source / sink is consumed by THIS call and never leaks into a later one)
    const PnBnSrc arm_in = g_pn_arm_in;
    unsigned long long *const arm_out = g_pn_arm_out;
    g_pn_arm_in = PnBnSrc{nullptr, nullptr, nullptr, 0.0, 0.f, 0}; g_pn_arm_out = nullptr;
    if (precision < 0 || precision > 2) return BC_ERR_SHAPE;
    if (!out || !x || !w) return BC_ERR_NULL;
    if (N <= 0 || Hx <= 0 || Wx <= 0 || Hy <= 0 || Wy <= 0 || Cx <= 0 || Cy <= 0) return BC_ERR_SHAPE;
    if (!(ks == 3 || ks == 1) || !(stride == 1 || stride == 2) || !(direction == 0 || direction == 1)) return BC_ERR_SHAPE;
    if (ks == 1 && stride != 2) return BC_ERR_SHAPE;      // (the net's only pointwise convs are the stride-2 shortcuts)
    const int pad = ks == 3 ? 1 : 0;
    if (Hy != (Hx + 2 * pad - ks) / stride + 1 || Wy != (Wx + 2 * pad - ks) / stride + 1) return BC_ERR_SHAPE;
    if ((in_scale == nullptr) != (in_shift == nullptr)) return BC_ERR_NULL;
    if (!pn_aligned16(out) || !pn_aligned16(x) || !pn_aligned16(w) || (in_scale && (!pn_aligned16(in_scale) || !pn_aligned16(in_shift)))) return BC_ERR_ALIGN;
    PnConvArgs a;
    memset(&a, 0, sizeof(a));
    a.x = x; a.w = w; a.out = out; a.in_scale = in_scale; a.in_shift = in_shift; a.add = add; a.add_mask = add_mask; a.stats = stats;
    a.in_bn = arm_in; a.out_acc = arm_out;
    if (a.in_bn.acc && (direction != 0 || a.in_bn.C != Cx || Cx > PN_COEF_MAX || in_scale)) return BC_ERR_SHAPE;
    if (a.out_acc && direction != 0) return BC_ERR_SHAPE;
    a.N = N; a.in_relu = in_relu; a.accumulate = accumulate;
    {
        static const int dbg = [] { const char *e = getenv("PN_DBG"); return e ? atoi(e) : 0; }();
        a.dbg = dbg;
        a.stamps = g_pn_stamps;
    }
    int S = 1, max_hc = 0, max_wc = 0, max_npix = 0;
    if (direction == 0) {
        a.Hi = Hx; a.Wi = Wx; a.K = Cx; a.Nn = Cy; a.Hout = Hy; a.Wout = Wy; a.S = S = stride; a.out_s = 1; a.n_cls = 1;
        PnTapSet &c = a.cls[0];
        c.ntaps = ks * ks; c.Hc = Hy; c.Wc = Wy; c.out_oy = 0; c.out_ox = 0;
        for (int t = 0; t < c.ntaps; ++t) { c.dy[t] = (int8_t)(t / ks - pad); c.dx[t] = (int8_t)(t % ks - pad); c.wt[t] = (int8_t)t; }
    } else {
        a.Hi = Hy; a.Wi = Wy; a.K = Cy; a.Nn = Cx; a.Hout = Hx; a.Wout = Wx; a.S = 1; a.out_s = stride; a.n_cls = stride * stride;
        for (int py = 0; py < stride; ++py)
            for (int px = 0; px < stride; ++px) {
                PnTapSet &c = a.cls[py * stride + px];
                c.ntaps = 0; c.out_oy = py; c.out_ox = px;
                c.Hc = (Hx - py + stride - 1) / stride; c.Wc = (Wx - px + stride - 1) / stride;
                for (int ky = 0; ky < ks; ++ky)
                    for (int kx = 0; kx < ks; ++kx) {
                        // x pixel (Y, X) receives gz[(Y + pad - ky) / stride][(X + pad - kx) / stride] through tap (ky, kx) when both divide
                        const int ny = py + pad - ky, nx = px + pad - kx;
                        if (((ny % stride) + stride) % stride != 0 || ((nx % stride) + stride) % stride != 0) continue;
                        const int dy = ny >= 0 ? ny / stride : -((-ny) / stride), dx = nx >= 0 ? nx / stride : -((-nx) / stride);
                        c.dy[c.ntaps] = (int8_t)dy; c.dx[c.ntaps] = (int8_t)dx; c.wt[c.ntaps] = (int8_t)(ky * ks + kx);
                        ++c.ntaps;
                    }
            }
    }
    if (a.K % 32 != 0 || a.Nn % 32 != 0) return BC_ERR_SHAPE;
    for (int k = 0; k < a.n_cls; ++k) {
        max_hc = a.cls[k].Hc > max_hc ? a.cls[k].Hc : max_hc;
        max_wc = a.cls[k].Wc > max_wc ? a.cls[k].Wc : max_wc;
    }
    const bool ts = pn_tap_split(N, max_hc, max_wc, a.Nn, a.n_cls);
    const int TH = ts ? 1 : PN_TH;
    for (int k = 0; k < a.n_cls; ++k) {
        PnTapSet &c = a.cls[k];
        int ymin = 0, ymax = 0, xmin = 0, xmax = 0;
        for (int t = 0; t < c.ntaps; ++t) {
            ymin = t == 0 ? c.dy[t] : (c.dy[t] < ymin ? c.dy[t] : ymin); ymax = t == 0 ? c.dy[t] : (c.dy[t] > ymax ? c.dy[t] : ymax);
            xmin = t == 0 ? c.dx[t] : (c.dx[t] < xmin ? c.dx[t] : xmin); xmax = t == 0 ? c.dx[t] : (c.dx[t] > xmax ? c.dx[t] : xmax);
        }
        c.dmin_y = ymin; c.dmin_x = xmin;
        c.PH = (TH - 1) * S + (ymax - ymin) + 1; c.PW = (PN_TW - 1) * S + (xmax - xmin) + 1;
        c.pw_magic = pn_magic(c.PW, c.PH * c.PW + 8);
        if (!c.pw_magic) return BC_ERR_SHAPE;
        max_npix = c.PH * c.PW > max_npix ? c.PH * c.PW : max_npix;
    }
    a.tiles_y = (max_hc + TH - 1) / TH; a.tiles_x = (max_wc + PN_TW - 1) / PN_TW;
    a.npix_pad = (max_npix + 7) / 8 * 8 + 1;
    const long long n_tiles = (long long)N * a.tiles_y * a.tiles_x;
    if (n_tiles > 0x7fffffffLL || (long long)N * a.Hi * a.Wi * a.K > 0x7fffffffLL || (long long)9 * a.K * a.Nn > 0x7fffffffLL) return BC_ERR_RANGE;     // (32-bit staging offsets)
    const long long n_wg = n_tiles < PN_PERSIST_WGS ? n_tiles : PN_PERSIST_WGS;
    if (stats && (direction != 0 || stats_capacity < n_wg * 2 * a.Nn)) return BC_ERR_SHAPE;      // (forward: one class, every workgroup leaves a partial)
    hipStream_t st = (hipStream_t)stream;
    // one 32-channel output block per workgroup and all nine taps of a channel chunk per stage: the small maps of the net (8 k, 2 k,
    // 512 pixels) get four times the workgroups, and a workgroup's K loop is Cin / KC stages
    // one-row tiles (small maps) take 32-channel chunks: their patch is small, and half as many stages means half as many exposed memory
    // round trips (a stage of a tap-split workgroup is shorter than the latency of its successor's loads); stride 2 with four-row tiles: the
    // patch is 2.9 x the stride-1 one -- 16-channel chunks, 112 KB double buffered
#define PN_GO(KC_, S_, TS_)                                                          \
    (precision == 1 ? pn_conv_launch<1, KC_, S_, TS_, 1>(a, st)                      \
                    : (precision == 2 ? pn_conv_launch<1, KC_, S_, TS_, 2>(a, st) : pn_conv_launch<1, KC_, S_, TS_, 0>(a, st)))
    // (32-input-channel layers as ONE 32-channel chunk with resident weights -- 90 KB, one workgroup per CU -- measured slower: 40-43 us against
    //  31-33 with two 16-channel chunks and two workgroups per CU)
    if (S == 1) return ts ? PN_GO(32, 1, true) : PN_GO(16, 1, false);
    return ts ? PN_GO(32, 2, true) : PN_GO(16, 2, false);
#undef PN_GO
}

// number of stats partial rows a forward launch of this geometry writes (rows of [2][Cy]; small maps run one-row tiles)
BC_EXPORT long long bc_pn_conv_partials(int N, int Hy, int Wy, int Cy)
{
    const int TH = pn_tap_split(N, Hy, Wy, Cy, 1) ? 1 : PN_TH;
    const long long n_tiles = (long long)N * ((Hy + TH - 1) / TH) * ((Wy + PN_TW - 1) / PN_TW);
    return n_tiles < PN_PERSIST_WGS ? n_tiles : PN_PERSIST_WGS;
}

// weight gradient dW[tap][Cx][Cy] = sum over pixels of prologue(x)[pixel + tap] * gz[pixel]: part (groups x taps x Cx x Cy) is workspace
BC_EXPORT int bc_pn_wgrad_nhwc(float *dw, float *part, long long part_capacity, const float *x, const float *gz, int N, int Hx, int Wx, int Cx, int Hy, int Wy,
                               int Cy, int ks, int stride, const float *in_scale, const float *in_shift, int in_relu, void *stream)
{
    if (!part || !x || !gz) return BC_ERR_NULL;      // (dw NULL: the partials stay in `part` for bc_pn_update)
    if (N <= 0 || Hx <= 0 || Wx <= 0 || Hy <= 0 || Wy <= 0 || Cx <= 0 || Cy <= 0 || Cx % 32 != 0 || Cy % 32 != 0) return BC_ERR_SHAPE;
    if (!(ks == 3 || ks == 1) || !(stride == 1 || stride == 2) || (ks == 1 && stride != 2)) return BC_ERR_SHAPE;
    const int pad = ks == 3 ? 1 : 0;
    if (Hy != (Hx + 2 * pad - ks) / stride + 1 || Wy != (Wx + 2 * pad - ks) / stride + 1) return BC_ERR_SHAPE;
    if ((in_scale == nullptr) != (in_shift == nullptr)) return BC_ERR_NULL;
    if (!pn_aligned16(x) || !pn_aligned16(gz) || (in_scale && (!pn_aligned16(in_scale) || !pn_aligned16(in_shift)))) return BC_ERR_ALIGN;
    PnWgradArgs a;
    memset(&a, 0, sizeof(a));
    a.x = x; a.gz = gz; a.part = part; a.in_scale = in_scale; a.in_shift = in_shift; a.in_relu = in_relu;
    a.N = N; a.Hx = Hx; a.Wx = Wx; a.Cx = Cx; a.Hy = Hy; a.Wy = Wy; a.Cy = Cy; a.S = stride; a.ntaps = ks * ks; a.pad = pad;
    a.tiles_y = (Hy + PN_TH - 1) / PN_TH; a.tiles_x = (Wy + PN_TW - 1) / PN_TW;
    a.n_tiles = N * a.tiles_y * a.tiles_x;
    a.PH = (PN_TH - 1) * stride + ks; a.PW = (PN_TW - 1) * stride + ks; a.npix = a.PH * a.PW;
    a.pw_magic = pn_magic(a.PW, a.npix + 8);
    if (!a.pw_magic) return BC_ERR_SHAPE;
    // groups: about one workgroup per CU over all (ci, co) blocks (more tiles per workgroup: the final reduction and the partials are paid once)
    const int blocks = (Cx / 32) * (Cy / 32);
    int groups = (256 + blocks - 1) / blocks;      // (512: measured the same)
    groups = groups < 1 ? 1 : (groups > a.n_tiles ? a.n_tiles : groups);
    a.tiles_per_group = (a.n_tiles + groups - 1) / groups;
    groups = (a.n_tiles + a.tiles_per_group - 1) / a.tiles_per_group;
    const long long n = (long long)a.ntaps * Cx * Cy;
    if (part_capacity < n * groups) return BC_ERR_SHAPE;
    size_t lds = ((size_t)a.npix * 32 + 128 * 32) * sizeof(float);
    if (lds < 4 * 16 * 64 * sizeof(float)) lds = 4 * 16 * 64 * sizeof(float);
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid((unsigned)groups, (unsigned)(Cx / 32), (unsigned)(Cy / 32));
#define PN_WG(T_, S_)                                                                                                                          \
    do {                                                                                                                                       \
        static bool attr_set[16];                                                                                                              \
        int dev = 0;                                                                                                                           \
        (void)hipGetDevice(&dev);                                                                                                              \
        if (dev >= 0 && dev < 16 && !attr_set[dev]) {                                                                                          \
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_pn_wgrad<T_, S_>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 512); \
            attr_set[dev] = true;                                                                                                              \
        }                                                                                                                                      \
        hipLaunchKernelGGL((k_pn_wgrad<T_, S_>), grid, dim3(256), lds, st, a);                                                                 \
    } while (0)
    if (ks == 3 && stride == 1) PN_WG(9, 1);
    else if (ks == 3) PN_WG(9, 2);
    else PN_WG(1, 2);
#undef PN_WG
    if (dw) hipLaunchKernelGGL(k_pn_reduce_groups, dim3((unsigned)((n + 31) / 32)), dim3(256), 0, st, dw, part, (int)n, groups);
    return pn_status();
}

// number of pixel-tile groups (partial copies of the gradient) a launch of this geometry leaves in its workspace
BC_EXPORT int bc_pn_wgrad_groups(int N, int Hy, int Wy, int Cx, int Cy)
{
    const int n_tiles = N * ((Hy + PN_TH - 1) / PN_TH) * ((Wy + PN_TW - 1) / PN_TW), blocks = (Cx / 32) * (Cy / 32);
    int groups = (256 + blocks - 1) / blocks;
    groups = groups < 1 ? 1 : (groups > n_tiles ? n_tiles : groups);
    const int per = (n_tiles + groups - 1) / groups;
    return (n_tiles + per - 1) / per;
}

BC_EXPORT long long bc_pn_wgrad_workspace(int N, int Hy, int Wy, int Cx, int Cy, int ks)
{
    return (long long)bc_pn_wgrad_groups(N, Hy, Wy, Cx, Cy) * ks * ks * Cx * Cy;
}

BC_EXPORT int bc_pn_bn_finalize(const float *part, long long n_part, int C, double count, const float *gamma, const float *beta, float eps, float momentum,
                                float *running_mean, float *running_var, long long *batches, float *scale, float *shift, float *save_mean, float *save_invstd,
                                void *stream)
{
    if (!part || !scale || !shift || !save_mean || !save_invstd) return BC_ERR_NULL;
    if (C <= 0 || C > 1024 || 1024 % C != 0 || n_part <= 0 || n_part > 0x7fffffffLL || count <= 0) return BC_ERR_SHAPE;
    hipLaunchKernelGGL(k_pn_bn_finalize, dim3(1), dim3(1024), 0, (hipStream_t)stream, part, (int)n_part, C, count, gamma, beta, eps, momentum, running_mean,
                       running_var, batches, scale, shift, save_mean, save_invstd);
    return pn_status();
}

BC_EXPORT int bc_pn_join(float *out, const float *za, const float *sa, const float *ta, const float *zb, const float *sb, const float *tb, int mode, int C,
                         long long pixels, void *stream)
{
    if (!out || !za || !sa || !ta || !zb || (mode >= 1 && (!sb || !tb))) return BC_ERR_NULL;
    if (C <= 0 || C % 4 != 0 || pixels <= 0 || mode < 0 || mode > 2) return BC_ERR_SHAPE;
    const long long total4 = pixels * (C / 4);
    hipLaunchKernelGGL(k_pn_join, dim3((unsigned)((total4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (float4 *)out, (const float4 *)za, sa, ta,
                       (const float4 *)zb, sb, tb, mode, C / 4, total4);
    return pn_status();
}

// BatchNorm without finalize launches.  bc_pn_arm_bn(in_acc, gamma, beta, count, eps, C, out_acc): the NEXT bc_pn_conv_nhwc of this host thread (forward) takes
// the prologue's scale / shift from the fixed-point sums in_acc ([2][C] 64-bit words in units of 2^-24; NULL: none; in_scale / in_shift must then be NULL
// in that call) and ADDS the sums of its output and of its squares to out_acc ([2][Cy]; NULL: none).  Integer additions: the result does not depend on
// their order.  bc_pn_bn_finalize_acc turns the accumulators of ALL layers into the arrays the backward pass reads and zeroes them.
BC_EXPORT int bc_pn_arm_bn(const void *in_acc, const float *gamma, const float *beta, double count, float eps, int C, void *out_acc)
{
    if (in_acc && (count <= 0 || C <= 0 || C > PN_COEF_MAX)) return BC_ERR_SHAPE;
    g_pn_arm_in = PnBnSrc{static_cast<const unsigned long long *>(in_acc), gamma, beta, count, eps, C};
    g_pn_arm_out = static_cast<unsigned long long *>(out_acc);
    return BC_OK;
}

BC_EXPORT int bc_pn_join_acc(float *out, const float *za, const void *acc_a, const float *gamma_a, const float *beta_a, const float *zb, const void *acc_b,
                             const float *gamma_b, const float *beta_b, double count, float eps, int mode, int C, long long pixels, void *stream)
{
    if (!out || !za || !acc_a || !zb || (mode >= 1 && !acc_b)) return BC_ERR_NULL;
    if (C <= 0 || C % 4 != 0 || C > PN_COEF_MAX || pixels <= 0 || mode < 0 || mode > 2 || count <= 0) return BC_ERR_SHAPE;
    const long long total4 = pixels * (C / 4);
    long long wgs = (total4 + 255) / 256;
    if (wgs > 1024) wgs = 1024;         // (every workgroup derives the coefficients once: grid-stride over the map)
    const PnBnSrc ba{static_cast<const unsigned long long *>(acc_a), gamma_a, beta_a, count, eps, C};
    const PnBnSrc bb{static_cast<const unsigned long long *>(acc_b), gamma_b, beta_b, count, eps, C};
    hipLaunchKernelGGL(k_pn_join_acc, dim3((unsigned)wgs), dim3(256), 0, (hipStream_t)stream, (float4 *)out, (const float4 *)za, ba, (const float4 *)zb, bb, mode,
                       C / 4, total4);
    return pn_status();
}

// layers: DEVICE array of n_layers records {acc, gamma, beta, running_mean, running_var, batches, scale, shift, mean, invstd (pointers), count (double),
// eps, momentum (float), C, pad (int)} = 104 bytes each (bc_pn_bn_layer_bytes)
BC_EXPORT int bc_pn_bn_layer_bytes(void) { return (int)sizeof(PnBnLayer); }
BC_EXPORT int bc_pn_bn_finalize_acc(const void *layers, int n_layers, void *stream)
{
    if (!layers) return BC_ERR_NULL;
    if (n_layers <= 0) return BC_ERR_SHAPE;
    hipLaunchKernelGGL(k_pn_bn_finalize_acc, dim3((unsigned)n_layers), dim3(PN_COEF_MAX), 0, (hipStream_t)stream, static_cast<const PnBnLayer *>(layers));
    return pn_status();
}

// backward of training-mode BatchNorm (+ the ReLU behind it): gz, and dgamma / dbeta; part = workspace of bc_pn_bn_bwd_partials(pixels) x 2 x C floats,
// coef = 3 C floats.  mask_mode 0 none, 1 own output (z * scale + shift > 0), 2 external map (mask > 0)
// partial rows of a backward launch: 64 pixels per workgroup or more, at most 1024 rows
BC_EXPORT long long bc_pn_bn_bwd_partials(long long pixels)
{
    long long per = (pixels + 1023) / 1024;
    per = per < 64 ? 64 : per;
    return (pixels + per - 1) / per;
}

BC_EXPORT int bc_pn_bn_bwd(float *gz, float *dgamma, float *dbeta, float *part, float *coef, const float *g, const float *z, const float *mask, int mask_mode,
                           const float *scale, const float *shift, const float *mean, const float *invstd, const float *gamma, int C, long long pixels,
                           void *stream)
{
    if (!gz || !part || !coef || !g || !z || !mean || !invstd) return BC_ERR_NULL;
    if ((mask_mode == 1 && (!scale || !shift)) || (mask_mode == 2 && !mask)) return BC_ERR_NULL;
    if (C <= 0 || C % 4 != 0 || 256 % (C / 4) != 0 || 1024 % C != 0 || pixels <= 0 || mask_mode < 0 || mask_mode > 2) return BC_ERR_SHAPE;
    PnBnBwdArgs a;
    a.g = g; a.z = z; a.mask = mask; a.scale = scale; a.shift = shift; a.mean = mean; a.invstd = invstd; a.part = part;
    a.pixels = pixels; a.C = C; a.mask_mode = mask_mode;
    const long long n_part = bc_pn_bn_bwd_partials(pixels);
    a.pix_per_wg = (int)((pixels + n_part - 1) / n_part);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(k_pn_bn_bwd_reduce, dim3((unsigned)n_part), dim3(256), 0, st, a);
    hipLaunchKernelGGL(k_pn_bn_bwd_finalize, dim3(1), dim3(1024), 0, st, part, (int)n_part, C, (double)pixels, gamma, mean, invstd, dgamma, dbeta, coef);
    const long long total4 = pixels * (C / 4);
    hipLaunchKernelGGL(k_pn_bn_bwd_apply, dim3((unsigned)((total4 + 255) / 256)), dim3(256), 0, st, a, coef, gz);
    return pn_status();
}

BC_EXPORT int bc_pn_head_fwd(float *logits, const float *z, const float *scale, const float *shift, const float *w, const float *bias, int N, int Hi, int Wi,
                             int C, void *stream)
{
    if (!logits || !z || !scale || !shift || !w) return BC_ERR_NULL;
    if (N <= 0 || Hi <= 0 || Wi <= 0 || C <= 0) return BC_ERR_SHAPE;
    const int Ho = (Hi - 1) / 2 + 1, Wo = (Wi - 1) / 2 + 1;
    hipLaunchKernelGGL(k_pn_head_fwd, dim3((unsigned)(N * Ho * Wo)), dim3(64), 0, (hipStream_t)stream, logits, z, scale, shift, w, bias, N, Hi, Wi, C, Ho, Wo);
    return pn_status();
}

BC_EXPORT int bc_pn_head_fwd_acc(float *logits, const float *z, const void *acc, const float *gamma, const float *beta, double count, float eps, const float *w,
                                 const float *bias, int N, int Hi, int Wi, int C, void *stream)
{
    if (!logits || !z || !acc || !w) return BC_ERR_NULL;
    if (N <= 0 || Hi <= 0 || Wi <= 0 || C <= 0 || C > PN_COEF_MAX || count <= 0) return BC_ERR_SHAPE;
    const int Ho = (Hi - 1) / 2 + 1, Wo = (Wi - 1) / 2 + 1;
    const PnBnSrc bn{static_cast<const unsigned long long *>(acc), gamma, beta, count, eps, C};
    hipLaunchKernelGGL(k_pn_head_fwd_acc, dim3((unsigned)(N * Ho * Wo)), dim3(64), 0, (hipStream_t)stream, logits, z, bn, w, bias, N, Hi, Wi, C, Ho, Wo);
    return pn_status();
}

BC_EXPORT int bc_pn_head_bwd(float *ga, float *dw, float *db, const float *gl, const float *z, const float *scale, const float *shift, const float *w, int N,
                             int Hi, int Wi, int C, void *stream)
{
    if (!ga || !dw || !gl || !z || !scale || !shift || !w) return BC_ERR_NULL;
    if (N <= 0 || Hi <= 0 || Wi <= 0 || C <= 0 || C > 1024 || 1024 % C != 0) return BC_ERR_SHAPE;
    const int Ho = (Hi - 1) / 2 + 1, Wo = (Wi - 1) / 2 + 1;
    const long long total = (long long)N * Hi * Wi * C;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(k_pn_head_bwd_data, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, ga, gl, w, N, Hi, Wi, C, Ho, Wo);
    hipLaunchKernelGGL(k_pn_head_bwd_weight, dim3(10), dim3(1024), 0, st, dw, db, gl, z, scale, shift, N, Hi, Wi, C, Ho, Wo);
    return pn_status();
}

BC_EXPORT int bc_pn_infogain(float *ig, const void *cur, const void *prev, int dtype, int N, int C, int H, int W, long long sn, long long sc, long long sh,
                             long long sw, int h, int w, float rh, float rw, void *stream)
{
    if (!ig || !cur || !prev) return BC_ERR_NULL;
    if (N <= 0 || C <= 0 || C > 32 || H <= 0 || W <= 0 || h <= 0 || w <= 0 || (long long)N * h * w > (1LL << 26)) return BC_ERR_SHAPE;
    if (dtype < 0 || dtype > 2) return BC_ERR_ELEM;
    PnIgArgs a;
    a.cur = cur; a.prev = prev; a.dtype = dtype; a.ig = ig; a.sn = sn; a.sc = sc; a.sh = sh; a.sw = sw; a.N = N; a.C = C; a.H = H; a.W = W; a.h = h; a.w = w; a.rh = rh; a.rw = rw;
    hipLaunchKernelGGL(k_pn_infogain, dim3((unsigned)(((long long)N * h * w * 32 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a);
    return pn_status();
}

BC_EXPORT int bc_pn_reward_seed(float *gl, float *loss, float *reward, const float *logits, const uint8_t *grid, const float *ig, const double *cost_dev,
                                double cost_host, double target, double gamma, int N, int h, int w, int GH, int GW, void *stream)
{
    if (!gl || !logits || !grid || !ig) return BC_ERR_NULL;
    if (N <= 0 || h <= 0 || w <= 0 || GH <= 0 || GW <= 0 || GH > h || GW > w) return BC_ERR_SHAPE;
    hipLaunchKernelGGL(k_pn_reward_seed, dim3(1), dim3(256), 0, (hipStream_t)stream, gl, loss, reward, logits, grid, ig, cost_dev, cost_host, target, gamma, N, h,
                       w, GH, GW);
    return pn_status();
}

BC_EXPORT int bc_pn_rmsprop(float *p, const float *g, float *sq, float *mom, long long n, float lr, float alpha, float eps, float wd, float momentum,
                            void *stream)
{
    if (!p || !g || !sq || (momentum > 0.f && !mom)) return BC_ERR_NULL;
    if (n <= 0 || n > 0x7fffffffLL) return BC_ERR_SHAPE;
    hipLaunchKernelGGL(k_pn_rmsprop, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, p, g, sq, mom, (int)n, lr, alpha, eps, wd, momentum);
    return pn_status();
}

// segs: DEVICE array of n_segs records of 80 bytes: int64 off, off_t, param address, s_co, s_ci, s_ky, s_kx; int32 taps, kw, cin, cin_pad, cout, numel
BC_EXPORT int bc_pn_sync_params(float *flat, float *flat_t, const void *segs, int n_segs, int dir, void *stream)
{
    if (!flat || !segs) return BC_ERR_NULL;
    if (n_segs <= 0 || !(dir == 0 || dir == 1)) return BC_ERR_SHAPE;
    static_assert(sizeof(PnSeg) == 96, "PnSeg layout: 7 x int64 + 6 x int32 + int64 + 2 x int32");
    hipLaunchKernelGGL(k_pn_sync_params, dim3(64, (unsigned)n_segs), dim3(256), 0, (hipStream_t)stream, flat, flat_t, (const PnSeg *)segs, dir);
    return pn_status();
}

/* measurement only: device buffer that receives 8 x uint64 (100 MHz s_memrealtime) per workgroup of the next bc_pn_conv_nhwc launches:
 * [0] entry, [1] first patch requested, [2] first stage in the LDS, [3] first tile multiplied, [4] first tile stored, [5] all tiles done; NULL = off */
BC_EXPORT int bc_pn_set_stamps(void *buf) { g_pn_stamps = (unsigned long long *)buf; return BC_OK; }

// reduce (weight-gradient partials) + RMSprop + export in one launch; segs as bc_pn_sync_params, with ws_off / groups filled for the conv weights
BC_EXPORT int bc_pn_update(float *p, float *g, float *sq, float *mom, float *flat_t, const float *ws, const void *segs, int n_segs, float lr, float alpha,
                           float eps, float wd, float momentum, void *stream)
{
    if (!p || !g || !sq || !segs || (momentum > 0.f && !mom)) return BC_ERR_NULL;
    if (n_segs <= 0) return BC_ERR_SHAPE;
    hipLaunchKernelGGL(k_pn_update, dim3(1152, (unsigned)n_segs), dim3(256), 0, (hipStream_t)stream, p, g, sq, mom, flat_t, ws, (const PnSeg *)segs, lr, alpha, eps,
                       wd, momentum);
    return pn_status();
}

BC_EXPORT int bc_pn_seg_bytes(void) { return (int)sizeof(PnSeg); }

BC_EXPORT int bc_pn_features_nhwc(float *out, int N, int h, int w, int Cpad, const void *const *ptrs, const long long *strides, const int *dims,
                                  const float *scales, void *stream)
{
    if (!out || !ptrs || !strides || !dims || !scales) return BC_ERR_NULL;
    if (N <= 0 || h <= 0 || w <= 0 || Cpad <= 0 || Cpad % 4 != 0) return BC_ERR_SHAPE;
    PnFeatGeom g;
    g.N = N; g.h = h; g.w = w; g.Cpad = Cpad;
    int ctot = 0;
    for (int k = 0; k < 4; ++k) {
        PnFeatSrc &s = g.src[k];
        s.ptr = ptrs[k];
        if (!s.ptr) return BC_ERR_NULL;
        s.sn = strides[4 * k]; s.sc = strides[4 * k + 1]; s.sh = strides[4 * k + 2]; s.sw = strides[4 * k + 3];
        s.C = dims[4 * k]; s.H = dims[4 * k + 1]; s.W = dims[4 * k + 2]; s.dtype = dims[4 * k + 3];
        if (s.C <= 0 || s.H <= 0 || s.W <= 0 || s.dtype < 0 || s.dtype > 3) return BC_ERR_SHAPE;
        s.scale_h = scales[3 * k]; s.scale_w = scales[3 * k + 1]; s.offset = scales[3 * k + 2];
        ctot += s.C;
    }
    if (ctot > Cpad) return BC_ERR_SHAPE;
    const long long total = (long long)N * h * w * (Cpad / 4);
    if (total > 0x7fffffffLL * 64) return BC_ERR_RANGE;
    hipLaunchKernelGGL(k_pn_features, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, out, g);
    return pn_status();
}

BC_EXPORT int bc_pn_probs(float *probs, float *log_probs, const float *logits, const uint8_t *grid, int n, void *stream)
{
    if (!probs || !log_probs || !logits || !grid) return BC_ERR_NULL;
    if (n <= 0) return BC_ERR_SHAPE;
    hipLaunchKernelGGL(k_pn_probs, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, probs, log_probs, logits, grid, n);
    return pn_status();
}
