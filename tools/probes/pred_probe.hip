// Knock-out probe for k_pred3x3 (csrc/pred3x3.inc): the kernel as shipped, without its arithmetic (-DPRED_KNOCK_COMPUTE), on the C5
// map (1,256,256,512) and a quarter-size one.  Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 [-DPRED_KNOCK_COMPUTE] -o pred_probe pred_probe.hip
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <hip/hip_bfloat16.h>
#include <stdint.h>
#include <stdio.h>
#include <type_traits>
enum { BC_F32 = 0, BC_F16 = 1, BC_BF16 = 2 };
namespace {
template <int DT> struct CvType;
template <> struct CvType<BC_F32> { typedef float T; static constexpr int E = 4, EPV = 4, UV = 8; };
template <> struct CvType<BC_F16> { typedef __half T; static constexpr int E = 2, EPV = 8, UV = 4; };
template <int DT> __device__ __forceinline__ float cv_to_float(const typename CvType<DT>::T *p) { return DT == BC_F32 ? *reinterpret_cast<const float *>(p) : __half2float(*reinterpret_cast<const __half *>(p)); }
template <int DT> __device__ __forceinline__ void cv_store(typename CvType<DT>::T *p, float v) { if (DT == BC_F32) *reinterpret_cast<float *>(p) = v; else *reinterpret_cast<__half *>(p) = __float2half(v); }
#include "../../blockcopy-video-processing-pytorch_amd/csrc/pred3x3.inc"
}

// access-pattern reference: the same patches and bytes as k_pred3x3 (8-row patches with halo, 4 waves x 2 rows x 32 pixels), but a load
// instruction covers ONE WHOLE PIXEL (64 lanes x 16 B = its 1 KB, linear in memory); no LDS, no arithmetic beyond an xor
__global__ __launch_bounds__(256) void k_pattern_pixel(float *out, const uint4 *__restrict__ x, PredGeom g)
{
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t tx = blockIdx.x % g.tiles_x, ty = blockIdx.x / g.tiles_x;
    const int oy0 = (int)ty * 6, ox0 = (int)tx * 30;
    uint4 a = {0u, 0u, 0u, 0u};
    for (uint32_t p0 = 0; p0 < 64; p0 += 16) {
        uint4 v[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const uint32_t p = wave * 64 + p0 + j;
            const int iy = min(max(oy0 - 1 + (int)(p / 32), 0), (int)g.H - 1), ix = min(max(ox0 - 1 + (int)(p % 32), 0), (int)g.W - 1);
            v[j] = x[((size_t)iy * g.W + ix) * (g.C / 4) + lane];
        }
#pragma unroll
        for (int j = 0; j < 16; ++j) { a.x ^= v[j].x; a.y ^= v[j].y; a.z ^= v[j].z; a.w ^= v[j].w; }
    }
    if ((a.x ^ a.y ^ a.z ^ a.w) == 0x12345678u) out[threadIdx.x] = 1.0f;
}

static void run_pattern(int H, int W, int C)
{
    float *x, *out, *flush;
    hipMalloc(&x, (size_t)H * W * C * 4);
    hipMalloc(&out, 4096);
    hipMalloc(&flush, (size_t)1 << 30);
    hipMemset(x, 0, (size_t)H * W * C * 4);
    PredGeom g{1, (uint32_t)H, (uint32_t)W, (uint32_t)C, (uint32_t)((W + 29) / 30), (uint32_t)((H + 5) / 6)};
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    float best = 1e9f, sum = 0;
    for (int rep = 0; rep < 8; ++rep) {
        hipMemsetAsync(flush, rep, (size_t)1 << 30, 0);
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(k_pattern_pixel, dim3(g.tiles_x * g.tiles_y), dim3(256), 0, 0, out, (const uint4 *)x, g);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (rep >= 2) { sum += ms; best = ms < best ? ms : best; }
    }
    printf("%-40s %dx%dx%d: mean %.1f us, best %.1f us (cold)\n", "whole-pixel loads, same patches", H, W, C, sum / 6 * 1e3, best * 1e3);
    hipFree(x); hipFree(out); hipFree(flush);
}

template <int COUT>
static void run(const char *name, int H, int W, int C)
{
    float *x, *out, *w, *flush;
    hipMalloc(&x, (size_t)H * W * C * 4);
    hipMalloc(&out, (size_t)H * W * COUT * 4);
    hipMalloc(&w, (size_t)C * 9 * COUT * 4);
    hipMalloc(&flush, (size_t)1 << 30);
    hipMemset(x, 0, (size_t)H * W * C * 4);
    hipMemset(w, 0, (size_t)C * 9 * COUT * 4);
    PredGeom g{1, (uint32_t)H, (uint32_t)W, (uint32_t)C, (uint32_t)((W + 29) / 30), (uint32_t)((H + 5) / 6)};
    const size_t lds = 256 * PRED_PS * 4;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    float best = 1e9f, sum = 0;
    for (int rep = 0; rep < 8; ++rep) {
        hipMemsetAsync(flush, rep, (size_t)1 << 30, 0);
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL((k_pred3x3<BC_F32, COUT, 1>), dim3(g.tiles_x * g.tiles_y, 1), dim3(256), lds, 0, out, (const uint4 *)x, w, (const float *)nullptr, g);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (rep >= 2) { sum += ms; best = ms < best ? ms : best; }
    }
    printf("%-40s %dx%dx%d -> %d: mean %.1f us, best %.1f us (cold)\n", name, H, W, C, COUT, sum / 6 * 1e3, best * 1e3);
    hipFree(x); hipFree(out); hipFree(w); hipFree(flush);
}

int main()
{
#ifdef PRED_KNOCK_COMPUTE
    const char *name = "no arithmetic (loads + LDS writes only)";
#else
    const char *name = "as shipped";
#endif
    run<1>(name, 256, 512, 256);
    run<2>(name, 256, 512, 256);
    run<1>(name, 128, 256, 256);
    run_pattern(256, 512, 256);
    run_pattern(128, 256, 256);
    return 0;
}
