// Ground-truth probe for the fp32 matrix pipe on MI355X: how many TFLOP/s does a grid of 8-wave workgroups (one per CU,
// two waves per SIMD) sustain on v_mfma_f32_32x32x2_f32 with (a) registers only, (b) an LDS fragment read per 4 MFMAs,
// (c) a 16-byte global (L2-resident, same addresses on every CU) load per 4 MFMAs, (d) both -- i.e. the operand traffic
// of csrc/conv3x3_v2.inc without its staging / barriers.  Build: hipcc --offload-arch=gfx950 -O3 -o mfma_probe mfma_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int R, bool LDS, bool GLB, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void probe(float *out, const float4 *__restrict__ w, int steps, int wstride)
{
    extern __shared__ float4 lds4[];
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < 4096; i += 64 * WAVES) lds4[i] = make_float4(i * 0.001f, 1.f, 2.f, 3.f);
    __syncthreads();
    f32x16 acc[R];
    for (int r = 0; r < R; ++r)
        for (int i = 0; i < 16; ++i) acc[r][i] = 0.f;
    float4 a = make_float4(1.f, 2.f, 3.f, 4.f), b = make_float4(0.5f, 0.25f, 0.125f, 1.f);
    const float4 *wp = w + lane + (tid >> 6) * 64;
    float4 bn = GLB ? wp[0] : b;
    for (int s = 0; s < steps; ++s) {
        if (GLB) { b = bn; bn = wp[((s + 1) % wstride) * 64 * 8]; }
        if (LDS) a = lds4[(lane * 9 + s * 7) & 4095];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            acc[r] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, acc[r], 0, 0, 0);
            acc[r] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, acc[r], 0, 0, 0);
            acc[r] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b.z, acc[r], 0, 0, 0);
            acc[r] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b.w, acc[r], 0, 0, 0);
        }
    }
    float sum = 0.f;
    for (int r = 0; r < R; ++r)
        for (int i = 0; i < 16; ++i) sum += acc[r][i];
    if (sum == 12345.678f) out[tid] = sum;
}

template <int R, bool LDS, bool GLB, int WAVES>
static void run(const char *name, int wgs, int steps, float *out, const float4 *w, size_t lds_bytes)
{
    hipFuncSetAttribute(reinterpret_cast<const void *>(&probe<R, LDS, GLB, WAVES>), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int iters = 20;
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0, 0);
        for (int i = 0; i < iters; ++i) hipLaunchKernelGGL((probe<R, LDS, GLB, WAVES>), dim3(wgs), dim3(64 * WAVES), lds_bytes, 0, out, w, steps, 36 * 4);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
    }
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double us = ms * 1e3 / iters;
    const double flop = (double)wgs * WAVES * steps * 4.0 * R * 32 * 32 * 2 * 2;
    printf("%-44s wgs %5d waves %d steps %5d R %d : %8.1f us  %6.1f TFLOP/s (%4.1f%% of 157.3)  mfma-cycles/wave %.0f => %.2f GHz-equivalent at 2 waves/SIMD\n", name, wgs, WAVES, steps,
           R, us, flop / us / 1e6, flop / us / 1e6 / 157.3 * 100, (double)steps * 4 * R * 64, (double)steps * 4 * R * 64 * (WAVES / 4.0) / us / 1e3);
}

int main()
{
    float *out;
    float4 *w;
    hipMalloc(&out, 1 << 20);
    hipMalloc(&w, 64 << 20);
    hipMemset(w, 0, 64 << 20);
    const size_t big = 100 * 1024, small = 66 * 1024;
    run<1, false, false, 8>("regs only, 1 WG/CU x 8 waves", 256, 144, out, w, big);
    run<1, false, false, 8>("regs only, 1 WG/CU x 8 waves (long)", 256, 576, out, w, big);
    run<1, true, false, 8>("+LDS read / 4 MFMA", 256, 144, out, w, big);
    run<1, false, true, 8>("+global 16B / 4 MFMA (shared addrs)", 256, 144, out, w, big);
    run<1, true, true, 8>("+LDS +global", 256, 144, out, w, big);
    run<1, true, true, 8>("+LDS +global (long)", 256, 576, out, w, big);
    run<2, true, true, 8>("+LDS +global R=2", 256, 144, out, w, big);
    run<4, true, true, 8>("+LDS +global R=4", 256, 144, out, w, big);
    run<1, false, false, 4>("regs only, 4 waves, 1 WG/CU (1 wave/SIMD)", 256, 288, out, w, big);
    run<4, false, false, 4>("regs only R=4, 4 waves, 1 WG/CU", 256, 72, out, w, big);
    run<1, false, false, 2>("regs only, 2-wave WGs x1024 (v1 shape)", 1024, 144, out, w, 17 * 1024);
    run<1, true, true, 2>("+LDS +global, 2-wave WGs x1024 (v1 shape)", 1024, 144, out, w, 17 * 1024);
    run<1, false, false, 8>("regs only, 2 rounds (512 WGs)", 512, 144, out, w, big);
    run<1, false, false, 8>("regs only, 268 WGs", 268, 144, out, w, big);
    run<1, false, false, 8>("regs only, 8 waves, small LDS (2 WG/CU ok) x512", 512, 144, out, w, small);
    return 0;
}
