// Probe for the Winograd kernel's matrix instruction: what does v_mfma_f32_16x16x4_f32 sustain with 16 independent accumulators per
// wave (the 16 Winograd frequencies), two waves per SIMD, (a) registers only, (b) with the operand arithmetic of the input transform
// (2 VALU ops per MFMA) interleaved, (c) with 2 dependent MFMAs per accumulator back to back (t = 0, 1 of a channel pair)?
// Compare with the 32x32x2 figures of mfma_probe.hip.  Build: hipcc --offload-arch=gfx950 -O3 -o mfma_probe3 mfma_probe3.hip
#include <hip/hip_runtime.h>
#include <stdio.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int MODE>
__global__ __launch_bounds__(512) void probe(float *out, int steps, float seed, unsigned long long *stamps = nullptr)
{
    const int tid = threadIdx.x;
    const unsigned long long t_begin = __builtin_amdgcn_s_memtime();
    f32x4 acc[16];
    for (int f = 0; f < 16; ++f) acc[f] = f32x4{0.f, 0.f, 0.f, 0.f};
    float a[16], b[16];
    for (int f = 0; f < 16; ++f) { a[f] = seed + f + tid * 0.001f; b[f] = seed * 0.5f + f; }
    for (int s = 0; s < steps; ++s) {
        if (MODE == 1 || MODE == 3) {      // 2 VALU ops per MFMA, results feed the MFMAs (like Bt d B)
#pragma unroll
            for (int f = 0; f < 16; ++f) { const float t = a[f] - a[(f + 5) & 15]; a[f] = t + a[(f + 9) & 15] * 1e-9f; }
        }
#pragma unroll
        for (int f = 0; f < 16; ++f) {
            acc[f] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[f], b[f], acc[f], 0, 0, 0);
            if (MODE >= 2) acc[f] = __builtin_amdgcn_mfma_f32_16x16x4f32(b[f], a[f], acc[f], 0, 0, 0);
        }
    }
    float sum = 0.f;
    for (int f = 0; f < 16; ++f) sum += acc[f][0] + acc[f][1] + acc[f][2] + acc[f][3];
    if (sum == 12345.678f) out[tid] = sum;
    if (stamps && tid == 0) stamps[blockIdx.x] = __builtin_amdgcn_s_memtime() - t_begin;      // ticks this workgroup lived
}

__global__ __launch_bounds__(512) void probe32(float *out, int steps, float seed)
{
    const int tid = threadIdx.x;
    f32x16 acc[4];
    for (int r = 0; r < 4; ++r)
        for (int i = 0; i < 16; ++i) acc[r][i] = 0.f;
    float a = seed + tid * 0.001f, b = seed * 0.5f;
    for (int s = 0; s < steps; ++s)
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int k = 0; k < 4; ++k) acc[r] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[r], 0, 0, 0);
    float sum = 0.f;
    for (int r = 0; r < 4; ++r)
        for (int i = 0; i < 16; ++i) sum += acc[r][i];
    if (sum == 12345.678f) out[tid] = sum;
}

template <typename F>
static void run(const char *name, F launch, double macs_per_wave)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int iters = 20;
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0, 0);
        for (int i = 0; i < iters; ++i) launch();
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
    }
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double us = ms * 1e3 / iters;
    const double flop = 256.0 * 8 * macs_per_wave * 2;
    printf("%-60s %8.1f us  %6.1f TFLOP/s (%4.1f%% of 157.3)\n", name, us, flop / us / 1e6, flop / us / 1e6 / 157.3 * 100);
}

int main()
{
    float *out;
    hipMalloc(&out, 1 << 20);
    const int steps = 512;
    run("32x32x2, 4 accumulators x 4 dependent, regs only", [&] { hipLaunchKernelGGL(probe32, dim3(256), dim3(512), 0, 0, out, steps, 1.0f); }, (double)steps * 16 * 2048);
    run("16x16x4, 16 accumulators, regs only", [&] { hipLaunchKernelGGL((probe<0>), dim3(256), dim3(512), 0, 0, out, steps, 1.0f); }, (double)steps * 16 * 1024);
    run("16x16x4, 16 accumulators + 2 VALU ops per MFMA", [&] { hipLaunchKernelGGL((probe<1>), dim3(256), dim3(512), 0, 0, out, steps, 1.0f); }, (double)steps * 16 * 1024);
    run("16x16x4, 16 accumulators x 2 dependent", [&] { hipLaunchKernelGGL((probe<2>), dim3(256), dim3(512), 0, 0, out, steps, 1.0f); }, (double)steps * 32 * 1024);
    run("16x16x4, 16 accumulators x 2 dependent + 1 VALU op per MFMA", [&] { hipLaunchKernelGGL((probe<3>), dim3(256), dim3(512), 0, 0, out, steps, 1.0f); }, (double)steps * 32 * 1024);
    // what does s_memtime count?  ticks of a workgroup's life vs the event-timed duration of a one-round launch of the register-only loop
    unsigned long long *st;
    hipMalloc(&st, 256 * 8);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL((probe<0>), dim3(256), dim3(512), 0, 0, out, 4096, 1.0f, st);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        unsigned long long h[256];
        hipMemcpy(h, st, sizeof(h), hipMemcpyDeviceToHost);
        double mean = 0;
        for (int i = 0; i < 256; ++i) mean += (double)h[i] / 256;
        printf("s_memtime: a workgroup of the register-only loop lives %.0f ticks in a %.1f us launch => %.3f ticks/ns; its %d MFMAs x 32 cycles x 2 waves per SIMD = %.0f cycles => %.2f GHz\n",
               mean, ms * 1e3, mean / (ms * 1e6), 4096 * 16, 4096.0 * 16 * 32 * 2, 4096.0 * 16 * 32 * 2 / (ms * 1e6));
    }
    return 0;
}
