// Probe for conv3x3_wino4.inc: how fast does ONE wave per SIMD issue v_mfma_f32_16x16x4_f32 when consecutive MFMAs form chains on one
// accumulator -- chain length 1 (all independent, 18 accumulators round robin), 2, 4 (the kernel: four channels of a frequency back to
// back), 4 with two chains interleaved (A0 B0 A1 B1 ...), all on one accumulator -- with one and with two waves per SIMD?
// Cycles per MFMA from s_memtime of wave 0.  Build: hipcc --offload-arch=gfx950 -O3 -o mfma_probe5 mfma_probe5.hip
#include <hip/hip_runtime.h>
#include <stdio.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(512) void probe(float *out, int steps, float seed, unsigned long long *stamps)
{
    const int tid = threadIdx.x;
    f32x4 acc[18];
    for (int f = 0; f < 18; ++f) acc[f] = f32x4{0.f, 0.f, 0.f, 0.f};
    float a[4], b[4];
    for (int j = 0; j < 4; ++j) { a[j] = seed + j + tid * 0.001f; b[j] = seed * 0.5f + j; }
    __syncthreads();
    const unsigned long long t_begin = __builtin_amdgcn_s_memtime();
    for (int s = 0; s < steps; ++s) {
        if (MODE == 0) {          // independent: 72 MFMAs round robin over 18 accumulators
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int f = 0; f < 18; ++f) acc[f] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j], b[j], acc[f], 0, 0, 0);
        } else if (MODE == 1) {   // chains of 2
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int f = 0; f < 18; ++f)
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[f] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[2 * h + j], b[2 * h + j], acc[f], 0, 0, 0);
        } else if (MODE == 2) {   // chains of 4 (the kernel)
#pragma unroll
            for (int f = 0; f < 18; ++f)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[f] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j], b[j], acc[f], 0, 0, 0);
        } else if (MODE == 3) {   // two chains of 4 interleaved
#pragma unroll
            for (int f = 0; f < 18; f += 2)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    acc[f] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j], b[j], acc[f], 0, 0, 0);
                    acc[f + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j], b[j], acc[f + 1], 0, 0, 0);
                }
        } else {                  // one accumulator
#pragma unroll
            for (int k = 0; k < 72; ++k) acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[k & 3], b[k & 3], acc[0], 0, 0, 0);
        }
    }
    const unsigned long long t_end = __builtin_amdgcn_s_memtime();
    float sum = 0.f;
    for (int f = 0; f < 18; ++f) sum += acc[f][0] + acc[f][1] + acc[f][2] + acc[f][3];
    if (sum == 12345.678f) out[tid] = sum;
    if ((tid & 63) == 0) stamps[blockIdx.x * 8 + (tid >> 6)] = t_end - t_begin;
}

template <int MODE>
static void run(const char *name, int threads)
{
    float *out;
    unsigned long long *st, h[8 * 256];
    hipMalloc(&out, 4096);
    hipMalloc(&st, sizeof(h));
    const int steps = 2000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    float ms = 0;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(probe<MODE>, dim3(256), dim3(threads), 0, 0, out, steps, 1.0f, st);
        hipEventRecord(e1, 0);
        hipDeviceSynchronize();
        hipEventElapsedTime(&ms, e0, e1);
    }
    hipMemcpy(h, st, sizeof(h), hipMemcpyDeviceToHost);
    double sum = 0;
    for (int b = 0; b < 256; ++b) sum += (double)h[b * 8];
    const double flops = 256.0 * (threads / 64) * steps * 72.0 * 2048.0;
    printf("%-40s %d waves/SIMD: %.1f cycles per MFMA and wave (%.1f per SIMD); %.3f ms wall = %.1f TFLOP/s; %.2f GHz by ticks / wall\n", name, threads / 256,
           sum / 256 / (steps * 72.0), sum / 256 / (steps * 72.0) / (threads / 256), ms, flops / ms / 1e9, sum / 256 / ms / 1e6);
    hipFree(out);
    hipFree(st);
}

int main()
{
    for (int threads : {256, 512}) {
        if (threads == 256) {
            run<0>("independent (18 accumulators)", 256); run<1>("chains of 2", 256); run<2>("chains of 4", 256); run<3>("two chains of 4 interleaved", 256); run<4>("one accumulator", 256);
        } else {
            run<0>("independent (18 accumulators)", 512); run<1>("chains of 2", 512); run<2>("chains of 4", 512); run<3>("two chains of 4 interleaved", 512); run<4>("one accumulator", 512);
        }
    }
    return 0;
}
