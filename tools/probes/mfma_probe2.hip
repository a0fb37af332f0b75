// Probe 2: operand delivery variants for the fp32 MFMA conv loop (see mfma_probe.hip).  All: 256 WGs x 8 waves, LDS > 80 KB.
//   NA LDS fragment reads (ds_read_b128, prefetched one step ahead) and NB global 16-byte loads (ring of D steps ahead) per step;
//   one step = 4 * RM * RN MFMAs (RM = NA blocks share each B, RN = NB blocks share each A).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int RM, int RN, int D, bool BLDS>
__global__ __launch_bounds__(512) void probe(float *out, const float4 *__restrict__ w, int steps, int wrap)
{
    extern __shared__ float4 lds4[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 6144; i += 512) lds4[i] = make_float4(i * 0.001f, 1.f, 2.f, 3.f);
    __syncthreads();
    f32x16 acc[RM][RN];
    for (int j = 0; j < RM; ++j)
        for (int n = 0; n < RN; ++n)
            for (int i = 0; i < 16; ++i) acc[j][n][i] = 0.f;
    const float4 *wp = w + lane + wave * 64 * RN;
    float4 bq[D + 1][RN];
    for (int d = 0; d < D; ++d)
        for (int n = 0; n < RN; ++n) bq[d][n] = BLDS ? lds4[(lane + d * 64 + n * 512) & 4095] : wp[(d % wrap) * 64 * 8 * RN + n * 64];
    float4 a[2][RM];
    for (int j = 0; j < RM; ++j) a[0][j] = lds4[(lane * 9 + j * 40) & 4095];
    for (int s0 = 0; s0 < steps; s0 += (D + 1) * 2) {
#pragma unroll
        for (int u = 0; u < (D + 1) * 2; ++u) {
            const int s = s0 + u;
#pragma unroll
            for (int n = 0; n < RN; ++n)
                bq[(u + D) % (D + 1)][n] = BLDS ? lds4[4096 + ((lane + (s + D) * 64 + n * 512) & 2047)] : wp[((s + D) % wrap) * 64 * 8 * RN + n * 64];
#pragma unroll
            for (int j = 0; j < RM; ++j) a[(u + 1) & 1][j] = lds4[(lane * 9 + (s + 1) * 7 + j * 40) & 4095];
            asm volatile("" ::: "memory");
#pragma unroll
            for (int j = 0; j < RM; ++j)
#pragma unroll
                for (int n = 0; n < RN; ++n) {
                    const float4 av = a[u & 1][j], b = bq[u % (D + 1)][n];
                    acc[j][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, b.x, acc[j][n], 0, 0, 0);
                    acc[j][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, b.y, acc[j][n], 0, 0, 0);
                    acc[j][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, b.z, acc[j][n], 0, 0, 0);
                    acc[j][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, b.w, acc[j][n], 0, 0, 0);
                }
        }
    }
    float sum = 0.f;
    for (int j = 0; j < RM; ++j)
        for (int n = 0; n < RN; ++n)
            for (int i = 0; i < 16; ++i) sum += acc[j][n][i];
    if (sum == 12345.678f) out[tid] = sum;
}

template <int RM, int RN, int D, bool BLDS>
static void run(const char *name, int mfma_per_wave, float *out, const float4 *w)
{
    hipFuncSetAttribute(reinterpret_cast<const void *>(&probe<RM, RN, D, BLDS>), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    int steps = mfma_per_wave / (4 * RM * RN);
    steps = steps / ((D + 1) * 2) * ((D + 1) * 2);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int iters = 20;
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0, 0);
        for (int i = 0; i < iters; ++i) hipLaunchKernelGGL((probe<RM, RN, D, BLDS>), dim3(256), dim3(512), 100 * 1024, 0, out, w, steps, 144);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
    }
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double us = ms * 1e3 / iters;
    const double flop = 256.0 * 8 * steps * 4.0 * RM * RN * 32 * 32 * 2 * 2;
    printf("%-40s RMxRN %dx%d D %d steps %4d : %8.1f us  %6.1f TFLOP/s (%4.1f%% of 157.3)\n", name, RM, RN, D, steps, us, flop / us / 1e6, flop / us / 1e6 / 157.3 * 100);
}

int main()
{
    float *out;
    float4 *w;
    hipMalloc(&out, 1 << 20);
    hipMalloc(&w, 64 << 20);
    hipMemset(w, 0, 64 << 20);
    run<1, 1, 1, false>("1x1 global B ring 1", 576, out, w);
    run<1, 1, 2, false>("1x1 global B ring 2", 576, out, w);
    run<1, 1, 5, false>("1x1 global B ring 5", 576, out, w);
    run<1, 1, 8, false>("1x1 global B ring 8", 576, out, w);
    run<1, 1, 2, true>("1x1 LDS B", 576, out, w);
    run<2, 1, 2, false>("2x1 global B ring 2", 576, out, w);
    run<2, 1, 4, false>("2x1 global B ring 4", 576, out, w);
    run<1, 2, 2, false>("1x2 global B ring 2", 576, out, w);
    run<2, 2, 1, false>("2x2 global B ring 1", 576, out, w);
    run<2, 2, 2, false>("2x2 global B ring 2", 576, out, w);
    run<2, 2, 2, true>("2x2 LDS B", 576, out, w);
    run<2, 2, 2, false>("2x2 global B ring 2 (2304 mfma)", 2304, out, w);
    return 0;
}
