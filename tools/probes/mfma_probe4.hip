// Probe for a wider Winograd wave tile: what does the F(2x2,3x3) main loop sustain when a wave owns 32 Winograd tiles x 32 output
// channels x 16 frequencies on v_mfma_f32_32x32x2_f32 (256 accumulator registers -> one wave per SIMD, the whole 512-entry register
// file), with its REAL operand traffic -- 16 ds_read_b64 of the 4x4 window per 4-channel sub-step (bank-conflict-free layout of
// conv3x3_wino.inc), the 32 packed adds of Bt d B, 8 float4 of transformed weights per lane from L2 -- against the shipped form
// (16 tiles x 16 channels on 16x16x4, two waves per SIMD)?  Per matrix cycle the wide tile needs half the LDS reads and VALU
// operations and a quarter of the L1 bytes per CU.  Build: hipcc --offload-arch=gfx950 -O3 -o mfma_probe4 mfma_probe4.hip
#include <hip/hip_runtime.h>
#include <stdio.h>

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ void wn_pass(f32x2 &x0, f32x2 x1, f32x2 &x2, f32x2 &x3, f32x2 &y1)
{
    asm("v_pk_add_f32 %0, %0, %2 neg_lo:[0,1] neg_hi:[0,1]\n\t"
        "v_pk_add_f32 %1, %4, %2\n\t"
        "v_pk_add_f32 %2, %2, %4 neg_lo:[0,1] neg_hi:[0,1]\n\t"
        "v_pk_add_f32 %3, %4, %3 neg_lo:[0,1] neg_hi:[0,1]\n\t"
        "s_nop 1"
        : "+v"(x0), "=&v"(y1), "+v"(x2), "+v"(x3)
        : "v"(x1));
}

constexpr int CHP = 36;   // words per staged pixel (32 channels + pad)
constexpr int PW2 = 10;   // staged row of an 8x8 patch

// MODE 0: everything; 1: no weight loads (registers); 2: no LDS reads (registers); 3: neither (transform + MFMA only)
template <int MODE>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
void probe_wide(float *out, const float4 *__restrict__ wpk, int chunks, float seed)
{
    extern __shared__ float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 2 * 100 * CHP * 2; i += 256) lds[i] = seed + i * 1e-6f;      // 2 wave rows x 2 patches
    __syncthreads();
    const int m = lane & 31, h = lane >> 5;
    const int patch = (wave >> 1) * 2 + (m >> 4), ty = (m >> 2) & 3, tx = m & 3;
    const float *img = lds + (patch * 100 + 2 * ty * PW2 + 2 * tx) * CHP + 2 * h;
    const float4 *wl = wpk + (size_t)(wave & 1) * chunks * (8 * 8 * 64) + lane;
    f32x16 acc[16];
#pragma unroll
    for (int f = 0; f < 16; ++f)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[f][i] = 0.f;
    float4 bq[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) bq[q] = wl[q * 64];
    for (int chunk = 0; chunk < chunks; ++chunk) {
#pragma unroll
        for (int ss = 0; ss < 8; ++ss) {      // 4-channel sub-steps of a 32-channel chunk
            float4 b[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) b[q] = bq[q];
            if (MODE == 0 || MODE == 2) {
                const float4 *p = wl + ((size_t)chunk * 8 + ss + 1) * (8 * 64);
#pragma unroll
                for (int q = 0; q < 8; ++q) bq[q] = p[q * 64];
            }
            asm volatile("" ::: "memory");
            const float *bf = reinterpret_cast<const float *>(b);
            f32x2 d[4][4];
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    if (MODE == 0 || MODE == 1) d[r][c] = *reinterpret_cast<const f32x2 *>(img + (r * PW2 + c) * CHP + 4 * ss);
                    else d[r][c] = f32x2{seed + r + chunk, seed + c + ss};
                }
            f32x2 r1[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) wn_pass(d[0][c], d[1][c], d[2][c], d[3][c], r1[c]);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                f32x2 *row = r == 1 ? r1 : d[r];
                f32x2 v1;
                wn_pass(row[0], row[1], row[2], row[3], v1);
                const f32x2 v0 = row[0], v2 = row[2], v3 = row[3];
                acc[4 * r + 0] = __builtin_amdgcn_mfma_f32_32x32x2f32(v0.x, bf[2 * (4 * r + 0)], acc[4 * r + 0], 0, 0, 0);
                acc[4 * r + 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(v1.x, bf[2 * (4 * r + 1)], acc[4 * r + 1], 0, 0, 0);
                acc[4 * r + 2] = __builtin_amdgcn_mfma_f32_32x32x2f32(v2.x, bf[2 * (4 * r + 2)], acc[4 * r + 2], 0, 0, 0);
                acc[4 * r + 3] = __builtin_amdgcn_mfma_f32_32x32x2f32(v3.x, bf[2 * (4 * r + 3)], acc[4 * r + 3], 0, 0, 0);
                acc[4 * r + 0] = __builtin_amdgcn_mfma_f32_32x32x2f32(v0.y, bf[2 * (4 * r + 0) + 1], acc[4 * r + 0], 0, 0, 0);
                acc[4 * r + 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(v1.y, bf[2 * (4 * r + 1) + 1], acc[4 * r + 1], 0, 0, 0);
                acc[4 * r + 2] = __builtin_amdgcn_mfma_f32_32x32x2f32(v2.y, bf[2 * (4 * r + 2) + 1], acc[4 * r + 2], 0, 0, 0);
                acc[4 * r + 3] = __builtin_amdgcn_mfma_f32_32x32x2f32(v3.y, bf[2 * (4 * r + 3) + 1], acc[4 * r + 3], 0, 0, 0);
            }
        }
    }
    float sum = 0.f;
#pragma unroll
    for (int f = 0; f < 16; ++f)
#pragma unroll
        for (int i = 0; i < 16; ++i) sum += acc[f][i];
    if (sum == 12345.678f) out[tid] = sum;
}

template <typename F>
static void run(const char *name, F launch, double flop)
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
    printf("%-72s %8.1f us  %6.1f TFLOP/s issued (%4.1f%% of 157.3)\n", name, us, flop / us / 1e6, flop / us / 1e6 / 157.3 * 100);
}

int main()
{
    float *out;
    float4 *w;
    const int chunks = 64;      // (a long K so that the loop dominates)
    hipMalloc(&out, 1 << 20);
    hipMalloc(&w, (size_t)2 * (chunks + 1) * 8 * 8 * 64 * sizeof(float4));
    hipMemset(w, 0, (size_t)2 * (chunks + 1) * 8 * 8 * 64 * sizeof(float4));
    const size_t lds = 2 * 2 * 100 * CHP * sizeof(float) + 90 * 1024;      // (+ padding: one workgroup per CU)
    // per wave and sub-step: 32 MFMAs x 32x32x2x2 FLOP
    const double flop = 256.0 * 4 * chunks * 8 * 32 * 4096.0;
#define RUN(M_, name_)                                                                                                              \
    hipFuncSetAttribute(reinterpret_cast<const void *>(&probe_wide<M_>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);     \
    run(name_, [&] { hipLaunchKernelGGL((probe_wide<M_>), dim3(256), dim3(256), lds, 0, out, w, chunks, 1.0f); }, flop)
    RUN(0, "wide tile 32x32x2: LDS window reads + transform + weights from L2");
    RUN(1, "wide tile 32x32x2: LDS window reads + transform, weights in registers");
    RUN(2, "wide tile 32x32x2: transform + weights from L2, window in registers");
    RUN(3, "wide tile 32x32x2: transform + MFMA only");
    return 0;
}
