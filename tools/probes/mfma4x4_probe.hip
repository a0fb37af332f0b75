// Layout probe for v_mfma_f32_4x4x1_16B_f32: which lane supplies A[i][block], B[j][block], and where does D[i][j] of a block land?
// A lane's A value encodes (lane), B likewise; the result is decoded on the host.  Build: hipcc --offload-arch=gfx950 -O3 -w -o mfma4x4_probe mfma4x4_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void probe(float *out, int mode)
{
    const int lane = threadIdx.x;
    // mode 0: A = 1 only in lane `sel`, B = 1 everywhere -> D shows where row i(sel) of block(sel) goes
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int sel = 0; sel < 64; ++sel) {
        f32x4 d = {0.f, 0.f, 0.f, 0.f};
        const float a = mode == 0 ? (lane == sel ? 1.f : 0.f) : 1.f;
        const float b = mode == 0 ? 1.f : (lane == sel ? 1.f : 0.f);
        d = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, d, 0, 0, 0);
        for (int r = 0; r < 4; ++r) out[((mode * 64 + sel) * 64 + lane) * 4 + r] = d[r];
    }
}
int main()
{
    float *d, *h = (float *)malloc(2 * 64 * 64 * 4 * sizeof(float));
    hipMalloc(&d, 2 * 64 * 64 * 4 * sizeof(float));
    for (int mode = 0; mode < 2; ++mode) hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, mode);
    hipMemcpy(h, d, 2 * 64 * 64 * 4 * sizeof(float), hipMemcpyDeviceToHost);
    for (int mode = 0; mode < 2; ++mode)
        for (int sel = 0; sel < 64; sel += (sel < 8 ? 1 : 13)) {
            printf("%s lane %2d = 1 -> nonzero D at:", mode == 0 ? "A" : "B", sel);
            for (int lane = 0; lane < 64; ++lane)
                for (int r = 0; r < 4; ++r)
                    if (h[((mode * 64 + sel) * 64 + lane) * 4 + r] != 0.f) printf(" (lane %d, vgpr %d)", lane, r);
            printf("\n");
        }
    return 0;
}
