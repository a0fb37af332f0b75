// Standalone consumer of libblockcopy_hip.so through its C ABI only: plain hipMalloc'd buffers, no PyTorch.
// Checks on the GPU: host index tables == device index tables; gather -> fused scatter+copy reproduces
// where(executed, image, prev); all-executed halo gather == zero-padded unfold of the dense map.
// Build: hipcc --offload-arch=gfx950 -I include tests/abi_c/abi_roundtrip.cpp -L<libdir> -lblockcopy_hip -o abi_roundtrip
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "blockcopy_hip.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 2; } } while (0)
#define BC(x) do { int r_ = (x); if (r_ != BC_OK) { printf("blockcopy error %d (%s) at %s:%d\n", r_, bc_error_string(r_), __FILE__, __LINE__); return 3; } } while (0)

int main()
{
    const int N = 1, C = 5, GH = 3, GW = 4, bs = 8, p = 1, E = 4;
    const int H = GH * bs, W = GW * bs, T = N * GH * GW, bsp = bs + 2 * p;
    if (bc_abi_version() != BC_ABI_VERSION) { printf("ABI version mismatch\n"); return 1; }

    std::vector<uint8_t> grid(T);
    for (int g = 0; g < T; ++g) grid[g] = (g * 5 + 1) % 3 != 0;
    std::vector<int32_t> gi(T), me(T);
    const int n_exec = bc_grid_tables_host(grid.data(), T, gi.data(), me.data(), nullptr, nullptr);

    std::vector<float> image((size_t)N * C * H * W), prev(image.size());
    for (size_t i = 0; i < image.size(); ++i) { image[i] = (float)i; prev[i] = -(float)i - 1.f; }

    uint8_t *d_grid; int32_t *d_gi, *d_me, *d_gi2, *d_me2, *d_cnt;
    float *d_img, *d_prev, *d_out, *d_blocks;
    CK(hipMalloc(&d_grid, T)); CK(hipMalloc(&d_gi, 4 * T)); CK(hipMalloc(&d_me, 4 * T));
    CK(hipMalloc(&d_gi2, 4 * T)); CK(hipMalloc(&d_me2, 4 * T)); CK(hipMalloc(&d_cnt, 8));
    CK(hipMalloc(&d_img, image.size() * 4)); CK(hipMalloc(&d_prev, image.size() * 4)); CK(hipMalloc(&d_out, image.size() * 4));
    CK(hipMalloc(&d_blocks, (size_t)T * C * bs * bs * 4));
    CK(hipMemcpy(d_grid, grid.data(), T, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_gi, gi.data(), 4 * T, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_me, me.data(), 4 * T, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_img, image.data(), image.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_prev, prev.data(), image.size() * 4, hipMemcpyHostToDevice));
    hipStream_t st;
    CK(hipStreamCreate(&st));

    // 1. device index tables == host index tables
    BC(bc_grid_tables(d_grid, T, d_gi2, d_me2, nullptr, nullptr, d_cnt, st));
    std::vector<int32_t> gi2(T), me2(T), cnt(2);
    CK(hipStreamSynchronize(st));
    CK(hipMemcpy(gi2.data(), d_gi2, 4 * T, hipMemcpyDeviceToHost));
    CK(hipMemcpy(me2.data(), d_me2, 4 * T, hipMemcpyDeviceToHost));
    CK(hipMemcpy(cnt.data(), d_cnt, 8, hipMemcpyDeviceToHost));
    if (cnt[0] != n_exec || cnt[1] != T - n_exec) { printf("counts differ\n"); return 1; }
    for (int g = 0; g < T; ++g) if (gi[g] != gi2[g]) { printf("grid_idx differs at %d\n", g); return 1; }
    for (int k = 0; k < n_exec; ++k) if (me[k] != me2[k]) { printf("mapping_exec differs at %d\n", k); return 1; }

    // 2. gather -> fused scatter+copy == where(executed, image, prev)
    BC(bc_split(d_blocks, d_img, d_me, n_exec, N, C, H, W, bs, E, st));
    BC(bc_combine_copy(d_blocks, d_prev, d_out, d_gi, N, C, H, W, bs, E, st));
    std::vector<float> out(image.size());
    CK(hipStreamSynchronize(st));
    CK(hipMemcpy(out.data(), d_out, out.size() * 4, hipMemcpyDeviceToHost));
    for (int c = 0; c < C; ++c)
        for (int y = 0; y < H; ++y)
            for (int x = 0; x < W; ++x) {
                const size_t i = ((size_t)c * H + y) * W + x;
                const float want = grid[(y / bs) * GW + x / bs] ? image[i] : prev[i];
                if (out[i] != want) { printf("scatter+copy differs at c=%d y=%d x=%d\n", c, y, x); return 1; }
            }

    // 3. all-executed halo gather over a fresh ring cache == zero-padded windows of the dense map
    std::vector<uint8_t> all(T, 1);
    const int n_all = bc_grid_tables_host(all.data(), T, gi.data(), me.data(), nullptr, nullptr);
    CK(hipMemcpy(d_gi, gi.data(), 4 * T, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_me, me.data(), 4 * T, hipMemcpyHostToDevice));
    float *d_ring, *d_pad;
    CK(hipMalloc(&d_ring, (size_t)T * C * 4 * p * bs * 4)); CK(hipMalloc(&d_pad, (size_t)T * C * bsp * bsp * 4));
    BC(bc_split(d_blocks, d_img, d_me, n_all, N, C, H, W, bs, E, st));
    BC(bc_pad_ring(d_pad, d_blocks, d_ring, d_gi, d_me, n_all, N, C, GH, GW, bs, p, E, st));
    std::vector<float> padded((size_t)T * C * bsp * bsp);
    CK(hipStreamSynchronize(st));
    CK(hipMemcpy(padded.data(), d_pad, padded.size() * 4, hipMemcpyDeviceToHost));
    for (int t = 0; t < T; ++t)
        for (int c = 0; c < C; ++c)
            for (int hp = 0; hp < bsp; ++hp)
                for (int wp = 0; wp < bsp; ++wp) {
                    const int y = (t / GW) * bs + hp - p, x = (t % GW) * bs + wp - p;
                    const float want = (y < 0 || y >= H || x < 0 || x >= W) ? 0.f : image[((size_t)c * H + y) * W + x];
                    if (padded[(((size_t)t * C + c) * bsp + hp) * bsp + wp] != want) { printf("halo differs at tile %d\n", t); return 1; }
                }

    // 4. argument validation returns codes, never crashes
    if (bc_split(nullptr, nullptr, nullptr, 1, 1, 3, 8, 8, 4, 4, st) != BC_ERR_NULL) return 1;
    if (bc_pad(nullptr, nullptr, nullptr, nullptr, nullptr, 1, 1, 3, 2, 2, 4, 9, 4, st) != BC_ERR_SHAPE) return 1;
    printf("abi_roundtrip ok (%d of %d tiles executed)\n", n_exec, T);
    return 0;
}
