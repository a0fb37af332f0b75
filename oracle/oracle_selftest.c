/* Sanitizer self-test of the CPU oracle (test infrastructure): runs every entry point on a small multi-frame case
 * under -fsanitize=address,undefined and checks split->combine round trips.  Built by `make bc_oracle_asan`. */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

void bco_split(void *, const void *, const int32_t *, int, int, int, int, int, int, int);
void bco_combine(const void *, void *, const int32_t *, int, int, int, int, int, int, int);
void bco_transfer(void *, const void *, const void *, const int32_t *, int, int, int, int, int, int, int, int);
void bco_repad(void *, const void *, const void *, const int32_t *, const int32_t *, int, int, int, int, int, int, int, int);
int bco_grid_mappings(const uint8_t *, int, int32_t *, int32_t *);
int bco_transfer_idx(const int32_t *, const uint8_t *, int, int32_t *);

int main(void)
{
    enum { N = 2, C = 3, GH = 2, GW = 3, BS = 4, P = 2, E = 4, T = N * GH * GW, H = GH * BS, W = GW * BS, BSP = BS + 2 * P };
    float *image = malloc(sizeof(float) * N * C * H * W), *out = malloc(sizeof(float) * N * C * H * W);
    for (int i = 0; i < N * C * H * W; ++i) image[i] = (float)i;
    uint8_t grid[T];
    int32_t gi[T], prev_gi[T], m[T], tr[T];
    float *prev_c = NULL, *prev_t = NULL;
    int prev_exec = 0, prev_tr = 0;
    for (int frame = 0; frame < 4; ++frame) {
        for (int g = 0; g < T; ++g) grid[g] = frame == 0 ? 1 : (uint8_t)((g * 7 + frame * 3) % 3 != 0);
        int n_exec = bco_grid_mappings(grid, T, gi, m);
        float *blocks = malloc(sizeof(float) * (n_exec ? n_exec : 1) * C * BS * BS);
        bco_split(blocks, image, m, n_exec, N, C, H, W, BS, E);
        memset(out, 0, sizeof(float) * N * C * H * W);
        bco_combine(blocks, out, m, n_exec, N, C, H, W, BS, E);
        for (int k = 0; k < n_exec; ++k) { /* executed tiles round-trip */
            int g = m[k], n = g / (GH * GW), gh = (g / GW) % GH, gw = g % GW;
            for (int c = 0; c < C; ++c)
                for (int y = 0; y < BS; ++y)
                    for (int x = 0; x < BS; ++x) {
                        long o = ((long)(n * C + c) * H + gh * BS + y) * W + gw * BS + x;
                        if (out[o] != image[o]) { fprintf(stderr, "round trip failed\n"); return 1; }
                    }
        }
        int n_tr = 0;
        float *transfer = malloc(sizeof(float) * T * C * BS * BS);
        memset(transfer, 0, sizeof(float) * T * C * BS * BS);
        if (frame > 0) {
            n_tr = bco_transfer_idx(prev_gi, grid, T, tr);
            bco_transfer(transfer, prev_c, prev_t, tr, n_tr, N, C, GH, GW, BS, P, E);
        }
        float *padded = malloc(sizeof(float) * (n_exec ? n_exec : 1) * C * BSP * BSP);
        bco_repad(padded, blocks, transfer, gi, m, n_exec, N, C, GH, GW, BS, P, E);
        free(padded);
        free(prev_c);
        free(prev_t);
        prev_c = blocks; prev_t = transfer; prev_exec = n_exec; prev_tr = n_tr;
        memcpy(prev_gi, gi, sizeof gi);
    }
    (void)prev_exec; (void)prev_tr;
    free(prev_c); free(prev_t); free(image); free(out);
    puts("oracle selftest ok");
    return 0;
}
