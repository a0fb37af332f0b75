/*
 * bc_oracle.c -- CPU ORACLE (test infrastructure, NOT product code).
 *
 * Plain-C restatement of the reference's four block-copy kernels and of its
 * grid -> index-table builder.  Only tests/, __graft_entry__.smoke() and the
 * cpu side of bench.py may load this; the product path (libblockcopy_hip.so)
 * never links or calls it.
 *
 * Every function follows one reference kernel string, walking the same
 * flattened pixel index `i` and channel `c` the CUDA kernels walk
 * (reference: blockcopy/blockcopy/utils/cuda.py:50-72 for the two loops).
 * The kernels are pure copies (no arithmetic on the payload) so the oracle is
 * written on raw bytes with an element size E (4 = float, 2 = __half) and any
 * correct device kernel must match it bit for bit.
 *
 * Pinning status: the reference ships NO tests or golden vectors for these
 * kernels and its CUDA strings cannot execute in this image (no CuPy/NVRTC/
 * NVIDIA GPU).  The oracle is pinned instead by running the reference's own
 * Python (TensorWrapper, BlockCopyModel, get_grid_mappings, SwiftNet, policy)
 * on top of these functions (oracle/gen_golden.py) and checking the
 * reference-independent properties P1 (all-active == dense zero-padded conv)
 * and P2 (static clip invariance); see DESIGN.md "Oracle".
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define BCO_API __attribute__((visibility("default")))

static inline void cp(void *dst, const void *src, int E) { memcpy(dst, src, (size_t)E); }

/* split_kernel -- blockcopy/blockcopy/utils/block_funcs.py:57-83
 * blocks[b,c,h,w] = image[gn,c,gh*BS+h,gw*BS+w] with (gn,gh,gw) from mapping_exec[b]. */
BCO_API void bco_split(void *blocks, const void *image, const int32_t *mapping_exec,
                       int n_exec, int N, int C, int H, int W, int BS, int E)
{
    (void)N;
    const int GRID_W = W / BS, GRID_H = H / BS;
    const long npixels = (long)n_exec * BS * BS;
    char *o = (char *)blocks;
    const char *im = (const char *)image;
    for (long i = 0; i < npixels; ++i) {
        const long b = i / (BS * BS);
        const long h = (i / BS) % BS;
        const long w = i % BS;
        const long i_b = b * C * BS * BS + h * BS + w;
        const long i_g = mapping_exec[b];
        const long gn = i_g / (GRID_H * GRID_W);
        const long gh = (i_g / GRID_W) % GRID_H;
        const long gw = i_g % GRID_W;
        const long i_image = gn * C * W * H + (gh * BS + h) * W + (gw * BS + w);
        for (long c = 0; c < C; ++c)
            cp(o + (i_b + c * BS * BS) * E, im + (i_image + c * (long)W * H) * E, E);
    }
}

/* combine_kernel -- blockcopy/blockcopy/utils/block_funcs.py:130-158
 * out[gn,c,gh*BS+h,gw*BS+w] = blocks[b,c,h,w]; tiles not listed in mapping_exec are untouched. */
BCO_API void bco_combine(const void *blocks, void *out, const int32_t *mapping_exec,
                         int n_exec, int N, int C, int H, int W, int BS, int E)
{
    (void)N;
    const int GRID_W = W / BS, GRID_H = H / BS;
    const long npixels = (long)n_exec * BS * BS;
    const char *bl = (const char *)blocks;
    char *o = (char *)out;
    for (long i = 0; i < npixels; ++i) {
        const long b = i / (BS * BS);
        const long h = (i / BS) % BS;
        const long w = i % BS;
        const long i_b = b * C * BS * BS + h * BS + w;
        const long i_g = mapping_exec[b];
        const long gn = i_g / (GRID_H * GRID_W);
        const long gh = (i_g / GRID_W) % GRID_H;
        const long gw = i_g % GRID_W;
        const long i_image = gn * C * W * H + (gh * BS + h) * W + (gw * BS + w);
        for (long c = 0; c < C; ++c)
            cp(o + (i_image + c * (long)W * H) * E, bl + (i_b + c * BS * BS) * E, E);
    }
}

/* transfer_kernel -- blockcopy/blockcopy/utils/block_funcs.py:201-237
 * For each transferred tile b: b_prev = transfer_map[b]; >=0 reads prev_data[b_prev],
 * <0 reads prev_transfer[b_prev + N*GRID_H*GRID_W].  With PADDING>=0 the interior
 * (w,h in [PADDING, BS-PADDING-1]) is skipped (:218-224) and stays whatever `out` held. */
BCO_API void bco_transfer(void *out, const void *prev_data, const void *prev_transfer,
                          const int32_t *transfer_map, int n_transfer,
                          int N, int C, int GRID_H, int GRID_W, int BS, int PADDING, int E)
{
    const long npixels = (long)n_transfer * BS * BS;
    char *o = (char *)out;
    for (long i = 0; i < npixels; ++i) {
        const long b = i / (BS * BS);
        const long h = (i / BS) % BS;
        const long w = i % BS;
        if (PADDING >= 0) {
            if (w >= PADDING && w <= BS - PADDING - 1 && h >= PADDING && h <= BS - PADDING - 1)
                continue;
        }
        const long i_b = b * C * BS * BS + h * BS + w;
        long b_prev = transfer_map[b];
        const int is_exec = b_prev >= 0;
        if (!is_exec) b_prev += (long)N * GRID_H * GRID_W;
        const char *data = (const char *)(is_exec ? prev_data : prev_transfer);
        for (long c = 0; c < C; ++c)
            cp(o + (i_b + c * BS * BS) * E,
               data + (b_prev * C * BS * BS + c * BS * BS + h * BS + w) * E, E);
    }
}

/* repad_kernel -- blockcopy/blockcopy/utils/blockpad.py:77-156
 * out[b, c, h_pad, w_pad] over (BS+2*PAD)^2: interior from features[b]; halo from the
 * neighbouring tile (features if grid_idx>=0 else transfer[grid_idx + N*GH*GW]);
 * zeros where the halo crosses the image border. */
BCO_API void bco_repad(void *out, const void *features, const void *transfer,
                       const int32_t *grid_idx, const int32_t *exec_map, int n_exec,
                       int N, int C, int GRID_H, int GRID_W, int BS, int PAD, int E)
{
    const int BS_PAD = BS + 2 * PAD;
    const long npixels = (long)n_exec * BS_PAD * BS_PAD;
    char *o = (char *)out;
    for (long i = 0; i < npixels; ++i) {
        const long b_pad = i / (BS_PAD * BS_PAD);
        const long h_pad = (i / BS_PAD) % BS_PAD;
        const long w_pad = i % BS_PAD;
        const long i_b = b_pad * C * BS_PAD * BS_PAD + h_pad * BS_PAD + w_pad;

        long b = b_pad;
        long h = h_pad - PAD;
        long w = w_pad - PAD;
        const char *data = (const char *)features;

        const int left = w_pad < PAD;
        const int right = w_pad >= BS_PAD - PAD;
        const int top = h_pad < PAD;
        const int bottom = h_pad >= BS_PAD - PAD;

        int zero_pad = 0;
        if (left || right || top || bottom) {
            const long g_id = exec_map[b];
            const int grid_left = g_id % GRID_W == 0;
            const int grid_right = g_id % GRID_W == GRID_W - 1;
            const int grid_top = (g_id % ((long)GRID_H * GRID_W)) < GRID_W;
            const int grid_bottom = (g_id % ((long)GRID_H * GRID_W)) >= (long)GRID_H * GRID_W - GRID_W;
            zero_pad = (left & grid_left) || (right & grid_right) || (top & grid_top) || (bottom & grid_bottom);
            if (!zero_pad) {
                long g_id_in = g_id;
                g_id_in += (right - left);
                g_id_in += (long)GRID_W * (bottom - top);
                long b_in = grid_idx[g_id_in];
                const int is_exec = b_in >= 0;
                if (!is_exec) b_in += (long)N * GRID_H * GRID_W;
                if (left) w = BS - PAD + w_pad;
                else if (right) w = w_pad - BS_PAD + PAD;
                if (top) h = BS - PAD + h_pad;
                else if (bottom) h = h_pad - BS_PAD + PAD;
                data = (const char *)(is_exec ? features : transfer);
                b = b_in;
            }
        }
        for (long c = 0; c < C; ++c) {
            char *dst = o + (i_b + c * BS_PAD * BS_PAD) * E;
            if (zero_pad) memset(dst, 0, (size_t)E);
            else cp(dst, data + (b * C * BS * BS + c * BS * BS + h * BS + w) * E, E);
        }
    }
}

/* get_grid_mappings -- blockcopy/blockcopy/core/tensorwrapper.py:108-128
 * executed tiles get 0..n_exec-1 in raster order; the others get -n_total + k (k = raster
 * rank among non-executed); mapping_exec = flat indices of executed tiles.  Returns n_exec. */
BCO_API int bco_grid_mappings(const uint8_t *grid, int n_total, int32_t *grid_idx, int32_t *mapping_exec)
{
    int n_exec = 0, n_tr = 0;
    for (int g = 0; g < n_total; ++g) {
        if (grid[g]) { grid_idx[g] = n_exec; mapping_exec[n_exec] = g; ++n_exec; }
        else { grid_idx[g] = -n_total + n_tr; ++n_tr; }
    }
    return n_exec;
}

/* transfer_idx = prev_grid_idx[~grid] -- blockcopy/blockcopy/core/tensorwrapper.py:176-178.
 * Returns n_transfer. */
BCO_API int bco_transfer_idx(const int32_t *prev_grid_idx, const uint8_t *grid, int n_total, int32_t *transfer_idx)
{
    int n_tr = 0;
    for (int g = 0; g < n_total; ++g)
        if (!grid[g]) transfer_idx[n_tr++] = prev_grid_idx[g];
    return n_tr;
}

/* nms_kernel + host sweep -- Pedestron/mmdet/ops/nms/src/nms_kernel.cu:12-130 (the GPU path of mmdet.ops.nms).
 * boxes: (n,5) float [x1,y1,x2,y2,score], ALREADY sorted by score descending (the reference sorts with
 * scores.sort(0, descending=True) at :73-75 before launching).  IoU uses the +1 pixel convention (:12-21); box j is
 * suppressed by an earlier kept box i when IoU > thresh (strict, :61).  Writes the kept positions (indices into the
 * sorted order, ascending) and returns their count.  The 64-wide bitmask blocking of the kernel does not change the
 * result and is not reproduced. */
static float bco_iou(const float *a, const float *b)
{
    float left = a[0] > b[0] ? a[0] : b[0], right = a[2] < b[2] ? a[2] : b[2];
    float top = a[1] > b[1] ? a[1] : b[1], bottom = a[3] < b[3] ? a[3] : b[3];
    float width = right - left + 1.f > 0.f ? right - left + 1.f : 0.f;
    float height = bottom - top + 1.f > 0.f ? bottom - top + 1.f : 0.f;
    float interS = width * height;
    float Sa = (a[2] - a[0] + 1) * (a[3] - a[1] + 1);
    float Sb = (b[2] - b[0] + 1) * (b[3] - b[1] + 1);
    return interS / (Sa + Sb - interS);
}

BCO_API int bco_nms_sorted(const float *boxes, int n, float thresh, int32_t *keep)
{
    int num = 0;
    for (int i = 0; i < n; ++i) {
        int removed = 0;
        for (int k = 0; k < num && !removed; ++k)
            if (bco_iou(boxes + 5 * (long)keep[k], boxes + 5 * (long)i) > thresh) removed = 1;
        if (!removed) keep[num++] = i;
    }
    return num;
}


/* ---------------------------------------------------------------------------------------------------------------------
 * Policy step: Bernoulli sampling of tile logits + rounding the executed count up to a multiple + the bool grid
 * (index tables: bco_grid_mappings above).  The PRODUCT defines this operation (include/blockcopy_hip.h bc_policy_step,
 * csrc/blockcopy_hip.hip k_policy_step): the reference samples with torch's global RNG and Python's `random.sample`
 * (blockcopy/blockcopy/policy/policy.py:124-144, 283-288), which no other implementation can reproduce, so the pinned
 * contract is the product's counter-based RNG restated here operation by operation.  Compile without FP contraction. */
static uint64_t bco_policy_rand(uint64_t seed, uint64_t counter, uint32_t tile, uint32_t stream)
{
    uint64_t z = (seed ^ (counter * 0xD1342543DE82EF95ull)) + (2ull * tile + stream + 1ull) * 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

static float bco_sigmoid_repro(float x)
{
    float t = -x * 1.44269504f;
    t = fminf(fmaxf(t, -126.0f), 126.0f);
    const float nf = rintf(t);
    const float f = t - nf;
    float p = 1.54035304e-4f;
    p = fmaf(p, f, 1.33335581e-3f);
    p = fmaf(p, f, 9.61812911e-3f);
    p = fmaf(p, f, 5.55041087e-2f);
    p = fmaf(p, f, 2.40226507e-1f);
    p = fmaf(p, f, 6.93147181e-1f);
    p = fmaf(p, f, 1.0f);
    uint32_t bits;
    memcpy(&bits, &p, 4);
    bits += (uint32_t)(int32_t)nf << 23;
    float e;
    memcpy(&e, &bits, 4);
    const float d = 1.0f + e;
    return 1.0f / d;
}

/* grid[i] = 1 for executed tiles; returns n_exec; counts = {n_exec, n_sampled, nan flag}; probs (optional) = the sigmoid values */
BCO_API int bco_policy_step(const float *logits, int n_total, uint64_t seed, uint64_t counter, int multiple, int at_least_one,
                               uint8_t *grid, int32_t *counts, float *probs)
{
    int n = 0, nan_flag = 0;
    uint64_t *key = (uint64_t *)malloc(sizeof(uint64_t) * (size_t)n_total);
    for (int i = 0; i < n_total; ++i) {
        const float x = logits[i];
        const int is_nan = x != x;
        const float u = (float)(bco_policy_rand(seed, counter, (uint32_t)i, 0u) >> 40) * 5.9604644775390625e-8f;
        const float pr = bco_sigmoid_repro(x);
        if (probs) probs[i] = pr;
        const int on = !is_nan && u < pr;
        n += on;
        nan_flag |= is_nan;
        key[i] = on ? ~0ull : (((bco_policy_rand(seed, counter, (uint32_t)i, 1u) >> 24) << 16) | (uint64_t)i);
    }
    const int n_sampled = n;
    if (at_least_one && n == 0) { key[0] = ~0ull; n = 1; }
    int rounded = 0;
    if (n > 0) {
        rounded = multiple * (1 + (n - 1) / multiple);
        if (rounded > n_total) rounded = n_total;
    }
    for (int need = rounded - n; need > 0; --need) {      /* switch on the skipped tile with the smallest key, `need` times */
        int best = -1;
        for (int i = 0; i < n_total; ++i)
            if (key[i] != ~0ull && (best < 0 || key[i] < key[best])) best = i;
        key[best] = ~0ull;
        ++n;
    }
    for (int i = 0; i < n_total; ++i) grid[i] = key[i] == ~0ull;
    free(key);
    if (counts) { counts[0] = n; counts[1] = at_least_one && n_sampled == 0 ? 1 : n_sampled; counts[2] = nan_flag; }
    return n;
}
