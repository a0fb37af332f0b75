/*
 * blockcopy_hip.h -- C ABI of libblockcopy_hip.so, the MI355X (gfx950) implementation of the
 * block-copy operator boundary.
 *
 * Plain pointers and sizes only: no torch / HIP types in any signature.  `stream` is a
 * hipStream_t passed as void* (NULL = the default stream).  All device pointers are raw HBM
 * addresses of contiguous NCHW tensors; `elem_size` is the payload element size in bytes
 * (4 = float, 2 = half/bf16; 1 and 8 also accepted -- the ops are pure copies).
 * Every function enqueues asynchronously on `stream`, never allocates or frees device memory,
 * never synchronises, and returns BC_OK (0), a negative BC_ERR_* code for rejected arguments
 * (the reference raises Python asserts for the same conditions, utils/cuda.py:42-48,
 * utils/block_funcs.py:16-30,88-104,164-170, utils/blockpad.py:24-36), or a positive hipError_t.
 *
 * Section A mirrors, one for one, the four kernels the reference launches through CuPy
 * (its "FFI" for this path); section B is the MI355X-first form of the same path used by the
 * engine (fused scatter+copy, halo gather over a persistent ring cache, device-side index tables).
 *
 * Reference citations are relative to blockcopy/blockcopy/ in the reference tree.
 */
#ifndef BLOCKCOPY_HIP_H
#define BLOCKCOPY_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BC_ABI_VERSION 2   /* 2: ring caches keep ACTIVATED values; round-2/3 entry points (conv / stem / policy / indirect scatter+copy) */

enum {
    BC_OK = 0,
    BC_ERR_NULL = -1,      /* required pointer is NULL */
    BC_ERR_SHAPE = -2,     /* non-positive dim, H or W not a multiple of bs, pad < 1 or pad > bs */
    BC_ERR_ELEM = -3,      /* unsupported elem_size / dtype (halo ops: 1,2,4,8; copy ops: 1..2^20) */
    BC_ERR_RANGE = -4,     /* element offsets would not fit the reference's 31-bit index space */
    BC_ERR_ALIGN = -5      /* pointer not aligned to elem_size */
};

/* ---------------------------------------------------------------------------------------------
 * A. Reference operator boundary (drop-in for the four CuPy kernels)
 * ------------------------------------------------------------------------------------------- */

/* gather.  replaces split_kernel + SplitFunction.forward launch, utils/block_funcs.py:10-83.
 * blocks[b,c,h,w] = image[gn,c,gh*bs+h,gw*bs+w], (gn,gh,gw) = unravel(mapping_exec[b], (N,H/bs,W/bs)).
 * blocks: (n_exec,C,bs,bs) caller-allocated; image: (N,C,H,W); mapping_exec: int32[n_exec] on device. */
int bc_split(void *blocks, const void *image, const int32_t *mapping_exec, int n_exec,
             int N, int C, int H, int W, int bs, int elem_size, void *stream);

/* scatter in place.  replaces combine_kernel + CombineFunction.forward, utils/block_funcs.py:85-158.
 * out[gn,c,gh*bs+h,gw*bs+w] = blocks[b,c,h,w]; tiles not in mapping_exec keep their previous contents. */
int bc_combine(const void *blocks, void *out, const int32_t *mapping_exec, int n_exec,
               int N, int C, int H, int W, int bs, int elem_size, void *stream);

/* border-ring transfer.  replaces transfer_kernel + TransferFunction.forward, utils/block_funcs.py:161-237.
 * out[b] (ring of width `padding` only; interior untouched = don't-care) = transfer_idx[b] >= 0 ?
 * prev_computed[transfer_idx[b]] : prev_transfer[transfer_idx[b] + N*GH*GW].  padding < 0 copies whole tiles. */
int bc_transfer(void *out, const void *prev_computed, const void *prev_transfer,
                const int32_t *transfer_idx, int n_transfer,
                int N, int C, int GH, int GW, int bs, int padding, int elem_size, void *stream);

/* halo gather.  replaces repad_kernel + BlockPadFunction.forward, utils/blockpad.py:21-156.
 * out: (n_exec,C,bs+2*pad,bs+2*pad) caller-allocated (the reference allocates it inside, blockpad.py:45-46;
 * the binding does that with the torch caching allocator).  Interior from features[b]; halo from the
 * neighbouring tile: features[grid_idx[g']] if >= 0 else transfer[grid_idx[g'] + N*GH*GW]; zeros beyond
 * the image border. */
int bc_pad(void *out, const void *features, const void *transfer, const int32_t *grid_idx,
           const int32_t *mapping_exec, int n_exec,
           int N, int C, int GH, int GW, int bs, int pad, int elem_size, void *stream);

/* ---------------------------------------------------------------------------------------------
 * B. MI355X-first forms of the same path
 * ------------------------------------------------------------------------------------------- */

/* fused scatter + copy: ONE pass that writes the whole dense map `out` (N,C,H,W), taking each tile from
 * `blocks` when grid_idx[tile] >= 0 and from `prev` (previous frame's dense map, same shape) otherwise.
 * Equals prev.clone() followed by bc_combine (core/tensorwrapper.py:421-433) in one kernel and with
 * 2*N*C*H*W*elem_size bytes of traffic instead of 2x that plus the executed tiles.  out must not alias
 * prev.  prev may be NULL only if every tile is executed. */
int bc_combine_copy(const void *blocks, const void *prev, void *out, const int32_t *grid_idx,
                    int N, int C, int H, int W, int bs, int elem_size, void *stream);

/* The same pass as a hipGraph node.  The op's contract makes `out` a fresh map every frame (the reference hands a new tensor to
 * the caller, core/tensorwrapper.py:421-433) and `prev` the map of the frame before, but the arguments of a captured kernel node
 * are frozen; this form therefore reads both addresses from device memory at run time:
 *     slots (device, 8-byte aligned, 3 x uint64): [0] = prev, [1] = out, [2] = timing record or 0.
 * The caller refreshes `slots` before every replay (the engine sends it with the frame's index tables: one H->D copy).
 * `align` = power of two that every future prev / out address is a multiple of (>= 16 for full-width vectors).
 * prev is never dereferenced when every tile is executed (it must still be a valid address or equal to out).
 * Timing record (measurement only): an array of ceil(vectors / 256) cells of two uint64 (bc_combine_copy_cells gives the
 * count); workgroup i leaves {its entry time, the time its store was acknowledged} of the constant 100 MHz clock
 * (s_memrealtime) in cell i -- graph kernel nodes cannot carry start / stop events; launch time = max(exit) - min(entry). */
int bc_combine_copy_indirect(const void *blocks, const void *slots, const int32_t *grid_idx,
                             int N, int C, int H, int W, int bs, int elem_size, int align, void *stream);
/* Executed-tile count known only on the DEVICE.  A policy that decides on the device (bc_policy_step) leaves the count in device memory;
 * a host that does not want to wait for it launches every op of the frame sized for a CEILING (normally: every tile executed) and arms
 * each launch with
 *     bc_dyn_set(n_exec_dev, ceiling)      n_exec_dev: device int32, read when the kernel RUNS; ceiling: the n_exec the launch is sized for
 * which applies to the NEXT launch of one of the functions below only (they consume it at entry; bc_dyn_set(NULL, 0) disarms).  The
 * kernel then works on the first *n_exec_dev tiles / packed rows and its surplus workgroups exit at once, so ONE captured hipGraph
 * serves every count (the reference needs the count on the host for every launch: int(grid.sum()) policy/policy.py:84, tensor shapes).
 * Index tables: rows >= *n_exec_dev of mapping_exec are never read.  Capable: bc_split, bc_combine, bc_pad_ring, bc_pad_ring_act, bc_pad_ring_nhwc,
 * bc_pad_ring_add_nhwc, bc_maxpool3x3s2_ring_nhwc, bc_conv3x3_ring_nhwc, bc_conv3x3s2_ring_nhwc, bc_conv3x3_dil_ring_nhwc,
 * bc_conv1x1_nhwc, bc_stem7x7s2_nhwc, bc_head1x1_scatter_nhwc (the copy half then always runs), bc_affine_act_nhwc,
 * bc_interp_bilinear(_act)_nhwc; bc_tile_copy_indirect takes the pointer as an argument.  Tile-indexed launches must be sized exactly
 * for `ceiling` (BC_ERR_SHAPE otherwise); pointwise ones (affine, interp, conv1x1) scale their unit count by *n_exec_dev / ceiling.
 * The armed state is PER HOST THREAD (thread_local): an arm is consumed by the next capable launch of the thread that set it, so two
 * threads (or two engines driven from two threads) of one process never take each other's arm. */
int bc_dyn_set(const void *n_exec_dev, int ceiling);

/* The network-INPUT stage of a graph-replayed frame: dst[tile] = src[tile] for every executed tile (mapping_exec) of two dense maps of
 * one geometry (N,C,H,W) -- src = the caller's frame, dst = the persistent frame-state map.  Equals the reference's
 * to_blocks(split) + combine_ of the network input (core/blockcopy.py:62-68, frame_state = "latest executed frame per block") without
 * the packed tensor and without a staging copy of the frame: the source address is read at run time from the device word
 * `src_slot` (uint64, refreshed by the host with the frame's index tables), as in bc_combine_copy_indirect.
 * `align` = power of two every future source address is a multiple of.  n_exec_dev: NULL, or a device int32 holding the executed-tile
 * count of THIS replay (<= n_exec, which sizes the launch): frames whose count the host does not know when it launches. */
int bc_tile_copy_indirect(void *dst, const void *src_slot, const int32_t *mapping_exec, const int32_t *n_exec_dev, int n_exec,
                          int N, int C, int H, int W, int bs, int elem_size, int align, void *stream);
/* The network's last stage in one launch (csrc/head1x1.inc): activation prologue + pointwise conv to Cout <= 32 channels + the
 * out-of-place combine.  Replaces, for a model whose head is BN -> ReLU -> 1x1 conv on packed tiles followed by
 * `out.combine()` (SwiftNet logits: semantic_segmentation/lib/models/swiftnet/swiftnet.py via util.py:40-55, then
 * core/blockcopy.py:79 / core/tensorwrapper.py:421-433), the elementwise pass, the library conv and bc_combine_copy.
 *   features        packed channels-last tiles (n_exec, bs, bs, Cin); Cin in {64, 128} (fp32) / {64, 128, 256} (16-bit)
 *   weights_packed  the (Cout, Cin, 1, 1) weight zero-padded to 32 output channels in the one-tap operand order of
 *                   bc_conv1x1_nhwc (nb = 1): wpk[step][lane][j] = W[lane % 32][2*EPV*step + EPV*(lane / 32) + j]
 *   in_scale/in_shift/in_relu   per input channel prologue relu?(x*scale + shift) (fp32, NULL = identity)
 *   out_shift       per output channel bias (fp32[Cout]) or NULL
 *   scatter = 0     out = packed tiles (n_exec, bs, bs, Cout); prev, slots, grid_idx, mapping_exec unused
 *   scatter = 1     out = the fresh dense map (N, GH*bs, GW*bs, Cout) channels-last: executed tiles are written at their grid
 *                   position (mapping_exec), every skipped tile (grid_idx < 0) is copied from `prev` (same shape; may be NULL
 *                   only if every tile is executed).  slots != NULL: prev / out are read from slots[0] / slots[1] at run time
 *                   as in bc_combine_copy_indirect (hipGraph node), the out / prev arguments are ignored.  slots[2] != 0 (measurement only):
 *                   a timing record of 16-byte cells -- cell 0 = {capacity in cells, -} written by the CALLER, cell 1 + i = {entry, exit}
 *                   of workgroup i's first wave on the constant 100 MHz clock (workgroups beyond the capacity leave nothing): a graph
 *                   kernel node cannot carry events; launch time = max(exit) - min(entry) over the written cells (bench.py `roofline`).
 * bs: multiple of 8, and of 32 when larger than 32.  Results: fp32 accumulation over Cin in matrix-core order, one rounding. */
int bc_head1x1_scatter_nhwc(void *out, const void *features, const void *weights_packed, const void *prev, const void *slots,
                            const int32_t *grid_idx, const int32_t *mapping_exec, int n_exec, int N, int Cin, int Cout,
                            int GH, int GW, int bs, int dtype, const float *in_scale, const float *in_shift, int in_relu,
                            const float *out_shift, int scatter, void *stream);
/* Dense 3x3 conv (stride 1, zero padding 1) from Cin channels to 1..4 output channels on a channels-last map (N, H, W, Cin) ->
 * (N, H, W, Cout): the prediction convs a detector head applies to the combined map right after blockcopy.to_tensor (reference
 * Pedestron/mmdet/models/anchor_heads/csp_head.py:103-108 csp_cls / csp_reg / csp_offset, applied at :139,146,151; in the
 * reference that is nn.Conv2d -> cuDNN on the dense tensor).  A bandwidth op (the map is read once: (Cin + Cout) * E bytes per pixel).
 *   weights_packed  fp32 [Cin][3][3][Cout] (= weight.permute(1, 2, 3, 0), fp32 whatever the map's dtype)
 *   bias            fp32 [Cout] or NULL
 * Cin: multiple of 32.  fp32 accumulation per pixel in channel order within a tap, taps added in (ky, kx) order, bias first;
 * one rounding to the map's dtype. */
int bc_pred3x3_nhwc(void *out, const void *x, const float *weights_packed, const float *bias, int N, int H, int W, int Cin, int Cout,
                    int dtype, void *stream);
/* Pyramid pooling of a dense channels-last map x (H, W, C) in two launches (reference semantic_segmentation/lib/models/swiftnet/util.py
 * SpatialPyramidPooling.forward after its first block, run dense inside blockcopy_noblocks, core/blockcopy.py:104-139):
 *   bc_spp_levels_nhwc   lv[bin0[l] + by * gw_l + bx][CO] = conv1x1_l(relu(bn_l(adaptive_avg_pool2d(x, (gh_l, gw_l)))))  for every level l
 *                        grids = {gh_0, gw_0, gh_1, gw_1, ...} (L <= 4 levels); scale / shift fp32 [L][C] (the levels' folded BN, NULL =
 *                        identity); weights fp32 [L][C][CO].  ATen's bin limits; the pooled value and the BN -> ReLU result are rounded
 *                        to the map's dtype like the stock passes, the conv accumulates in fp32 in channel order.
 *   bc_spp_fuse_nhwc     out (H, W, N) = conv1x1_f(relu(bn_f(cat[x, upsample_bilinear(lv_0), ..., upsample_bilinear(lv_{L-1})])))
 *                        without materialising the upsampled maps or the concatenation (align_corners = False, ATen's source index;
 *                        sampled value and BN -> ReLU result rounded to the map's dtype); scale / shift fp32 [C + L * CO];
 *                        weights_packed = the one-tap stream (pack layout of bc_conv1x1_nhwc, fp32 whatever the map's dtype) of
 *                        the (N, K', 1, 1) weight zero-padded to K' = roundup(C + L * CO, 32) input channels; N a multiple of 64.
 * C a multiple of 4 with 256 % (C / 4) == 0. */
int bc_spp_levels_nhwc(void *lv, const void *x, const float *scale, const float *shift, const float *weights, int H, int W, int C, int CO,
                       int L, const int32_t *grids, int dtype, void *stream);
int bc_spp_fuse_nhwc(void *out, const void *x, const void *lv, const float *scale, const float *shift, const void *weights_packed, int H, int W,
                     int C, int CO, int L, const int32_t *grids, int N, int dtype, void *stream);
/* the same two over a batch of B independent maps in one launch each: x (B, H, W, C), lv (B, bins, CO), out (B, H, W, N) */
int bc_spp_levels_n_nhwc(void *lv, const void *x, const float *scale, const float *shift, const float *weights, int B, int H, int W, int C,
                         int CO, int L, const int32_t *grids, int dtype, void *stream);
int bc_spp_fuse_n_nhwc(void *out, const void *x, const void *lv, const float *scale, const float *shift, const void *weights_packed, int B,
                       int H, int W, int C, int CO, int L, const int32_t *grids, int N, int dtype, void *stream);
/* the fuse launch with a PACKED result (one map): only the pixels of the executed tiles are computed, out (n_exec, bs, bs, N) = what
 * bc_split of the dense result would give -- the re-packing gather after the dense module (core/blockcopy.py:118-121, to_blocks_like)
 * and the never-used pixels of the skipped tiles disappear.  H, W multiples of bs; mapping_exec as everywhere (flat grid positions). */
int bc_spp_fuse_packed_nhwc(void *out, const void *x, const void *lv, const float *scale, const float *shift, const void *weights_packed,
                            const int32_t *mapping_exec, int n_exec, int bs, int H, int W, int C, int CO, int L, const int32_t *grids, int N,
                            int dtype, void *stream);
/* number of timing cells (= workgroups) such a launch writes, or a negative error code */
int bc_combine_copy_cells(const void *blocks, int N, int C, int H, int W, int bs, int elem_size, int align);

/* halo gather over a persistent ring cache.  `ring` is a (N*GH*GW, C, 4*pad*bs) device buffer owned by the
 * caller and kept across frames for one padded layer: per grid position and channel the four contiguous segments
 * [top pad rows | bottom pad rows | left pad cols | right pad cols] of the tile most recently executed there
 * (row-major; everything a neighbour's halo of width `pad` can need, nothing else).  Same output as bc_transfer + bc_pad of the reference decomposition, but non-executed
 * neighbours are read from ring[g'] (indexed by grid position, no per-frame compaction) and every executed
 * tile refreshes ring[g] with its own border in the same launch -- the transfer kernel and its tensors
 * disappear. */
int bc_pad_ring(void *out, const void *features, void *ring, const int32_t *grid_idx,
                const int32_t *mapping_exec, int n_exec,
                int N, int C, int GH, int GW, int bs, int pad, int elem_size, void *stream);

/* index tables on device.  replaces get_grid_mappings (core/tensorwrapper.py:108-128) and the
 * prev_grid_idx[~grid] lookup (:176-178), which the reference runs on the CPU behind two D->H syncs.
 * grid: uint8/bool[n_total] on device.  Writes grid_idx int32[n_total], mapping_exec int32[<= n_total],
 * counts[0] = n_exec, counts[1] = n_transfer; if prev_grid_idx != NULL also transfer_idx int32[<= n_total].
 * One workgroup; no host synchronisation. */
int bc_grid_tables(const uint8_t *grid, int n_total, int32_t *grid_idx, int32_t *mapping_exec,
                   const int32_t *prev_grid_idx, int32_t *transfer_idx, int32_t *counts, void *stream);

/* the same tables on the host (for policies whose grid is already host-resident).  Returns n_exec. */
int bc_grid_tables_host(const uint8_t *grid, int n_total, int32_t *grid_idx, int32_t *mapping_exec,
                        const int32_t *prev_grid_idx, int32_t *transfer_idx);

/* per-tile bilinear resampling of a packed batch: in (planes, h, w) -> out (planes, H, W), planes = n_exec*C.
 * Replaces the reference's INTERPOLATE route (core/tensorwrapper.py:577-598: bilinear re-expressed as trilinear
 * on a (1,B,C,h,w) view because the stock bilinear kernel serialises over tiles x channels -- on MI355X it costs
 * 1.1 ms per call at SwiftNet's decoder shapes).  No halo: a tile's border is interpolated from the tile alone,
 * exactly as in the reference.  Arithmetic = PyTorch's upsample_bilinear2d (source index scale*(dst+0.5)-0.5
 * clamped at 0, or scale*dst with align_corners; fp32 accumulation; rh/rw are the already-resolved
 * input/output scales).  dtype: 0 = float32, 1 = float16, 2 = bfloat16. */
enum { BC_F32 = 0, BC_F16 = 1, BC_BF16 = 2 };
int bc_interp_bilinear(void *out, const void *in, long long planes, int h, int w, int H, int W,
                       int align_corners, float rh, float rw, int dtype, void *stream);

/* fused per-block ops around the convs.  The reference runs bias / batch-norm / ReLU /
 * residual add as separate PyTorch launches on the packed batch (core/tensorwrapper.py:478-527 passes them through);
 * here the engine keeps them pending and folds them into these two entry points.
 *
 * bc_pad_ring_act = bc_pad_ring with an activation PROLOGUE: every gathered real value x of channel c becomes
 * relu?(x*scale[c] + shift[c]) (fp32 arithmetic; scale/shift are float32[C] device vectors, either may be NULL),
 * zeros beyond the image border stay zero.  The ring cache keeps the ACTIVATED values (what the padded op sees): values read
 * from packed tiles are transformed, values read from ring records never are, so a record is valid whichever route wrote
 * it (this prologue, or a plain gather of an input whose activation had been materialised by its producer).  With
 * scale = shift = NULL and relu = 0 it is the pure copy.
 * bc_affine_act: out = relu?(in*scale[c] + shift[c] + add) over a packed (B,C,hw) tensor; add may be NULL; out may
 * alias in. */
int bc_pad_ring_act(void *out, const void *features, void *ring, const int32_t *grid_idx,
                    const int32_t *mapping_exec, int n_exec, int N, int C, int GH, int GW, int bs, int pad,
                    int dtype, const float *scale, const float *shift, int relu, void *stream);
int bc_affine_act(void *out, const void *in, const void *add, const float *scale, const float *shift, int relu,
                  long long B, int C, long long hw, int dtype, void *stream);

/* channels-last (NHWC) forms.  A channels-last (N,C,H,W) map is an NCHW (N,1,H,W) map of C*elem_size-byte units, so
 * bc_split / bc_combine / bc_combine_copy (and bc_transfer) serve it unchanged when called with C = 1 and
 * elem_size = C*E (any unit size from 1 to 2^20 bytes is accepted by those copy ops).  The three ops below need the
 * channel index and have their own channels-last kernels; packed tiles are (n_exec, bs, bs, C) in memory, padded tiles
 * (n_exec, bs+2p, bs+2p, C), ring records (N*GH*GW, 4*pad*bs, C).  In this layout every access of the halo gather is an
 * aligned vector (C*elem_size must be a multiple of 2 bytes; 16 for full speed).
 * bc_pad_ring_nhwc: scale = shift = NULL and relu = 0 -> pure copy (dtype ignored). */
int bc_pad_ring_nhwc(void *out, const void *features, void *ring, const int32_t *grid_idx,
                     const int32_t *mapping_exec, int n_exec, int N, int C, int GH, int GW, int bs, int pad,
                     int elem_size, int dtype, const float *scale, const float *shift, int relu, void *stream);
int bc_affine_act_nhwc(void *out, const void *in, const void *add, const float *scale, const float *shift, int relu,
                       long long pixels, int C, int dtype, void *stream);
int bc_interp_bilinear_nhwc(void *out, const void *in, long long planes, int C, int h, int w, int H, int W,
                            int align_corners, float rh, float rw, int dtype, void *stream);
/* the same with an epilogue on the interpolated value y (rounded to the tensor dtype first, as a separate launch would see
 * it): out = relu?(y*scale[c] + shift[c] + add), add (planes, H, W, C) or NULL -- "upsample, then += skip" in one launch. */
int bc_interp_bilinear_act_nhwc(void *out, const void *in, long long planes, int C, int h, int w, int H, int W,
                                int align_corners, float rh, float rw, int dtype, const float *scale, const float *shift,
                                const void *add, int relu, void *stream);

/* prediction map of a segmentation frame: out[n][y][x] (int64) = arg-max over the C classes of the logits map `in`, bilinearly
 * resampled to (H, W) with torch's upsample_bilinear2d arithmetic (align_corners as given; rh / rw = the source-index scales, in / out
 * for a `size=` call) -- the reference driver's F.interpolate(out, size, mode='bilinear') + out.max(dim=1)[1]
 * (semantic_segmentation/test_swiftnet.py:190-194) without the (N, C, H, W) intermediate.  `in` may have any layout: sn / sc / sy /
 * sx are its element strides.  Ties: the first maximal class; NaN is maximal (torch.max). */
int bc_upsample_argmax(long long *out, const void *in, int N, int C, int h, int w, int H, int W, long long sn, long long sc,
                       long long sy, long long sx, int align_corners, float rh, float rw, int dtype, void *stream);

/* fused halo gather + 3x3 / stride 1 / pad 1 convolution of a packed channels-last tile batch on the matrix cores:
 * out = epilogue(conv3x3(prologue(halo-padded tiles))) in ONE launch, without materialising the padded tensor.
 * Replaces, for one padded conv layer of the reference (core/tensorwrapper.py:478-527: BlockPad.apply, then the stock
 * F.conv2d with padding 0 on the padded batch), the sequence bc_pad_ring_nhwc + library conv [+ bc_affine_act_nhwc].
 *   features (n_exec, bs, bs, Cin), out (n_exec, bs, bs, Cout), ring (N*GH*GW, 4*bs, Cin) as for bc_pad_ring_nhwc
 *   with pad = 1 (read for non-executed neighbours, refreshed with the ACTIVATED (post-prologue) border of every executed tile);
 *   prologue: x -> relu?(x*in_scale[cin] + in_shift[cin]) on real values, zeros beyond the image border stay zero;
 *   epilogue: y -> relu?(y*out_scale[cout] + out_shift[cout] + out_add[pixel, cout]); any of them may be NULL/0.
 *   weights_packed: 9 * Cin * Cout elements of the tensor dtype in the MFMA operand order
 *       wpk[nb][unit][tap][step][lane][j] = W[cout = 32*nb + lane%32][cin = 32*unit + 2*EPV*step + EPV*(lane/32) + j][ky = tap/3][kx = tap%3]
 *   with EPV = 16 / elem_size (nb < Cout/32, unit < Cin/32, tap 0..8, step < 16/EPV, lane < 64, j < EPV) -- a pure permutation of
 *   the (Cout, Cin, 3, 3) weight.
 * Arithmetic: BC_F32: exact fp32 (v_mfma_f32_32x32x2_f32 = k-ordered fma chain); BC_F16 / BC_BF16: v_mfma_f32_32x32x16_f16 / _bf16
 * with fp32 accumulation, epilogue in fp32, ONE rounding to the tensor dtype at the store; the prologue rounds each gathered
 * element to the tensor dtype exactly like bc_pad_ring_nhwc does.  Summation order: cin-unit / tap / channel (K groups of one
 * workgroup are added in a fixed order: deterministic).
 * Constraints: Cin % 32 == 0 (16-bit: % 64), Cout % 64 == 0, bs = 4 or a multiple of 8 (<= 248), 16-byte aligned pointers;
 * bs = 2 (ResNet-50's last stage at block 64) for fp16, and for fp32 where the launch is always served by the split form (code | 0x2000:
 * the only form compiled for 2x2-pixel tiles; bf16: BC_ERR_SHAPE). */
int bc_conv3x3_ring_nhwc(void *out, const void *features, void *ring, const void *weights_packed,
                         const int32_t *grid_idx, const int32_t *mapping_exec, int n_exec, int N, int Cin, int Cout,
                         int GH, int GW, int bs, int dtype, const float *in_scale, const float *in_shift, int in_relu,
                         const float *out_scale, const float *out_shift, const void *out_add, int out_relu, void *stream);

/* the same for a 3x3 / STRIDE 2 / pad 1 conv (the first conv of a ResNet stage): features (n_exec, bs, bs, Cin) ->
 * out (n_exec, bs/2, bs/2, Cout); halo, ring (pad 1, over the INPUT tiles), prologue, epilogue and weight stream as above.
 * bs even with bs/2 = 4 or a multiple of 8, or bs/2 = 2 as above. */
int bc_conv3x3s2_ring_nhwc(void *out, const void *features, void *ring, const void *weights_packed,
                           const int32_t *grid_idx, const int32_t *mapping_exec, int n_exec, int N, int Cin, int Cout,
                           int GH, int GW, int bs, int dtype, const float *in_scale, const float *in_shift, int in_relu,
                           const float *out_scale, const float *out_shift, const void *out_add, int out_relu, void *stream);

/* pointwise (1x1, pad 0) conv of a channels-last tensor on the matrix cores: the GEMM of the fused kernel with one tap, so that the
 * BN / ReLU recorded BEFORE the conv (prologue) and the bias / residual add / ReLU recorded AFTER it (epilogue) cost no launch of
 * their own (ResNet bottlenecks and decoder blocks are  ... -> ReLU -> conv1x1 -> ...).  features = n_tiles tiles of bs x bs pixels x Cin
 * (stride 1: any view with the same number of pixels, e.g. a dense map as 8x8 tiles; stride 2: the real packed tiles, out has
 * bs/2 x bs/2 per tile); weights_packed: the bc_conv3x3_ring_nhwc stream with a single tap; prologue / epilogue as there.
 * Cin % 32 == 0 (16-bit: % 64), Cout % 64 == 0, bs/stride a multiple of 8 (or 4, or 2 as above, with stride 2).  A stride-2 launch stages only
 * the pixels it reads (every second pixel of every second row). */
int bc_conv1x1_nhwc(void *out, const void *features, const void *weights_packed, int n_tiles, int Cin, int Cout, int bs, int stride,
                    int dtype, const float *in_scale, const float *in_shift, int in_relu, const float *out_scale,
                    const float *out_shift, const void *out_add, int out_relu, void *stream);
int bc_conv1x1_candidates(int dtype, int stride, int n_tiles, int Cin, int Cout, int bs_in, int *out, int max_out);
/* A decoder's  x = F.interpolate(x, 2x, 'bilinear'); x += conv1x1(skip)  (semantic_segmentation/models/util.py _Upsample.forward; the
 * reference resamples the packed tiles, every tile by itself, then adds) as the EPILOGUE of the 1x1 conv: arms the NEXT bc_conv1x1_nhwc
 * (consumed at its entry, like bc_dyn_set) with "+ bilinear(src)":  out[t][y][x][c] = conv(...)*out_scale + out_shift + bilinear(src[t])[y][x][c]
 * (+ out_add, ReLU), src = the coarser packed map (n_exec, src_bs, src_bs, Cout) in the tensor dtype, out_bs = the real tile size of
 * `out` (a power of two >= 2; the armed call itself sees 8x8 re-tiles: bs = 8, stride 1, the direct one-tap form), rh / rw / align_corners =
 * the source-index scale of F.interpolate (in / out for align_corners = 0).  Index arithmetic, clamping at the TILE border and rounding
 * are those of bc_interp_bilinear_nhwc on the packed batch.  No resampling launch, no round trip of the sum through memory.
 * BC_ERR_SHAPE from the armed call when it cannot carry the term (stride 2, a GEMM-form decomposition forced, tile size not 2^k).
 * Per host thread, like bc_dyn_set: only the arming thread's next bc_conv1x1_nhwc takes the term. */
int bc_conv_upsample_arm(const void *src, int src_bs, int out_bs, int align_corners, float rh, float rw);

/* network-input stage in ONE launch: (bs+6)^2 window gather from the frame-state map + the 7x7 / stride 2 / pad 3 conv of the
 * 3-channel frame (ResNet stem) + epilogue, on the matrix cores.  Replaces, for the first padded op of the reference pipeline,
 * split + transfer + repad (p = 3) + F.conv2d (core/tensorwrapper.py:304-381, 529-575; utils/blockpad.py:77-156): for the network
 * input the halo of a tile IS the frame-state map (the dense map holding the most recently executed pixels of every tile, i.e.
 * bc_combine of this frame's packed input into last frame's map), zeros beyond the image; no ring cache is involved.
 *   frame_state (N, 3, H, W) contiguous NCHW; out (n_exec, bs/2, bs/2, 64) channels-last = relu?(conv * out_scale[c] + out_shift[c] + out_add);
 *   weights_packed: per output-row tap ky the 21 (kx, c) values as one K segment, zero-padded to 24 (fp32) / 32 (16-bit):
 *     fp32   wpk[nb][ky][t4<3][lane][j<4] = Wseg[32*nb + lane%32][ky][8*t4 + 2*j + lane/32]
 *     16-bit wpk[nb][ky][s<2][lane][j<8]  = Wseg[32*nb + lane%32][ky][16*s + 8*(lane/32) + j],   Wseg[co][ky][3*kx + c] = W[co][c][ky][kx]
 *   Cout = 64, bs/2 a multiple of 32.
 *   bc_tune_set("stem_split", 1): an fp32 frame is multiplied on the 16-bit matrix pipe at fp32 accuracy (every operand split hi + lo into two
 *   fp16 numbers, 16 x = hi + lo, three MFMAs per product; measured error <= 1e-6 of the result's largest element): weights_packed then holds,
 *   behind the fp32 stream, the hi and the lo stream in the 16-bit order (blockcopy.backend.pack_stem7x7_weights; 2 x 14 x 64 vectors each). */
int bc_stem7x7s2_nhwc(void *out, const void *frame_state, const void *weights_packed, const int32_t *mapping_exec, int n_exec,
                      int N, int H, int W, int bs, int Cout, int dtype, const float *out_scale, const float *out_shift,
                      const void *out_add, int out_relu, void *stream);

/* group_norm on packed tiles.  The reference folds the tile axis into the spatial axis so that the statistics of a group run over
 * ALL executed tiles of the frame (core/tensorwrapper.py:600-633: F.group_norm on the (1, C, B*h*w, 1) view).  On channels-last
 * packed tiles (a contiguous (n_pix, C) matrix) the op is a per-channel affine map; this entry computes it in one read of the
 * tensor: scale[c] = gamma[c] * rstd[g(c)], shift[c] = beta[c] - mean[g(c)] * scale[c] (biased variance, fp32 partial sums per
 * workgroup, combined in double in a fixed order: deterministic).  The engine records (scale, shift) as pending work: the next
 * kernel applies it as its prologue.  C * elem_size a multiple of 16 and <= 4096 bytes with 256 % (C*elem_size/16) == 0;
 * gamma / beta may be NULL; workspace: >= 512 * C * 2 floats of scratch. */
int bc_group_norm_affine_nhwc(const void *features, long long n_pix, int C, int groups, int dtype, float eps, const float *gamma,
                              const float *beta, float *scale, float *shift, float *workspace, long long workspace_floats, void *stream);

/* training-mode BatchNorm2d forward (batch statistics) of a contiguous NCHW float32 tensor, optional fused ReLU: the per-frame
 * normalisation of the online-RL policy net (reference policy/policy.py keeps the net in train() mode; policy/net.py, resnet.py).
 * y = relu?((x - mean[c]) * invstd[c] * gamma[c] + beta[c]); save_mean / save_invstd (what aten::native_batch_norm_backward
 * takes), running_mean / running_var (momentum update, unbiased variance) and num_batches_tracked (+1) are updated when non-NULL.
 * Two launches (partial sums; normalise) instead of the library's five; deterministic.  workspace: >= C * 64 * 2 floats. */
int bc_bn_train_fwd(void *y, const void *x, int N, int C, long long HW, const float *gamma, const float *beta, float *running_mean,
                    float *running_var, long long *num_batches_tracked, float *save_mean, float *save_invstd, float momentum,
                    float eps, int relu, float *workspace, long long workspace_floats, void *stream);

/* channels-last form of the above, statistics only: per-channel scale = gamma*invstd, shift = beta - mean*scale of a (n_pix, C)
 * matrix (+ save_mean / save_invstd / running statistics / batch counter), to be applied with bc_affine_act_nhwc(relu).  Same two
 * kernels as bc_group_norm_affine_nhwc with one channel per group.  workspace: >= 512 * C * 2 floats. */
int bc_bn_train_stats_nhwc(const void *features, long long n_pix, int C, int dtype, float eps, const float *gamma, const float *beta,
                           float *running_mean, float *running_var, long long *num_batches_tracked, float momentum, float *save_mean,
                           float *save_invstd, float *scale, float *shift, float *workspace, long long workspace_floats, void *stream);

/* F.adaptive_avg_pool2d(in, (OH, OW)) of a channels-last (N, C, H, W) tensor (ATen's bin limits: start = floor(i*H/OH),
 * end = ceil((i+1)*H/OH)); out (N, C, OH, OW) channels-last.  The pyramid pooling of SwiftNet runs on a dense map inside a
 * blockcopy_noblocks module (reference semantic_segmentation models; core/blockcopy.py:104-139): three calls per frame that cost
 * 11.5 us each with the stock kernel.  C * elem_size a multiple of 16 bytes, <= 4096, 256 % (C*elem_size/16) == 0. */
int bc_adaptive_avg_pool_nhwc(void *out, const void *in, int N, int C, int H, int W, int OH, int OW, int dtype, void *stream);

/* decompositions of the fused conv kernel that cover a layer (stride 1 or 2; bs_in = input tile size): codes written to out,
 * count returned.  What bc_tune_set("conv2_cfg", code) may force; the engine times exactly these when it measures a layer shape.
 * code = decomposition index (bits 0-7) | 0x100 if the launch runs without the one-workgroup-per-CU LDS floor (two workgroups
 * may then share a CU; listed only for multi-round launches whose workgroups need <= 78 KB) | 0x200 for the Winograd F(2x2,3x3)
 * form (fp32, stride 1, 3x3: csrc/conv3x3_wino.inc; index = its own decomposition table, 11..13 = the variants whose input
 * transform is computed once per workgroup) | 0x400 for its wide wave tile (csrc/conv3x3_wino32.inc) | 0x800 (pointwise convs,
 * fp32, stride 1: bc_conv1x1_candidates) for the plain-GEMM form csrc/gemm1x1.inc (index 0..3 = workgroup tile 128x128, 128x64,
 * 64x128, 64x64; reads the same packed one-tap weight stream) | 0x1000 for the Winograd F(4x4,3x3) form (fp32, stride 1, 3x3, tiles
 * of a multiple of 16 pixels or 8x8 tiles: csrc/conv3x3_wino4.inc; index 0..2 = (columns of 16*NB output channels, frequency groups,
 * NB) = (4,2,1), (2,4,1), (2,4,2) of its eight waves; | 0x4000 on top, i.e. 0x5000 | index: the same with its 36 element-wise products on the
 * 16-bit matrix pipe -- transformed input and weights split hi + lo in fp16 as for 0x2000 below, three v_mfma_f32_16x16x16_f16 where the fp32 pipe runs
 * four v_mfma_f32_16x16x4_f32; needs |x| < 655) | 0x2000 (fp32 tensors, 3x3 and pointwise, both strides): decomposition `index` of the
 * direct form on the 16-BIT matrix pipe at fp32 accuracy -- every staged value is split x = hi + lo into two fp16 numbers (16 x = hi + lo: |x| < 4094,
 * 22 bits of mantissa) and a product is hi*hi + hi*lo + lo*hi on v_mfma_f32_32x32x16_f16 with fp32 accumulation (three MFMAs of 16 channels where the
 * fp32 pipe runs eight of 2; measured error 3e-7 .. 1e-6 of the result's largest element, the fp32 direct form's own level); reads its own weight
 * stream (below).  At most 128 codes.
 * The Winograd forms read further weight streams placed behind the direct one in weights_packed (fp32 3x3 only; 9 + 16 + 16 + 36 = 77
 * floats per (cin, cout) pair in all: direct, F(2x2) 16-channel tile, F(2x2) wide tile (conv3x3_wino32.inc), F(4x4)):
 *   wino[nb16][chunk][step < 4][q < 8][lane = 16*kq + n][e < 4] = (G g Gt)[f][cin = 32*chunk + 8*step + 2*kq + t][cout = 16*nb16 + n],
 *   2*f + t = 4*q + e, f = 4*xi + nu, G = [1 0 0; .5 .5 .5; .5 -.5 .5; 0 0 1]  (16 * Cin * Cout floats after the 9 * Cin * Cout);
 *   wino4[cb][chunk][f < 36][lane = 16*kq + n][j < 4] = (G4 g G4t)[f][cin = 16*chunk + 4*kq + j][cout = 16*cb + n], f = 6*xi + nu,
 *   G4 = [1/4 0 0; -1/6 -1/6 -1/6; -1/6 1/6 -1/6; 1/24 1/12 1/6; 1/24 -1/12 1/6; 0 0 1]  (36 * Cin * Cout floats after the first 41 * Cin * Cout;
 *   products computed in fp64, rounded once).  A caller that never forces a 0x1000 code may pass a buffer without the last stream.
 *   split (codes | 0x2000): the direct stream position by position, every 16-byte vector of four weights w0..w3 replaced by the eight fp16 numbers
 *   [hi0..3 | lo0..3] with 16 w = hi + lo (9 * Cin * Cout floats after the first 77 * Cin * Cout; pointwise convs: Cin * Cout floats after the first
 *   Cin * Cout);
 *   F(4x4) split (codes 0x5000 | index): the wino4 stream position by position, [hi0..3 | lo0..3] with 256 U = hi + lo (36 * Cin * Cout floats after the
 *   first 86 * Cin * Cout: 122 floats per pair in all). */
/* The same fused halo + 3x3 conv with DILATION 2 (padding = dilation = 2, stride 1): the dilated last stage of a detector backbone
 * (Pedestron/mmdet/models/backbones/resnet.py:155-162; reference path: BlockPadFunction with pad 2 + F.conv2d(dilation=2),
 * core/tensorwrapper.py:529-575).  Everything as bc_conv3x3_ring_nhwc except: taps 2 pixels apart, halo / zero border 2 pixels
 * wide, `ring` = (N*GH*GW, 8*bs, Cin) records of the pad-2 layout of bc_pad_ring_nhwc (bit-identical refresh), bs a multiple of 8,
 * Cout a multiple of 32.  dilation = 1 forwards to bc_conv3x3_ring_nhwc.  bc_conv3x3_dil_candidates lists the decompositions
 * (indices into the direct form's table; fp32: also `index | 0x2000`, the same decompositions and the 2x2 wave tiles 8 / 9 in the split
 * form on the 16-bit matrix pipe, selected through bc_tune_set("conv2_cfg", code) like every other form) that cover a layer. */
int bc_conv3x3_dil_ring_nhwc(void *out, const void *features, void *ring, const void *weights_packed,
                             const int32_t *grid_idx, const int32_t *mapping_exec, int n_exec, int N, int Cin, int Cout,
                             int GH, int GW, int bs, int dilation, int dtype, const float *in_scale, const float *in_shift, int in_relu,
                             const float *out_scale, const float *out_shift, const void *out_add, int out_relu, void *stream);
int bc_conv3x3_dil_candidates(int dtype, int dilation, int n_exec, int Cin, int Cout, int bs, int *out, int max_out);
int bc_conv3x3_candidates(int dtype, int stride, int n_exec, int Cin, int Cout, int bs_in, int *out, int max_out);

/* halo gather with a residual-add prologue and a by-product: v = relu?(features*scale[c] + shift[c] + add) is computed while
 * gathering; `out` receives the padded batch of v and `act_out` (n_exec, bs, bs, C) the plain v of every executed tile
 * (what the next block's shortcut reads), so the end of a residual block costs one launch instead of bc_affine_act_nhwc
 * + bc_pad_ring_nhwc.  `add` has the layout of `features`.  The ring cache of this op holds ACTIVATED values (exactly
 * what bc_pad_ring_nhwc of the materialised v would store), so both forms may serve the same layer on different frames.
 * C*elem_size a multiple of 16 bytes. */
int bc_pad_ring_add_nhwc(void *out, void *act_out, const void *features, const void *add, void *ring,
                         const int32_t *grid_idx, const int32_t *mapping_exec, int n_exec, int N, int C, int GH, int GW,
                         int bs, int pad, int dtype, const float *scale, const float *shift, int relu, void *stream);

/* fused halo gather + max_pool2d(kernel 3, stride 2, padding 1) of a packed channels-last batch (the ResNet stem pool, the
 * one padded op of the path that is not a conv; reference: BlockPad.apply + F.max_pool2d(padding=0),
 * core/tensorwrapper.py:478-527).  features (n_exec, bs, bs, C) -> out (n_exec, bs/2, bs/2, C); ring and prologue exactly
 * as for bc_pad_ring_nhwc with pad = 1 (zeros beyond the image border take part in the max, the ring keeps the activated values);
 * bit-identical to bc_pad_ring_nhwc followed by a pad-0 pool.  bs even, C*elem_size a multiple of 16 bytes. */
int bc_maxpool3x3s2_ring_nhwc(void *out, const void *features, void *ring, const int32_t *grid_idx,
                              const int32_t *mapping_exec, int n_exec, int N, int C, int GH, int GW, int bs, int dtype,
                              const float *scale, const float *shift, int relu, void *stream);

/* the detector neck's L2Norm AND channel concatenation in one pass (Pedestron/mmdet/models/necks/csp_neck.py:83-85: x.pow(2).sum(1).sqrt()
 * + eps, x / norm, * weight, then torch.cat of the three levels): for the n_pix pixels of a channels-last tensor x (n_pix, C),
 *     out[p][c_off + c] = weight[c] * (x[p][c] / (sqrt(sum_c x[p][c]^2) + eps))
 * written into channels [c_off, c_off + C) of the channels-last tensor out (n_pix, C_total).  fp32 arithmetic (sum of squares in a fixed
 * order), one rounding at the store.  C, c_off, C_total multiples of 16 / elem_size, C <= 256 * 16 / elem_size. */
int bc_l2norm_cat_nhwc(void *out, const void *x, const float *weight, long long n_pix, int C, int C_total, int c_off, float eps, int dtype,
                       void *stream);
/* The detector neck's transposed convs without a transposed-conv launch (Pedestron/mmdet/models/necks/csp_neck.py:37-39, 68-83): t = the 16-tap
 * patches of every input pixel, t (n_img, h, w, 16 C) with channel (4 ky + kx) C + co = sum_ci x[ci] W[ci][co][ky][kx] -- ONE pointwise conv to
 * 16 C channels (bc_conv1x1_nhwc: exactly the transposed conv's multiplications on the matrix cores) -- and this pass gathers the taps of every
 * output pixel (stride 4, pad 0: depth-to-space; stride 2, pad 1: up to 2 x 2 overlapping taps, nothing from outside the image = the packed tile),
 * adds the bias, rounds to the tensor type, L2-normalises over the channels and writes channels [c_off, c_off + C) of out (n_img, stride h, stride w,
 * C_total): conv_transpose2d + L2Norm + cat of the reference in one read of t.  C <= 256 (fp32) / 512 (16-bit). */
int bc_l2norm_cat_deconv_nhwc(void *out, const void *t, const float *bias, const float *weight, int n_img, int h, int w, int C, int C_total,
                              int c_off, int stride, float eps, int dtype, void *stream);

/* detector post-processing (config C5).  replaces nms_kernel + the host sweep of
 * Pedestron/mmdet/ops/nms/src/nms_kernel.cu:23-130: boxes (n,5) float32 [x1,y1,x2,y2,score] ALREADY sorted by score
 * descending, n <= 4096; IoU with the +1 pixel convention, suppression when IoU > iou_thr.  mask_ws: device scratch of
 * n*ceil(n/64) + n 64-bit words (the upper-triangular suppression words and, per box, its suppressors inside its own block of 64).
 * Writes keep[0..count) = kept positions in the sorted order (ascending) and *count, all on the device and in ONE launch: the
 * workgroup that finishes its tile of the pair matrix last resolves the greedy rule (the reference copies the mask to the host and
 * sweeps there).  Launches on different streams may overlap; at most 64 distinct streams per process. */
int bc_nms_sorted(const float *boxes, int n, float iou_thr, unsigned long long *mask_ws, int32_t *keep,
                  int32_t *count, void *stream);
/* the same with the box count known only on the DEVICE: the launch and mask_ws are sized for n_max rows, the kernel works on the first
 * min(*n_dev, n_max) of them (*n_dev = 0: *count = 0).  With bc_csp_decode below a detector's decode needs no host round trip
 * between the head's top-k and the kept boxes (the reference syncs for the score mask, csp_head.py:268-284, and again inside nms). */
int bc_nms_sorted_dev(const float *boxes, int n_max, const int32_t *n_dev, float iou_thr, unsigned long long *mask_ws, int32_t *keep,
                      int32_t *count, void *stream);
/* centre / scale / offset head decode between top-k and NMS (Pedestron/mmdet/models/anchor_heads/csp_head.py:229-284 get_bboxes_single:
 * csp_height2bbox + clamp + score mask): candidate k = flat map position top[k] (row-major, map_w columns) with score scores[k] (sorted
 * descending: the output of top-k), heights[k] = exp(scale prediction), off_y / off_x[k] its offsets.  Writes dets[k] = [x1, y1, x2, y2,
 * score]: centre (col * stride + stride / 2 + off_x * stride, row ... off_y ...), height heights * stride, width wh_ratio * height, clamped
 * to [0, img_w - 1] x [0, img_h - 1]; and *n_sel = the number of scores > score_thr (= the rows the reference keeps: a prefix).  Every
 * operation is the single fp32 operation of the reference's tensor expression, in its order: identical boxes.  One launch for ~25. */
int bc_csp_decode(const float *scores, const long long *top, const float *heights, const float *off_y, const float *off_x, int k,
                  int map_w, int stride, float wh_ratio, int img_h, int img_w, float score_thr, float *dets, int32_t *n_sel, void *stream);

/* the same WITH the top-k in the launch (csp_head.py:262-267 cls.sigmoid().topk(nms_pre) + the gathers of scale and offset at the kept positions):
 * cls = the centre logits of the map (n = map_h * map_w values, row-major; BC_F32 / BC_F16 / BC_BF16, converted to float as `.float()`), reg = the
 * scale predictions (float, n), off = the offset map: value (channel c, position i) at off[c * off_channel_stride + i * off_pixel_stride] (c = 0:
 * y, 1: x; any of NCHW / channels-last).  Selects the k (<= 4096, <= n) largest sigmoid(cls) and writes dets / *n_sel as bc_csp_decode does for
 * that selection (scores = 1 / (1 + exp(-x)) in fp32, heights = exp(reg)), plus the positions (top_out, k x int32, may be NULL).  The selection
 * runs on the LOGITS: the fp32 score expression is monotone non-decreasing in the logit on this device (bc_csp_score_monotone below checks
 * every float), so the k largest logits are k largest scores and rows ordered by logit are ordered by score.  Order among EQUAL scores, which
 * torch.topk leaves unspecified: the larger logit first; equal logits: lowest position first.  One workgroup, one launch: replaces sigmoid +
 * top-k (~10 library launches) + 3 gathers + exp + decode. */
int bc_csp_topk_decode(const void *cls, int cls_dtype, const float *reg, const float *off, long long off_channel_stride,
                       long long off_pixel_stride, int n, int k, int map_w, int stride, float wh_ratio, int img_h, int img_w, float score_thr,
                       float *dets, int32_t *n_sel, int32_t *top_out, void *stream);
/* self-test of the property above: *violations (device) = the number of neighbouring float pairs x < x' in [-inf, +inf] whose fp32 scores
 * 1 / (1 + exp(-x)) DEcrease (4.3e9 evaluations, a few ms).  0 on gfx950 with this build (tests/test_gpu_ops.py asserts it). */
int bc_csp_score_monotone(unsigned long long *violations, void *stream);

/* device policy step: the per-frame decision of the online-RL policies without leaving the GPU.
 * Replaces, in one launch: Bernoulli(logits).sample() + `.cpu()` (policy/policy.py:283-288), quantize_number_exec_grid
 * (:124-144: Python random.sample on the host) and get_grid_mappings (core/tensorwrapper.py:108-128, CPU TorchScript).
 *   logits float32[n_total] (raster order of the (N,1,GH,GW) grid); seed/counter: counter-based RNG (same arguments -> same
 *   decision; the engine passes its frame counter); multiple = max(1, int(n_total * quantize_number_exec));
 *   outputs: grid uint8[n_total] (1 = execute), grid_idx / mapping_exec as bc_grid_tables, counts int32[4] =
 *   {n_exec, n_sampled (before rounding up), any-NaN-logit flag, low 32 bits of counter}; host_mailbox (optional): the same four
 *   ints stored to device-visible pinned HOST memory so that the engine learns n_exec (it selects the captured graph)
 *   without a device->host copy.  n_total <= 8192.  Exact arithmetic definition: csrc/blockcopy_hip.hip k_policy_step,
 *   restated on the CPU by oracle/bc_oracle.c bc_oracle_policy_step (bit-identical decisions). */
int bc_policy_step(const float *logits, int n_total, unsigned long long seed, unsigned long long counter, int multiple,
                   int at_least_one, uint8_t *grid, int32_t *grid_idx, int32_t *mapping_exec, int32_t *counts,
                   int32_t *host_mailbox, void *stream);

/* input features of the policy net in one gather (replaces the four nearest-neighbour F.interpolate calls, casts, -0.5
 * centring and concat of policy/net.py:82-113): out float32 (N, sum C_k, h, w) contiguous = concat over k = 0..3 of
 * nearest(src_k) + offset_k, src index = min((int)floorf(dst * scale), in - 1) per axis (ATen's legacy 'nearest').
 * Per source k: ptrs[k]; strides[4k..] = element strides (n, c, h, w); dims[4k..] = {C, H, W, dtype} with dtype BC_F32 /
 * BC_F16 / BC_BF16 / 3 = uint8 (bool); scales[3k..] = {scale_h, scale_w, offset}. */
int bc_policy_features(float *out, int N, int h, int w, const void *const *ptrs, const long long *strides, const int *dims,
                       const float *scales, void *stream);

/* ---------------------------------------------------------------------------------------------
 * B2. The online-RL policy's CNN on own kernels (csrc/policy_net.hip): forward, backward and optimizer step of the reference's
 * PolicyNet (policy/net.py:17-125: resnet8 trunk policy/resnet.py:60-115 + three stride-2 head stages) and of PolicyTrainRL.optim
 * (policy/policy.py:319-370: information gain policy/information_gain.py:22-41, REINFORCE loss, backward, RMSprop), which the
 * reference runs as ~45 + ~120 PyTorch/cuDNN launches per frame / training step.  All maps are DENSE channels-last fp32
 * [N][H][W][C], C a multiple of 32; weights W[tap = 3 ky + kx][Cin][Cout], transposed copy WT[tap][Cout][Cin] for the data gradient.
 * Stateless entry points (the host sequences them and captures the sequence in a hipGraph: blockcopy/policy/native.py).
 * ------------------------------------------------------------------------------------------- */

/* implicit-GEMM conv on the fp32 matrix cores (exact fp32: a k-ordered fmaf chain).  ks 3 (pad 1) or 1 (pad 0, stride 2), stride 1 / 2.
 *   direction 0 (forward):        out (N,Hy,Wy,Cy) = conv(prologue(x (N,Hx,Wx,Cx)), w = W[tap][Cx][Cy])
 *   direction 1 (data gradient):  out (N,Hx,Wx,Cx) = conv_transpose(x = gz (N,Hy,Wy,Cy), w = WT[tap][Cy][Cx]); stride 2 runs as the four
 *                                 output-parity classes (1 / 2 / 2 / 4 taps, no multiplications by inserted zeros)
 *   prologue: relu?(x * in_scale[c] + in_shift[c]) applied while the patch is staged (the producer's BatchNorm + ReLU), zero padding after it;
 *   epilogue: out = acc (+ add, or add where add_mask > 0: the residual branch's gradient behind a ReLU) (+ out if accumulate);
 *   precision 0: v_mfma_f32_32x32x2_f32 (exact fp32 products); 1: every operand split hi + lo into two fp16 numbers (both scaled by 16:
 *   |values| < 4094, 22 bits of mantissa) and each product taken as hi*hi + hi*lo + lo*hi on v_mfma_f32_32x32x16_f16 with fp32 accumulation --
 *   fp32-level accuracy (measured <= 3e-6 of the output's largest element) at 5.3 x the matrix rate; meant for forward convs behind BatchNorm;
 *   2: the same with bf16 halves (fp32's exponent range: no scale, no under- / overflow; 16 bits of mantissa, ~2e-5 relative) -- meant for the
 *   data gradient, whose operand spans 1e-7 .. 1e-3;
 *   stats (forward only): per-workgroup partial sums [bc_pn_conv_partials(N,Hy,Wy,Cy)][2][Cy] of out and out^2 (training-mode BatchNorm
 *   statistics of the OUTPUT, finished by bc_pn_bn_finalize). */
int bc_pn_conv_nhwc(float *out, const float *x, const float *w, int N, int Hx, int Wx, int Cx, int Hy, int Wy, int Cy, int ks, int stride,
                    int direction, const float *in_scale, const float *in_shift, int in_relu, const float *add, const float *add_mask,
                    int accumulate, float *stats, long long stats_capacity, int precision, void *stream);
long long bc_pn_conv_partials(int N, int Hy, int Wy, int Cy);
/* weight gradient dw[tap][Cx][Cy] = sum over output pixels of prologue(x)[pixel * stride + tap - pad][ci] * gz[pixel][co] as a GEMM over
 * pixels, split over pixel-tile groups and summed in a FIXED order (two launches, no atomics): part = workspace of
 * bc_pn_wgrad_workspace(...) floats. */
int bc_pn_wgrad_nhwc(float *dw, float *part, long long part_capacity, const float *x, const float *gz, int N, int Hx, int Wx, int Cx, int Hy, int Wy,
                     int Cy, int ks, int stride, const float *in_scale, const float *in_shift, int in_relu, void *stream);
long long bc_pn_wgrad_workspace(int N, int Hy, int Wy, int Cx, int Cy, int ks);
int bc_pn_wgrad_groups(int N, int Hy, int Wy, int Cx, int Cy);      /* partial copies a launch leaves in `part` (dw = NULL: not summed, see bc_pn_update) */
/* training-mode BatchNorm from the conv's partial sums: scale / shift (the consumers' prologue), save_mean / save_invstd (backward), running
 * statistics (momentum, unbiased variance) and the batch counter updated in place (F.batch_norm(training=True) semantics). */
int bc_pn_bn_finalize(const float *part, long long n_part, int C, double count, const float *gamma, const float *beta, float eps, float momentum,
                      float *running_mean, float *running_var, long long *batches, float *scale, float *shift, float *save_mean, float *save_invstd,
                      void *stream);
/* The same BatchNorm WITHOUT a finalize launch between a conv and its consumers (a forward pass of the policy net: 11 launches less).  The producer adds
 * its per-workgroup sums (sum x, sum x^2 of its output, per channel) as 64-bit FIXED-POINT numbers in units of 2^-24 to accumulators acc[16][2][C]
 * (16 replicas, workgroup w adds to replica w mod 16: a sixteenth of the same-address atomics; readers add the replicas up) -- integer
 * additions: the result is independent of their order, i.e. run-to-run identical without a fixed reduction order -- and every consumer derives
 * scale / shift of its input channels from the accumulators at kernel start (bc_pn_bn_finalize's arithmetic in double precision: all consumers and
 * bc_pn_bn_finalize_acc get bit-identical coefficients).  C <= 128.
 *   bc_pn_arm_bn: one shot, per host thread, for the NEXT bc_pn_conv_nhwc (direction 0): in_acc (NULL: none; in_scale / in_shift must then be NULL in that
 *     call) with gamma / beta / count (= N * H * W of the producer) / eps / C of the input's BatchNorm, out_acc (NULL: none) for the conv's own output.
 *   bc_pn_join_acc / bc_pn_head_fwd_acc: bc_pn_join / bc_pn_head_fwd with (acc, gamma, beta) in place of (scale, shift); mode 0: acc_b unused.
 *   bc_pn_bn_finalize_acc: ONE launch at the end of the pass for all layers -- scale / shift / mean / invstd arrays (what the backward pass and the
 *     weight gradient's prologue read), running statistics and batch counters as bc_pn_bn_finalize, accumulators ZEROED for the next pass.  `layers`:
 *     device array of bc_pn_bn_layer_bytes()-byte records {acc, gamma, beta, running_mean, running_var, batches, scale, shift, mean, invstd (pointers);
 *     count (double); eps, momentum (float); C, 0 (int)}. */
int bc_pn_arm_bn(const void *in_acc, const float *gamma, const float *beta, double count, float eps, int C, void *out_acc);
int bc_pn_join_acc(float *out, const float *za, const void *acc_a, const float *gamma_a, const float *beta_a, const float *zb, const void *acc_b,
                   const float *gamma_b, const float *beta_b, double count, float eps, int mode, int C, long long pixels, void *stream);
int bc_pn_head_fwd_acc(float *logits, const float *z, const void *acc, const float *gamma, const float *beta, double count, float eps, const float *w,
                       const float *bias, int N, int Hi, int Wi, int C, void *stream);
int bc_pn_bn_layer_bytes(void);
int bc_pn_bn_finalize_acc(const void *layers, int n_layers, void *stream);
/* residual join: out = relu(za * sa + ta + B), B = zb (mode 0) | zb * sb + tb (mode 1: projection shortcut) | relu(zb * sb + tb) (mode 2) */
int bc_pn_join(float *out, const float *za, const float *sa, const float *ta, const float *zb, const float *sb, const float *tb, int mode, int C,
               long long pixels, void *stream);
/* backward of BatchNorm (+ the ReLU behind it) in three launches: gz = d loss / d conv output, dgamma, dbeta.  g = gradient w.r.t. the activation;
 * mask_mode 0 none | 1 own output (z * scale + shift > 0) | 2 external map (mask > 0: the block output behind the residual join);
 * part = bc_pn_bn_bwd_partials(pixels) * 2 * C floats, coef = 3 * C floats (workspaces). */
long long bc_pn_bn_bwd_partials(long long pixels);
int bc_pn_bn_bwd(float *gz, float *dgamma, float *dbeta, float *part, float *coef, const float *g, const float *z, const float *mask, int mask_mode,
                 const float *scale, const float *shift, const float *mean, const float *invstd, const float *gamma, int C, long long pixels,
                 void *stream);
/* the last head stage (3x3 / stride 2 / pad 1 to ONE channel + bias = the tile logits) on relu(z * scale + shift), and its backward */
int bc_pn_head_fwd(float *logits, const float *z, const float *scale, const float *shift, const float *w, const float *bias, int N, int Hi, int Wi,
                   int C, void *stream);
int bc_pn_head_bwd(float *ga, float *dw, float *db, const float *gl, const float *z, const float *scale, const float *shift, const float *w, int N,
                   int Hi, int Wi, int C, void *stream);
/* information gain of a segmentation output (information_gain.py:22-41): KL(prev || cur) of the bilinearly resampled (ATen upsample_bilinear2d,
 * align_corners = False, rh / rw = 1 / scale_factor), log-softmaxed logit maps, mean over the classes; cur / prev share the element strides
 * and the element type `dtype` (BC_F32 / BC_F16 / BC_BF16; arithmetic in fp32). */
int bc_pn_infogain(float *ig, const void *cur, const void *prev, int dtype, int N, int C, int H, int W, long long sn, long long sc, long long sh,
                   long long sw, int h, int w, float rh, float rw, void *stream);
/* REINFORCE seed (policy.py:334-349): reward = adaptive_max_pool2d(ig + rc) per tile, rc = -(cost - target) |cost - target| gamma, negated on
 * skipped tiles; gl = d mean(-log_prob * reward) / d logits = (sigmoid(l) - grid) * reward / n; loss and reward are optional outputs;
 * cost: *cost_dev (device float64) when given, else cost_host. */
int bc_pn_reward_seed(float *gl, float *loss, float *reward, const float *logits, const uint8_t *grid, const float *ig, const double *cost_dev,
                      double cost_host, double target, double gamma, int N, int h, int w, int GH, int GW, void *stream);
/* torch.optim.RMSprop (centered = False) over a flat buffer, torch's operation order */
int bc_pn_rmsprop(float *p, const float *g, float *sq, float *mom, long long n, float lr, float alpha, float eps, float wd, float momentum,
                  void *stream);
/* flat buffer <-> torch parameters in one launch (dir 0: flat -> parameters + transposed copies, 1: parameters -> flat + transposed copies);
 * segs = DEVICE array of bc_pn_seg_bytes()-byte records: int64 off, off_t, param address, s_co, s_ci, s_ky, s_kx; int32 taps (0 = plain
 * vector), kw, cin, cin_pad, cout, numel; int64 ws_off; int32 groups, 0 (the last three: bc_pn_update). */
int bc_pn_sync_params(float *flat, float *flat_t, const void *segs, int n_segs, int dir, void *stream);
int bc_pn_seg_bytes(void);
/* the tail of a training step in ONE launch: for every segment, gradient = the fixed-order sum of its `groups` partial copies at ws + ws_off
 * (weight gradients left by bc_pn_wgrad_nhwc with dw = NULL; groups = 0: the gradient is already in g), torch.optim.RMSprop on p / sq / mom, and the
 * export of the new values into the module's parameter tensors and the transposed copies (bc_pn_rmsprop + bc_pn_sync_params dir 0). */
int bc_pn_update(float *p, float *g, float *sq, float *mom, float *flat_t, const float *ws, const void *segs, int n_segs, float lr, float alpha,
                 float eps, float wd, float momentum, void *stream);
/* decision bookkeeping of a frame: probs = sigmoid(logits), log_probs = log-probability of the decided grid under Bernoulli(logits)
 * (= -binary_cross_entropy_with_logits; torch.distributions.Bernoulli's own formulas, policy/policy.py:283-288) in one launch */
int bc_pn_probs(float *probs, float *log_probs, const float *logits, const uint8_t *grid, int n, void *stream);
/* measurement only: device buffer receiving 8 x uint64 stamps (100 MHz) per workgroup of the following bc_pn_conv_nhwc launches; NULL = off */
int bc_pn_set_stamps(void *buf);
/* the policy input (bc_policy_features) straight into the channels-last layout, channels sum C_k .. Cpad - 1 zero */
int bc_pn_features_nhwc(float *out, int N, int h, int w, int Cpad, const void *const *ptrs, const long long *strides, const int *dims,
                        const float *scales, void *stream);

/* tuning / A-B knob (measurement infrastructure; defaults are the shipped behaviour): key in
 *   "conv_impl"      1 = first-generation fused conv kernel, 2 = CU-balanced kernel (default)
 *   "conv2_cfg"      -1 = choose the decomposition per launch (default), 0..15 = force one (BC_ERR_SHAPE at launch if it does not fit)
 *   "stem_split"     1 = bc_stem7x7s2_nhwc runs fp32 frames on the 16-bit matrix pipe (see there; default 0: the caller must have packed the streams)
 *   "xcd_remap"      1 = XCD-aware workgroup order in the fused conv kernels (workgroups that share an XCD take a contiguous,
 *                    output-channel-group-major run of logical ids); 0 (default) = launch order.  Speed only, results identical
 *   "conv2_min_lds"  dynamic LDS floor in bytes (default 84 KiB: one 8-wave workgroup per CU)
 * Not part of the reference's boundary. */
int bc_tune_set(const char *key, int value);
/* "conv_stamps": device buffer (8 x uint64 per workgroup) that receives in-kernel s_memtime stamps of the balanced conv kernel
 * (prologue / first stage / main loop / reduction / store); NULL (default) disables them.  Measurement only. */
int bc_tune_set_ptr(const char *key, void *ptr);
/* read a knob back; "conv_last_cfg" = decomposition index the most recent bc_conv3x3_ring_nhwc launch used (-1 = first-generation kernel) */
int bc_tune_get(const char *key, int *value);

/* ---------------------------------------------------------------------------------------------
 * C. Introspection / measurement
 * ------------------------------------------------------------------------------------------- */

enum { BC_OP_SPLIT = 0, BC_OP_COMBINE = 1, BC_OP_TRANSFER = 2, BC_OP_PAD = 3,
       BC_OP_COMBINE_COPY = 4, BC_OP_PAD_RING = 5, BC_OP_GRID_TABLES = 6, BC_OP_INTERP = 7, BC_OP_AFFINE = 8, BC_OP_NMS = 9, BC_OP_CONV3X3 = 10, BC_OP_HEAD = 11, BC_OP_PRED = 12, BC_OP_COUNT = 13 };

int bc_abi_version(void);
const char *bc_error_string(int code);
const char *bc_op_name(int op);

/* Per-op device timing with hipEvent pairs recorded on the launch stream around each kernel of the
 * selected ops (bit i of op_mask = op i).  bc_prof_read synchronises the recorded events and returns the
 * number of launches, their summed device time (ms) and summed algorithmic bytes since the last reset. */
int bc_prof_enable(unsigned op_mask);
int bc_prof_reset(void);
int bc_prof_read(int op, long long *launches, double *total_ms, double *total_bytes);
/* second total of the same launches; BC_OP_CONV3X3: matrix FLOPs actually ISSUED (total_bytes holds the FLOPs of the direct
 * definition 2*px*k*k*Cin*Cout; the Winograd F(2x2) forms issue 16/36 of them, the F(4x4) form 36/144, the stem kernel its zero-padded K segments) */
int bc_prof_read_aux(int op, double *total_aux);

#ifdef __cplusplus
}
#endif
#endif /* BLOCKCOPY_HIP_H */
