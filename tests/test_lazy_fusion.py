"""Lazy fusion (core/fusion.py) must be invisible: every recorded form gives the value the unfused op sequence gives.
CPU tier: the block ops are served by the checker backend; fusion on vs off is compared on the same packed tensors."""
import pytest
import torch
import torch.nn.functional as F


def _packed(C=8, n_frames=1, nhwc=True, seed=0):
    """A packed all-active TensorWrapper (2x3 grid of 4x4 tiles) on the checker backend."""
    import blockcopy

    g = torch.Generator().manual_seed(seed)
    x = torch.randn(1, C, 8, 12, generator=g)
    if nhwc:
        x = x.contiguous(memory_format=torch.channels_last)
    w = blockcopy.to_tensorwrapper(x)
    w.process_temporal_features(None)
    grid = torch.ones(1, 1, 2, 3, dtype=torch.bool)
    return w.to_blocks(grid, grid)


def _bn_params(C, seed=1):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(C, generator=g) * 0.3, torch.rand(C, generator=g) + 0.5, torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g) * 0.2)


def _both(fn):
    """fn(blocks) -> dense tensor, evaluated with lazy fusion on and off."""
    from blockcopy.core import fusion

    outs = []
    for flag in (True, False):
        prev = fusion.set_enabled(flag)
        try:
            outs.append(fn(_packed()).clone())
        finally:
            fusion.set_enabled(prev)
    return outs


@pytest.mark.parametrize("tail", ["bn", "bn_relu", "bn_relu_conv", "relu_bn", "bn_bn", "add_bn"])
def test_interpolate_then_batchnorm(oracle_backend, tail):
    """F.interpolate -> eval batch_norm on a packed channels-last tensor lost the deferred resampling
    (the BN record was built on the unwritten placeholder).  SwiftNet's _Upsample with use_skip=False is this sequence."""
    rm, rv, w, b = _bn_params(8)
    conv_w = torch.randn(8, 8, 3, 3, generator=torch.Generator().manual_seed(5)) * 0.1

    def fn(blk):
        y = F.interpolate(blk, scale_factor=2, mode="bilinear", align_corners=False)
        if tail == "relu_bn":
            y = F.relu(y)
        if tail == "add_bn":
            y = y + F.interpolate(blk * 0.5, scale_factor=2, mode="bilinear", align_corners=False)
        y = F.batch_norm(y, rm, rv, w, b, False, 0.1, 1e-5)
        if tail == "bn_bn":
            y = F.batch_norm(y, rm * 0.5, rv, None, None, False, 0.1, 1e-5)
        if tail in ("bn_relu", "bn_relu_conv"):
            y = F.relu(y)
        if tail == "bn_relu_conv":
            y = F.conv2d(y, conv_w, None, 1, 1)
        return y.combine().to_tensor()

    fused, plain = _both(fn)
    assert torch.isfinite(fused).all()
    assert (fused - plain).abs().max().item() <= 1e-5 * max(1.0, plain.abs().max().item())


def test_tensor_valued_attributes_show_the_value(oracle_backend):
    """.data / .T / .mT of a lazily fused tensor must not expose the pre-activation storage."""
    from blockcopy.core import fusion

    assert fusion.ENABLED
    blk = _packed()
    y = F.relu(blk)
    assert y._pending is not None                      # recorded, not launched
    assert float(y.data.min()) == 0.0                  # was the raw minimum (negative) before the fix
    y2 = F.relu(_packed())
    assert float(y2.mT.min()) == 0.0
    y3 = F.interpolate(_packed(), scale_factor=2, mode="bilinear", align_corners=False)
    assert y3._pending is not None and y3._pending.interp is not None
    ref = F.interpolate(_packed()._plain(), scale_factor=2, mode="bilinear", align_corners=False)
    assert torch.allclose(y3.data._plain(), ref, atol=1e-6)
    # non-tensor attributes stay free of side effects
    y4 = F.relu(_packed())
    assert tuple(y4.shape) == (6, 8, 4, 4) and y4.dtype == torch.float32 and y4._pending is not None


def test_inplace_write_to_recorded_residual_is_loud(oracle_backend):
    """s = bn(x) + y records y by alias; a real in-place write to y before s is consumed must not
    silently change s."""
    rm, rv, w, b = _bn_params(8)
    x, y = _packed(seed=0), _packed(seed=3)
    s = F.batch_norm(x, rm, rv, w, b, False, 0.1, 1e-5) + y
    assert s._pending is not None and s._pending.add is not None
    y.mul_(100.0)
    with pytest.raises(RuntimeError, match="modified in place"):
        s.combine()
    # untouched operand: fine, and equal to the eager sum
    x, y = _packed(seed=0), _packed(seed=3)
    s = F.batch_norm(x, rm, rv, w, b, False, 0.1, 1e-5) + y
    want = F.batch_norm(x._plain(), rm, rv, w, b, False, 0.1, 1e-5) + y._plain()
    assert torch.allclose(s._plain(), want, atol=1e-6)


def test_pad_memo_distinguishes_prologues(oracle_backend):
    """two padded ops on the same raw tensor with DIFFERENT pending affines must not share a gather,
    also when the coefficient tensors are short-lived (ids recycled)."""
    from blockcopy.core import fusion

    conv_w = torch.randn(4, 8, 3, 3, generator=torch.Generator().manual_seed(7)) * 0.1
    blk = _packed()
    outs = []
    for k in range(4):
        rm, rv, w, b = _bn_params(8, seed=10 + k)
        fusion.clear_cache()                       # coefficient vectors of the previous round die -> ids may be reused
        y = F.batch_norm(blk, rm, rv, w, b, False, 0.1, 1e-5)
        got = F.conv2d(y, conv_w, None, 1, 1).combine().to_tensor()
        prev = fusion.set_enabled(False)
        try:
            blk2 = _packed()
            want = F.conv2d(F.batch_norm(blk2, rm, rv, w, b, False, 0.1, 1e-5), conv_w, None, 1, 1).combine().to_tensor()
        finally:
            fusion.set_enabled(prev)
        assert (got - want).abs().max().item() <= 1e-5, k
        del rm, rv, w, b, y
    # identical prologue on the same tensor DOES share one gather (the CSP head's three branches)
    rm, rv, w, b = _bn_params(8, seed=99)
    blk = _packed()
    y = F.batch_norm(blk, rm, rv, w, b, False, 0.1, 1e-5)
    F.conv2d(y, conv_w, None, 1, 1)
    memo = blk.get_features()._pad_memo
    F.conv2d(y, conv_w * 2, None, 1, 1)
    assert blk.get_features()._pad_memo is memo


@pytest.mark.parametrize("tail", ["plain", "bias_relu", "residual_relu", "bn_residual_relu_conv", "two_consumers", "stride2_residual"])
def test_deferred_conv_takes_the_recorded_work_as_its_epilogue(oracle_backend, tail):
    """A fused halo+conv launch is deferred (fusion.Pending.conv) and carries bias / BN / residual add / ReLU recorded after it
    as its epilogue -- the end of a residual block is one launch.  Same values as the unfused op sequence, incl. a tensor
    consumed twice (next conv + next shortcut) and a stride-2 conv."""
    import blockcopy.backend as bk
    from blockcopy.core import fusion

    g = torch.Generator().manual_seed(11)
    w1, b1 = torch.randn(8, 8, 3, 3, generator=g) * 0.1, torch.randn(8, generator=g) * 0.1
    w2 = torch.randn(8, 8, 3, 3, generator=g) * 0.1
    rm, rv, bw, bb = _bn_params(8, seed=4)
    launches = []
    chk = bk.get_backend()
    orig = chk.conv3x3_ring
    chk.conv3x3_ring = lambda *a, **k: (launches.append(k.get("epilogue")), orig(*a, **k))[1]

    def fn(blk):
        st = 2 if tail == "stride2_residual" else 1
        y = F.conv2d(blk, w1, None if tail == "plain" else b1, st, 1)
        if tail == "bn_residual_relu_conv":
            y = F.batch_norm(y, rm, rv, bw, bb, False, 0.1, 1e-5)
        if "residual" in tail or tail == "two_consumers":
            y += (blk if st == 1 else F.avg_pool2d(blk._plain(), 2).contiguous(memory_format=torch.channels_last))
        if tail != "plain":
            y = F.relu(y, inplace=True)
        if tail == "bn_residual_relu_conv":
            y = F.conv2d(y, w2, None, 1, 1)
        if tail == "two_consumers":
            y = F.conv2d(y, w2, None, 1, 1) + y
        return y.combine().to_tensor()

    prev_mode, fusion.CONV_MODE = fusion.CONV_MODE, "native"      # (the tiny tiles of this test would go to the library route)
    try:
        fused, plain = _both(fn)
    finally:
        chk.conv3x3_ring = orig
        fusion.CONV_MODE = prev_mode
    assert torch.isfinite(fused).all() and (fused - plain).abs().max().item() <= 1e-5 * max(1.0, plain.abs().max().item())
    if tail in ("residual_relu", "stride2_residual"):
        epi = launches[0]           # the fused pass: ONE conv launch whose epilogue holds bias + identity + ReLU
        assert epi is not None and epi[1] is not None and epi[2] is not None and epi[3] is True


def test_deferred_conv_input_modified_in_place_is_a_loud_error(oracle_backend):
    """A deferred conv / resampling reads its input when the value is needed, not where the op was called: a real in-place
    write to that input in between must not silently change the result (same rule as the recorded residual operand)."""
    from blockcopy.core import fusion

    w1 = torch.randn(8, 8, 3, 3, generator=torch.Generator().manual_seed(3)) * 0.1
    prev_mode, fusion.CONV_MODE = fusion.CONV_MODE, "native"
    try:
        blk = _packed()
        y = F.conv2d(blk, w1, None, 1, 1)
        assert y._pending is not None and y._pending.conv is not None     # recorded, not launched
        blk.as_subclass(torch.Tensor).mul_(2.0)                            # a real in-place write to the conv's input
        with pytest.raises(RuntimeError, match="modified in place"):
            y.combine()
        blk2 = _packed()
        z = F.interpolate(blk2, scale_factor=2, mode="bilinear", align_corners=False)
        assert z._pending is not None and z._pending.interp is not None
        blk2.as_subclass(torch.Tensor).add_(1.0)
        with pytest.raises(RuntimeError, match="modified in place"):
            z.combine()
    finally:
        fusion.CONV_MODE = prev_mode


def test_unconsumed_deferred_conv_still_refreshes_its_ring(oracle_backend):
    """a deferred fused conv takes its layer's ring cache when it is RECORDED but refreshes it when it is
    LAUNCHED; a result nobody consumes must still be launched by the end of the frame body (BlockFeatures.flush_deferred),
    or later frames that skip those tiles would gather stale halo records for the layer."""
    from blockcopy.core import fusion

    w1 = torch.randn(8, 8, 3, 3, generator=torch.Generator().manual_seed(3)) * 0.1
    prev_mode, fusion.CONV_MODE = fusion.CONV_MODE, "native"
    try:
        blk = _packed()
        feats = blk.get_features()
        y = F.conv2d(blk, w1, None, 1, 1)
        assert y._pending is not None and y._pending.conv is not None and len(feats._deferred) == 1
        ring = feats.rings[0]
        ring.fill_(float("nan"))
        del y                                            # the value is never asked for
        assert feats.flush_deferred() == 1 and not feats._deferred
        assert torch.isfinite(ring).all()                # every tile was executed: every record refreshed
        # consumed results are not launched twice
        blk2 = _packed()
        z = F.conv2d(blk2, w1, None, 1, 1)
        z.combine()
        assert blk2.get_features().flush_deferred() == 0
        # and the flush honours the in-place guard of the deferred input
        blk3 = _packed()
        u = F.conv2d(blk3, w1, None, 1, 1)
        blk3.as_subclass(torch.Tensor).mul_(2.0)
        with pytest.raises(RuntimeError, match="modified in place"):
            blk3.get_features().flush_deferred()
    finally:
        fusion.CONV_MODE = prev_mode
