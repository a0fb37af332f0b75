"""The RL policy's CNN on own kernels (csrc/policy_net.hip, blockcopy/policy/native.py) against PyTorch: every kernel kind against the
same operation in float64 on the CPU, then the whole forward / REINFORCE step against autograd + torch.optim.RMSprop on the module.

Tolerances (written here, floating point): conv / gradients 2e-5 of the tensor's largest magnitude against a float64 reference (the
kernels are exact-fp32 fmaf chains in a different summation order); BatchNorm statistics 1e-5 relative; end-to-end logits 1e-4,
end-to-end gradients 1e-3 of the tensor's largest magnitude (ten batch-statistics BatchNorms in between)."""
import copy
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _lib():
    import blockcopy.backend as bk

    be = bk.get_backend()
    assert be.name == "hip"
    return be, be.lib


def _st():
    return torch.cuda.current_stream().cuda_stream


def _nhwc(t):      # (N,C,H,W) -> contiguous (N,H,W,C) on the GPU
    return t.permute(0, 2, 3, 1).contiguous().float().cuda()


def _nchw(t):      # GPU (N,H,W,C) -> CPU float64 (N,C,H,W)
    return t.permute(0, 3, 1, 2).double().cpu()


def _wk(w):        # torch (Cout,Cin,kh,kw) -> kernel layout [tap][Cin][Cout]
    co, ci, kh, kw = w.shape
    return w.permute(2, 3, 1, 0).reshape(kh * kw, ci, co).contiguous().float().cuda()


def _wt(w):        # torch (Cout,Cin,kh,kw) -> transposed kernel layout [tap][Cout][Cin]
    co, ci, kh, kw = w.shape
    return w.permute(2, 3, 0, 1).reshape(kh * kw, co, ci).contiguous().float().cuda()


CONV_CASES = [  # N, H, W, Cin, Cout, ks, stride
    (1, 20, 52, 32, 32, 3, 1),
    (2, 8, 40, 32, 64, 3, 2),
    (1, 12, 64, 64, 64, 3, 1),
    (1, 16, 36, 32, 64, 1, 2),
    (1, 10, 34, 64, 128, 3, 2),
    (1, 6, 32, 128, 128, 3, 1),
    (2, 9, 33, 128, 128, 3, 2),
    (1, 7, 31, 64, 32, 3, 1),
    (2, 12, 40, 32, 32, 3, 1),
    (2, 64, 128, 64, 64, 3, 1),
    (3, 5, 70, 64, 64, 3, 1),
]


def _conv_ref(x, w, ks, stride, scale=None, shift=None, relu=False):
    xa = x.double()
    if scale is not None:
        xa = xa * scale.double().view(1, -1, 1, 1) + shift.double().view(1, -1, 1, 1)
    if relu:
        xa = xa.relu()
    return F.conv2d(xa, w.double(), None, stride, 1 if ks == 3 else 0), xa


@pytest.mark.parametrize("case", CONV_CASES)
@pytest.mark.parametrize("prologue", [False, True])
@pytest.mark.parametrize("precision", [0, 1])
def test_pn_conv_forward_and_statistics(case, prologue, precision):
    be, lib = _lib()
    N, H, W, Ci, Co, ks, s = case
    g = torch.Generator().manual_seed(sum(case) + prologue)
    x = torch.randn((N, Ci, H, W), generator=g)
    w = torch.randn((Co, Ci, ks, ks), generator=g) * 0.1
    sc = (torch.rand(Ci, generator=g) + 0.5) if prologue else None
    sh = (torch.randn(Ci, generator=g) * 0.3) if prologue else None
    want, _ = _conv_ref(x, w, ks, s, sc, sh, prologue)
    Hy, Wy = want.shape[2:]
    out = torch.full((N, Hy, Wy, Co), float("nan"), device="cuda")
    n_part = lib.bc_pn_conv_partials(N, Hy, Wy, Co)
    stats = torch.full((n_part * 2 * Co,), float("nan"), device="cuda")
    xg, wg = _nhwc(x), _wk(w)
    scg, shg = (sc.cuda(), sh.cuda()) if prologue else (None, None)
    rc = lib.bc_pn_conv_nhwc(out.data_ptr(), xg.data_ptr(), wg.data_ptr(), N, H, W, Ci, Hy, Wy, Co, ks, s, 0, scg.data_ptr() if prologue else None,
                             shg.data_ptr() if prologue else None, int(prologue), None, None, 0, stats.data_ptr(), stats.numel(), precision, _st())
    assert rc == 0
    got = _nchw(out)
    tol = 2e-5 * float(want.abs().max())
    assert float((got - want).abs().max()) <= tol
    part = stats.view(n_part, 2, Co).double().cpu()
    assert torch.allclose(part[:, 0].sum(0), got.sum((0, 2, 3)), rtol=1e-5, atol=1e-3)
    assert torch.allclose(part[:, 1].sum(0), (got * got).sum((0, 2, 3)), rtol=1e-5, atol=1e-3)


@pytest.mark.parametrize("case", CONV_CASES)
@pytest.mark.parametrize("precision", [0, 2])
def test_pn_conv_data_gradient(case, precision):
    """direction 1 == autograd's gradient of the input (stride 2: four parity classes), with the masked residual term and accumulation."""
    be, lib = _lib()
    N, H, W, Ci, Co, ks, s = case
    g = torch.Generator().manual_seed(7 + sum(case))
    x = torch.randn((N, Ci, H, W), generator=g, dtype=torch.float64, requires_grad=True)
    w = torch.randn((Co, Ci, ks, ks), generator=g, dtype=torch.float64) * 0.1
    y = F.conv2d(x, w, None, s, 1 if ks == 3 else 0)
    gy = torch.randn(y.shape, generator=g, dtype=torch.float64)
    (want,) = torch.autograd.grad(y, x, gy)
    Hy, Wy = y.shape[2:]
    add = torch.randn((N, Ci, H, W), generator=g)
    mask = torch.randn((N, Ci, H, W), generator=g)
    gyg, wtg, addg, maskg = _nhwc(gy), _wt(w), _nhwc(add), _nhwc(mask)
    out = torch.full((N, H, W, Ci), float("nan"), device="cuda")
    args = (gyg.data_ptr(), wtg.data_ptr(), N, H, W, Ci, Hy, Wy, Co, ks, s, 1, None, None, 0)
    assert lib.bc_pn_conv_nhwc(out.data_ptr(), *args, None, None, 0, None, 0, precision, _st()) == 0
    tol = (2e-5 if precision == 0 else 1e-4) * max(1.0, float(want.abs().max()))      # (split bf16: 16 bits of mantissa per operand)
    assert float((_nchw(out) - want).abs().max()) <= tol
    # + residual gradient behind a ReLU, then accumulated once more on top
    assert lib.bc_pn_conv_nhwc(out.data_ptr(), *args, addg.data_ptr(), maskg.data_ptr(), 0, None, 0, precision, _st()) == 0
    want2 = want + add.double() * (mask > 0)
    assert float((_nchw(out) - want2).abs().max()) <= tol
    assert lib.bc_pn_conv_nhwc(out.data_ptr(), *args, None, None, 1, None, 0, precision, _st()) == 0
    assert float((_nchw(out) - (want2 + want)).abs().max()) <= 2 * tol


@pytest.mark.parametrize("case", CONV_CASES)
@pytest.mark.parametrize("prologue", [False, True])
def test_pn_weight_gradient(case, prologue):
    be, lib = _lib()
    N, H, W, Ci, Co, ks, s = case
    g = torch.Generator().manual_seed(11 + sum(case) + prologue)
    x = torch.randn((N, Ci, H, W), generator=g)
    w = (torch.randn((Co, Ci, ks, ks), generator=g, dtype=torch.float64) * 0.1).requires_grad_()
    sc = (torch.rand(Ci, generator=g) + 0.5) if prologue else None
    sh = (torch.randn(Ci, generator=g) * 0.3) if prologue else None
    xa = x.double()
    if prologue:
        xa = (xa * sc.double().view(1, -1, 1, 1) + sh.double().view(1, -1, 1, 1)).relu()
    y = F.conv2d(xa, w, None, s, 1 if ks == 3 else 0)
    gy = torch.randn(y.shape, generator=g, dtype=torch.float64)
    (want,) = torch.autograd.grad(y, w, gy)
    Hy, Wy = y.shape[2:]
    ws = torch.empty(lib.bc_pn_wgrad_workspace(N, Hy, Wy, Ci, Co, ks), device="cuda")
    dw = torch.full((ks * ks, Ci, Co), float("nan"), device="cuda")
    xg, gyg = _nhwc(x), _nhwc(gy)
    scg, shg = (sc.cuda(), sh.cuda()) if prologue else (None, None)
    rc = lib.bc_pn_wgrad_nhwc(dw.data_ptr(), ws.data_ptr(), ws.numel(), xg.data_ptr(), gyg.data_ptr(), N, H, W, Ci, Hy, Wy, Co, ks, s,
                              scg.data_ptr() if prologue else None, shg.data_ptr() if prologue else None, int(prologue), _st())
    assert rc == 0
    got = dw.view(ks, ks, Ci, Co).permute(3, 2, 0, 1).double().cpu()
    assert float((got - want).abs().max()) <= 2e-5 * max(1.0, float(want.abs().max()))
    # run to run identical (fixed-order split reduction, no atomics)
    dw2 = torch.empty_like(dw)
    lib.bc_pn_wgrad_nhwc(dw2.data_ptr(), ws.data_ptr(), ws.numel(), xg.data_ptr(), gyg.data_ptr(), N, H, W, Ci, Hy, Wy, Co, ks, s,
                         scg.data_ptr() if prologue else None, shg.data_ptr() if prologue else None, int(prologue), _st())
    assert torch.equal(dw, dw2)


@pytest.mark.parametrize("C,pixels,mask_mode", [(32, 5000, 1), (64, 777, 2), (128, 300, 0), (32, 70000, 2), (128, 512, 1), (128, 1024, 1), (128, 2048, 1), (64, 1024, 1), (128, 1024, 0), (128, 1024, 2)])
def test_pn_batchnorm_forward_statistics_and_backward(C, pixels, mask_mode):
    """bc_pn_bn_finalize from partial sums == F.batch_norm(training=True) (scale / shift, saved statistics, running statistics, counter);
    bc_pn_bn_bwd == autograd through batch_norm (+ ReLU / external mask)."""
    be, lib = _lib()
    g = torch.Generator().manual_seed(C + pixels)
    z = (torch.randn((pixels, C), generator=g) * 2 + torch.randn(C, generator=g)).cuda()
    gamma, beta = (torch.rand(C, generator=g) + 0.5).cuda(), torch.randn(C, generator=g).cuda()
    rm, rv = torch.randn(C, generator=g).cuda(), (torch.rand(C, generator=g) + 0.5).cuda()
    nb = torch.tensor(3, dtype=torch.int64, device="cuda")
    # partial sums as a conv epilogue would leave them: chunks of 128 pixels
    n_part = (pixels + 127) // 128
    zp = torch.zeros((n_part * 128, C), device="cuda")
    zp[:pixels] = z
    part = torch.stack([zp.view(n_part, 128, C).sum(1), (zp * zp).view(n_part, 128, C).sum(1)], 1).contiguous()
    scale, shift, mean, invstd = (torch.empty(C, device="cuda") for _ in range(4))
    rm2, rv2 = rm.clone(), rv.clone()
    assert lib.bc_pn_bn_finalize(part.data_ptr(), n_part, C, float(pixels), gamma.data_ptr(), beta.data_ptr(), 1e-5, 0.02, rm2.data_ptr(), rv2.data_ptr(),
                                 nb.data_ptr(), scale.data_ptr(), shift.data_ptr(), mean.data_ptr(), invstd.data_ptr(), _st()) == 0
    z64 = z.double().cpu().requires_grad_()
    rm_ref, rv_ref = rm.double().cpu(), rv.double().cpu()
    y = F.batch_norm(z64.t().reshape(1, C, pixels, 1), rm_ref, rv_ref, gamma.double().cpu(), beta.double().cpu(), True, 0.02, 1e-5)
    y = y.reshape(C, pixels).t()
    got_y = z.double().cpu() * scale.double().cpu() + shift.double().cpu()
    assert float((got_y - y.detach()).abs().max()) <= 1e-4 * max(1.0, float(y.abs().max()))
    assert torch.allclose(rm2.double().cpu(), rm_ref, rtol=1e-5, atol=1e-6) and torch.allclose(rv2.double().cpu(), rv_ref, rtol=1e-5, atol=1e-6)
    assert int(nb) == 4
    # backward
    gout = torch.randn((pixels, C), generator=g).cuda()
    ext = torch.randn((pixels, C), generator=g).cuda()
    if mask_mode == 1:
        act = y.relu()
    elif mask_mode == 2:
        act = y * (ext.double().cpu() > 0)
    else:
        act = y
    (want,) = torch.autograd.grad(act, z64, gout.double().cpu())
    n_bp = lib.bc_pn_bn_bwd_partials(pixels)
    bp, coef = torch.empty(n_bp * 2 * C, device="cuda"), torch.empty(3 * C, device="cuda")
    gz, dg, db = torch.empty_like(z), torch.empty(C, device="cuda"), torch.empty(C, device="cuda")
    assert lib.bc_pn_bn_bwd(gz.data_ptr(), dg.data_ptr(), db.data_ptr(), bp.data_ptr(), coef.data_ptr(), gout.data_ptr(), z.data_ptr(),
                            ext.data_ptr() if mask_mode == 2 else None, mask_mode, scale.data_ptr(), shift.data_ptr(), mean.data_ptr(), invstd.data_ptr(),
                            gamma.data_ptr(), C, pixels, _st()) == 0
    assert float((gz.double().cpu() - want).abs().max()) <= 5e-5 * max(1.0, float(want.abs().max()))


def _small_policy(H=128, W=256, block=32, N=1, classes=19, lr=1e-4, momentum=0.0, wd=1e-3):
    """PolicyTrainRL with the reference's net on a small frame; ``block`` 32 -> the policy input is the frame's resolution."""
    from blockcopy.core.argparser import default_settings
    from blockcopy.policy.policy import build_policy_from_settings

    torch.manual_seed(0)
    settings = default_settings(block_policy="rl_semseg", block_size=block, block_num_classes=classes, block_optim_lr=lr, block_optim_momentum=momentum,
                                block_optim_wd=wd, block_target=0.4)
    pol = build_policy_from_settings(settings).cuda()
    pol.net.train()
    return pol


def _native(pol, frame_shape):
    from blockcopy.policy import native

    why = native.supported(pol.net, pol.optimizer)
    assert why is None, why
    return native.NativePolicyNet(pol.net, pol.optimizer, frame_shape, "cuda")


def _flat_to_param_grads(nat):
    """flat gradient buffer -> {parameter name: tensor in the parameter's shape}"""
    out = {}
    G = nat.G
    for c in nat.convs:
        g = G[c.off:c.off + c.numel].view(c.ks, c.ks, c.Cxp, c.Cy)[:, :, :c.Cx]
        out[c.name + ".weight"] = g.permute(3, 2, 0, 1)
    for b in nat.bns:
        out[b.name + ".weight"] = G[b.off_g:b.off_g + b.C]
        out[b.name + ".bias"] = G[b.off_b:b.off_b + b.C]
    C = nat.last_C
    out["layers.2.0.weight"] = G[nat.off_last_w:nat.off_last_w + 9 * C].view(3, 3, C, 1).permute(3, 2, 0, 1)
    out["layers.2.0.bias"] = G[nat.off_last_b:nat.off_last_b + 1]
    return out


@pytest.mark.parametrize("N,H,W,open_gates", [(1, 128, 256, False), (1, 96, 160, False), (2, 128, 256, True), (2, 96, 160, True), (2, 96, 160, False)])
def test_native_policy_forward_and_gradients_match_autograd(N, H, W, open_gates):
    """Whole net: logits, BatchNorm running statistics and EVERY parameter gradient of the REINFORCE loss against autograd on the module
    in FLOAT64 on the CPU (the truth), with PyTorch's own fp32 GPU result beside it: batch statistics over as few as 30-120 pixels in the
    last stages amplify the summation-order noise of any fp32 implementation, so the bar is relative to what the fp32 library route
    itself achieves (never looser than 2e-2, never tighter than 1e-3 of a tensor's largest gradient)."""
    pol = _small_policy(H=H, W=W, N=N, lr=0.0)          # lr 0: the step leaves the parameters alone, the gradients stay in the flat buffer
    net = pol.net
    if open_gates:
        # every ReLU gate wide open (BatchNorm shift +4): no pre-activation can sit within rounding of zero, see below
        with torch.no_grad():
            for m in net.modules():
                if isinstance(m, torch.nn.BatchNorm2d):
                    m.bias.fill_(4.0)
    ref32 = copy.deepcopy(net)
    ref = copy.deepcopy(net).double().cpu()
    nat = _native(pol, (N, 3, H, W))
    g = torch.Generator().manual_seed(1)
    x = torch.randn((N, 26, H, W), generator=g).cuda()
    feat = torch.zeros((N, H, W, 32), device="cuda")
    feat[..., :26] = x.permute(0, 2, 3, 1)
    x64 = x.double().cpu()
    for rep in range(3):          # eager pass, captured pass, replay
        logits = nat.forward_on(feat).clone()
        want = ref.layers(ref.backbone(x64))
        assert logits.shape == want.shape
        assert float((logits.double().cpu() - want).abs().max()) <= 1e-4 * max(1.0, float(want.abs().max())), rep
    for (k, b_ref), (_, b_nat) in zip(ref.named_buffers(), net.named_buffers()):
        assert torch.allclose(b_ref.double(), b_nat.double().cpu(), rtol=1e-4, atol=1e-5), k
    # REINFORCE seed on random rewards: information gain from two random logit maps
    outputs = torch.randn((N, 19, H // 4, W // 4), generator=g).cuda().contiguous(memory_format=torch.channels_last)
    outputs_prev = (outputs + 0.5 * torch.randn(outputs.shape, generator=g).cuda()).contiguous(memory_format=torch.channels_last)
    grid = (torch.rand((N, 1, H // 32, W // 32), generator=g) > 0.5).cuda()
    from blockcopy.policy.information_gain import InformationGainSemSeg

    ig_ref = InformationGainSemSeg(19)({"outputs": outputs.double().cpu(), "outputs_prev": outputs_prev.double().cpu()})
    cost, target, gamma = 0.55, 0.4, 5.0

    def autograd_grads(model, xin, ig, grid_):
        logits_t = model.layers(model.backbone(xin))
        r = -(cost - target)
        reward = F.adaptive_max_pool2d(ig + r * abs(r) * gamma, output_size=grid_.shape[2:])
        reward = torch.where(grid_, reward, -reward)
        log_probs = -F.binary_cross_entropy_with_logits(logits_t, grid_.to(logits_t.dtype), reduction="none")
        loss = (-log_probs * reward.detach()).mean()
        model.zero_grad()
        loss.backward()
        return loss

    for rep in range(3):
        loss = autograd_grads(ref, x64, ig_ref, grid.cpu())
        autograd_grads(ref32, x, ig_ref.float().cuda(), grid)
        nat.forward_on(feat)
        ig, loss_nat = nat.step(grid, outputs, outputs_prev, cost, target, gamma)
        assert torch.allclose(ig.double().cpu(), ig_ref, rtol=1e-4, atol=1e-6)
        assert abs(float(loss_nat) - float(loss)) <= 1e-5 * max(1.0, abs(float(loss)))
        grads = _flat_to_param_grads(nat)
        worst = (0.0, "")
        for (name, p), (_, p32) in zip(ref.named_parameters(), ref32.named_parameters()):
            if name.startswith("backbone.fc"):
                continue
            gn, g64 = grads[name].double().cpu(), p.grad
            scale = max(float(g64.abs().max()), 1e-12)
            err, err32 = float((gn - g64).abs().max()) / scale, float((p32.grad.double().cpu() - g64).abs().max()) / scale
            worst = max(worst, (err, err32, name))
            # A ReLU gate whose pre-activation is within fp32 rounding of zero opens in one implementation and not in the other: single
            # elements of a map's gradient then move by that element's whole contribution (measured on the N = 2 cases with the default
            # initialisation: two of 131 k gates of one head stage differ from the float64 run, everything else agrees to 3e-8; PyTorch's
            # fp32 route flips gates of its own).  So the element-wise bar (1e-3 of the largest element; measured 2e-6) applies to the runs
            # where no gate flips -- the two single-image seeds and the runs with every gate open -- and the whole-tensor bars always.
            if open_gates or N == 1:
                assert err <= 1e-3, (rep, name, err, err32)
            if g64.numel() > 64:
                assert float((gn - g64).norm() / g64.norm()) <= 3e-2, (rep, name)
                assert float((gn * g64).sum() / gn.norm() / g64.norm()) > 0.999, (rep, name)
        print(f"N={N} {H}x{W} rep {rep}: worst relative gradient error (native, torch fp32, layer) = {worst}")


def test_pn_conv_with_armed_batchnorm_accumulators():
    """bc_pn_arm_bn at the kernel level: a conv whose prologue derives scale / shift from fixed-point sums (16 replicas of [sum x | sum x^2] in
    units of 2^-24) and whose epilogue adds the sums of ITS output to accumulators == the same conv with explicit coefficients from the same
    statistics; the accumulators hold the output's sums to 2^-24 per workgroup; an arm is ONE shot -- consumed by the next call even when that
    call fails, never by a later one."""
    be, lib = _lib()
    N, H, W, Ci, Co = 2, 24, 40, 32, 64
    g = torch.Generator().manual_seed(77)
    x = torch.randn((N, Ci, H, W), generator=g) * 1.5 + 0.3
    w = torch.randn((Co, Ci, 3, 3), generator=g) * 0.1
    gamma, beta = torch.rand(Ci, generator=g) + 0.5, torch.randn(Ci, generator=g) * 0.2
    count, eps = float(N * H * W), 1e-5
    # accumulators of x as a producer would have left them: its sums spread over the replicas
    s1, s2 = x.double().sum((0, 2, 3)), (x.double() ** 2).sum((0, 2, 3))
    acc = torch.zeros((16, 2, Ci), dtype=torch.int64)
    share = torch.rand((16, 1), generator=g).double()
    share /= share.sum()
    acc[:, 0] = torch.round(share * s1 * 2.0 ** 24).long()
    acc[:, 1] = torch.round(share * s2 * 2.0 ** 24).long()
    S1, S2 = acc[:, 0].sum(0).double() / 2.0 ** 24, acc[:, 1].sum(0).double() / 2.0 ** 24
    mean = S1 / count
    var = (S2 / count - mean * mean).clamp_min(0)
    scale = gamma.double() / torch.sqrt(var + eps)
    shift = beta.double() - mean * scale
    want, _ = _conv_ref(x, w, 3, 1, scale.float(), shift.float(), True)
    xg, wg, accg = _nhwc(x), _wk(w), acc.cuda()
    gg, bg = gamma.cuda(), beta.cuda()
    out_acc = torch.zeros((16, 2, Co), dtype=torch.int64, device="cuda")
    out = torch.full((N, H, W, Co), float("nan"), device="cuda")
    args = (out.data_ptr(), xg.data_ptr(), wg.data_ptr(), N, H, W, Ci, H, W, Co, 3, 1, 0, None, None, 1, None, None, 0, None, 0, 1, _st())
    assert lib.bc_pn_arm_bn(accg.data_ptr(), gg.data_ptr(), bg.data_ptr(), count, eps, Ci, out_acc.data_ptr()) == 0
    assert lib.bc_pn_conv_nhwc(*args) == 0
    got = _nchw(out)
    assert float((got - want).abs().max()) <= 2e-5 * float(want.abs().max())
    sums = out_acc.cpu().sum(0).double() / 2.0 ** 24
    assert torch.allclose(sums[0], got.sum((0, 2, 3)), rtol=1e-5, atol=1e-3) and torch.allclose(sums[1], (got * got).sum((0, 2, 3)), rtol=1e-5, atol=1e-3)
    # one shot: an arm in front of a FAILING call is gone afterwards (explicit coefficients in the next call would otherwise be refused)
    assert lib.bc_pn_arm_bn(accg.data_ptr(), gg.data_ptr(), bg.data_ptr(), count, eps, Ci, None) == 0
    assert lib.bc_pn_conv_nhwc(None, *args[1:]) != 0
    sc32, sh32 = scale.float().cuda(), shift.float().cuda()
    out2 = torch.full_like(out, float("nan"))
    assert lib.bc_pn_conv_nhwc(out2.data_ptr(), xg.data_ptr(), wg.data_ptr(), N, H, W, Ci, H, W, Co, 3, 1, 0, sc32.data_ptr(), sh32.data_ptr(), 1, None, None, 0,
                               None, 0, 1, _st()) == 0
    assert float((_nchw(out2) - want).abs().max()) <= 2e-5 * float(want.abs().max())
    assert int(out_acc.cpu().sum(0)[0].sum()) == int(torch.round(sums[0] * 2.0 ** 24).sum())      # (untouched by the un-armed call)


@pytest.mark.parametrize("N,H,W", [(1, 128, 256), (2, 96, 160)])
def test_batchnorm_through_accumulators_equals_the_partial_sum_route(N, H, W, monkeypatch):
    """The forward pass with its BatchNorm statistics in fixed-point accumulators (bc_pn_arm_bn: integer atomics, the consumers derive their
    coefficients, ONE finish launch per pass) against the per-layer route (per-workgroup partial sums + bc_pn_bn_finalize after every conv) on the
    same module: logits within 2e-6 (the sums differ only by the 2^-24 quantum of a workgroup's partial sum), running statistics and batch counters
    alike, three passes each (eager, captured, replay), and the accumulators back at zero after every pass."""
    from blockcopy.policy import native

    pol_a, pol_b = _small_policy(H=H, W=W, N=N, lr=0.0), _small_policy(H=H, W=W, N=N, lr=0.0)
    pol_b.net.load_state_dict(pol_a.net.state_dict())
    monkeypatch.setattr(native, "BN_ACC", True)
    nat_a = _native(pol_a, (N, 3, H, W))
    monkeypatch.setattr(native, "BN_ACC", False)
    nat_b = _native(pol_b, (N, 3, H, W))
    g = torch.Generator().manual_seed(5)
    for rep in range(3):
        feat = torch.zeros((N, H, W, 32), device="cuda")
        feat[..., :26] = torch.randn((N, H, W, 26), generator=g).cuda()
        la, lb = nat_a.forward_on(feat).clone(), nat_b.forward_on(feat).clone()
        assert float((la - lb).abs().max()) <= 2e-6 * max(1.0, float(lb.abs().max())), rep
        assert all(int(bn.acc.abs().max()) == 0 for bn in nat_a.bns), rep
    for (k, ba), (_, bb) in zip(pol_a.net.named_buffers(), pol_b.net.named_buffers()):
        assert torch.allclose(ba.double(), bb.double(), rtol=1e-6, atol=1e-7), k
    for bn_a, bn_b in zip(nat_a.bns, nat_b.bns):
        for name in ("scale", "shift", "mean", "invstd"):
            assert torch.allclose(getattr(bn_a, name), getattr(bn_b, name), rtol=2e-6, atol=1e-7), (bn_a.name, name)


@pytest.mark.parametrize("momentum,wd", [(0.0, 0.0), (0.5, 1e-3)])
def test_pn_rmsprop_kernel_matches_torch(momentum, wd):
    """bc_pn_rmsprop == torch.optim.RMSprop (uncentred) on the same gradients, three steps with carried state."""
    be, lib = _lib()
    g = torch.Generator().manual_seed(17)
    n = 10007
    p0 = torch.randn(n, generator=g).cuda()
    ref = torch.nn.Parameter(p0.clone())
    opt = torch.optim.RMSprop([ref], lr=1e-3, alpha=0.99, eps=1e-8, weight_decay=wd, momentum=momentum)
    p, sq, mom = p0.clone(), torch.zeros(n, device="cuda"), torch.zeros(n, device="cuda")
    for it in range(3):
        grad = (torch.randn(n, generator=g) * 10 ** float(torch.randint(-6, 0, (1,), generator=g))).cuda()
        ref.grad = grad.clone()
        opt.step()
        assert lib.bc_pn_rmsprop(p.data_ptr(), grad.data_ptr(), sq.data_ptr(), mom.data_ptr(), n, 1e-3, 0.99, 1e-8, wd, momentum, _st()) == 0
        assert torch.allclose(p, ref.detach(), rtol=1e-5, atol=1e-7), (it, float((p - ref).abs().max()))
        assert torch.allclose(sq, opt.state[ref]["square_avg"], rtol=1e-5, atol=1e-12)
        if momentum > 0:
            want = opt.state[ref]["momentum_buffer"]
            assert torch.allclose(mom, want, rtol=1e-5, atol=1e-6), (it, float((mom - want).abs().max()), float(((mom - want).abs() / want.abs().clamp_min(1e-3)).max()))


def test_native_policy_step_moves_the_module_like_torch_rmsprop():
    """REINFORCE updates with a real learning rate, momentum and weight decay on: the module's parameters (exported by the native step) move
    the way autograd + torch.optim.RMSprop moves a copy.  RMSprop's first steps are sign-like (lr * g / sqrt(0.01 g^2)) and the library's
    gradients carry their own rounding (measured here: up to 1e-3 of a tensor's largest element on a first call), so elements whose gradient is
    near zero step the other way in either implementation: the comparison is by direction and sign agreement, per tensor; the arithmetic of the
    update itself is compared exactly in test_pn_rmsprop_kernel_matches_torch and the gradients in the test above."""
    N, H, W = 1, 128, 256
    pol = _small_policy(H=H, W=W, lr=1e-3, momentum=0.5, wd=1e-3)
    net = pol.net
    ref = copy.deepcopy(net)
    opt = torch.optim.RMSprop([p for p in ref.parameters()], lr=1e-3, weight_decay=1e-3, momentum=0.5)
    nat = _native(pol, (N, 3, H, W))
    g = torch.Generator().manual_seed(5)
    from blockcopy.policy.information_gain import InformationGainSemSeg

    for it in range(2):
        x = torch.randn((N, 26, H, W), generator=g).cuda()
        feat = torch.zeros((N, H, W, 32), device="cuda")
        feat[..., :26] = x.permute(0, 2, 3, 1)
        outputs = torch.randn((N, 19, H // 4, W // 4), generator=g).cuda()
        outputs_prev = (outputs + 0.5 * torch.randn(outputs.shape, generator=g).cuda())
        grid = (torch.rand((N, 1, H // 32, W // 32), generator=g) > 0.5).cuda()
        logits_t = ref.layers(ref.backbone(x))
        logits = nat.forward_on(feat)
        assert float((logits - logits_t).abs().max()) <= 1e-4 * max(1.0, float(logits_t.abs().max())), it
        ig_ref = InformationGainSemSeg(19)({"outputs": outputs, "outputs_prev": outputs_prev})
        cost, target, gamma = 0.3 + 0.1 * it, 0.4, 5.0
        r = -(cost - target)
        reward = F.adaptive_max_pool2d(ig_ref + r * abs(r) * gamma, output_size=grid.shape[2:])
        reward = torch.where(grid, reward, -reward)
        loss = (F.binary_cross_entropy_with_logits(logits_t, grid.float(), reduction="none") * reward.detach()).mean()
        opt.zero_grad()
        loss.backward()
        before = {k: v.detach().clone() for k, v in ref.named_parameters()}
        opt.step()
        nat.step(grid, outputs, outputs_prev, torch.tensor(cost, dtype=torch.float64, device="cuda"), target, gamma)
        for (name, p_ref), (_, p) in zip(ref.named_parameters(), net.named_parameters()):
            if name.startswith("backbone.fc"):
                continue
            d_ref, d = (p_ref - before[name]).flatten(), (p.detach() - before[name]).flatten()
            assert float(d_ref.abs().max()) > 1e-4, name      # (the step is not a no-op)
            assert float((d * d_ref).sum() / d.norm() / d_ref.norm()) > 0.95, (it, name)
            assert float((torch.sign(d) == torch.sign(d_ref)).float().mean()) > 0.9, (it, name)
        # same parameters again for the next round (the native side imports them: version counters), its own optimizer state carried on
        with torch.no_grad():
            for (_, p_ref), (_, p) in zip(ref.named_parameters(), net.named_parameters()):
                p.copy_(p_ref)
            for (_, b_ref), (_, b) in zip(ref.named_buffers(), net.named_buffers()):
                b.copy_(b_ref)


def test_native_policy_tracks_external_parameter_writes():
    """load_state_dict on the module is seen by the native forward (version counters), and the module sees the native step's update."""
    N, H, W = 1, 64, 128
    pol = _small_policy(H=H, W=W, lr=1e-3)
    nat = _native(pol, (N, 3, H, W))
    feat = torch.randn((N, H, W, 32), device="cuda")
    feat[..., 26:] = 0
    a = nat.forward_on(feat).clone()
    gen = torch.Generator().manual_seed(2)      # (a scaling would do nothing: a batch-statistics BatchNorm follows every conv)
    sd = {k: (v + 0.05 * torch.randn(v.shape, generator=gen).to(v.device) if v.dtype.is_floating_point and k.endswith("conv1.weight") else v)
          for k, v in pol.net.state_dict().items()}
    pol.net.load_state_dict(sd)
    assert not nat.params_current()
    b = nat.forward_on(feat).clone()
    assert nat.params_current() and float((a - b).abs().max()) > 1e-4
    x = feat[..., :26].permute(0, 3, 1, 2).contiguous()
    want = pol.net.layers(pol.net.backbone(x))
    assert float((b - want).abs().max()) <= 1e-4 * max(1.0, float(want.abs().max()))
    before = pol.net.backbone.conv1.weight.detach().clone()
    outputs = torch.randn((N, 19, H // 4, W // 4), device="cuda")
    nat.step(torch.ones((N, 1, H // 32, W // 32), dtype=torch.bool, device="cuda"), outputs, outputs + 1.0, 0.5, 0.4, 5.0)
    assert float((pol.net.backbone.conv1.weight - before).abs().max()) > 0
    assert nat.params_current()


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16])
@pytest.mark.parametrize("cl", [False, True])
def test_pn_infogain_and_features(dtype, cl):
    be, lib = _lib()
    from blockcopy.policy.information_gain import InformationGainSemSeg

    g = torch.Generator().manual_seed(3)
    N, C, H, W = 2, 19, 64, 96
    cur = torch.randn((N, C, H, W), generator=g).cuda().to(dtype)
    prev = (cur.float() + 0.7 * torch.randn((N, C, H, W), generator=g).cuda()).to(dtype)
    if cl:
        cur, prev = cur.contiguous(memory_format=torch.channels_last), prev.contiguous(memory_format=torch.channels_last)
    want = InformationGainSemSeg(C)({"outputs": cur.float(), "outputs_prev": prev.float()})
    ig = torch.empty((N, 1, H // 4, W // 4), device="cuda")
    code = {torch.float32: 0, torch.float16: 1}[dtype]
    assert lib.bc_pn_infogain(ig.data_ptr(), cur.data_ptr(), prev.data_ptr(), code, N, C, H, W, *cur.stride(), H // 4, W // 4, 4.0, 4.0, _st()) == 0
    assert torch.allclose(ig, want, rtol=1e-4, atol=1e-6)


def test_native_route_in_the_rl_loop_matches_the_autograd_route(monkeypatch):
    """PolicyTrainRL on a SwiftNet clip: the native route (default) and the autograd route (BLOCKCOPY_NATIVE_POLICY=0) see the same
    grids (teacher forcing) and must agree on logits, probabilities, information gain and the direction of the update."""
    import random

    from bc_workloads import harness, seeded
    from blockcopy.policy import native

    def run(enabled):
        monkeypatch.setattr(native, "ENABLED", enabled)
        torch.manual_seed(0)
        random.seed(0)
        model = harness.build_model("resnet18", block_policy="rl_semseg", block_size=32, block_target=0.4, device="cuda", channels_last=True, block_graph=1,
                                    block_train_interval=2)
        model.policy.device_step = False
        gen = torch.Generator().manual_seed(9)
        grids = [(torch.rand((1, 1, 4, 8), generator=gen) > 0.5) for _ in range(8)]
        frame = {"t": 0}
        model.policy.quantize_number_exec_grid = lambda sampled: grids[frame["t"]].clone()
        w0 = model.policy.net.state_dict()["backbone.conv1.weight"].clone()
        rec = []
        model.reset_temporal()
        with torch.no_grad():
            for t in range(8):
                frame["t"] = t
                y = model(seeded.synthetic_frame(40 + t, (1, 3, 128, 256)).cuda())
                pm = model.policy_meta
                rec.append((y.clone(), pm["grid_probs"].detach().clone() if t > 0 else None,
                            pm["information_gain"].detach().clone() if "information_gain" in pm else None))
        used = [n for n in model.policy._natives.values() if n is not None]
        return rec, model.policy.net.state_dict()["backbone.conv1.weight"] - w0, used

    rec_n, d_n, used_n = run(True)
    rec_a, d_a, used_a = run(False)
    assert used_n and used_n[0].steps >= 3 and not used_a
    for t, ((y1, p1, i1), (y2, p2, i2)) in enumerate(zip(rec_n, rec_a)):
        assert float((y1 - y2).abs().max()) <= 1e-4 * max(1.0, float(y2.abs().max())), t
        if p1 is not None:
            assert float((p1 - p2).abs().max()) <= (1e-4 if t < 2 else 0.05), t
        if i1 is not None:
            assert torch.allclose(i1, i2, rtol=1e-3, atol=1e-5), t
    cos = float((d_n * d_a).sum() / d_n.norm() / d_a.norm())
    assert cos > 0.9, cos


@pytest.mark.parametrize("dtype,cl", [(torch.float32, False), (torch.float16, True)])
def test_pn_features_and_probs(dtype, cl):
    """The channels-last, channel-padded policy input (bc_pn_features_nhwc) == the reference recipe (4 x F.interpolate(nearest) + casts +
    centring + concat, policy/net.py:82-113), bit for bit; bc_pn_probs == sigmoid / -BCE-with-logits."""
    pol = _small_policy(H=128, W=256, block=64)           # block 64: the policy input is half the frame's resolution
    nat = _native(pol, (1, 3, 128, 256))
    g = torch.Generator().manual_seed(4)
    mf = torch.channels_last if cl else torch.contiguous_format
    meta = {"inputs": torch.randn((1, 3, 128, 256), generator=g).cuda().to(dtype).contiguous(memory_format=mf),
            "frame_state": torch.randn((1, 3, 128, 256), generator=g).cuda().to(dtype).contiguous(memory_format=mf),
            "output_repr": torch.randn((1, 19, 32, 64), generator=g).cuda().to(dtype).contiguous(memory_format=mf),
            "grid": (torch.rand((1, 1, 2, 4), generator=g) > 0.5).cuda()}
    import blockcopy.policy.net as pnet

    old = pnet.FUSED_FEATURES
    pnet.FUSED_FEATURES = False
    try:
        want = pol.net.build_features(meta)             # the stock ops
    finally:
        pnet.FUSED_FEATURES = old
    assert nat.features(meta)
    got = nat.feat
    assert got.shape == (1, 64, 128, 32) and torch.equal(got[..., :26], want.permute(0, 2, 3, 1).float()) and float(got[..., 26:].abs().max()) == 0.0
    # decision bookkeeping
    nat.logits.copy_(torch.randn(nat.n_total, generator=g).cuda() * 3)
    probs, logp = nat.decision_probs(meta["grid"])
    lg = nat.logits.view(1, 1, 2, 4)
    assert torch.allclose(probs, torch.sigmoid(lg), rtol=1e-6, atol=1e-7)
    assert torch.allclose(logp, -F.binary_cross_entropy_with_logits(lg, meta["grid"].float(), reduction="none"), rtol=1e-5, atol=1e-6)
