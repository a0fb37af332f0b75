"""Pending elementwise work on a packed tensor (SURVEY.md section 8(f)-3 "fused per-block ops around the convs").

Between two convs a CNN runs bias add / batch-norm / residual add / ReLU as separate launches; on the packed batch each
of them is a 5-7 us kernel that re-reads and re-writes the whole activation (~75 launches and ~0.6 ms per SwiftNet-RN18
frame at C2).  The engine instead records them on the TensorWrapper as

        value = relu?( raw * scale[c] + shift[c] + add )

and folds the record into the next consumer: the halo gather of a padded op applies (scale, shift, relu) to the values
it gathers (``bc_pad_ring_act``), anything else triggers ONE fused pass (``bc_affine_act``).  Inference only: the
per-channel vectors are derived from module parameters once and cached.
"""
from __future__ import annotations

import os
import weakref

import torch

ENABLED = os.environ.get("BLOCKCOPY_FUSE", "1") != "0"


def set_enabled(flag: bool) -> bool:
    global ENABLED
    prev, ENABLED = ENABLED, bool(flag)
    return prev


class Pending:
    """value = relu?( base * scale[c] + shift[c] + add ),  base = the stored data, or -- deferred interpolation --
    ``bilinear(interp[0])`` with ``interp = (source, H, W, align_corners, rh, rw)``: the resampling launch has not
    happened yet (the stored data is an unwritten placeholder of the right shape) and will carry the rest of the record
    as its epilogue, e.g. a decoder's ``x = upsample(x); x += skip`` becomes one kernel."""

    __slots__ = ("scale", "shift", "add", "relu", "interp", "add_version", "conv", "src_guard", "up", "up_guard")

    def __init__(self, scale=None, shift=None, add=None, relu=False, interp=None, add_version=None, conv=None, src_guard=None, up=None,
                 up_guard=None):
        self.scale, self.shift, self.add, self.relu, self.interp = scale, shift, add, relu, interp
        # "+ bilinear(up[0])" carried by a deferred pointwise conv: up = the interp record of a deferred resampling that was added to
        # the conv's (not yet computed) result -- a decoder's `x = upsample(x); x += conv1x1(skip)` is then the conv launch alone
        # (backend.conv1x1(upsample=...)); like `add` it ends the affine-only state of the record
        self.up, self.up_guard = up, up_guard
        self.add_version = add._version if (add is not None and add_version is None) else add_version
        # (tensor, version) of the input a deferred producer will read at launch time (see checked_source)
        self.src_guard = src_guard if src_guard is not None else ((interp[0], interp[0]._version) if interp is not None else None)
        # deferred fused halo+conv: ``conv = (launch, kwargs)``; like ``interp`` the stored data is an unwritten placeholder and
        # the launch happens when the value is needed, with the rest of the record as the kernel's epilogue -- the end of a
        # residual block (conv -> + bias -> + identity -> ReLU) is then ONE launch and no separate elementwise pass
        self.conv = conv

    def copy(self):
        return Pending(self.scale, self.shift, self.add, self.relu, self.interp, self.add_version, self.conv, self.src_guard, self.up, self.up_guard)

    @property
    def deferred(self):
        """True when the base value itself has not been computed yet (deferred resampling or conv)."""
        return self.interp is not None or self.conv is not None

    def set_add(self, t):
        """Record the residual operand by alias, together with its version counter: the sum is formed when the record
        is consumed, so a real in-place write to the operand in between would silently change it."""
        self.add = t
        self.add_version = t._version

    def checked_add(self):
        """The residual operand, or a loud error if it was modified in place after the add was recorded."""
        if self.add is not None and self.add._version != self.add_version:
            raise RuntimeError("blockcopy lazy fusion: a tensor recorded as a residual operand (x + y) was modified in place "
                               "before the sum was consumed; materialise the sum first (e.g. call .clone() on it) or set "
                               "BLOCKCOPY_FUSE=0")
        return self.add

    def defer_conv(self, launch, kwargs, source, registry=None):
        """Record a deferred producer launch (``launch(epilogue=..., **kwargs)``) that will read ``source`` when it runs.
        ``conv = (launch, kwargs, state)``; copies of the record share ``state`` (``state["launched"]``).  A launch with a side
        effect on temporal state (the fused halo+conv refreshes its layer's ring cache) is also put on ``registry`` -- the
        frame's list of outstanding producers -- so that the frame can run it even if nobody ever asks for its value."""
        self.src_guard = (source, source._version)
        self.conv = (launch, kwargs, {"launched": False, "guard": self.src_guard})
        if registry is not None:
            registry.append(self.conv)
        return self

    def check_source(self):
        """Loud error if the input of a deferred producer (conv / resampling) was written in place after the op was recorded:
        the launch happens when the value is needed, so it would silently see the new contents."""
        if ((self.src_guard is not None and self.src_guard[0]._version != self.src_guard[1])
                or (self.up_guard is not None and self.up_guard[0]._version != self.up_guard[1])):
            raise RuntimeError("blockcopy lazy fusion: the input of a deferred conv / interpolation was modified in place before the "
                               "result was consumed; consume the result first or set BLOCKCOPY_DEFER_CONV=0 / BLOCKCOPY_FUSE=0")

    @property
    def affine_only(self):
        return self.add is None and not self.relu and self.up is None


_cache = {}


def clear_cache():
    """Forget derived per-channel vectors (call after changing weights of a wrapped model in place)."""
    _cache.clear()


def _key(*tensors):
    # identity + storage address + version: a freed model's parameters may hand their addresses to a new model
    return tuple((id(t), t.data_ptr(), t._version) if t is not None else None for t in tensors)


def _lookup(k, tensors):
    hit = _cache.get(k)
    if hit is None:
        return None
    value, refs = hit
    for r, t in zip(refs, tensors):
        if (r is None) != (t is None) or (r is not None and r() is not t):
            del _cache[k]      # stale: an id was recycled
            return None
    return value


def _store(k, tensors, value):
    if torch.cuda.is_available() and torch.cuda.is_current_stream_capturing():
        return value   # memory of a graph's private pool must not outlive the capture as a cached constant
    if len(_cache) > 4096:
        _evict()
    _cache[k] = (value, tuple(weakref.ref(t) if t is not None else None for t in tensors))
    return value


def _evict():
    """Make room WITHOUT forgetting what live models still use: entries of dead tensors and of superseded versions of live ones (a
    parameter that is trained in place -- the online policy -- leaves one entry per optimizer step) go first; only if that frees less
    than a quarter, the oldest half goes too.  (Clearing everything, as this used to do, made a later graph capture re-derive -- inside
    the capture -- the packed weights of a model that had been warmed up: found by the GPU suite when enough tests had filled the cache.)"""
    def stale(k, refs):
        for part, r in zip([p for p in k if isinstance(p, tuple) or p is None][-len(refs):], refs):
            if r is None:
                continue
            t = r()
            if t is None or part is None or t._version != part[2] or t.data_ptr() != part[1]:
                return True
        return False

    for k in [k for k, (_, refs) in _cache.items() if stale(k, refs)]:
        del _cache[k]
    if len(_cache) > 3072:
        for k in list(_cache)[:len(_cache) // 2]:
            del _cache[k]


def channel_vector(t: torch.Tensor) -> torch.Tensor:
    """fp32 contiguous copy of a per-channel parameter, derived once per parameter object."""
    k = ("vec",) + _key(t)
    v = _lookup(k, (t,))
    if v is None:
        with torch.no_grad():
            v = _store(k, (t,), t.detach().float().contiguous().clone() if t.dtype != torch.float32 else t.detach().contiguous())
    return v


def packed_conv3x3_weight(weight: torch.Tensor, pack) -> torch.Tensor:
    """MFMA operand stream of a (Cout, Cin, 3, 3) weight for the fused halo+conv kernel, derived once per parameter."""
    k = ("wpk", getattr(pack, "__name__", "")) + _key(weight)      # (per packing: one parameter may feed two kernel forms)
    v = _lookup(k, (weight,))
    if v is None:
        with torch.no_grad():
            v = _store(k, (weight,), pack(weight))
    return v


CONV_MODE = os.environ.get("BLOCKCOPY_CONV", "auto")   # auto | native | winograd | winograd-wide | winograd4 | winograd4-split | split | library
CONV_TUNE = os.environ.get("BLOCKCOPY_CONV_TUNE", "1") != "0"   # auto mode: measure a layer shape the plan table does not know (eager runs only)
_conv_plans = {}      # (n_exec, bs, cin, cout, n_total, dtype, stride, ks) -> None (library conv) | decomposition code (-1 = library's own choice)
DEFER_CONV = os.environ.get("BLOCKCOPY_DEFER_CONV", "1") != "0"   # fused convs launch lazily with the recorded elementwise work as epilogue
UPSAMPLE_EPILOGUE = os.environ.get("BLOCKCOPY_UPSAMPLE_EPILOGUE", "1") != "0"   # upsample(x) + conv1x1(skip): the resampling rides in the conv's epilogue
POINTWISE = os.environ.get("BLOCKCOPY_POINTWISE", "1") != "0"     # 1x1 convs through the fused kernel's one-tap form (prologue / epilogue fusion)
GROUP_NORM = os.environ.get("BLOCKCOPY_GROUP_NORM", "1") != "0"   # group_norm on packed tiles as a recorded per-channel affine map (one stats pass)
ADAPTIVE_POOL = os.environ.get("BLOCKCOPY_ADAPTIVE_POOL", "1") != "0"   # adaptive_avg_pool2d of dense channels-last maps (pyramid pooling) in the library's kernel
STEM_KERNEL = os.environ.get("BLOCKCOPY_STEM", "1") != "0"        # network input: window gather + 7x7 stem conv in one kernel
SPP_FUSED = os.environ.get("BLOCKCOPY_SPP", "1") != "0"            # SwiftNet's pyramid pooling (after its first block) as two launches instead of 15
PRED_KERNEL = os.environ.get("BLOCKCOPY_PRED", "1") != "0"        # dense 3x3 convs to <= 4 channels on a map handed out by to_tensor (detector prediction convs)
HEAD_KERNEL = os.environ.get("BLOCKCOPY_HEAD", "1") != "0"        # network output: prologue + 1x1 conv to <= 32 channels + out-of-place combine in one kernel
# tuner: charge the library route the elementwise pass that follows a conv in a CNN (bias / folded BN, residual add, ReLU: it rides in the
# fused kernels' epilogue but costs the library route one more pass over the result).  Measured on C4: 204 -> 212 fps (profiles/r03)
TUNE_EPILOGUE_COST = os.environ.get("BLOCKCOPY_TUNE_EPILOGUE_COST", "1") != "0"
CONV_TUNE_LOG = []    # (key, {candidate: microseconds}, choice) of every measurement, for the bench report
WINOGRAD_FLAG = 0x600  # decomposition codes with one of these bits run a Winograd F(2x2,3x3) form (0x200: csrc/conv3x3_wino.inc,
WINOGRAD_WIDE = 0x400  # 0x400: the wide wave tile of csrc/conv3x3_wino32.inc)
WINOGRAD_F4 = 0x1000   # codes with this bit: the Winograd F(4x4,3x3) form of csrc/conv3x3_wino4.inc
SPLIT_FLAG = 0x2000    # codes with this bit: the direct form of an fp32 layer on the 16-bit matrix pipe, operands split hi + lo (conv3x3_v2.inc BC_F32S)

# Plan table persistence.  A plan decides which KERNEL FORM a layer runs in (library conv / direct MFMA form / Winograd form), and
# the forms differ by fp32 rounding, so a run is only reproducible -- from run to run and from rank to rank -- with a fixed table.
# The package ships one measured on MI355X for the BASELINE configs (plans/gfx950.json, written by tools/tune_plans.py); it is
# loaded at import.  BLOCKCOPY_CONV_PLAN=<file> loads another one instead, BLOCKCOPY_CONV_PLAN=none starts empty.  Shapes the table
# does not know are measured once (BLOCKCOPY_CONV_TUNE=1, default) or follow a fixed rule (=0); `save_conv_plans` writes the table
# back.  PLAN_STATS counts where this process's decisions came from.
DEFAULT_PLAN_FILE = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "plans", "gfx950.json")
PLAN_FILE = os.environ.get("BLOCKCOPY_CONV_PLAN", DEFAULT_PLAN_FILE)
PLAN_STATS = {"from_table": 0, "tuned_live": 0, "fixed_rule": 0}
PLAN_KEYS_SEEN = {}    # layer-shape key -> lookups in this process (tools/refine_plans.py: which entries a workload actually uses)
_DTYPE_NAMES = {torch.float32: "f32", torch.float16: "f16", torch.bfloat16: "bf16"}
_DTYPE_BY_NAME = {v: k for k, v in _DTYPE_NAMES.items()}
_loaded_keys = set()


def _key_to_str(key):
    n_exec, bs, cin, cout, n_total, dtype, stride, ks = key
    return f"{n_exec},{bs},{cin},{cout},{n_total},{_DTYPE_NAMES[dtype]},{stride},{ks}"


def _key_from_str(text):
    f = text.split(",")
    return (int(f[0]), int(f[1]), int(f[2]), int(f[3]), int(f[4]), _DTYPE_BY_NAME[f[5]], int(f[6]), int(f[7]))


def clear_conv_plans():
    _conv_plans.clear()
    _loaded_keys.clear()
    del CONV_TUNE_LOG[:]
    for k in PLAN_STATS:
        PLAN_STATS[k] = 0


def load_conv_plans(path, replace=False):
    """Read a plan table written by ``save_conv_plans``; returns the number of entries.  Entries of the file win over
    plans already in memory."""
    import json

    with open(path) as f:
        doc = json.load(f)
    if replace:
        clear_conv_plans()
    for text, plan in doc["plans"].items():
        key = _key_from_str(text)
        _conv_plans[key] = None if plan is None else int(plan)
        _loaded_keys.add(key)
    return len(doc["plans"])


def save_conv_plans(path, note=None):
    """Write every plan this process knows (loaded and measured) as JSON: {"plans": {"n_exec,bs,cin,cout,n_total,dtype,stride,ks":
    code | null}} with null = halo gather + library conv."""
    import json

    doc = {"format": 1, "note": note or "blockcopy conv plan table (fusion.conv3x3_plan)",
           "plans": {_key_to_str(k): v for k, v in sorted(_conv_plans.items(), key=lambda kv: _key_to_str(kv[0]))}}
    tmp = f"{path}.tmp{os.getpid()}"
    with open(tmp, "w") as f:
        json.dump(doc, f, indent=0, sort_keys=True)
        f.write("\n")
    os.replace(tmp, path)
    return len(doc["plans"])


def conv_plan_hash():
    """Short digest of the plan table in memory (bench.py prints it: equal hashes = equal kernel forms per layer)."""
    import hashlib

    text = ";".join(f"{_key_to_str(k)}={v}" for k, v in sorted(_conv_plans.items(), key=lambda kv: _key_to_str(kv[0])))
    return hashlib.sha256(text.encode()).hexdigest()[:16]


def _load_default_plans():
    if PLAN_FILE and PLAN_FILE.lower() != "none":
        if os.path.exists(PLAN_FILE):
            load_conv_plans(PLAN_FILE)
        elif PLAN_FILE != DEFAULT_PLAN_FILE:
            raise FileNotFoundError(f"BLOCKCOPY_CONV_PLAN={PLAN_FILE} does not exist")


_load_default_plans()


def default_native_conv3x3(n_exec: int, bs: int, cin: int, cout: int) -> bool:
    """Untuned choice between the hand-written fused halo+conv and halo gather + library conv (profiles/r02/kbench_conv_*:
    the CU-balanced kernel wins on every SwiftNet / CSP layer shape except 4x4 tiles with few executed tiles)."""
    return not (bs <= 4 and n_exec * bs * bs < 1024)


def conv3x3_plan(n_exec: int, bs: int, cin: int, cout: int, n_total: int, dtype, tuner=None, stride: int = 1, ks: int = 3, candidates=None):
    """How to run one padded 3x3 conv layer (or, ``ks=1``, one pointwise conv): ``None`` = halo gather + library conv, ``int`` =
    the fused halo+conv kernel with that decomposition code (-1: the library's cost model; codes with WINOGRAD_FLAG: Winograd form).

    ``BLOCKCOPY_CONV`` = ``library`` | ``native`` (direct MFMA form everywhere) | ``winograd`` / ``winograd-wide`` / ``winograd4`` (every layer that
    lists a Winograd candidate of the 16-channel F(2x2) / the wide F(2x2) / the F(4x4) form runs its first one, the rest the direct form) | ``auto``: the plan table decides (see PLAN_FILE above); a shape it
    does not know is MEASURED once (``tuner()`` times the library route and every decomposition on the live tensors -- the conv
    library's own solver-search idea -- never during graph capture) or, untuned, follows a fixed rule."""
    if CONV_MODE == "library":
        return None
    if CONV_MODE == "native":
        return -1
    if CONV_MODE == "split":      # every fp32 layer in the direct form on the 16-bit matrix pipe (codes | 0x2000), the rest as "native"
        if dtype == torch.float32 and candidates is not None:
            sp = [c for c in candidates() if c >= 0 and (c & SPLIT_FLAG) and not (c & 0x100)]
            if sp:
                return sp[0]
        return -1
    if CONV_MODE in ("winograd", "winograd-wide", "winograd4", "winograd4-split"):
        if ks == 3 and stride == 1 and dtype == torch.float32 and candidates is not None:
            flag = WINOGRAD_WIDE if CONV_MODE == "winograd-wide" else (WINOGRAD_F4 if CONV_MODE.startswith("winograd4") else 0x200)
            cands = [c for c in candidates() if c >= 0]
            wino = [c for c in cands if c & flag]
            if CONV_MODE == "winograd4-split":      # the F(4x4) form with its products on the 16-bit matrix pipe (codes 0x5000 | c)
                wino = [c for c in cands if (c & 0x5000) == 0x5000]
            if wino:
                return wino[0]
            if CONV_MODE.startswith("winograd4"):      # (tiles the F(4x4) form does not cover: 4x4 tiles, sizes that are no multiple of 16: the F(2x2) form)
                wino = [c for c in cands if c & 0x200]
                if wino:
                    return wino[0]
        return -1
    key = (n_exec, bs, cin, cout, n_total, dtype, stride, ks)
    PLAN_KEYS_SEEN[key] = PLAN_KEYS_SEEN.get(key, 0) + 1
    if key in _conv_plans:
        if key in _loaded_keys:
            PLAN_STATS["from_table"] += 1
        return _conv_plans[key]
    capturing = torch.cuda.is_available() and torch.cuda.is_current_stream_capturing()
    if tuner is not None and CONV_TUNE and not capturing:
        times = tuner()
        if times:
            best = min(times, key=times.get)
            plan = None if best == "library" else int(best)
            _conv_plans[key] = plan
            CONV_TUNE_LOG.append((key, times, best))
            PLAN_STATS["tuned_live"] += 1
            return plan
    PLAN_STATS["fixed_rule"] += 1
    if ks == 1:
        return -1 if n_exec * bs * bs >= 4096 else None       # untuned rule for pointwise convs: the library for tiny maps
    return -1 if default_native_conv3x3(n_exec, bs // stride, cin, cout) else None


def batchnorm_affine(running_mean, running_var, weight, bias, eps):
    """Eval-mode batch norm as y = x*scale + shift."""
    ts = (running_mean, running_var, weight, bias)
    k = ("bn", float(eps)) + _key(*ts)
    v = _lookup(k, ts)
    if v is None:
        with torch.no_grad():
            inv = torch.rsqrt(running_var.detach().float() + eps)
            scale = inv if weight is None else weight.detach().float() * inv
            shift = -running_mean.detach().float() * scale
            if bias is not None:
                shift = shift + bias.detach().float()
        v = _store(k, ts, (scale.contiguous(), shift.contiguous()))
    return v


def compose_affine(p_scale, p_shift, scale, shift):
    """(x*p_scale + p_shift)*scale + shift  ->  (x*S + T); operands are cached vectors, so the result is cached too."""
    ts = (p_scale, p_shift, scale, shift)
    k = ("cmp",) + _key(*ts)
    v = _lookup(k, ts)
    if v is None:
        with torch.no_grad():
            S = scale if p_scale is None else p_scale * scale
            T = shift if p_shift is None else p_shift * scale + shift
        v = _store(k, ts, (S.contiguous(), T.contiguous()))
    return v


def add_shifts(a, b):
    if a is None:
        return b
    if b is None:
        return a
    k = ("add",) + _key(a, b)
    v = _lookup(k, (a, b))
    if v is None:
        with torch.no_grad():
            v = _store(k, (a, b), (a + b).contiguous())
    return v
