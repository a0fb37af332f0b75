"""Execution policies: each frame they write ``policy_meta['grid']`` (bool, (N,1,GH,GW), on the frame's device)
and the counters ``num_exec / num_total / perc_exec``.

Contract and policy names follow the reference (policy/policy.py:14-370).  MI355X-first difference: every policy
here decides its grid on the host (or pulls it to the host exactly once) and publishes that copy as
``policy_meta['grid_host']``; the engine builds its index tables from it, so the block path itself never
synchronises with the GPU (the reference syncs in ``int(grid.sum())`` :84, ``grid.cpu()`` :136 and twice more in
``_process_grid``)."""
from __future__ import annotations

import abc
import logging
import os
import random
from abc import abstractmethod

import torch
import torch.nn.functional as F
from torch.distributions import Bernoulli

from blockcopy.policy.information_gain import InformationGain, InformationGainObjectDetection, InformationGainSemSeg
from blockcopy.policy.net import PolicyNet, build_policy_net_from_settings
from blockcopy.utils.profiler import timings


def _backend():
    from blockcopy.backend import get_backend

    return get_backend()


def build_policy_from_settings(settings: dict):
    """Policy object for ``settings['block_policy']`` in {all, none, random, fixed, rl_semseg, rl_objectdetection}."""
    name = settings["block_policy"]
    logging.info(f"> Policy: {name} with execution percentage target {settings['block_target']} and block size {settings['block_size']}")
    quantize_number_exec = 1 / 16
    common = dict(block_size=settings["block_size"], verbose=settings.get("block_policy_verbose", False))
    if name == "all":
        return PolicyAll(**common)
    if name == "none":
        return PolicyNone(**common)
    if name == "random":
        return PolicyRandom(quantize_number_exec=quantize_number_exec, **common)
    if name == "fixed":
        return PolicyFixed(block_target=settings["block_target"], seed=settings.get("block_seed", 0), **common)
    if name.startswith("rl_"):
        net = build_policy_net_from_settings(settings)
        optimizer = build_policy_optimizer_from_settings(settings, net)
        if name == "rl_semseg":
            ig = InformationGainSemSeg(num_classes=settings["block_num_classes"])
        elif name == "rl_objectdetection":
            ig = InformationGainObjectDetection(num_classes=settings["block_num_classes"])
        else:
            raise AttributeError(f'Policy with name "{name}" not defined!')
        return PolicyTrainRL(block_target=settings["block_target"], cost_momentum=settings["block_cost_momentum"],
                             optimizer=optimizer, complexity_weight=settings["block_complexity_weight"],
                             quantize_number_exec=quantize_number_exec, policy_net=net, information_gain=ig,
                             graph_forward=bool(settings.get("block_graph", 0)), **common)
    raise NotImplementedError(f"Policy {name} not implemented")


def build_policy_optimizer_from_settings(settings: dict, net: PolicyNet) -> torch.optim.Optimizer:
    return torch.optim.RMSprop(net.parameters(), lr=settings["block_optim_lr"], weight_decay=settings["block_optim_wd"],
                               centered=False, momentum=settings["block_optim_momentum"])


class LazyCount:
    """Executed-tile count of a frame whose decision was taken on the DEVICE and not waited for (``PolicyTrainRL.wait_free``): an
    int-like that asks the device the first time the HOST really needs the number -- one event wait, long satisfied by then in the
    normal case (statistics, the running cost before a training step).  ``row`` = the pinned mailbox row the policy-step kernel wrote
    [n_exec, -, NaN flag, -], ``event`` = recorded right behind that launch."""

    lazy = True

    def __init__(self, row: torch.Tensor, event, total: int):
        self._row, self._event, self._value, self.total = row, event, None, int(total)
        self.device_row = row          # [n_exec, -, NaN flag, -] where the kernel left it (device memory in wait-free mode): for device-side arithmetic

    def resolve(self) -> int:
        if self._value is None:
            self._event.synchronize()
            n_exec, _, nan, _ = self._row.tolist()
            assert nan == 0, "Policy net returned NaN's, maybe optimization problem?"
            self._value, self._row, self._event, self.device_row = int(n_exec), None, None, None
        return self._value

    @property
    def resolved(self) -> bool:
        return self._value is not None or self._event.query()

    def __int__(self):
        return self.resolve()

    __index__ = __int__

    def __float__(self):
        return float(self.resolve())

    def __eq__(self, other):
        return self.resolve() == other

    def __hash__(self):
        return id(self)

    def __repr__(self):
        return f"LazyCount({self._value if self._value is not None else 'pending'})"


class LazyFraction:
    """``count / total`` of a LazyCount, float-able (``policy_meta['perc_exec']`` of a frame decided on the device)."""

    lazy = True

    def __init__(self, count: LazyCount):
        self.count = count

    def __float__(self):
        return float(self.count.resolve()) / self.count.total

    def __repr__(self):
        return f"LazyFraction({self.count!r} / {self.count.total})"


class PolicyStats:
    """Running fraction of executed tiles."""

    def __init__(self):
        self.count_images = 0
        self._exec = 0
        self._pending = []      # LazyCounts not yet added to _exec (folded in when somebody reads the statistics)
        self.total = 0

    @property
    def exec(self) -> int:
        if self._pending:
            self._exec += sum(c.resolve() for c in self._pending)
            del self._pending[:]
        return self._exec

    @exec.setter
    def exec(self, value):
        del self._pending[:]
        self._exec = int(value)

    def add_policy_meta(self, policy_meta: dict) -> dict:
        grid = policy_meta["grid"]
        host = policy_meta.get("grid_host", None)
        known = policy_meta.pop("num_exec_known", None)
        num_total = int(grid.numel())
        self.count_images += grid.size(0)
        self.total += num_total
        policy_meta["num_total"] = num_total
        if getattr(known, "lazy", False):
            # decided on the device and not waited for: the host learns the number when it first asks (LazyCount)
            policy_meta["num_exec"] = known
            policy_meta["perc_exec"] = LazyFraction(known)
            self._pending.append(known)
            if len(self._pending) > 32:         # (keep the list short: counts this old resolve without waiting)
                self._exec += sum(c.resolve() for c in self._pending[:-8])
                del self._pending[:-8]
            return policy_meta
        # count known from the device policy step / host mirror available -> no sync; otherwise the one D->H sync of the frame
        num_exec = int(known) if known is not None else (int(host.sum()) if host is not None else int(grid.sum()))
        policy_meta["num_exec"] = num_exec
        policy_meta["perc_exec"] = float(num_exec) / num_total
        self._exec += num_exec
        return policy_meta

    def get_exec_percentage(self):
        return float(self.exec) / self.total

    def __repr__(self) -> str:
        return f"Policy stats: average exec percentage [0 - 1] : {self.get_exec_percentage():0.3f}"


class Policy(torch.nn.Module, metaclass=abc.ABCMeta):
    """Base class.  Subclasses implement ``forward(policy_meta) -> policy_meta``; ``optim`` is a no-op by default."""

    def __init__(self, block_size, verbose=False, quantize_number_exec=0):
        super().__init__()
        self.block_size = block_size
        self.net = None
        self.optimizer = None
        self.verbose = verbose
        self.stats = PolicyStats()
        self.fp16_enabled = False
        self.quantize_number_exec = quantize_number_exec

    def is_trainable(self):
        return self.net is not None

    def grid_shape(self, policy_meta):
        N, C, H, W = policy_meta["inputs"].shape
        assert H % self.block_size == 0, f"input height ({H}) not a multiple of block size {self.block_size}!"
        assert W % self.block_size == 0, f"input width  ({W}) not a multiple of block size {self.block_size}!"
        return (N, 1, H // self.block_size, W // self.block_size)

    @staticmethod
    def publish(policy_meta: dict, grid_host: torch.Tensor) -> dict:
        """Store a host-decided bool grid as ``grid_host`` and its device copy as ``grid``."""
        device = policy_meta["inputs"].device
        grid_host = grid_host.to(torch.bool)
        if device.type == "cuda":
            from blockcopy.backend import pinned_ring

            ring = pinned_ring(grid_host.numel(), torch.bool, True)      # reused pinned staging (no per-frame page-locking)
            pinned = ring.next().view(grid_host.shape).copy_(grid_host)
            policy_meta["grid"] = pinned.to(device, non_blocking=True)
            ring.uploaded()
            policy_meta["grid_host"] = grid_host if not grid_host.is_pinned() else grid_host.clone()   # (callers may keep it: never the ring's buffer)
        else:
            policy_meta["grid"] = grid_host
            policy_meta["grid_host"] = grid_host
        return policy_meta

    def quantize_number_exec_grid(self, grid: torch.Tensor) -> torch.Tensor:
        """Round the number of executed tiles UP to a multiple of ``quantize_number_exec * total`` by switching on
        randomly chosen skipped tiles, so the packed batch size takes few distinct values (one MIOpen solver
        search / one captured graph per value).  Operates in place on a host bool grid (reference :124-144)."""
        if self.quantize_number_exec > 0:
            with timings.env("policy/quantize_number_exec", 3):
                flat = grid.view(-1)
                total = flat.numel()
                idx_not_exec = torch.nonzero(~flat).squeeze(1).tolist()
                num_exec = total - len(idx_not_exec)
                multiple = max(1, int(total * self.quantize_number_exec))
                num_exec_rounded = min(total, multiple * (1 + (num_exec - 1) // multiple))
                idx = random.sample(idx_not_exec, num_exec_rounded - num_exec)
                if idx:
                    flat[idx] = True
        return grid

    @abstractmethod
    def forward(self, policy_meta: dict) -> dict:
        raise NotImplementedError

    def optim(self, policy_meta, train=True, **kwargs):
        return policy_meta


class PolicyAll(Policy):
    """Execute every tile."""

    def forward(self, policy_meta: dict) -> dict:
        self.publish(policy_meta, torch.ones(self.grid_shape(policy_meta), dtype=torch.bool))
        return self.stats.add_policy_meta(policy_meta)


class PolicyNone(Policy):
    """Execute nothing once an ``outputs_prev`` exists (i.e. from the third frame of a clip on, as in the reference :188)."""

    def forward(self, policy_meta: dict) -> dict:
        first = policy_meta.get("outputs_prev", None) is None
        self.publish(policy_meta, torch.full(self.grid_shape(policy_meta), bool(first), dtype=torch.bool))
        return self.stats.add_policy_meta(policy_meta)


class PolicyRandom(Policy):
    """Each tile executed with probability 1/2, count rounded up to the quantisation step (reference :195-216)."""

    def forward(self, policy_meta: dict) -> dict:
        shape = self.grid_shape(policy_meta)
        if policy_meta.get("outputs_prev", None) is None:
            grid = torch.ones(shape, dtype=torch.bool)
        else:
            grid = torch.randn(shape) > 0
        grid = self.quantize_number_exec_grid(grid)
        self.publish(policy_meta, grid)
        return self.stats.add_policy_meta(policy_meta)


class PolicyFixed(Policy):
    """Exactly ``round(block_target * total)`` tiles per frame, chosen by a seeded permutation (frame f of a clip
    uses seed ``seed + f``); the first frame of a clip executes everything.  Reproducible stand-in for `random`
    used by the benchmarks."""

    def __init__(self, block_size, block_target, seed=0, verbose=False):
        super().__init__(block_size, verbose)
        self.block_target = block_target
        self.seed = seed
        self._frame = 0

    def forward(self, policy_meta: dict) -> dict:
        shape = self.grid_shape(policy_meta)
        total = shape[0] * shape[2] * shape[3]
        if policy_meta["outputs"] is None:
            self._frame = 0
            grid = torch.ones(shape, dtype=torch.bool)
        else:
            self._frame += 1
            g = torch.Generator(device="cpu")
            g.manual_seed(self.seed + self._frame)
            grid = torch.zeros(total, dtype=torch.bool)
            grid[torch.randperm(total, generator=g)[:int(round(self.block_target * total))]] = True
            grid = grid.view(shape)
        self.publish(policy_meta, grid)
        return self.stats.add_policy_meta(policy_meta)


class PolicyTrainRL(Policy, metaclass=abc.ABCMeta):
    """REINFORCE policy trained online: reward = information gain + gamma * signed squared deviation of the running
    execution rate from ``block_target`` (reference :219-370)."""

    def __init__(self, block_size: int, block_target: float, optimizer: torch.optim.Optimizer, complexity_weight: float,
                 policy_net: PolicyNet, information_gain: InformationGain, cost_momentum: float = 0.9,
                 at_least_one: bool = False, quantize_number_exec: float = 0, verbose: bool = False,
                 graph_forward: bool = False):
        super().__init__(block_size, verbose, quantize_number_exec)
        # MI355X-first: frames that will not be trained on (train_interval - 1 of every train_interval) need no autograd
        # graph; their policy-net trunk runs as a captured hipGraph over a static input (same logits, ~40 fewer launches)
        self.graph_forward = graph_forward
        self._fwd_graphs = {}
        # (Training frames keep the eager autograd forward / backward.  Round 3 measured both graph forms of the REINFORCE step on C3
        #  -- the whole step recomputed inside one captured graph, and torch.cuda.make_graphed_callables' forward / backward graph
        #  pair -- at 514 and 491 fps against 535 eager: the frame is bound by the GPU time of the fp32 policy net and by the
        #  host's wait for the executed-tile count, not by launch cost, and replaying 100-300 small kernels from a hipGraph is no
        #  cheaper on this runtime than launching them.  DESIGN.md section 6.)
        # MI355X-first: sampling + count quantisation + index tables in ONE device kernel (bc_policy_step) on GPU frames;
        # the host only waits for the executed-tile count (it selects the captured graph).  False = the reference's host route.
        self.device_step = os.environ.get("BLOCKCOPY_DEVICE_POLICY", "1") != "0"
        # MI355X-first, on top of the device step: do not wait for its executed-tile count either.  Needs an engine whose launches read
        # the count from the device (core/graphs.py dynamic mode, block_graph = 2: BlockCopyModel switches this on); the host then
        # learns the counts lazily (LazyCount) -- for the statistics, and for the running cost right before a training step
        self.wait_free = False
        self._pending_use = []        # perc_exec values (LazyFraction) not yet folded into the running cost
        # wait-free mode: the SAME running cost as a float64 scalar on the device (identical IEEE operations in the same order, so the
        # same bits), fed from the count words the policy-step kernel leaves in device memory -- the training step's complexity reward
        # then needs no number from the host at all
        self._rc_dev = None
        self._pending_dev = []        # floats (host-known fractions) / device count rows not yet folded into _rc_dev
        self.rng_seed = None          # counter-based RNG of the device step: drawn from torch's generator at first use
        self.rng_counter = 0
        self._step_bufs = {}
        self._natives = {}            # frame geometry -> own-kernel forward / backward / optimizer of the policy net (policy/native.py) or None
        assert 0 <= block_target <= 1
        self.block_target = block_target
        self.information_gain = information_gain
        self.momentum = cost_momentum
        self._running_cost = None
        self.net = policy_net
        self.complexity_weight_gamma = complexity_weight
        self.optimizer = optimizer
        self.at_least_one = at_least_one

    @property
    def running_cost(self):
        """Exponential average of the executed fraction (reference policy.py:327-330); frames decided on the device without a wait
        are folded in, in order, when the value is read."""
        for use in self._pending_use:
            self._fold(float(use))
        del self._pending_use[:]
        return self._running_cost

    @running_cost.setter
    def running_cost(self, value):
        del self._pending_use[:]
        del self._pending_dev[:]
        self._running_cost = value
        self._rc_dev = None

    def _running_cost_dev(self, device) -> torch.Tensor:
        """The running cost as a float64 0-d tensor on ``device``, pending frames folded in (no host synchronisation)."""
        for use, total in self._pending_dev:
            u = use if not isinstance(use, torch.Tensor) else use[0].to(torch.float64) / float(total)
            if self._rc_dev is None:
                self._rc_dev = torch.as_tensor(u, dtype=torch.float64, device=device).clone()
            self._rc_dev = self._rc_dev * self.momentum + (1 - self.momentum) * u
        del self._pending_dev[:]
        return self._rc_dev

    def _fold(self, block_use: float):
        if self._running_cost is None:
            self._running_cost = block_use
        self._running_cost = self._running_cost * self.momentum + (1 - self.momentum) * block_use

    def forward(self, policy_meta: dict):
        shape = self.grid_shape(policy_meta)
        if policy_meta["outputs"] is None:
            # no temporal history: execute everything
            self.publish(policy_meta, torch.ones(shape, dtype=torch.bool))
        else:
            will_train = policy_meta.get("train_hint", True)
            with torch.enable_grad():
                with timings.env("policy/net", 3):
                    assert self.net.training
                    on_gpu = policy_meta["inputs"].is_cuda
                    native = self._native_for(policy_meta) if on_gpu else None
                    grid_logits = native.forward(policy_meta) if native is not None else None
                    if grid_logits is not None:
                        # own kernels, no autograd graph: optim() runs the backward of THIS forward from the buffers it left (native.py)
                        policy_meta["_native"] = native
                    elif self.graph_forward and on_gpu and not will_train:
                        grid_logits = self._forward_nograd_graph(policy_meta)
                    else:
                        grid_logits = self.net(policy_meta)
                if self.device_step and grid_logits.is_cuda and hasattr(_backend(), "policy_step"):
                    with timings.env("policy/sample", 3):
                        self._device_step(policy_meta, grid_logits, shape)
                    return self.stats.add_policy_meta(policy_meta)
                with timings.env("policy/sample", 3):
                    m = Bernoulli(logits=grid_logits)
                    sample = m.sample()
                    # the single D->H transfer of the frame: sampled grid (+ a NaN flag for the logits)
                    packed = torch.cat([sample.reshape(-1), torch.isnan(grid_logits).any().reshape(1).to(sample.dtype)]).cpu()
                    assert packed[-1] == 0, "Policy net returned NaN's, maybe optimization problem?"
                    grid_host = packed[:-1].reshape(shape) > 0
                if self.at_least_one and not grid_host.any():
                    grid_host[0, 0, 0, 0] = True
                grid_host = self.quantize_number_exec_grid(grid_host)
                self.publish(policy_meta, grid_host)
                grid_f = policy_meta["grid"].to(grid_logits.dtype)
                policy_meta["grid_log_probs"] = m.log_prob(grid_f) if grid_logits.requires_grad or not self.graph_forward else None
                policy_meta["grid_probs"] = m.probs
        return self.stats.add_policy_meta(policy_meta)

    def _native_for(self, policy_meta: dict):
        """blockcopy.policy.native.NativePolicyNet for this frame geometry (built once per geometry; None where it does not apply)."""
        frame = policy_meta["inputs"]
        key = (tuple(frame.shape), frame.device)
        if key not in self._natives:
            from blockcopy.policy import native

            self._natives[key] = native.try_build(self.net, self.optimizer, frame.shape, frame.device) if frame.dim() == 4 else None
        return self._natives[key]

    def _native_step(self, native, policy_meta: dict):
        """REINFORCE update on own kernels (information gain, seed, backward, RMSprop, parameter export: native.NativePolicyNet.step)."""
        grid = policy_meta["grid"]
        if self.wait_free and (self._rc_dev is not None or self._pending_dev):
            cost = self._running_cost_dev(grid.device)
        else:
            cost = float(self.running_cost)
        out, prev = policy_meta["outputs"], policy_meta["outputs_prev"]
        ig = None
        direct = (isinstance(self.information_gain, InformationGainSemSeg) and torch.is_tensor(out) and torch.is_tensor(prev) and out.is_cuda and out.dim() == 4
                  and out.shape == prev.shape and out.dtype == prev.dtype and out.stride() == prev.stride()
                  and out.dtype in (torch.float32, torch.float16, torch.bfloat16))
        if not direct:
            with torch.no_grad():
                ig = self._get_information_gain(policy_meta)
        ig, loss = native.step(grid, out, prev, cost, self.block_target, self.complexity_weight_gamma,
                               scale_factor=getattr(self.information_gain, "scale_factor", 0.25), ig=ig)
        policy_meta["information_gain"] = ig
        policy_meta["loss_policy"] = loss

    MAILBOX_ROWS = 64      # pinned rows the device step reports its counts into, one per frame, reused round robin
    GRID_RING = 8          # device buffers the wait-free step writes its grid into, round robin (a frame's grid lives eight frames)

    def _device_step(self, policy_meta: dict, grid_logits: torch.Tensor, shape):
        """Decision on the device: one launch samples, rounds the executed count up to the quantisation step and builds the
        index tables; an async copy mirrors the grid into pinned memory; ONE event wait then gives the host the executed-tile
        count (mailbox) and the mirror.  Distribution-identical to the host route (reference policy.py:283-288 + :124-144);
        bit-identical to the CPU restatement oracle/bc_oracle.c bco_policy_step for the same (seed, counter)."""
        be = _backend()
        dev = grid_logits.device
        n_total = grid_logits.numel()
        if self.rng_seed is None:
            self.rng_seed = int(torch.randint(0, 2 ** 62, (1,)).item())
        st = self._step_bufs.get((dev, n_total))
        if st is None:
            # tables = [grid_idx | mapping_exec | counts(4)]: the layout the graph's table buffer has, so the engine takes it in ONE copy
            tables = torch.zeros(2 * n_total + 4, dtype=torch.int32, device=dev)
            st = self._step_bufs[(dev, n_total)] = {
                "grid": torch.zeros(n_total, dtype=torch.uint8, device=dev), "tables": tables, "counts": tables[2 * n_total:],
                "mailbox": torch.zeros((self.MAILBOX_ROWS, 4), dtype=torch.int32).pin_memory(), "owners": [None] * self.MAILBOX_ROWS,
                # wait-free mode: the rows live in DEVICE memory (the running cost is folded from them by device arithmetic, and the host
                # only reads a row when it asks for a LazyCount)
                "mailbox_dev": torch.zeros((self.MAILBOX_ROWS, 4), dtype=torch.int32, device=dev),
                "host": torch.zeros(n_total, dtype=torch.uint8).pin_memory(), "event": torch.cuda.Event()}
        multiple = max(1, int(n_total * self.quantize_number_exec)) if self.quantize_number_exec > 0 else 1
        logits = grid_logits.detach().float().contiguous()
        row_k = self.rng_counter % self.MAILBOX_ROWS
        if st["owners"][row_k] is not None:
            st["owners"][row_k].resolve()      # (MAILBOX_ROWS frames old: answered long ago) the row is about to be rewritten
        row = (st["mailbox_dev"] if self.wait_free else st["mailbox"])[row_k]
        if self.wait_free:
            # nobody waits: the engine's launches read the count from the device (dynamic graph), the host asks lazily.  No copy either: the
            # tables are written straight into the captured frame's table buffer when the engine has named one (``tables_target``, same layout;
            # the frame's graph is behind this launch in the stream), and the grid into one of GRID_RING buffers (it is read by this frame's
            # bookkeeping and the next frame's policy input -- eight frames of life instead of a clone per frame)
            tables = policy_meta.pop("tables_target", None)
            if tables is None or tables.numel() != 2 * n_total + 4 or tables.device != dev or tables.dtype != torch.int32:
                tables = st["tables"]
            if "grid_ring" not in st:
                st["grid_ring"] = [torch.zeros(n_total, dtype=torch.uint8, device=dev) for _ in range(self.GRID_RING)]
            ring = st["grid_ring"]
            grid_u8 = ring[self.rng_counter % self.GRID_RING]
            be.policy_step(logits, self.rng_seed, self.rng_counter, multiple, self.at_least_one, grid_u8, tables[:2 * n_total], tables[2 * n_total:], row)
            self.rng_counter += 1
            ev = torch.cuda.Event()
            ev.record()
            lazy = st["owners"][row_k] = LazyCount(row, ev, n_total)
            policy_meta["grid"] = grid_u8.view(torch.bool).view(shape)
            policy_meta.pop("grid_host", None)
            policy_meta["grid_tables"] = (tables, None)
            policy_meta["num_exec_known"] = lazy
        else:
            policy_meta.pop("tables_target", None)
            be.policy_step(logits, self.rng_seed, self.rng_counter, multiple, self.at_least_one, st["grid"], st["tables"][:2 * n_total], st["counts"], row)
            self.rng_counter += 1
            st["owners"][row_k] = None
            st["host"].copy_(st["grid"], non_blocking=True)
            st["event"].record()
            st["event"].synchronize()          # the frame's single wait: n_exec selects the captured graph
            n_exec, _, nan, _ = row.tolist()
            assert nan == 0, "Policy net returned NaN's, maybe optimization problem?"
            grid = st["grid"].view(torch.bool).view(shape).clone()
            policy_meta["grid"] = grid
            policy_meta["grid_host"] = st["host"].view(torch.bool).view(shape).clone()
            policy_meta["grid_tables"] = (st["tables"], n_exec)      # consumed by the engine instead of rebuilding them on the host
            policy_meta["num_exec_known"] = n_exec
        # probabilities / log-probabilities of the decision are bookkeeping for optim(): they are derived THERE, i.e. their small
        # launches queue behind the frame's block pipeline instead of in front of it (the host is on the critical path right here:
        # the GPU has nothing to do until the frame's graph is launched)
        policy_meta["_decision_logits"] = grid_logits

    @torch.no_grad()
    def _forward_nograd_graph(self, policy_meta: dict) -> torch.Tensor:
        """Policy logits for a frame that will not be trained on: feature build (a few small ops) eagerly, the
        resnet8 trunk + head as a captured graph over a static input.  BatchNorm stays in training mode (batch
        statistics, running-stat updates), exactly as in the autograd path."""
        x = self.net.build_features(policy_meta)
        key = (tuple(x.shape), x.dtype, x.device)
        st = self._fwd_graphs.get(key)
        if st is None:
            st = self._fwd_graphs[key] = {"x": torch.empty_like(x), "graph": None, "logits": None, "warm": 0}
        if st["graph"] is None and st["warm"] < 1:
            st["warm"] += 1
            return self.net.layers(self.net.backbone(x))      # first encounter: eager (solver search, lazy module loads)
        st["x"].copy_(x)
        if st["graph"] is None:
            torch.cuda.synchronize(x.device)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                st["logits"] = self.net.layers(self.net.backbone(st["x"]))
            st["graph"] = g
        st["graph"].replay()
        return st["logits"]

    def _get_information_gain(self, policy_meta: dict) -> torch.Tensor:
        with timings.env("policy/information_gain", 3):
            ig = self.information_gain(policy_meta)
            assert ig.dim() == 4
            return ig

    def _get_reward_complexity(self, policy_meta: dict):
        if self.wait_free and (self._rc_dev is not None or self._pending_dev):
            r = -(self._running_cost_dev(policy_meta["grid"].device) - self.block_target)       # float64 on the device: same operations, same bits
            return r * r.abs()
        r = -float(self.running_cost - self.block_target)
        return r * abs(r)

    def optim(self, policy_meta: dict, train=True) -> dict:
        logits = policy_meta.pop("_decision_logits", None)
        native = policy_meta.pop("_native", None)
        if logits is not None and native is not None:
            # (no autograd graph on the native route: the backward runs from the forward's own buffers; one launch for both maps)
            policy_meta["grid_probs"], policy_meta["grid_log_probs"] = native.decision_probs(policy_meta["grid"])
        elif logits is not None:        # (device step: see _device_step)
            with torch.enable_grad():   # (the log-probabilities of a frame that will be trained on carry the policy net's autograd graph)
                # Bernoulli(logits).probs / .log_prob(grid) written out (torch/distributions/bernoulli.py: sigmoid; minus the binary cross
                # entropy with logits): the distribution object VALIDATES its arguments with `.all()` -- a device synchronisation in the
                # constructor and another one in log_prob, i.e. the host would wait here for the whole frame it has just enqueued
                want_lp = logits.requires_grad or not self.graph_forward
                policy_meta["grid_log_probs"] = (-F.binary_cross_entropy_with_logits(logits, policy_meta["grid"].to(logits.dtype), reduction="none")
                                                 if want_lp else None)
                policy_meta["grid_probs"] = torch.sigmoid(logits)
        policy_meta["output_repr"] = self.information_gain.get_output_repr(policy_meta)
        grid = policy_meta["grid"]
        assert grid.dim() == 4
        block_use = policy_meta["perc_exec"]
        if self.wait_free:
            if self._rc_dev is None and not self._pending_dev and (self._running_cost is not None or self._pending_use):
                # (entering the wait-free mode with a history: the device copy starts from the host value -- one read, once)
                self._rc_dev = torch.as_tensor(self.running_cost, dtype=torch.float64, device=grid.device)
            lazy = getattr(block_use, "lazy", False)
            self._pending_use.append(block_use if lazy else float(block_use))       # folded in, in order, whenever the HOST reads the running cost
            self._pending_dev.append((block_use.count.device_row, block_use.count.total) if lazy else (float(block_use), 1))
        else:
            _ = self.running_cost                     # fold what is pending first: the average depends on the order
            self._fold(float(block_use))
            self._rc_dev = None
            del self._pending_dev[:]

        if policy_meta["outputs_prev"] is not None and train and native is not None:
            self._native_step(native, policy_meta)
        elif policy_meta["outputs_prev"] is not None and train:
            with torch.enable_grad():
                ig = self._get_information_gain(policy_meta)
                policy_meta["information_gain"] = ig
                reward_complexity_weighted = self._get_reward_complexity(policy_meta) * self.complexity_weight_gamma
                reward = ig + reward_complexity_weighted
                assert reward.dim() == 4
                log_probs = policy_meta["grid_log_probs"]
                reward = F.adaptive_max_pool2d(reward, output_size=log_probs.shape[2:])
                reward = torch.where(grid, reward, -reward)   # skipped tiles are rewarded for LOW gain
                loss_policy = (-log_probs * reward.detach()).mean()
                with timings.env("policy/optimizer_backward", 3):
                    loss_policy.backward()
                with timings.env("policy/optimizer_step", 3):
                    self.optimizer.step()
                    self.optimizer.zero_grad(set_to_none=True)
                policy_meta["loss_policy"] = loss_policy.detach()

                if self.verbose:
                    assert not torch.isnan(loss_policy)
                    probs = policy_meta["grid_probs"]
                    exec_mean, skip_mean = probs[grid].mean(), probs[~grid].mean()
                    print(f"BLOCKS/running_cost: {self.running_cost: 0.3f} \nBLOCKS/block_use: {block_use:0.3f} \n"
                          f"BLOCKS/information_gain_max: {ig.max()} \nBLOCKS/information_gain_min: {ig.min()} \n"
                          f"BLOCKS/reward_complexity_weighted: {reward_complexity_weighted} \n"
                          f"BLOCKS/avg_prob_exec: {exec_mean:0.3f} \nBLOCKS/avg_prob_skip: {skip_mean:0.3f} \n")
                    print(self.stats)
                    if self.stats.count_images > 300 and exec_mean - skip_mean < 0.3:
                        print("Warning: Block execution policy seems not well trained yet.")
        return policy_meta
