"""DreamerEngine -- host-side schedule of the MI355X-native Dreamer training step.

This is plumbing around the C ABI (``include/bigdreamer_hip.h``): it owns the flat fp32 parameter /
gradient / Adam buffers (one per optimiser, reference order ``src/dreamer.py:160-165``), the packed
weight copies the kernels read, the saved-activation workspaces, and issues the kernels of one
``Dreamer.train_step`` (``src/dreamer.py:253-393``) in dependency order on the current stream.  All
arithmetic happens in the HIP library; torch supplies device memory, streams, the noise generator
and (multi-GPU) ``torch.distributed`` all-reduces over RCCL.

No autograd graph is built: the backward schedule is explicit (the reference's three ``backward()`` calls
become observe/imagine/MLP backward kernels + weight-gradient GEMMs).
"""
from __future__ import annotations

import collections.abc
import ctypes as C
import math
import os
import weakref
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch

from . import _cabi as cabi
from .parallel import DataParallel
from .synth import DENSE_LAYERS, MODEL_MODULES, Dims, model_modules, param_shapes

lib = cabi.lib
ptr = cabi.ptr

DEFAULT_HP = dict(
    kl_balance=0.8, kl_loss_weight=0.1, free_nats=3.0, grad_clip_norm=100.0, discount=0.995, disclam=0.95,
    model_learning_rate=2e-4, actor_learning_rate=4e-5, value_learning_rate=1e-4, adam_epsilon=1e-5,
    weight_decay=1e-6, entropy_weight=1e-5, polyak_avg=1.0, min_std_dev=0.1, discount_weight=5.0,
)

# actor constants (src/models.py:479-503)
ACT_RAW_INIT_STD = float(torch.log(torch.exp(torch.tensor(5.0)) - 1))
ACT_MIN_STD = 1e-4
ACT_MEAN_SCALE = 5.0

# scalar-board slots (raw sums; see include/bigdreamer_hip.h "losses")
SLOT_OBS, SLOT_REW, SLOT_KL, SLOT_RET, SLOT_ENT, SLOT_VAL, SLOT_GN_MODEL, SLOT_GN_ACTOR, SLOT_GN_CRITIC = range(9)
SLOT_DISC, SLOT_WOBJ = 9, 10       # use_discount=True: Bernoulli loss sum (model phase); weighted actor objective sum
N_SLOTS = 16


class ParamGroup:
    """Flat parameter / gradient / Adam-moment buffers of one optimiser, with named views.

    With `conv_storage`, 4-D (convolution) tensors are STORED permuted (d0, d2, d3, d1) -- Conv2d (co, ky, kx, ci),
    ConvTranspose2d (ci, ky, kx, co): the layout the hand-written conv kernels use as plain [N][K] matrices
    (csrc/conv.hip) -- while `p` / `g` stay views with the reference's logical shape (state_dict, tests); `ps` / `gs`
    are the contiguous storage views.  Clip, Adam and the all-reduce act on the flat buffers and do not care."""

    def __init__(self, specs: List[Tuple[str, str, Tuple[int, ...]]], device, with_opt: bool = True,
                 conv_storage: bool = False):
        self.specs = specs
        # every tensor starts on a 16-byte boundary of the flat buffers (pad floats stay zero: zero gradient, zero Adam
        # update): 16-byte vector loads and LDS-DMA of a weight matrix as it lies in the buffer (csrc/gemm.hip)
        al = lambda k: (k + 3) & ~3
        n = sum(al(int(np.prod(s))) for _, _, s in specs)
        self.numel = n
        self.flat = torch.zeros(n, dtype=torch.float32, device=device)
        self.grad = torch.zeros(n, dtype=torch.float32, device=device) if with_opt else None
        self.m = torch.zeros(n, dtype=torch.float32, device=device) if with_opt else None
        self.v = torch.zeros(n, dtype=torch.float32, device=device) if with_opt else None
        self.step = 0
        self.p: Dict[Tuple[str, str], torch.Tensor] = {}
        self.g: Dict[Tuple[str, str], torch.Tensor] = {}
        self.ps: Dict[Tuple[str, str], torch.Tensor] = {}
        self.gs: Dict[Tuple[str, str], torch.Tensor] = {}
        self._layout: Dict[Tuple[str, str], tuple] = {}
        off = 0
        for mod, name, shape in specs:
            k = int(np.prod(shape))
            perm = conv_storage and len(shape) == 4
            sshape = (shape[0], shape[2], shape[3], shape[1]) if perm else shape
            self._layout[(mod, name)] = (off, k, sshape, perm)

            def views(buf):
                st = buf[off:off + k].view(sshape)
                return st, (st.permute(0, 3, 1, 2) if perm else st)

            self.ps[(mod, name)], self.p[(mod, name)] = views(self.flat)
            if with_opt:
                self.gs[(mod, name)], self.g[(mod, name)] = views(self.grad)
            off += al(k)

    def logical(self, buf: torch.Tensor, mod: str, name: str) -> torch.Tensor:
        """View of `buf` (a flat buffer laid out like `flat`: grad, m, v) with the reference's shape of (mod, name)."""
        off, k, sshape, perm = self._layout[(mod, name)]
        st = buf[off:off + k].view(sshape)
        return st.permute(0, 3, 1, 2) if perm else st


def _adam_state_dict(g: ParamGroup, lr: float, hp: dict, over: Optional[dict] = None) -> dict:
    """The group's optimiser state in torch.optim.Adam's state_dict layout (parameter index = reference parameter
    order, moments in the reference's logical shapes), so that the reference's ``model_optimizer.load_state_dict``
    accepts it (src/planet.py:114)."""
    state = {}
    for i, (mod, name, _shape) in enumerate(g.specs):
        state[i] = {"step": torch.tensor(float(g.step)), "exp_avg": g.logical(g.m, mod, name).detach().cpu().clone().contiguous(),
                    "exp_avg_sq": g.logical(g.v, mod, name).detach().cpu().clone().contiguous()}
    over = over or {}
    group = {"lr": over.get("lr", lr), "betas": (0.9, 0.999), "eps": over.get("eps", hp["adam_epsilon"]),
             "weight_decay": over.get("weight_decay", hp["weight_decay"]), "amsgrad": False,
             "maximize": False, "foreach": None, "capturable": False, "differentiable": False, "fused": None,
             "params": list(range(len(g.specs)))}
    return {"state": state, "param_groups": [group]}


def _load_adam_state_dict(g: ParamGroup, sd: dict) -> Optional[dict]:
    """Restore moments + step; returns the checkpoint's param_group hyper-parameters (lr, eps, weight_decay) -- torch's
    ``Optimizer.load_state_dict`` adopts them (src/planet.py:114), so the caller does too."""
    pg = (sd.get("param_groups") or [None])[0]
    hyper = {k: float(pg[k]) for k in ("lr", "eps", "weight_decay") if pg and k in pg} or None
    if pg and tuple(pg.get("betas", (0.9, 0.999))) != (0.9, 0.999):
        raise NotImplementedError(f"Adam betas {pg['betas']} in the checkpoint: the kernels implement (0.9, 0.999), the "
                                  "reference's only setting (src/dreamer.py:56-67)")
    st = sd["state"]
    if not st:                      # a freshly built optimiser: nothing to restore
        g.m.zero_(); g.v.zero_(); g.step = 0
        return hyper
    assert len(st) == len(g.specs), f"optimizer state has {len(st)} parameters, this group {len(g.specs)}"
    steps = set()
    for i, (mod, name, shape) in enumerate(g.specs):
        e = st[i] if i in st else st[str(i)]
        assert tuple(e["exp_avg"].shape) == tuple(shape), (mod, name, tuple(e["exp_avg"].shape), shape)
        g.logical(g.m, mod, name).copy_(e["exp_avg"].to(torch.float32))
        g.logical(g.v, mod, name).copy_(e["exp_avg_sq"].to(torch.float32))
        steps.add(int(float(e["step"])))
    assert len(steps) == 1, f"per-parameter Adam step counts differ: {steps}"
    g.step = steps.pop()
    return hyper


class _LogRecord:
    """Pinned host copy of the scalar board of ONE train step, filled by three asynchronous D2H copies (one per
    optimiser phase, each on the stream that wrote the slots) -- reading it never stalls the pipeline streams."""

    ROW = {"model": 0, "actor": 1, "critic": 2}

    def __init__(self):
        self.host = torch.zeros(3, N_SLOTS + 1, dtype=torch.float32).pin_memory()     # last column: cluster error word
        self.events: List[Optional[torch.cuda.Event]] = [None, None, None]
        self.counts: Optional[dict] = None
        self.owner = None          # weakref to the LazyLogs handed to the caller
        self.need = (0, 1, 2)


class LazyLogs(collections.abc.MutableMapping):
    """The reference's log dict (src/dreamer.py:293-296,359-360,383), resolved on first read.

    ``Dreamer.train_step`` returns one per step; nothing synchronises until a key is read, so the reference loop's
    ``for _ in range(collect_interval): logs = model.train_step()`` (src/main.py:105-108) runs the cross-step pipeline
    and only the burst's last dict -- the one the loop reads -- waits for the GPU.  Keys and values are exactly those
    of the eager dict; extra keys may be written (``logs["weight_update_per_sec"] = ...``, src/main.py:108)."""

    def __init__(self, eng: "DreamerEngine", rec: _LogRecord, drop_prefix: Tuple[str, ...] = ()):
        self._eng, self._rec, self._drop = eng, rec, drop_prefix
        self._vals: Optional[Dict[str, float]] = None
        self._extra: Dict[str, object] = {}

    def resolve(self) -> Dict[str, float]:
        if self._vals is None:
            vals = self._eng._resolve_record(self._rec)
            self._vals = {k: v for k, v in vals.items() if not k.startswith(self._drop)} if self._drop else vals
            self._rec = None
        return self._vals

    def __getitem__(self, k):
        return self._extra[k] if k in self._extra else self.resolve()[k]

    def __setitem__(self, k, v):
        self._extra[k] = v

    def __delitem__(self, k):
        del self._extra[k]

    def __iter__(self):
        yield from self.resolve()
        yield from (k for k in self._extra if k not in self._vals)

    def __len__(self):
        return len(set(self.resolve()) | set(self._extra))

    def __repr__(self):
        return repr(dict(self.items()))


class WgradBatch:
    """All weight-gradient GEMMs of one backward pass -> one bd_wgrad_grouped launch (+ one grouped reduce).
    The descriptor table lives in HBM and is rebuilt only when a buffer pointer or size changes."""

    def __init__(self, eng: "DreamerEngine", name: str, ws_attr: str = "_wgrad_ws"):
        self.eng, self.name, self.ws_attr = eng, name, ws_attr
        self.items: List[tuple] = []
        self._cache: Dict[tuple, tuple] = {}

    def add(self, dpre, ldp, act, lda, M, N, K, dW, ldw, db=None, act2=None, lda2=0, M1=None, gather=None) -> None:
        """gather = (nseg, seglen, gh, gw, IH, IW, C): `act` is an NHWC image and row m takes its stride-2 window
        (conv weight gradients, include/bigdreamer_hip.h)."""
        self.items.append((ptr(dpre), ldp, ptr(act), lda, M if M1 is None else M1, ptr(act2), lda2, M, N, K, ptr(dW), ldw,
                           ptr(db), tuple(gather) if gather else (0, 0, 0, 0, 0, 0, 0)))

    def run(self) -> None:
        eng = self.eng
        if not self.items:
            return
        key = tuple(self.items)
        hit = self._cache.get(key)
        if hit is None:
            n = len(self.items)
            descs = (cabi.WgradDesc * n)()
            for i, it in enumerate(self.items):
                (descs[i].dpre, descs[i].ldp, descs[i].act1, descs[i].lda1, descs[i].M1, descs[i].act2, descs[i].lda2,
                 descs[i].M, descs[i].N, descs[i].K, descs[i].dW, descs[i].ldw, descs[i].db) = it[:13]
                (descs[i].g_nseg, descs[i].g_seglen, descs[i].g_gh, descs[i].g_gw, descs[i].g_IH, descs[i].g_IW,
                 descs[i].g_C) = it[13]
            tb, tr, wsf = C.c_int(0), C.c_int(0), C.c_size_t(0)
            cabi.check(lib.bd_wgrad_plan(descs, n, C.byref(tb), C.byref(tr), C.byref(wsf)))
            # (a pageable H2D copy synchronises with the device: tables are cached per operand-pointer set -- the
            # double-buffered features and the replay's output ring make a handful of sets that then repeat)
            table = torch.frombuffer(bytearray(bytes(descs)), dtype=torch.uint8).to(eng.dev)
            if getattr(eng, self.ws_attr).numel() < wsf.value:
                setattr(eng, self.ws_attr, torch.zeros(wsf.value, dtype=torch.float32, device=eng.dev))
            if len(self._cache) >= 16:
                self._cache.pop(next(iter(self._cache)))
            hit = self._cache[key] = (table, n, tb.value, tr.value)
        table, n, tb, tr = hit
        ws = ptr(getattr(eng, self.ws_attr))
        if eng._timers_on:       # bracket the slab-GEMM kernel alone (one launch) with HIP events; the reduce follows
            with eng.span("wgrad_gemm_" + self.name):
                cabi.check(lib.bd_wgrad_grouped_phase(table.data_ptr(), n, tb, tr, ws, 1, cabi.stream()))
            cabi.check(lib.bd_wgrad_grouped_phase(table.data_ptr(), n, tb, tr, ws, 2, cabi.stream()))
        else:
            cabi.check(lib.bd_wgrad_grouped(table.data_ptr(), n, tb, tr, ws, cabi.stream()))
        self.items = []


# Pipeline streams are per PROCESS and device, not per engine.  torch hands out streams from a fixed pool and HIP maps them
# onto a handful of hardware queues in creation order: a second engine built in the same process (Planet then Dreamer, an
# evaluation agent beside a training agent, the legs of a benchmark) used to get the NEXT pool streams, whose queue
# assignment made two of its three hot streams share a hardware queue -- the same kernels ran 10-15 % slower (round 2:
# "cause not found").  Engines now share one stream per (device, role, priority); engines used alternately are merely
# ordered against each other on them, which is what a single-threaded caller does anyway.  BD_SHARE_STREAMS=0: one set per
# engine (diagnosis).
_STREAM_CACHE: Dict[tuple, "torch.cuda.Stream"] = {}


def _engine_stream(dev: torch.device, role: str, priority: int) -> "torch.cuda.Stream":
    if os.environ.get("BD_SHARE_STREAMS", "1") == "0":
        return torch.cuda.Stream(device=dev, priority=priority)
    key = (dev.index if dev.index is not None else torch.cuda.current_device(), role, priority)
    st = _STREAM_CACHE.get(key)
    if st is None:
        st = _STREAM_CACHE[key] = torch.cuda.Stream(device=dev, priority=priority)
    return st


class DreamerEngine:
    def __init__(self, dims: Dims, hp: Optional[dict] = None, device="cuda", params: Optional[dict] = None,
                 world_size: int = 1, process_group=None, phase_groups: Optional[dict] = None):
        self.d = dims
        self.pixel = bool(dims.pixel)   # 64x64 pixel observations: conv encoder / decoder on csrc/conv.hip (conv_stack.py)
        self.hp = dict(DEFAULT_HP)
        if hp:
            self.hp.update({k: v for k, v in hp.items() if k in self.hp})
        self.dev = torch.device(device)
        assert self.dev.type == "cuda", "the HIP path needs a GPU (there is no CPU fallback)"
        self.world_size = world_size
        self.pg = process_group
        # One RCCL communicator per optimiser (+ the caller's for the KL scalar) unless the caller brought its own.  All
        # collectives of ONE communicator run in issue order on its internal stream, so the critic's all-reduce of step k -- off
        # the critical path, at the end of a low-priority phase -- sat in front of the KL / world-model collectives of step k + 1
        # whenever the critic phase ran late, and dynamics learning stalled behind it: the one-rank rehearsal read 4.3 or
        # 3.6 ms/step instead of 2.9 depending on how HIP had mapped the streams to hardware queues (tools/r03_dp_rehearsal.sh).
        # Issuing the actor / critic updates one host step late (below) fixes the ORDER of issue, separate communicators remove
        # the coupling.  Collective: every rank builds its engine at the same point.  BD_PHASE_GROUPS=0: one communicator.
        rehearsal = os.environ.get("BD_FORCE_DP", "0") == "1" and torch.distributed.is_initialized()
        if (phase_groups is None and (world_size > 1 or rehearsal) and os.environ.get("BD_PHASE_GROUPS", "1") != "0"
                and torch.distributed.get_backend(process_group) == "nccl"):
            phase_groups = DataParallel.make_phase_groups("nccl", process_group)
        self.dp = DataParallel(world_size, torch.distributed.get_rank(process_group) if world_size > 1 else 0,
                               process_group, phase_groups)
        if (world_size > 1 or rehearsal) and torch.distributed.get_backend(process_group) == "nccl":
            # A process group created without `device_id` builds its communicators (and their internal streams) at the first
            # collective -- which would be in the middle of the first train step, AFTER the engine's streams exist; created
            # in that order the rehearsal ran 3.64 instead of 2.86 ms/step once the process had more than four hardware
            # queues.  One tiny all-reduce per communicator here puts them first in every case.
            self.dp.force = self.dp.force or rehearsal
            tiny = torch.zeros(4, dtype=torch.float32, device=self.dev)
            for g in (None, "model", "actor", "critic"):
                self.dp.allreduce_sum_(tiny, g)
            torch.cuda.synchronize(self.dev)
        d = dims
        shapes = param_shapes(d)
        # pixel mode: the conv stacks run on this library's gather-GEMM kernels (csrc/conv.hip, conv_stack.py) and their
        # weights are STORED (d0, ky, kx, d1) (ParamGroup).  (The MIOpen / torch-autograd comparator of rounds 1-2 is a test
        # helper now: tests/torch_conv_stacks.py injects it through the same `self.conv` interface.)
        self.groups = {
            "model": ParamGroup([(m, n, s) for m in model_modules(d) for n, s in shapes[m]], self.dev,
                                conv_storage=self.pixel),
            "actor": ParamGroup([("actor", n, s) for n, s in shapes["actor"]], self.dev),
            "critic": ParamGroup([("critic", n, s) for n, s in shapes["critic"]], self.dev),
            "critic_target": ParamGroup([("critic_target", n, s) for n, s in shapes["critic"]], self.dev, False),
        }
        self._mod_group = {m: "model" for m in model_modules(d)}
        self._mod_group.update(actor="actor", critic="critic", critic_target="critic_target")
        if params is not None:
            self.load_params(params)
        self.scalars = torch.zeros(N_SLOTS, dtype=torch.float32, device=self.dev)
        self.red_ws = torch.zeros(int(lib.bd_reduce_ws_floats()), dtype=torch.float32, device=self.dev)
        self._wgrad_ws = torch.zeros(1, dtype=torch.float32, device=self.dev)
        self._wgrad_ws_side = torch.zeros(1, dtype=torch.float32, device=self.dev)
        self._wgrad_ws_early = torch.zeros(1, dtype=torch.float32, device=self.dev)
        self.red_ws_side = torch.zeros(int(lib.bd_reduce_ws_floats()), dtype=torch.float32, device=self.dev)
        self._wbatch = {"model": WgradBatch(self, "model"), "model_early": WgradBatch(self, "model_early", "_wgrad_ws_early"),
                        "actor": WgradBatch(self, "actor", "_wgrad_ws_bh"),
                        "critic": WgradBatch(self, "critic", "_wgrad_ws_side")}
        # The critic update only needs the imagined features and the lambda-returns, and the actor's backward pass uses
        # the critic TARGET: the two are independent, so the critic phase runs on a second HIP stream underneath the
        # latency-bound imagination backward.  Measured on MI355X at batch=50: no gain (6.92 vs 6.85 ms/step; the
        # head-chain and wgrad kernels already fill the chip and slow imagine_bwd down by contention), so it is off
        # by default (BD_OVERLAP_CRITIC=1 enables it; parity-tested either way).
        self.overlap_critic = os.environ.get("BD_OVERLAP_CRITIC", "0") == "1"
        self._side = _engine_stream(self.dev, "side", int(os.environ.get("BD_SIDE_PRIO", "0")))
        self._s_heads = _engine_stream(self.dev, "heads", -1)     # heads of the first half of a split rollout
        self.img_split = os.environ.get("BD_IMG_SPLIT", "0") == "1"
        self._img_split_rows = 0
        # pixel mode: decoder weight gradients under the observe scan
        self._s_early = _engine_stream(self.dev, "early", int(os.environ.get("BD_EARLY_PRIO", "0")))
        # Cross-step software pipeline (on unless BD_PIPELINE=0).  Dynamics learning of step k+1 reads only the world
        # model that step k's model optimiser wrote, never what step k's behaviour learning (imagination, actor,
        # critic) produces, while behaviour learning k needs the world model k and the posteriors k.  So the two
        # halves run on two HIP streams: the latency-bound observe scan (52 CUs) of step k+1 runs underneath the
        # imagination of step k (154 CUs) instead of leaving 200 CUs idle.  Same kernels, same operands, same order of
        # every read-after-write: results are bit-identical to the serial schedule (parity-tested both ways).
        #   s_wm:  [wait BH(k-1) done] encoder, observe fwd/bwd, heads, wgrad  [wait BH(k) done with the world model]
        #          clip+Adam+pack -> ev_wm_done
        #   s_bh:  [wait ev_wm_done] imagine fwd, heads, lambda-return -> ev_ret, imagine bwd -> ev_bh_wm_free; actor update
        #          -> ev_bh_done
        #   _side: [wait ev_ret] critic fwd/bwd, wgrad, clip+Adam+pack -> ev_cr_done  (under the next step's imagination)
        # `feat` (posterior features, read by behaviour learning), the imagined features and the lambda-returns (read by
        # the critic update) are double-buffered by step parity.
        self.pipeline = os.environ.get("BD_PIPELINE", "1") != "0"
        self._s_wm = _engine_stream(self.dev, "wm", int(os.environ.get("BD_WM_PRIO", "-1")))   # the scan is latency-bound: dispatch it first
        # state observations: the actor chain bounds the step; pixels: the conv-heavy dynamics chain does, and behaviour
        # learning should only fill its gaps (BD_BH_PRIO overrides: -1 high, 0 normal)
        bh_prio = int(os.environ.get("BD_BH_PRIO", "0" if self.pixel else "-1"))
        self._s_bh = _engine_stream(self.dev, "bh", bh_prio)      # critic: _side
        self._ev_bh_wm_free: Optional[torch.cuda.Event] = None
        self._ev_bh_done: List[Optional[torch.cuda.Event]] = [None, None]
        self._ev_cr_done: List[Optional[torch.cuda.Event]] = [None, None]
        # data-parallel: actor / critic optimiser steps (their all-reduces) are issued one host step late, see
        # _optimizer_step_or_defer; BD_DEFER_OPT=1 forces the same order on one GPU (tests)
        # (needed only where the three optimisers share ONE communicator: with one communicator each -- the default, above --
        # the all-reduces cannot queue behind one another and the updates are issued where the serial schedule has them; held
        # back, the actor's update of step k would wait for the host to queue all of dynamics learning k+1 first: +2 ms per
        # step in the rehearsal of the pixel configuration)
        shared_comm = not all(g in self.dp.groups for g in ("model", "actor", "critic"))
        self.defer_opt = (world_size > 1 and shared_comm) or os.environ.get("BD_DEFER_OPT", "0") == "1"
        if os.environ.get("BD_FORCE_DP", "0") == "1" and torch.distributed.is_initialized():
            self.dp.force = True                       # single-rank rehearsal of the data-parallel schedule
            self.defer_opt = shared_comm or os.environ.get("BD_DEFER_OPT", "0") == "1"
        if (world_size > 1 or self.dp.force) and torch.cuda.current_stream(self.dev) == torch.cuda.default_stream(self.dev):
            # The legacy null stream synchronises implicitly with every BLOCKING stream of the process, and RCCL's is
            # one: with collectives in flight, each replay gather / stream hand-over the caller issues on the null
            # stream waits for the communicator to drain, i.e. for behaviour learning of the previous step -- the
            # cross-step pipeline collapses to the serial schedule (measured with a one-rank communicator: 3.47 ->
            # 5.33 ms/step).  Data-parallel runs therefore move the calling thread to a non-blocking stream.
            self._main_stream = _engine_stream(self.dev, "main", 0)
            torch.cuda.set_stream(self._main_stream)
        self._pending_opt: List[tuple] = []
        self._opt_over: Dict[str, dict] = {}
        # perf-mode noise: Philox keyed by the process seed (torch.manual_seed before building the agent, src/main.py:56-58)
        # and the rank, counter = step index per phase
        # (read at the FIRST perf-mode draw, so that seeding torch any time before training reproduces the run)
        self._rng_seed: Optional[int] = None
        self._rng_step = {"wm": 0, "bh": 0}
        self._rng_entropy_step = 0
        self._log_ring: List[_LogRecord] = []
        self._log_i = 0
        self._cur_rec: Optional[_LogRecord] = None
        self._wm_done_hist: List[torch.cuda.Event] = []
        self._parity = 0
        self._wgrad_ws_bh = torch.zeros(1, dtype=torch.float32, device=self.dev)
        self.red_ws_bh = torch.zeros(int(lib.bd_reduce_ws_floats()), dtype=torch.float32, device=self.dev)
        self._buf: Dict[str, torch.Tensor] = {}
        # cluster variant of the observe scan (several CUs per 16-row tile): on unless BD_OBS_CLUSTER=0
        # steps the host may run ahead of the GPU before train_step waits (0: no limit): two keep the launch queues fed and stay
        # far below the depth at which the HIP runtime stalls the host to retire commands in bulk (5-6 steps of configs[1])
        self.host_ahead = int(os.environ.get("BD_HOST_AHEAD", "2"))
        self._bh_hist: List["torch.cuda.Event"] = []
        self.use_obs_cluster = os.environ.get("BD_OBS_CLUSTER", "1") != "0"
        self._obs_ws: Optional[torch.Tensor] = None
        self._obs_err_off: Optional[int] = None
        self._timers_on, self._timer_every, self._timer_tick = False, 1, 0
        self._timer_events: Dict[str, List[Tuple[torch.cuda.Event, torch.cuda.Event]]] = {}
        self._build_pack_tables()
        self.conv = None
        if self.pixel:
            from .conv_stack import ConvStacks
            self.conv = ConvStacks(self)
        for g in ("model", "actor", "critic", "critic_target"):
            self.pack(g)

    # ------------------------------------------------------------------------------------------ timing
    class _Span:
        """HIP-event pair recorded on the launch stream around a group of kernel launches."""

        def __init__(self, eng, name):
            self.eng, self.name = eng, name

        def __enter__(self):
            self.on = self.eng._timers_on and self.eng._timer_tick % self.eng._timer_every == 0
            if self.on:
                self.e0 = torch.cuda.Event(enable_timing=True)
                self.e1 = torch.cuda.Event(enable_timing=True)
                self.e0.record()

        def __exit__(self, *exc):
            if self.on:
                self.e1.record()
                self.eng._timer_events.setdefault(self.name, []).append((self.e0, self.e1))

    def span(self, name: str) -> "DreamerEngine._Span":
        return DreamerEngine._Span(self, name)

    def enable_timers(self, on: bool, every: int = 1) -> None:
        """HIP-event spans around the kernel groups of every `every`-th train step (events cost queue slots and host
        time, so a benchmark samples)."""
        self._timers_on, self._timer_every, self._timer_tick = on, max(1, every), 0
        if on:
            self._timer_events = {}

    def timer_summary(self) -> Dict[str, Tuple[float, int]]:
        """{span: (average ms, count)}; call after a device synchronise."""
        return {k: (sum(a.elapsed_time(b) for a, b in v) / len(v), len(v)) for k, v in self._timer_events.items()}

    # ------------------------------------------------------------------------------------------ params
    def W(self, mod: str, name: str) -> torch.Tensor:
        return self.groups[self._mod_group[mod]].p[(mod, name)]

    def G(self, mod: str, name: str) -> torch.Tensor:
        return self.groups[self._mod_group[mod]].g[(mod, name)]

    def Ws(self, mod: str, name: str) -> torch.Tensor:
        """Contiguous STORAGE view of a parameter (differs from W() for conv tensors)."""
        return self.groups[self._mod_group[mod]].ps[(mod, name)]

    def Gs(self, mod: str, name: str) -> torch.Tensor:
        return self.groups[self._mod_group[mod]].gs[(mod, name)]

    def load_params(self, params: dict) -> None:
        for mod, sd in params.items():
            for name, v in sd.items():
                self.W(mod, name).copy_(torch.as_tensor(np.asarray(v), dtype=torch.float32))

    _OPT = {"model": "model_learning_rate", "actor": "actor_learning_rate", "critic": "value_learning_rate"}

    def optimizer_state_dict(self, group: str) -> dict:
        """torch.optim.Adam-layout state of one optimiser ("model" | "actor" | "critic"); synchronises."""
        self.flush_optimizers()
        self.join()
        torch.cuda.current_stream().synchronize()
        return _adam_state_dict(self.groups[group], self.hp[self._OPT[group]], self.hp, self._opt_over.get(group))

    def load_optimizer_state_dict(self, group: str, sd: dict) -> None:
        """Moments, step count AND the checkpoint's lr / eps / weight_decay (torch's Optimizer.load_state_dict adopts the
        saved param_group, so a run resumed by the reference continues with the checkpoint's settings, src/planet.py:114)."""
        self.flush_optimizers()
        self.join()
        hyper = _load_adam_state_dict(self.groups[group], sd)
        if hyper:
            mine = {"lr": self.hp[self._OPT[group]], "eps": self.hp["adam_epsilon"], "weight_decay": self.hp["weight_decay"]}
            diff = {k: (mine[k], v) for k, v in hyper.items() if v != mine[k]}
            if diff:
                import warnings
                warnings.warn(f"{group} optimiser: adopting the checkpoint's hyper-parameters (configured, checkpoint): {diff}")
            self._opt_over[group] = hyper

    def state_dict(self, mod: str) -> Dict[str, torch.Tensor]:
        g = self.groups[self._mod_group[mod]]
        return {n: g.p[(m, n)] for (m, n, _) in g.specs if m == mod}

    def buf(self, name: str, *shape) -> torch.Tensor:
        t = self._buf.get(name)
        if t is None or tuple(t.shape) != tuple(shape):
            t = torch.zeros(*shape, dtype=torch.float32, device=self.dev)
            self._buf[name] = t
        return t

    # ------------------------------------------------------------------------------------------ packing
    def _build_pack_tables(self) -> None:
        d = self.d
        Be, S, A, Hd = d.Be, d.S, d.A, d.Hd
        ent: Dict[str, List[Tuple[str, torch.Tensor, bool]]] = {g: [] for g in ("model", "actor", "critic", "critic_target")}

        def add(group, key, view, fwd=True, tr=False):
            if fwd:
                ent[group].append((key, view, False))
            if tr:
                ent[group].append((key + ".T", view, True))

        tm = lambda n: self.W("transition_model", n)
        We = tm("fc_embed_state_action.0.weight")
        cat = d.categorical
        # Categorical latents: the state columns of a first layer are GATHERED (csrc/scan_cat.hip), from plain
        # transposed copies [S x out] refreshed with the packs (self._plain); only their transposes are packed (dgrad)
        self._plain: Dict[str, Tuple[torch.Tensor, torch.Tensor, str]] = {}
        add("model", "embed_s", We[:, :S], fwd=not cat, tr=True)
        add("model", "embed_a", We[:, S:], tr=True)
        if cat:
            self._plain["embed_sT"] = (torch.zeros(S, Be, dtype=torch.float32, device=self.dev), We[:, :S], "model")
        for gi, gname in enumerate("rzn"):
            add("model", f"i{gname}", tm("rnn.weight_ih")[gi * Be:(gi + 1) * Be], tr=True)
            add("model", f"h{gname}", tm("rnn.weight_hh")[gi * Be:(gi + 1) * Be], tr=True)
        add("model", "p1", tm("belief_prior.model.0.weight"), tr=True)
        Wp2 = tm("belief_prior.model.2.weight")
        add("model", "p2", Wp2, tr=True)
        if not cat:
            add("model", "p2m", Wp2[:S], tr=True)
            add("model", "p2s", Wp2[S:], tr=True)
        Wq1 = tm("belief_posterior.model.0.weight")
        add("model", "q1h", Wq1[:, :Be], tr=True)
        add("model", "q1e", Wq1[:, Be:], tr=True)
        Wq2 = tm("belief_posterior.model.2.weight")
        if cat:
            add("model", "q2", Wq2, tr=True)
        else:
            add("model", "q2m", Wq2[:S], tr=True)
            add("model", "q2s", Wq2[S:], tr=True)
        for l in range(DENSE_LAYERS + 1):
            if not self.pixel:
                add("model", f"enc{l}", self.W("encoder", f"model.{2 * l}.weight"), tr=(l > 0))
                add("model", f"obs{l}", self.W("observation_model", f"model.{2 * l}.weight"), tr=True)
            add("model", f"rew{l}", self.W("reward_model", f"model.{2 * l}.weight"), tr=True)
            if d.use_discount:
                add("model", f"dsc{l}", self.W("discount_model", f"model.{2 * l}.weight"), tr=True)
            add("critic", f"cri{l}", self.W("critic", f"model.{2 * l}.weight"), tr=(l > 0))
            add("critic_target", f"tgt{l}", self.W("critic_target", f"model.{2 * l}.weight"), tr=True)
        if cat:     # heads on [h; one-hot s]: layer 0 = belief columns (packed) + a gather of the state columns (plain W^T)
            heads = [("model", "reward_model", "rew"), ("critic", "critic", "cri"), ("critic_target", "critic_target", "tgt")]
            if d.use_discount:
                heads.append(("model", "discount_model", "dsc"))
            if not self.pixel:
                heads.append(("model", "observation_model", "obs"))
            for grp, mod, prefix in heads:
                W0 = self.W(mod, "model.0.weight")
                add(grp, f"{prefix}0h", W0[:, :Be])
                self._plain[f"{prefix}0sT"] = (torch.zeros(S, Hd, dtype=torch.float32, device=self.dev), W0[:, Be:], grp)
        Wa0 = self.W("actor", "model.0.weight")
        add("actor", "a0h", Wa0[:, :Be])
        if cat:
            self._plain["a0sT"] = (torch.zeros(S, Hd, dtype=torch.float32, device=self.dev), Wa0[:, Be:], "actor")
        else:
            add("actor", "a0s", Wa0[:, Be:])
        for l in range(1, DENSE_LAYERS):
            add("actor", f"a{l}", self.W("actor", f"model.{2 * l}.weight"), tr=True)
        Wa4 = self.W("actor", f"model.{2 * DENSE_LAYERS}.weight")
        add("actor", "a4m", Wa4[:A], tr=True)
        add("actor", "a4s", Wa4[A:], tr=True)
        add("actor", "a4", Wa4, fwd=False, tr=True)     # (mean | raw) rows together: the hidden-layer backward as one chain

        self.pk: Dict[str, torch.Tensor] = {}
        self._pack_tables = {}
        for group, lst in ent.items():
            total = sum(cabi.packed_floats(*(v.shape if not tr else v.shape[::-1])) for _, v, tr in lst)
            store = torch.zeros(total, dtype=torch.float32, device=self.dev)
            descs = (cabi.PackDesc * len(lst))()
            off = 0
            for i, (key, v, tr) in enumerate(lst):
                N, K = v.shape
                n = cabi.packed_floats(N, K)   # same count for the transpose
                self.pk[key] = store[off:off + n]
                descs[i] = cabi.PackDesc(v.data_ptr(), self.pk[key].data_ptr(), v.stride(0), N, K, int(tr))
                off += n
            raw = torch.frombuffer(bytearray(bytes(descs)), dtype=torch.uint8).to(self.dev)
            self._pack_tables[group] = (raw, len(lst), store)

    def pack(self, group: str) -> None:
        raw, n, _ = self._pack_tables[group]
        cabi.check(lib.bd_pack_weights(raw.data_ptr(), n, cabi.stream()))
        for dst, src, grp in self._plain.values():          # Categorical latents: plain transposes for the gathers
            if grp == group:
                dst.copy_(src.t())
        if group == "model" and self.conv is not None:
            self.conv.pack()

    # ------------------------------------------------------------------------------------------ kernels
    def _dense_spec(self, mod: str, prefix: str, in_width: int, out_width: int):
        """[(packed W key, bias view, N, K, act)] for a DenseModel (4 hidden ELU layers + linear)."""
        sizes = [in_width] + [self.d.Hd] * DENSE_LAYERS + [out_width]
        return [(f"{prefix}{l}", self.W(mod, f"model.{2 * l}.bias"), sizes[l + 1], sizes[l],
                 cabi.ACT_ELU if l < DENSE_LAYERS else cabi.ACT_NONE) for l in range(DENSE_LAYERS + 1)]

    def mlp_forward(self, M, in0, ld0, w0, layers, saves, out, ldo, in1=None, ld1=0, w1=0, raw_packs=False,
                    gather=None) -> None:
        """layers: [(packed-weight key | packed tensor when raw_packs, bias, N, K, act)].
        gather = (class indices [M x D] uint8, plain W0s^T [D*C x N0], D, C): a one-hot input segment of layer 0."""
        a = cabi.MlpFwdArgs()
        a.M, a.in0, a.ld0, a.w0 = M, ptr(in0), ld0, w0
        a.in1, a.ld1, a.w1 = ptr(in1), ld1, w1
        if gather is not None:
            a.gidx, a.gWT, a.gD, a.gC = ptr(gather[0]), ptr(gather[1]), gather[2], gather[3]
        a.n_layers = len(layers)
        for i, (key, bias, N, K, act) in enumerate(layers):
            a.layer[i] = cabi.Layer(ptr(key if raw_packs else self.pk[key]), ptr(bias), N, K, act,
                                    ptr(saves[i]) if saves else None)
        a.out, a.ldo = ptr(out), ldo
        cabi.check(lib.bd_mlp_forward(C.byref(a), cabi.stream()))

    def mlp_backward(self, M, dout, lddo, layers, saves, dpres, din0=None, ld0=0, w0=0, din1=None, ld1=0, w1=0,
                     accumulate=False, dout_scale=1.0, raw_packs=False) -> None:
        """layers as in mlp_forward; with raw_packs the first entry is the packed TRANSPOSED weight tensor."""
        a = cabi.MlpBwdArgs()
        a.M, a.dout, a.lddo, a.dout_scale = M, ptr(dout), lddo, dout_scale
        a.n_layers = len(layers)
        for i, (key, _bias, N, K, act) in enumerate(layers):
            wt = key if raw_packs else self.pk.get(key + ".T")
            a.layer[i] = cabi.LayerBwd(ptr(wt) if wt is not None else None,
                                       ptr(saves[i]) if (saves and saves[i] is not None) else None, N, K, act,
                                       ptr(dpres[i]) if (dpres and dpres[i] is not None) else None)
        a.din0, a.ld0, a.w0 = ptr(din0), ld0, w0
        a.din1, a.ld1, a.w1 = ptr(din1), ld1, w1
        a.accumulate = int(accumulate)
        cabi.check(lib.bd_mlp_backward(C.byref(a), cabi.stream()))

    def wgrad(self, dpre, ldp, act, lda, M, N, K, dW, ldw, db=None, accumulate=False) -> None:
        need = int(lib.bd_wgrad_ws_floats(M, N, K))
        if self._wgrad_ws.numel() < need:
            self._wgrad_ws = torch.zeros(need, dtype=torch.float32, device=self.dev)
        cabi.check(lib.bd_wgrad(ptr(dpre), ldp, ptr(act), lda, M, N, K, ptr(dW), ldw, ptr(db), int(accumulate),
                                ptr(self._wgrad_ws), self._wgrad_ws.numel(), cabi.stream()))

    def _dense_wgrads(self, batch: "WgradBatch", mod: str, M: int, dpres, inp, ld_in, saves, sizes) -> None:
        """Weight/bias gradients of a DenseModel from its pre-activation gradients (queued on `batch`)."""
        for l in range(len(sizes) - 1):
            act, lda = (inp, ld_in) if l == 0 else (saves[l - 1], sizes[l])
            batch.add(dpres[l], sizes[l + 1], act, lda, M, sizes[l + 1], sizes[l],
                      self.G(mod, f"model.{2 * l}.weight"), sizes[l], self.G(mod, f"model.{2 * l}.bias"))

    def _allreduce(self, t: torch.Tensor, key: Optional[str] = None) -> None:
        self.dp.allreduce_sum_(t, key)

    def optimizer_step(self, group: str, slot: int, lr: float, red_ws: Optional[torch.Tensor] = None,
                       rec: Optional[_LogRecord] = None) -> None:
        """all-reduce (data-parallel) + global-norm clip + Adam + re-pack of one optimiser, on the current stream.
        `rec`: log record that receives this phase's snapshot of the scalar board (lazy logs)."""
        g = self.groups[group]
        red_ws = self.red_ws if red_ws is None else red_ws
        self._allreduce(g.grad, group)       # grads already carry 1/global-count: SUM over ranks = global-mean gradient
        g.step += 1
        hp = self.hp
        ov = self._opt_over.get(group)          # hyper-parameters adopted from a checkpoint (load_optimizer_state_dict)
        if ov:
            lr = ov.get("lr", lr)
        eps, wd = (ov or {}).get("eps", hp["adam_epsilon"]), (ov or {}).get("weight_decay", hp["weight_decay"])
        cabi.check(lib.bd_sumsq(ptr(g.grad), g.numel, ptr(self.scalars), slot, ptr(red_ws), cabi.stream()))
        cabi.check(lib.bd_adam_step(ptr(g.flat), ptr(g.grad), ptr(g.m), ptr(g.v), g.numel, lr, 0.9, 0.999,
                                    eps, wd, g.step, hp["grad_clip_norm"],
                                    ptr(self.scalars), slot, cabi.stream()))
        self.pack(group)
        rec = rec if rec is not None else self._cur_rec
        if rec is not None:
            self._snapshot(rec, _LogRecord.ROW[group])

    # ------------------------------------------------------------------------------------------ lazy logs
    def _new_record(self) -> _LogRecord:
        """Next record of a ring of eight; a record still referenced by an unread LazyLogs is resolved first."""
        if len(self._log_ring) < 8:
            self._log_ring.append(_LogRecord())
            rec = self._log_ring[-1]
        else:
            rec = self._log_ring[self._log_i]
            self._log_i = (self._log_i + 1) % 8
            owner = rec.owner() if rec.owner is not None else None
            if owner is not None and owner._vals is None:
                owner.resolve()
        rec.events = [None, None, None]
        rec.owner, rec.counts = None, None
        return rec

    def _snapshot(self, rec: _LogRecord, row: int) -> None:
        """Asynchronous D2H of the scalar board (+ the cluster scan's sticky error word) into `rec`, on the current
        stream, behind the kernels of this phase that wrote its slots."""
        rec.host[row, :N_SLOTS].copy_(self.scalars, non_blocking=True)
        if row == 0 and self._obs_ws is not None and self._obs_err_off is not None:
            rec.host[row, N_SLOTS:].copy_(self._cluster_error_word(), non_blocking=True)
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream())
        rec.events[row] = ev

    def _resolve_record(self, rec: _LogRecord) -> Dict[str, float]:
        # The actor / critic updates of the LATEST step may still be held back on the host (deferred schedule).  Issuing
        # them contains all-reduces: with several ranks a dict read -- which only one rank may make (rank 0 logging), or
        # which a recycled record makes on the ranks that still hold its dict -- must never do that: the ranks' collectives
        # would be issued in different orders on the shared communicator.  One rank: flush (no collective involved).
        if any(p[6] is rec for p in self._pending_opt):
            if self.world_size > 1:
                raise RuntimeError("log dict of a train step whose actor / critic updates are still queued (data-parallel "
                                   "deferred schedule): call train_step(), update_critic(), update_belief_and_act() or "
                                   "flush_optimizers() on EVERY rank before reading it -- reading a log dict never issues "
                                   "collectives")
            self.flush_optimizers()
        for row in rec.need:
            assert rec.events[row] is not None, "log record read before its step was queued"
            rec.events[row].synchronize()
        h = rec.host.numpy()
        s = h[0, :N_SLOTS].copy()
        for slot in (SLOT_RET, SLOT_ENT, SLOT_GN_ACTOR, SLOT_WOBJ):
            s[slot] = h[1, slot]
        for slot in (SLOT_VAL, SLOT_GN_CRITIC):
            s[slot] = h[2, slot]
        self._raise_on_cluster_error(int(h[0, N_SLOTS:].view(np.uint32)[0]))
        return self._logs_from(s, rec.counts)

    def _cluster_error_word(self) -> torch.Tensor:
        """The sticky error word of the cluster scans as one float32-typed element (bit pattern of the uint32 word).
        Data-parallel: a time-out on ANY rank must stop EVERY rank at the same log read (a rank that raised alone would
        leave the others waiting in their next all-reduce), so the ranks' "some member timed out" flags are summed on the
        world-model communicator -- right behind the gradient all-reduce the model optimiser step has just issued, in
        the same order on all ranks -- and the result is reported as forward+backward (bits 1|2)."""
        w = self._obs_ws[self._obs_err_off:self._obs_err_off + 1]
        if self.world_size <= 1:
            return w
        flag = (w.view(torch.int32) != 0).to(torch.float32)
        self._allreduce(flag, "model")
        return torch.where(flag > 0, torch.full_like(flag, 3).to(torch.int32), torch.zeros_like(flag).to(torch.int32)).view(torch.float32)

    def _raise_on_cluster_error(self, word: int) -> None:
        if word:
            off = self._obs_err_off
            self._obs_ws[off:off + 1].zero_()                    # reported once
            which = "+".join(n for b, n in ((1, "forward"), (2, "backward")) if word & b)
            raise RuntimeError(f"bigdreamer_hip: a member of the cluster observe scan timed out waiting for its peers "
                               f"({which} scan): the step's posteriors / gradients are wrong (BD_OBS_CLUSTER=0 selects "
                               "the single-workgroup scan)")

    def update_critic(self) -> None:
        """polyak_update(critic_target, critic, polyak_avg) (src/dreamer.py:423-427).  Data-parallel: issues the pending
        critic update first (a collective -- call on every rank, as src/main.py:110-112 does)."""
        self.flush_optimizers()
        self.join()
        t, s = self.groups["critic_target"], self.groups["critic"]
        cabi.check(lib.bd_polyak(ptr(t.flat), ptr(s.flat), t.numel, float(self.hp["polyak_avg"]), cabi.stream()))
        self.pack("critic_target")

    # ------------------------------------------------------------------------------------------ forward pieces
    def encode(self, obs2d: torch.Tensor, M: int, save: bool = True):
        """Encoder DenseModel (src/planet.py:195-200) + hoisted embedding half of the posterior's first
        layer.  Returns (embeddings [M x E], pre_emb [M x Hd])."""
        d = self.d
        layers = self._dense_spec("encoder", "enc", d.O, d.E) + [("q1e", None, d.Hd, d.E, cabi.ACT_NONE)]
        acts = [self.buf(f"enc_act{l}", M, d.Hd) for l in range(DENSE_LAYERS)]
        emb = self.buf("emb", M, d.E)
        pre = self.buf("pre_emb", M, d.Hd)
        self.mlp_forward(M, obs2d, d.O, d.O, layers, acts + [emb, None], pre, d.Hd)
        return emb, pre

    # ---- pixel observations: the conv stacks (conv_stack.ConvStacks; same interface: tests/torch_conv_stacks.py) ----
    def encode_pixels(self, obs4d: torch.Tensor, grad: bool = True):
        """CnnImageEncoder (src/models.py:527-564) on (M,3,64,64) + the hoisted posterior projection.  grad=False: API use
        (separate activation buffers, nothing kept for a backward)."""
        d = self.d
        M = obs4d.shape[0]
        emb = self.conv.encode(obs4d.float(), tag="" if grad else "api_")
        pre = self.buf("pre_emb" if grad else "api_cv_pre_emb", M, d.Hd)
        self.mlp_forward(M, emb, d.E, d.E, [("q1e", None, d.Hd, d.E, cabi.ACT_NONE)], None, pre, d.Hd)
        return emb, pre

    def decode_pixels(self, feat: torch.Tensor) -> torch.Tensor:
        """ObservationModel (src/models.py:319-362) for API callers: (M, Be+S) -> (M,3,64,64) in the reference's NCHW."""
        from . import conv as _conv
        return _conv.to_nchw(self.conv.decode(feat.contiguous().float(), tag="api_"))

    def observe(self, actions, nonterm, pre_emb, eps_post, init_belief, init_state, T: int, B: int, save: bool = True,
                tag: str = "", prior_only: bool = False, feat_tag: str = ""):
        """TransitionModel.forward recurrence.  Returns feat [T*B x (Be+S)], mean, std of the fed-back state:
        the posterior (embeddings given) or, with prior_only=True (embeddings=None, src/models.py:241,296-297), the
        prior -- the same kernel run with the prior head's weights and a zero embedding projection."""
        d, pk = self.d, self.pk
        if d.categorical:
            return self._observe_cat(actions, nonterm, pre_emb, eps_post, init_belief, init_state, T, B, save, tag,
                                     prior_only, feat_tag)
        tm = lambda n: self.W("transition_model", n)
        M = T * B
        a = cabi.ObserveFwdArgs()
        a.T, a.B, a.Be, a.S, a.A, a.Hd = T, B, d.Be, d.S, d.A, d.Hd
        a.w_embed_s, a.w_embed_a, a.b_embed = ptr(pk["embed_s"]), ptr(pk["embed_a"]), ptr(tm("fc_embed_state_action.0.bias"))
        a.w_ir, a.w_iz, a.w_in = ptr(pk["ir"]), ptr(pk["iz"]), ptr(pk["in"])
        a.w_hr, a.w_hz, a.w_hn = ptr(pk["hr"]), ptr(pk["hz"]), ptr(pk["hn"])
        a.b_ih, a.b_hh = ptr(tm("rnn.bias_ih")), ptr(tm("rnn.bias_hh"))
        if prior_only:
            a.w_q1h, a.b_q1 = ptr(pk["p1"]), ptr(tm("belief_prior.model.0.bias"))
            a.w_q2m, a.w_q2s, a.b_q2 = ptr(pk["p2m"]), ptr(pk["p2s"]), ptr(tm("belief_prior.model.2.bias"))
            pre_emb = self.buf(tag + "zero_pre_emb", M, d.Hd)       # stays zero: nothing ever writes it
        else:
            a.w_q1h, a.b_q1 = ptr(pk["q1h"]), ptr(tm("belief_posterior.model.0.bias"))
            a.w_q2m, a.w_q2s, a.b_q2 = ptr(pk["q2m"]), ptr(pk["q2s"]), ptr(tm("belief_posterior.model.2.bias"))
        a.init_belief, a.init_state, a.actions = ptr(init_belief), ptr(init_state), ptr(actions)
        a.nonterm, a.pre_emb, a.eps_post = ptr(nonterm), ptr(pre_emb), ptr(eps_post)
        a.min_std = self.hp["min_std_dev"]
        feat = self.buf(tag + feat_tag + "feat", M, d.Be + d.S)
        qm, qs = self.buf(tag + "post_mean", M, d.S), self.buf(tag + "post_std", M, d.S)
        a.feat, a.post_mean, a.post_std = ptr(feat), ptr(qm), ptr(qs)
        if save:
            a.sv_s, a.sv_x = ptr(self.buf("sv_s", M, d.S)), ptr(self.buf("sv_x", M, d.Be))
            a.sv_gates, a.sv_q = ptr(self.buf("sv_gates", M, 4 * d.Be)), ptr(self.buf("sv_q", M, d.Hd))
        with self.span("observe_fwd"):
            if self._cluster_ok(B):
                ws = self._cluster_ws(B)
                cabi.check(lib.bd_observe_forward_cluster(C.byref(a), ptr(ws), ws.numel(), cabi.stream()))
            else:
                cabi.check(lib.bd_observe_forward(C.byref(a), cabi.stream()))
        return feat, qm, qs

    def _observe_cat(self, actions, nonterm, pre_emb, q_post, init_belief, init_state, T, B, save, tag, prior_only, feat_tag):
        """Categorical latents: TransitionModel.forward recurrence on bd_observe_cat_forward.  Returns feat
        [T*B x (Be+S)] ([h; one-hot s]), the logits of the fed-back state twice (the Gaussian signature's mean / std
        slots), and leaves the class indices in buf(tag + feat_tag + "sidx")."""
        d, pk = self.d, self.pk
        tm = lambda n: self.W("transition_model", n)
        M = T * B
        a = cabi.ObserveCatFwdArgs()
        a.T, a.B, a.Be, a.D, a.C, a.A, a.Hd = T, B, d.Be, d.cat_D, d.cat_C, d.A, d.Hd
        a.w_embed_sT, a.w_embed_a = ptr(self._plain["embed_sT"][0]), ptr(pk["embed_a"])
        a.b_embed = ptr(tm("fc_embed_state_action.0.bias"))
        a.w_ir, a.w_iz, a.w_in = ptr(pk["ir"]), ptr(pk["iz"]), ptr(pk["in"])
        a.w_hr, a.w_hz, a.w_hn = ptr(pk["hr"]), ptr(pk["hz"]), ptr(pk["hn"])
        a.b_ih, a.b_hh = ptr(tm("rnn.bias_ih")), ptr(tm("rnn.bias_hh"))
        if prior_only:
            a.w_q1h, a.b_q1 = ptr(pk["p1"]), ptr(tm("belief_prior.model.0.bias"))
            a.w_q2, a.b_q2 = ptr(pk["p2"]), ptr(tm("belief_prior.model.2.bias"))
            pre_emb = self.buf(tag + "zero_pre_emb", M, d.Hd)
        else:
            a.w_q1h, a.b_q1 = ptr(pk["q1h"]), ptr(tm("belief_posterior.model.0.bias"))
            a.w_q2, a.b_q2 = ptr(pk["q2"]), ptr(tm("belief_posterior.model.2.bias"))
        a.init_belief, a.init_state, a.actions = ptr(init_belief), ptr(init_state), ptr(actions)
        a.nonterm, a.pre_emb, a.q_post = ptr(nonterm), ptr(pre_emb), ptr(q_post)
        feat = self.buf(tag + feat_tag + "feat", M, d.Be + d.S)
        logits = self.buf(tag + "post_logits", M, d.S)
        sidx = self._buf_u8(tag + feat_tag + "sidx", M, d.cat_D)
        a.feat, a.post_logits, a.sidx = ptr(feat), ptr(logits), ptr(sidx)
        if save:
            a.sv_s, a.sv_x = ptr(self.buf("sv_s", M, d.S)), ptr(self.buf("sv_x", M, d.Be))
            a.sv_gates, a.sv_q = ptr(self.buf("sv_gates", M, 4 * d.Be)), ptr(self.buf("sv_q", M, d.Hd))
        with self.span("observe_fwd"):
            Cm = self._cat_cluster(B)
            if Cm:
                ws = self._cat_cluster_ws(B, Cm)
                cabi.check(lib.bd_observe_cat_forward_cluster(C.byref(a), Cm, ptr(ws), ws.numel(), cabi.stream()))
            else:
                cabi.check(lib.bd_observe_cat_forward(C.byref(a), cabi.stream()))
        return feat, logits, logits

    def _cat_cluster(self, B: int) -> int:
        """Members per 16-row tile of the Categorical cluster observe scan (csrc/observe_cat_cluster.hip), 0 = use the
        one-workgroup-per-tile kernels: same rule as _cluster_ok (tiles * Cm one-per-CU members leave half the chip to the
        other pipeline streams)."""
        if not self.use_obs_cluster:
            return 0
        d = self.d
        return int(lib.bd_observe_cat_cluster_size(B, d.Be, d.cat_D, d.cat_C, int(os.environ.get("BD_OBS_CLUSTER_MAX_WGS", "128"))))

    def _cat_cluster_ws(self, B: int, Cm: int) -> torch.Tensor:
        d = self.d
        need = int(lib.bd_observe_cat_cluster_ws_floats(B, d.Be, d.Hd, d.cat_D, Cm))
        off = int(lib.bd_observe_cluster_err_offset(B))
        if self._obs_ws is None or self._obs_ws.numel() < need or self._obs_err_off != off:
            self._obs_ws = torch.zeros(need, dtype=torch.float32, device=self.dev)     # zero: flags AND the sticky error word
            self._obs_err_off = off
        return self._obs_ws

    def _buf_u8(self, name: str, *shape) -> torch.Tensor:
        t = self._buf.get(name)
        if t is None or tuple(t.shape) != tuple(shape):
            t = torch.zeros(*shape, dtype=torch.uint8, device=self.dev)
            self._buf[name] = t
        return t

    def _imagine_cat(self, start_feat, start_sidx, N: int, Hm: int, noise, save: bool, tag: str, feat_tag: str):
        d, pk = self.d, self.pk
        tm = lambda n: self.W("transition_model", n)
        ac = lambda n: self.W("actor", n)
        Mi = Hm * N
        a = cabi.ImagineCatFwdArgs()
        a.N, a.Hm, a.Be, a.D, a.C, a.A, a.Hd, a.n_samples = N, Hm, d.Be, d.cat_D, d.cat_C, d.A, d.Hd, d.n_entropy
        a.w_embed_sT, a.w_embed_a = ptr(self._plain["embed_sT"][0]), ptr(pk["embed_a"])
        a.b_embed = ptr(tm("fc_embed_state_action.0.bias"))
        a.w_ir, a.w_iz, a.w_in = ptr(pk["ir"]), ptr(pk["iz"]), ptr(pk["in"])
        a.w_hr, a.w_hz, a.w_hn = ptr(pk["hr"]), ptr(pk["hz"]), ptr(pk["hn"])
        a.b_ih, a.b_hh = ptr(tm("rnn.bias_ih")), ptr(tm("rnn.bias_hh"))
        a.w_p1, a.b_p1 = ptr(pk["p1"]), ptr(tm("belief_prior.model.0.bias"))
        a.w_p2, a.b_p2 = ptr(pk["p2"]), ptr(tm("belief_prior.model.2.bias"))
        a.w_a0h, a.w_a0sT = ptr(pk["a0h"]), ptr(self._plain["a0sT"][0])
        for l in range(1, DENSE_LAYERS):
            a.w_a[l - 1] = ptr(pk[f"a{l}"])
        for l in range(DENSE_LAYERS):
            a.b_a[l] = ptr(ac(f"model.{2 * l}.bias"))
        a.w_a4m, a.w_a4s, a.b_a4 = ptr(pk["a4m"]), ptr(pk["a4s"]), ptr(ac(f"model.{2 * DENSE_LAYERS}.bias"))
        a.start_feat, a.start_sidx = ptr(start_feat), ptr(start_sidx)
        a.eps_action, a.eps_entropy, a.q_prior = ptr(noise["action"]), ptr(noise.get("entropy")), ptr(noise["img_prior"])
        a.act_raw_init_std, a.act_min_std, a.act_mean_scale = ACT_RAW_INIT_STD, ACT_MIN_STD, ACT_MEAN_SCALE
        ifeat = self.buf(tag + feat_tag + "ifeat", Mi, d.Be + d.S)
        a.feat = ptr(ifeat)
        a.sidx = ptr(self._buf_u8(tag + feat_tag + "isidx", Mi, d.cat_D))      # read by the critic update: double-buffered like ifeat
        a.prior_logits = ptr(self.buf(tag + "iprior_logits", Mi, d.S))
        ent, act = self.buf(tag + "entropy", Mi), self.buf(tag + "action", Mi, d.A)
        a.entropy, a.action = ptr(ent), ptr(act)
        if save:
            a.sv_actor = ptr(self.buf("sv_actor", DENSE_LAYERS, Mi, d.Hd))
            a.sv_act_stats = ptr(self.buf("sv_act_stats", Mi, 4 * d.A))
            a.sv_x, a.sv_gates = ptr(self.buf("isv_x", Mi, d.Be)), ptr(self.buf("isv_gates", Mi, 4 * d.Be))
            a.sv_p = ptr(self.buf("isv_p", Mi, d.Hd))
        with self.span("imagine_fwd"):
            cabi.check(lib.bd_imagine_cat_forward(C.byref(a), cabi.stream()))
        if save and noise.get("entropy") is None:       # perf mode: the scan alone ran; the estimator draws in-kernel
            self._entropy_estimate(noise, ent, Hm, N)
        self._img_split_rows = 0
        return ifeat, ent, act

    def _cluster_ok(self, B: int) -> bool:
        """Cluster scan only while its tiles*C one-per-CU members leave half the chip to the kernels the other pipeline
        streams run beside it (members claim a whole CU's LDS and advance in lock step: on a crowded chip they would
        queue for CUs behind unrelated workgroups); larger batches use the single-workgroup scan (observe.hip)."""
        if not self.use_obs_cluster:
            return False
        C_ = int(lib.bd_observe_cluster_size(B, self.d.Be))
        return C_ > 0 and ((B + 15) // 16) * C_ <= int(os.environ.get("BD_OBS_CLUSTER_MAX_WGS", "128"))

    def _cluster_ws(self, B: int) -> torch.Tensor:
        need = int(lib.bd_observe_cluster_ws_floats(B, self.d.Be))
        off = int(lib.bd_observe_cluster_err_offset(B))
        if self._obs_ws is None or self._obs_ws.numel() < need or self._obs_err_off != off:
            self._obs_ws = torch.zeros(need, dtype=torch.float32, device=self.dev)     # zero: flags AND the sticky error word
            self._obs_err_off = off
        return self._obs_ws

    def cluster_status(self, B: int) -> None:
        """Raise if a member of ANY cluster launch since the last check timed out waiting for its peers (synchronises;
        the error word is sticky across launches and cleared by this read)."""
        if self._obs_ws is not None:
            cabi.check(lib.bd_observe_cluster_status(ptr(self._obs_ws), B, cabi.stream()))

    def prior_head(self, feat, M: int, eps, tag: str = ""):
        """belief_prior on all beliefs at once (src/models.py:256): returns state, mean, std [M x S] -- Categorical
        latents: (state or None when eps is None, logits, logits)."""
        d = self.d
        tm = lambda n: self.W("transition_model", n)
        layers = [("p1", tm("belief_prior.model.0.bias"), d.Hd, d.Be, cabi.ACT_ELU),
                  ("p2", tm("belief_prior.model.2.bias"), d.head_out, d.Hd, cabi.ACT_NONE)]
        hid, out = self.buf(tag + "p_hid", M, d.Hd), self.buf(tag + "p_out", M, d.head_out)
        self.mlp_forward(M, feat, d.Be + d.S, d.Be, layers, [hid, None], out, d.head_out)
        if d.categorical:
            pst = None
            if eps is not None:     # the sampled prior state is not on the training path (src/models.py:256 draws it anyway)
                pst, probs = self.buf(tag + "prior_state", M, d.S), self.buf(tag + "prior_probs", M, d.S)
                cabi.check(lib.bd_categorical_head_forward(ptr(out), ptr(eps), M, d.cat_D, d.cat_C, ptr(pst), ptr(probs),
                                                           cabi.stream()))
            return pst, out, out
        pm, ps = self.buf(tag + "prior_mean", M, d.S), self.buf(tag + "prior_std", M, d.S)
        pst = self.buf(tag + "prior_state", M, d.S) if eps is not None else None     # (perf mode: the unused sample is not drawn)
        cabi.check(lib.bd_gauss_head_forward(ptr(out), ptr(eps), M, d.S, self.hp["min_std_dev"], ptr(pm), ptr(ps),
                                             ptr(pst), cabi.stream()))
        return pst, pm, ps

    def dense_forward(self, mod: str, prefix: str, tag: str, x, ldx: int, M: int, out_width: int, rows=None, sidx=None):
        """DenseModel forward with saved activations; `rows` = (r0, r1) runs that row range only (same buffers).
        `sidx` (Categorical latents, x = [h; one-hot s]): the state's class indices [M x D] -- layer 0 then contracts the
        belief columns only and gathers the state columns (bd_mlp_forward's one-hot segment).  The returned layer list is
        the full-width one (the backward's d/d x is dense over all of [h; s])."""
        d = self.d
        layers = self._dense_spec(mod, prefix, ldx, out_width)
        acts = [self.buf(f"{tag}_act{l}", M, d.Hd) for l in range(DENSE_LAYERS)]
        out = self.buf(f"{tag}_out", M, out_width)
        fl, w_in, gather = layers, ldx, None
        if sidx is not None:
            fl = [(f"{prefix}0h", layers[0][1], layers[0][2], d.Be, layers[0][4])] + layers[1:]
            w_in, gather = d.Be, (sidx, self._plain[f"{prefix}0sT"][0], d.cat_D, d.cat_C)
        if rows is None:
            self.mlp_forward(M, x, ldx, w_in, fl, acts + [None], out, out_width, gather=gather)
        else:
            r0, r1 = rows
            if gather is not None:
                gather = (sidx[r0:r1],) + gather[1:]
            self.mlp_forward(r1 - r0, x.view(M, ldx)[r0:r1], ldx, w_in, fl, [t[r0:r1] for t in acts] + [None],
                             out[r0:r1], out_width, gather=gather)
        return out, acts, layers

    def imagine(self, start_feat, N: int, Hm: int, noise, save: bool = True, tag: str = "", feat_tag: str = "",
                split: bool = False, start_sidx: Optional[torch.Tensor] = None):
        d, pk = self.d, self.pk
        if d.categorical:
            if start_sidx is None:
                # API callers (Dreamer.get_action / imagine_ahead / update_belief_and_act) hand a DENSE state: the kernel
                # takes, per factor, the all-zero case (the collect loop's initial state, src/main.py:91-95: the actor and
                # the embed layer see zeros, as in the reference) or the one non-zero class.  Anything denser cannot be
                # carried as class indices: refuse it rather than silently treat it as one-hot.
                st = start_feat[:, d.Be:].reshape(N, d.cat_D, d.cat_C)
                if int(((st != 0).sum(-1) > 1).any()):
                    raise ValueError("Categorical latents: the start state must be all-zero or one-hot per factor "
                                     f"({d.cat_D} x {d.cat_C}); got a factor with more than one non-zero class")
            return self._imagine_cat(start_feat, start_sidx, N, Hm, noise, save, tag, feat_tag)
        tm = lambda n: self.W("transition_model", n)
        ac = lambda n: self.W("actor", n)
        Mi = Hm * N
        a = cabi.ImagineFwdArgs()
        a.N, a.Hm, a.Be, a.S, a.A, a.Hd, a.n_samples = N, Hm, d.Be, d.S, d.A, d.Hd, d.n_entropy
        a.w_embed_s, a.w_embed_a, a.b_embed = ptr(pk["embed_s"]), ptr(pk["embed_a"]), ptr(tm("fc_embed_state_action.0.bias"))
        a.w_ir, a.w_iz, a.w_in = ptr(pk["ir"]), ptr(pk["iz"]), ptr(pk["in"])
        a.w_hr, a.w_hz, a.w_hn = ptr(pk["hr"]), ptr(pk["hz"]), ptr(pk["hn"])
        a.b_ih, a.b_hh = ptr(tm("rnn.bias_ih")), ptr(tm("rnn.bias_hh"))
        a.w_p1, a.b_p1 = ptr(pk["p1"]), ptr(tm("belief_prior.model.0.bias"))
        a.w_p2m, a.w_p2s, a.b_p2 = ptr(pk["p2m"]), ptr(pk["p2s"]), ptr(tm("belief_prior.model.2.bias"))
        a.w_a0h, a.w_a0s = ptr(pk["a0h"]), ptr(pk["a0s"])
        for l in range(1, DENSE_LAYERS):
            a.w_a[l - 1] = ptr(pk[f"a{l}"])
        for l in range(DENSE_LAYERS):
            a.b_a[l] = ptr(ac(f"model.{2 * l}.bias"))
        a.w_a4m, a.w_a4s, a.b_a4 = ptr(pk["a4m"]), ptr(pk["a4s"]), ptr(ac(f"model.{2 * DENSE_LAYERS}.bias"))
        a.start_feat = ptr(start_feat)
        a.eps_action, a.eps_entropy, a.eps_prior = ptr(noise["action"]), ptr(noise.get("entropy")), ptr(noise["img_prior"])
        a.min_std, a.act_raw_init_std = self.hp["min_std_dev"], ACT_RAW_INIT_STD
        a.act_min_std, a.act_mean_scale = ACT_MIN_STD, ACT_MEAN_SCALE
        ifeat = self.buf(tag + feat_tag + "ifeat", Mi, d.Be + d.S)
        a.feat = ptr(ifeat)
        a.prior_mean = ptr(self.buf(tag + "iprior_mean", Mi, d.S))
        a.prior_std = ptr(self.buf(tag + "iprior_std", Mi, d.S))
        ent, act = self.buf(tag + "entropy", Mi), self.buf(tag + "action", Mi, d.A)
        a.entropy, a.action = ptr(ent), ptr(act)
        if save:
            a.sv_actor = ptr(self.buf("sv_actor", DENSE_LAYERS, Mi, d.Hd))
            a.sv_act_stats = ptr(self.buf("sv_act_stats", Mi, 4 * d.A))
            a.sv_x, a.sv_gates = ptr(self.buf("isv_x", Mi, d.Be)), ptr(self.buf("isv_gates", Mi, 4 * d.Be))
            a.sv_p = ptr(self.buf("isv_p", Mi, d.Hd))
        H1 = Hm // 2 if (split and Hm >= 2) else 0
        with self.span("imagine_fwd"):
            if not H1:
                cabi.check(lib.bd_imagine_forward_scan(C.byref(a), cabi.stream()))
            else:
                # two time segments: the frozen reward / value heads of the first can run under the second
                # (_behaviour_phase).  Same kernel, same operands per step: bit-identical to one launch.
                a.sv_actor_stride = Mi * d.Hd
                a.Hm = H1
                cabi.check(lib.bd_imagine_forward_scan(C.byref(a), cabi.stream()))
                self._ev_img_half = torch.cuda.Event()
                self._ev_img_half.record(torch.cuda.current_stream())
                r0 = H1 * N
                f4 = 4      # bytes per float
                a.Hm = Hm - H1
                a.start_feat = ptr(ifeat) + (r0 - N) * (d.Be + d.S) * f4
                a.eps_action = ptr(noise["action"]) + r0 * d.A * f4
                if noise.get("entropy") is not None:
                    a.eps_entropy = ptr(noise["entropy"]) + r0 * d.n_entropy * d.A * f4
                a.eps_prior = ptr(noise["img_prior"]) + r0 * d.S * f4
                a.feat = ptr(ifeat) + r0 * (d.Be + d.S) * f4
                a.prior_mean = a.prior_mean + r0 * d.S * f4
                a.prior_std = a.prior_std + r0 * d.S * f4
                a.entropy, a.action = ptr(ent) + r0 * f4, ptr(act) + r0 * d.A * f4
                if save:
                    a.sv_actor = a.sv_actor + r0 * d.Hd * f4
                    a.sv_act_stats = a.sv_act_stats + r0 * 4 * d.A * f4
                    a.sv_x, a.sv_gates = a.sv_x + r0 * d.Be * f4, a.sv_gates + r0 * 4 * d.Be * f4
                    a.sv_p = a.sv_p + r0 * d.Hd * f4
                cabi.check(lib.bd_imagine_forward_scan(C.byref(a), cabi.stream()))
        if save:        # the entropy estimate of all Hm x N rows: off the recurrence (bd_actor_entropy), outside the scan's span
            self._entropy_estimate(noise, ent, Hm, N)
        self._img_split_rows = H1 * N
        return ifeat, ent, act

    def _entropy_estimate(self, noise, ent: torch.Tensor, Hm: int, N: int) -> None:
        """SampleDist.entropy of every imagined action (src/models.py:725-733) from the (mean, std) the scan left in
        sv_act_stats; explicit draws (parity) or, perf mode (noise["entropy"] is None), draws generated in the kernel."""
        d = self.d
        if noise.get("entropy") is not None:
            cabi.check(lib.bd_actor_entropy(ptr(noise["entropy"]), ptr(self._buf["sv_act_stats"]), ptr(ent), Hm, N, d.A,
                                            d.n_entropy, cabi.stream()))
        else:
            cabi.check(lib.bd_actor_entropy_rng(self.rng_seed, self._rng_entropy_step, self.RNG_STREAMS["entropy"],
                                                ptr(self._buf["sv_act_stats"]), ptr(ent), Hm, N, d.A, d.n_entropy, cabi.stream()))

    # ------------------------------------------------------------------------------------------ train step
    def plan(self, belief: torch.Tensor, state: torch.Tensor, horizon: int, iters: int, candidates: int, top: int,
             eps_action: torch.Tensor, eps_state: torch.Tensor, trace: Optional[list] = None) -> torch.Tensor:
        """MPCPlanner.forward (src/planner.py:28-90) on the current stream: `iters` x (bd_plan_rollout, bd_cem_refit).
        belief (B,Be), state (B,S); eps_action (iters,H,B,candidates,A); eps_state (iters,H,B*candidates,S).
        Returns the action-belief mean of every planning step, (H,B,A); row 0 is the planner's answer."""
        d, pk = self.d, self.pk
        tm = lambda n: self.W("transition_model", n)
        B = belief.shape[0]
        rows = B * candidates
        mean = self.buf("plan_mean", horizon, B, d.A)
        std = self.buf("plan_std", horizon, B, d.A)
        mean.zero_()                                    # q(a_t:t+H) ~ N(0, I), src/planner.py:41-46
        std.fill_(1.0)
        actions = self.buf("plan_actions", horizon, rows, d.A)
        returns = self.buf("plan_returns", rows)
        a = cabi.PlanArgs()
        a.rows, a.H, a.cand, a.Be, a.S, a.A, a.Hd = rows, horizon, candidates, d.Be, d.S, d.A, d.Hd
        a.w_embed_s, a.w_embed_a, a.b_embed = ptr(pk["embed_s"]), ptr(pk["embed_a"]), ptr(tm("fc_embed_state_action.0.bias"))
        a.w_ir, a.w_iz, a.w_in = ptr(pk["ir"]), ptr(pk["iz"]), ptr(pk["in"])
        a.w_hr, a.w_hz, a.w_hn = ptr(pk["hr"]), ptr(pk["hz"]), ptr(pk["hn"])
        a.b_ih, a.b_hh = ptr(tm("rnn.bias_ih")), ptr(tm("rnn.bias_hh"))
        a.w_p1, a.b_p1 = ptr(pk["p1"]), ptr(tm("belief_prior.model.0.bias"))
        a.w_p2m, a.w_p2s, a.b_p2 = ptr(pk["p2m"]), ptr(pk["p2s"]), ptr(tm("belief_prior.model.2.bias"))
        for l in range(DENSE_LAYERS + 1):
            a.w_r[l] = ptr(pk[f"rew{l}"])
            a.b_r[l] = ptr(self.W("reward_model", f"model.{2 * l}.bias"))
        a.min_std = self.hp["min_std_dev"]
        a.init_belief, a.init_state = ptr(belief), ptr(state)
        a.act_mean, a.act_std, a.actions = ptr(mean), ptr(std), ptr(actions)
        # Reward model inside the rollout (one return per candidate leaves the CU) once the candidate tiles fill the
        # chip; below that the rollout is latency bound, so it writes [h'; s'] and the reward model runs as one dense
        # chain over all H x rows rows (measured at B=1: 0.98 -> see DESIGN.md; BD_PLAN_FUSE=0/1 forces either form).
        fuse_env = os.environ.get("BD_PLAN_FUSE", "")
        fuse = (rows >= 16 * 256) if fuse_env == "" else fuse_env != "0"
        F = d.Be + d.S
        if fuse:
            a.returns, a.feat = ptr(returns), None
        else:
            feat = self.buf("plan_feat", horizon * rows, F)
            a.returns, a.feat = None, ptr(feat)
        st = cabi.stream()
        for it in range(iters):
            a.eps_action, a.eps_state = ptr(eps_action[it]), ptr(eps_state[it])
            cabi.check(lib.bd_plan_rollout(C.byref(a), st))
            if fuse:
                ret, steps = returns, 1
            else:
                ret, _, _ = self.dense_forward("reward_model", "rew", "plan_rew", feat, F, horizon * rows, 1)
                steps = horizon
            if trace is not None:
                trace.append(ret.view(steps, rows).sum(dim=0))
            cabi.check(lib.bd_cem_refit(ptr(ret), steps, ptr(actions), horizon, B, candidates, top, d.A, ptr(mean),
                                        ptr(std), st))
        return mean

    @property
    def rng_seed(self) -> int:
        if self._rng_seed is None:
            self.set_noise_seed(int(torch.initial_seed()))
        return self._rng_seed

    def set_noise_seed(self, seed: int) -> None:
        """Key of the perf-mode noise generator (default: torch.initial_seed() at the first draw); ranks get distinct keys."""
        self._rng_seed = (int(seed) + 0x9E3779B97F4A7C15 * (self.dp.rank + 1)) & 0xFFFFFFFFFFFFFFFF
        self._rng_step = {"wm": 0, "bh": 0}

    # noise streams of the perf-mode generator (csrc/bd_rng.h: counter = (index, stream id, step))
    RNG_STREAMS = {"obs_post": 1, "action": 2, "img_prior": 3, "entropy": 4, "obs_prior": 5}

    def make_noise(self, B: int, part: str = "all") -> Dict[str, torch.Tensor]:
        """Noise of one step in perf mode (parity tests pass explicit arrays): ONE bd_rng_fill launch per phase on the
        current stream -- Philox4x32-10, counter-based, keyed by the process seed and the step index, so a run is
        reproducible whatever the stream timing.  part: "wm" (observe scan), "bh" (imagination) or "all".
        Not drawn at all: the prior-state sample of the observe step (src/models.py:256 draws it, nothing on the training
        path reads it) and the 100-sample entropy draws -- bd_actor_entropy_rng generates those inside the estimator
        ("entropy": None), 13.7 MB per step at configs[1] that never exist in HBM."""
        d = self.d
        T, N, Hm = d.T, d.T * B, d.Hm
        shapes = {}
        if part in ("all", "wm"):
            shapes.update(obs_post=(T, B, d.S))
        if part in ("all", "bh"):
            shapes.update(action=(Hm, N, d.A), img_prior=(Hm, N, d.S))
        exp_kind = ("obs_post", "img_prior") if d.categorical else ()       # the sampler's Exp(1) variates, one per class
        out: Dict[str, Optional[torch.Tensor]] = {}
        a = cabi.RngFillArgs()
        a.n, a.seed, a.step = len(shapes), self.rng_seed, self._rng_step[part if part != "all" else "wm"]
        for i, (k, shp) in enumerate(shapes.items()):
            t = out[k] = self.buf("noise_" + k, *shp)
            a.t[i] = cabi.RngTensor(t.data_ptr(), t.numel(), cabi.BD_RNG_EXPONENTIAL if k in exp_kind else cabi.BD_RNG_NORMAL,
                                    self.RNG_STREAMS[k])
        cabi.check(lib.bd_rng_fill(C.byref(a), cabi.stream()))
        if part in ("all", "wm"):
            out["obs_prior"] = None
            self._rng_step["wm"] += 1
        if part in ("all", "bh"):
            out["entropy"] = None
            self._rng_entropy_step = self._rng_step["bh"]
            self._rng_step["bh"] += 1
        return out

    def _optimizer_step_or_defer(self, span: str, group: str, slot: int, lr: float, red_ws: torch.Tensor) -> None:
        """Actor / critic optimiser step of the behaviour phase.  Data-parallel + pipelined: queued and issued by the NEXT
        train_step right after its dynamics phase (or by join()).  All ranks share ONE communicator, whose collectives
        execute in issue order; issued here, actor k's all-reduce (ready at the end of behaviour learning k) would sit in
        front of the KL / world-model all-reduces of step k+1 and stall dynamics learning k+1 half-way.  Deferred, the
        order per host step is kl, model (k+1), actor, critic (k): each is ready by the time its turn comes."""
        if self.pipeline and self.defer_opt:
            self._pending_opt.append((span, group, slot, lr, red_ws, torch.cuda.current_stream(), self._cur_rec))
            return
        with self.span(span):
            self.optimizer_step(group, slot, lr, red_ws)

    def flush_optimizers(self) -> None:
        """Issue the actor / critic optimiser steps the data-parallel schedule holds back by one host step.  They contain
        the gradient all-reduces, i.e. this is a COLLECTIVE: it runs inside train_step, logs() (also when a lazy log dict
        is read), update_critic() and update_belief_and_act(), which every rank calls at the same points of the
        collect-update loop (src/main.py:103-143).  join() and the modules' forward() never issue collectives, so a call
        only one rank makes (e.g. rank 0 rendering a video with observation_model, src/main.py:244) cannot reorder the
        ranks' collectives on the shared communicator."""
        pend, self._pending_opt = self._pending_opt, []
        for span, group, slot, lr, red_ws, stream, rec in pend:
            with torch.cuda.stream(stream):
                with self.span(span):
                    self.optimizer_step(group, slot, lr, red_ws, rec)

    _flush_pending_opt = flush_optimizers

    def join(self) -> None:
        """Order everything the pipeline streams have been given before later work on the caller's stream (stream
        ordering only: no host synchronisation, no collectives)."""
        if self.pipeline:
            cur = torch.cuda.current_stream()
            cur.wait_stream(self._s_wm)
            cur.wait_stream(self._s_bh)
            cur.wait_stream(self._side)

    def train_step(self, batch: Dict[str, torch.Tensor], noise: Optional[Dict[str, torch.Tensor]] = None,
                   sync_logs: bool = True) -> Dict[str, float]:
        """One Dreamer.train_step (src/dreamer.py:253-393) on this rank's batch shard.

        batch: observations (L,B,O), actions (L,B,A), rewards (L,B), nonterminals (L,B,1) -- device fp32,
        contiguous, time-major as ExperienceReplay.sample returns them (src/memory.py:87-104).
        With the pipeline on, the two halves are queued on the engine's own streams and the call returns without
        ordering them against the caller's stream; `logs()` / `join()` do that (sync_logs=True calls logs())."""
        obs = batch["observations"]
        B = obs.shape[1]
        self._timer_tick += 1
        lazy = sync_logs == "lazy"
        rec = self._cur_rec = self._new_record() if lazy else None
        if not self.pipeline:
            if noise is None:
                noise = self.make_noise(B)
            feat = self._dynamics_phase(batch, noise, "")
            self._behaviour_phase(feat, noise, obs.shape[0] - 1, B, self.red_ws, None)
            return self._finish_step(rec, sync_logs)
        cur = torch.cuda.current_stream()
        s_wm, s_bh = self._s_wm, self._s_bh
        par = self._parity
        self._parity ^= 1
        s_wm.wait_stream(cur)                       # the batch (and explicit noise) were produced on the caller's stream
        for t in batch.values():
            t.record_stream(s_wm)
        if noise is not None:
            for t in noise.values():
                t.record_stream(s_wm)
                t.record_stream(s_bh)
        with torch.cuda.stream(s_wm):
            if self._ev_bh_done[par] is not None:   # feat[par] was last read by behaviour learning two steps ago
                s_wm.wait_event(self._ev_bh_done[par])
            nz = noise if noise is not None else self.make_noise(B, "wm")
            feat = self._dynamics_phase(batch, nz, f"p{par}_")
            ev_wm_done = torch.cuda.Event()
            ev_wm_done.record(s_wm)
        # A caller that does not wait for the logs runs many steps ahead of the GPU, and its sampler recycles a small ring
        # of batch buffers (ExperienceReplay._out: 4 deep): order whatever the caller enqueues next on ITS stream behind
        # dynamics learning of two steps ago, so that a gather can never overwrite a batch that has not been consumed.
        self._wm_done_hist.append(ev_wm_done)
        if len(self._wm_done_hist) > 2:
            ev_old = self._wm_done_hist.pop(0)
            if os.environ.get("BD_BATCH_GUARD", "1") != "0":      # "0": diagnosis only (tests show the race without it)
                cur.wait_event(ev_old)
        self._flush_pending_opt()                   # actor / critic updates of the previous step (data-parallel runs)
        with torch.cuda.stream(s_bh):
            s_bh.wait_event(ev_wm_done)
            nz = noise if noise is not None else self.make_noise(B, "bh")
            self._behaviour_phase(feat, nz, obs.shape[0] - 1, B, self.red_ws_bh, par)
            self._ev_bh_done[par] = torch.cuda.Event()
            self._ev_bh_done[par].record(s_bh)
        # Host throttle.  Enqueueing a step takes ~1.1 ms of host time against ~2.8 ms on the GPU, so a caller that does not
        # read the logs gets further ahead with every step; at ~8-10 steps (500-600 launches in flight) the HIP runtime
        # blocks the host for 7-15 ms while it retires a whole batch of commands, and the streams run dry around each of
        # those drains (3.7-4.0 ms steps, tools/step_trace.py).  Waiting here for the behaviour phase of `host_ahead` steps
        # ago keeps the lead small and constant instead: no drains, every step at the steady-state rate from the fifth on.
        if self.host_ahead > 0:
            self._bh_hist.append(self._ev_bh_done[par])
            if len(self._bh_hist) > self.host_ahead:
                self._bh_hist.pop(0).synchronize()
        return self._finish_step(rec, sync_logs)

    def _finish_step(self, rec: Optional[_LogRecord], sync_logs):
        if rec is not None:                      # sync_logs == "lazy"
            rec.counts = dict(self._counts)
            self._cur_rec = None
            out = LazyLogs(self, rec)
            rec.owner = weakref.ref(out)
            return out
        return self.logs() if sync_logs else {}

    def world_model_step(self, batch: Dict[str, torch.Tensor], noise: Optional[Dict[str, torch.Tensor]] = None,
                         sync_logs: bool = True) -> Dict[str, float]:
        """Planet.train_step (src/planet.py:310-368): dynamics learning only, on the caller's stream."""
        self.flush_optimizers()
        self.join()
        obs = batch["observations"]
        T, B = obs.shape[0] - 1, obs.shape[1]
        self._timer_tick += 1
        self._dynamics_phase(batch, noise if noise is not None else self.make_noise(B, "wm"), "")
        self._counts = dict(N=T * B, Mi=1, S=(self.d.cat_D if self.d.categorical else self.d.S),
                            sum_form=int(self.hp["kl_balance"] == -1))
        if not sync_logs:
            return {}
        return {k: v for k, v in self.logs().items() if k in ("observation_loss", "reward_loss", "kl_loss", "model_loss",
                                                              "grad_norm_model")}

    def _dynamics_phase(self, batch: Dict[str, torch.Tensor], noise: Dict[str, torch.Tensor], feat_tag: str) -> torch.Tensor:
        """Dynamics learning (src/dreamer.py:263-302) on the current stream; returns the posterior features."""
        d, hp = self.d, self.hp
        obs, actions, rewards, nonterm = (batch[k] for k in ("observations", "actions", "rewards", "nonterminals"))
        L, B = obs.shape[0], obs.shape[1]
        T = L - 1
        N = T * B
        F = d.Be + d.S
        W = self.world_size
        st = cabi.stream()
        sc, ws = ptr(self.scalars), ptr(self.red_ws)

        # ======================= dynamics learning (dreamer.py:263-302) =======================
        obs_t = obs[1:].reshape(N, d.O)                     # targets and encoder input
        with self.span("encoder_fwd"):
            emb, pre_emb = self.encode_pixels(obs[1:].reshape(N, 3, 64, 64)) if self.pixel else self.encode(obs_t, N)
        init_belief = self.buf("init_belief", B, d.Be).zero_()
        init_state = self.buf("init_state", B, d.S).zero_()
        feat, qm, qs = self.observe(actions[:-1], nonterm[:-1], pre_emb, noise["obs_post"], init_belief, init_state, T, B,
                                    feat_tag=feat_tag)
        cat = d.categorical
        with self.span("wm_heads_fwd"):
            _, pm, ps = self.prior_head(feat, N, None if cat else noise.get("obs_prior"))
            if self.pixel:
                om_out, om_acts, om_layers = self.conv.decode(feat).view(N, d.O), None, None
                obs_t = self.conv.acts_enc[0].view(N, d.O)      # the same images in the NHWC order of the prediction
            else:
                om_out, om_acts, om_layers = self.dense_forward("observation_model", "obs", "om", feat, F, N, d.O,
                                                                sidx=self._buf[feat_tag + "sidx"] if cat else None)
            rw_out, rw_acts, rw_layers = self.dense_forward("reward_model", "rew", "rw", feat, F, N, 1,
                                                            sidx=self._buf[feat_tag + "sidx"] if cat else None)

        inv_rows = self.dp.mean_grad_scale(N)
        dc_state = None
        if d.use_discount:      # _discount_loss (src/dreamer.py:239-251): Bernoulli(logits).log_prob(nonterminal), weight 5
            dc_out, dc_acts, dc_layers = self.dense_forward("discount_model", "dsc", "dc", feat, F, N, 1,
                                                            sidx=self._buf[feat_tag + "sidx"] if cat else None)
            d_dc = self.buf("d_dc_out", N, 1)
            cabi.check(lib.bd_bernoulli_nll(ptr(dc_out), ptr(nonterm[:-1]), N, hp["discount_weight"] * inv_rows, ptr(d_dc), sc,
                                            SLOT_DISC, ws, st))
            dc_state = (d_dc, dc_acts, dc_layers)
        d_om, d_rw = self.buf("d_om_out", N, d.O), self.buf("d_rw_out", N, 1)
        cabi.check(lib.bd_normal_nll(ptr(om_out), d.O, ptr(obs_t), d.O, N, d.O, inv_rows, ptr(d_om), d.O, sc, SLOT_OBS, ws, st))
        cabi.check(lib.bd_normal_nll(ptr(rw_out), 1, ptr(rewards[:-1]), 1, N, 1, inv_rows, ptr(d_rw), 1, sc, SLOT_REW, ws, st))
        sum_form = int(hp["kl_balance"] == -1)
        kl_n = d.cat_D if cat else d.S                      # KL terms per row: one per factor / per state dimension
        if cat:     # Categorical branch of _kl_loss (src/dreamer.py:102-106,119-144): qm / pm are the logits
            cabi.check(lib.bd_kl_categorical_forward(ptr(qm), ptr(pm), N, d.cat_D, d.cat_C, hp["free_nats"], sum_form, sc,
                                                     SLOT_KL, ws, st))
        else:
            cabi.check(lib.bd_kl_forward(ptr(qm), ptr(qs), ptr(pm), ptr(ps), N, d.S, hp["free_nats"], sum_form, sc, SLOT_KL,
                                         ws, st))
        if W > 1 and not sum_form:                          # the free-nats clamp acts on the GLOBAL mean
            self._kl_local = self.scalars[SLOT_KL:SLOT_KL + 1].clone()
            self._allreduce(self.scalars[SLOT_KL:SLOT_KL + 1], "model")
        kl_inv = 1.0 / (N * kl_n * W) if not sum_form else 1.0 / (N * W)
        kl_w = hp["kl_loss_weight"] * (1.0 / W if sum_form else 1.0)
        if cat:
            dqm, dpm = self.buf("dql", N, d.S), self.buf("dpl", N, d.S)
            dqs = dps = None
            cabi.check(lib.bd_kl_categorical_backward(ptr(qm), ptr(pm), N, d.cat_D, d.cat_C, hp["free_nats"], hp["kl_balance"],
                                                      kl_w, kl_inv, sc, SLOT_KL, ptr(dqm), ptr(dpm), st))
        else:
            dqm, dqs = self.buf("dqm", N, d.S), self.buf("dqs", N, d.S)
            dpm, dps = self.buf("dpm", N, d.S), self.buf("dps", N, d.S)
            cabi.check(lib.bd_kl_backward(ptr(qm), ptr(qs), ptr(pm), ptr(ps), N, d.S, hp["free_nats"], hp["kl_balance"],
                                          kl_w, kl_inv, sc, SLOT_KL, ptr(dqm), ptr(dqs), ptr(dpm), ptr(dps), st))

        # ---- backward of the world model ----
        dfeat = self.buf("dfeat", N, F)
        rw_dpre = [self.buf(f"rw_dpre{l}", N, d.Hd) for l in range(DENSE_LAYERS)] + [d_rw]
        if self.pixel:
            self.mlp_backward(N, d_rw, 1, rw_layers, rw_acts + [None], rw_dpre[:-1] + [None], din0=dfeat, ld0=F, w0=F)
            with self.span("decoder_bwd"):
                # the decoder's weight-gradient GEMMs (60 % of the pass) are ready here: they run on their own stream
                # underneath the observe-scan backward (52 CUs) instead of after it
                self.conv.backward_decoder(d_om.view(N, 64, 64, 3), feat, dfeat, self._wbatch["model_early"])
                self._s_early.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(self._s_early):
                    self._wbatch["model_early"].run()
        else:
            om_dpre = [self.buf(f"om_dpre{l}", N, d.Hd) for l in range(DENSE_LAYERS)] + [d_om]
            self.mlp_backward(N, d_om, d.O, om_layers, om_acts + [None], om_dpre[:-1] + [None], din0=dfeat, ld0=F, w0=F)
            self.mlp_backward(N, d_rw, 1, rw_layers, rw_acts + [None], rw_dpre[:-1] + [None], din0=dfeat, ld0=F, w0=F,
                              accumulate=True)
        dc_dpre = None
        if dc_state is not None:
            d_dc, dc_acts, dc_layers = dc_state
            dc_dpre = [self.buf(f"dc_dpre{l}", N, d.Hd) for l in range(DENSE_LAYERS)] + [d_dc]
            self.mlp_backward(N, d_dc, 1, dc_layers, dc_acts + [None], dc_dpre[:-1] + [None], din0=dfeat, ld0=F, w0=F,
                              accumulate=True)
        self._dc_wgrad = (dc_dpre, dc_state[1]) if dc_state is not None else None
        # prior head: KL gradient on (mean, std) -> belief part of dfeat
        p_out, p_hid = self._buf["p_out"], self._buf["p_hid"]
        HO = d.head_out
        d_p_hid = self.buf("d_p_hid", N, d.Hd)
        if cat:
            d_p_out = dpm                                   # the KL gradient w.r.t. the prior logits IS d(head output)
        else:
            d_p_out = self.buf("d_p_out", N, HO)
            cabi.check(lib.bd_gauss_head_backward(ptr(p_out), None, None, ptr(dpm), ptr(dps), N, d.S, ptr(d_p_out), st))
        tm = lambda n: self.W("transition_model", n)
        p_layers = [("p1", None, d.Hd, d.Be, cabi.ACT_ELU), ("p2", None, HO, d.Hd, cabi.ACT_NONE)]
        self.mlp_backward(N, d_p_out, HO, p_layers, [p_hid, None], [d_p_hid, None], din0=dfeat, ld0=F, w0=d.Be,
                          accumulate=True)
        # recurrence
        d_e, d_gi, d_gh = self.buf("d_embed_pre", N, d.Be), self.buf("d_gi", N, 3 * d.Be), self.buf("d_gh", N, 3 * d.Be)
        d_q1, d_q2 = self.buf("d_q1_pre", N, d.Hd), self.buf("d_q2_out", N, HO)
        pk = self.pk
        if cat:
            b = cabi.ObserveCatBwdArgs()
            b.T, b.B, b.Be, b.D, b.C, b.A, b.Hd = T, B, d.Be, d.cat_D, d.cat_C, d.A, d.Hd
            b.wt_embed_s = ptr(pk["embed_s.T"])
            b.wt_ir, b.wt_iz, b.wt_in = ptr(pk["ir.T"]), ptr(pk["iz.T"]), ptr(pk["in.T"])
            b.wt_hr, b.wt_hz, b.wt_hn = ptr(pk["hr.T"]), ptr(pk["hz.T"]), ptr(pk["hn.T"])
            b.wt_q1h, b.wt_q2 = ptr(pk["q1h.T"]), ptr(pk["q2.T"])
            b.init_belief, b.nonterm = ptr(init_belief), ptr(nonterm[:-1])
            b.feat, b.post_logits = ptr(feat), ptr(qm)
            b.sv_x, b.sv_gates, b.sv_q = ptr(self._buf["sv_x"]), ptr(self._buf["sv_gates"]), ptr(self._buf["sv_q"])
            b.dfeat, b.dpost_logits = ptr(dfeat), ptr(dqm)
            b.d_embed_pre, b.d_gi, b.d_gh, b.d_q1_pre, b.d_q2_out = ptr(d_e), ptr(d_gi), ptr(d_gh), ptr(d_q1), ptr(d_q2)
            with self.span("observe_bwd"):
                Cm = self._cat_cluster(B)
                if Cm:
                    ws_c = self._cat_cluster_ws(B, Cm)
                    cabi.check(lib.bd_observe_cat_backward_cluster(C.byref(b), Cm, ptr(ws_c), ws_c.numel(), st))
                else:
                    cabi.check(lib.bd_observe_cat_backward(C.byref(b), st))
        else:
            self._observe_backward_gaussian(T, B, init_belief, nonterm, noise, feat, qs, dfeat, dqm, dqs, d_e, d_gi, d_gh,
                                            d_q1, d_q2, st)
        self._dynamics_tail(batch, emb, obs_t, feat, init_belief, actions, d_e, d_gi, d_gh, d_q1, d_q2, d_p_hid, d_p_out,
                            p_hid, rw_dpre, rw_acts, om_dpre if (not self.pixel) else None,
                            om_acts if (not self.pixel) else None, N, B, F)
        return feat

    def _observe_backward_gaussian(self, T, B, init_belief, nonterm, noise, feat, qs, dfeat, dqm, dqs, d_e, d_gi, d_gh, d_q1,
                                   d_q2, st) -> None:
        d, hp, pk = self.d, self.hp, self.pk
        b = cabi.ObserveBwdArgs()
        b.T, b.B, b.Be, b.S, b.A, b.Hd = T, B, d.Be, d.S, d.A, d.Hd
        b.wt_embed_s = ptr(pk["embed_s.T"])
        b.wt_ir, b.wt_iz, b.wt_in = ptr(pk["ir.T"]), ptr(pk["iz.T"]), ptr(pk["in.T"])
        b.wt_hr, b.wt_hz, b.wt_hn = ptr(pk["hr.T"]), ptr(pk["hz.T"]), ptr(pk["hn.T"])
        b.wt_q1h, b.wt_q2m, b.wt_q2s = ptr(pk["q1h.T"]), ptr(pk["q2m.T"]), ptr(pk["q2s.T"])
        b.init_belief, b.nonterm, b.eps_post = ptr(init_belief), ptr(nonterm[:-1]), ptr(noise["obs_post"])
        b.feat, b.post_std = ptr(feat), ptr(qs)
        b.sv_x, b.sv_gates, b.sv_q = ptr(self._buf["sv_x"]), ptr(self._buf["sv_gates"]), ptr(self._buf["sv_q"])
        b.dfeat, b.dpost_mean, b.dpost_std = ptr(dfeat), ptr(dqm), ptr(dqs)
        b.min_std = hp["min_std_dev"]
        b.d_embed_pre, b.d_gi, b.d_gh, b.d_q1_pre, b.d_q2_out = ptr(d_e), ptr(d_gi), ptr(d_gh), ptr(d_q1), ptr(d_q2)
        with self.span("observe_bwd"):
            if self._cluster_ok(B):
                ws_c = self._cluster_ws(B)
                cabi.check(lib.bd_observe_backward_cluster(C.byref(b), ptr(ws_c), ws_c.numel(), st))
            else:
                cabi.check(lib.bd_observe_backward(C.byref(b), st))

    def _dynamics_tail(self, batch, emb, obs_t, feat, init_belief, actions, d_e, d_gi, d_gh, d_q1, d_q2, d_p_hid, d_p_out,
                       p_hid, rw_dpre, rw_acts, om_dpre, om_acts, N, B, F) -> None:
        """Encoder backward, the grouped weight-gradient launch of the world model, its optimiser step."""
        d, hp = self.d, self.hp
        HO = d.head_out
        # encoder (+ hoisted projection as its last layer)
        if self.pixel:
            d_emb = self.buf("d_emb", N, d.E)
            self.mlp_backward(N, d_q1, d.Hd, [("q1e", None, d.Hd, d.E, cabi.ACT_NONE)], [None], [None], din0=d_emb, ld0=d.E,
                              w0=d.E)
            with self.span("encoder_bwd"):
                self.conv.backward_encoder(d_emb, self._wbatch["model"])
        else:
            enc_layers = self._dense_spec("encoder", "enc", d.O, d.E) + [("q1e", None, d.Hd, d.E, cabi.ACT_NONE)]
            enc_acts = [self._buf[f"enc_act{l}"] for l in range(DENSE_LAYERS)]
            enc_dpre = [self.buf(f"enc_dpre{l}", N, d.Hd) for l in range(DENSE_LAYERS)] + [self.buf("d_emb", N, d.E)]
            self.mlp_backward(N, d_q1, d.Hd, enc_layers, enc_acts + [None, None], enc_dpre + [None])

        # ---- weight gradients (into the flat model gradient buffer): one grouped launch ----
        Gt = lambda n: self.G("transition_model", n)
        wb = self._wbatch["model"]
        wb.add(d_gi, 3 * d.Be, self._buf["sv_x"], d.Be, N, 3 * d.Be, d.Be, Gt("rnn.weight_ih"), d.Be, Gt("rnn.bias_ih"))
        # previous belief: the initial belief for step 0 (rows < B), the belief part of feat[t-1] afterwards
        wb.add(d_gh, 3 * d.Be, init_belief, d.Be, N, 3 * d.Be, d.Be, Gt("rnn.weight_hh"), d.Be, Gt("rnn.bias_hh"),
               act2=feat, lda2=F, M1=B)
        gWe = Gt("fc_embed_state_action.0.weight")
        wb.add(d_e, d.Be, self._buf["sv_s"], d.S, N, d.Be, d.S, gWe, d.S + d.A, Gt("fc_embed_state_action.0.bias"))
        wb.add(d_e, d.Be, actions[:-1], d.A, N, d.Be, d.A, gWe[:, d.S:], d.S + d.A)
        wb.add(d_p_hid, d.Hd, feat, F, N, d.Hd, d.Be, Gt("belief_prior.model.0.weight"), d.Be, Gt("belief_prior.model.0.bias"))
        wb.add(d_p_out, HO, p_hid, d.Hd, N, HO, d.Hd, Gt("belief_prior.model.2.weight"), d.Hd,
               Gt("belief_prior.model.2.bias"))
        gWq1 = Gt("belief_posterior.model.0.weight")
        wb.add(d_q1, d.Hd, feat, F, N, d.Hd, d.Be, gWq1, d.Be + d.E, Gt("belief_posterior.model.0.bias"))
        wb.add(d_q1, d.Hd, emb, d.E, N, d.Hd, d.E, gWq1[:, d.Be:], d.Be + d.E)
        wb.add(d_q2, HO, self._buf["sv_q"], d.Hd, N, HO, d.Hd, Gt("belief_posterior.model.2.weight"), d.Hd,
               Gt("belief_posterior.model.2.bias"))
        dense_sizes = lambda i, o: [i] + [d.Hd] * DENSE_LAYERS + [o]
        self._dense_wgrads(wb, "reward_model", N, rw_dpre, feat, F, rw_acts, dense_sizes(F, 1))
        if self._dc_wgrad is not None:
            self._dense_wgrads(wb, "discount_model", N, self._dc_wgrad[0], feat, F, self._dc_wgrad[1], dense_sizes(F, 1))
        if not self.pixel:      # (pixel mode: the conv stacks queued their weight gradients themselves)
            self._dense_wgrads(wb, "observation_model", N, om_dpre, feat, F, om_acts, dense_sizes(F, d.O))
            self._dense_wgrads(wb, "encoder", N, enc_dpre, obs_t, d.O, enc_acts, dense_sizes(d.O, d.E))
        with self.span("wgrad_model"):
            wb.run()
        if self.pixel:
            torch.cuda.current_stream().wait_stream(self._s_early)      # decoder weight gradients (queued above)
        if self.pipeline and self._ev_bh_wm_free is not None:
            # the previous step's imagination / reward-head kernels may still be reading the weights Adam overwrites
            torch.cuda.current_stream().wait_event(self._ev_bh_wm_free)
        with self.span("opt_model"):
            self.optimizer_step("model", SLOT_GN_MODEL, hp["model_learning_rate"])

    def _behaviour_phase(self, feat: torch.Tensor, noise: Dict[str, torch.Tensor], T: int, B: int,
                         red_ws: torch.Tensor, par: Optional[int]) -> None:
        """Behaviour learning (src/dreamer.py:308-391) on the current stream: imagination with the post-update world
        model (packed by the model optimiser step) from the detached posteriors, actor update, critic update."""
        d, hp, pk = self.d, self.hp, self.pk
        Hm = d.Hm
        N = T * B
        Mi = Hm * N
        F = d.Be + d.S
        st = cabi.stream()
        sc, ws = ptr(self.scalars), ptr(red_ws)
        sum_form = int(hp["kl_balance"] == -1)
        # Pipelined (par = step parity): the critic update of this step runs on a third stream underneath the NEXT
        # step's imagination (it only needs the imagined features and the lambda-returns, and nothing on the actor's
        # chain reads the critic -- the value head there is the critic TARGET), so both are double-buffered by parity.
        ptag = "" if par is None else f"p{par}_"
        if par is not None and self._ev_cr_done[par] is not None:
            torch.cuda.current_stream().wait_event(self._ev_cr_done[par])     # critic of two steps ago: last reader
        ifeat, ent, act = self.imagine(feat, N, Hm, noise, feat_tag=ptag, split=self.img_split,
                                       start_sidx=self._buf[ptag + "sidx"] if d.categorical else None)
        r0 = self._img_split_rows
        isidx = self._buf[ptag + "isidx"] if d.categorical else None
        with self.span("img_heads_fwd"):
            if not r0:
                r_out, r_acts, r_layers = self.dense_forward("reward_model", "rew", "ir", ifeat, F, Mi, 1, sidx=isidx)
                v_out, v_acts, v_layers = self.dense_forward("critic_target", "tgt", "iv", ifeat, F, Mi, 1, sidx=isidx)
            else:       # rows of the first time segment on a helper stream, under the second segment of the rollout
                with torch.cuda.stream(self._s_heads):
                    self._s_heads.wait_event(self._ev_img_half)
                    self.dense_forward("reward_model", "rew", "ir", ifeat, F, Mi, 1, rows=(0, r0))
                    self.dense_forward("critic_target", "tgt", "iv", ifeat, F, Mi, 1, rows=(0, r0))
                r_out, r_acts, r_layers = self.dense_forward("reward_model", "rew", "ir", ifeat, F, Mi, 1, rows=(r0, Mi))
                v_out, v_acts, v_layers = self.dense_forward("critic_target", "tgt", "iv", ifeat, F, Mi, 1, rows=(r0, Mi))
                torch.cuda.current_stream().wait_stream(self._s_heads)
        returns = self.buf(ptag + "returns", Mi)
        cabi.check(lib.bd_lambda_return_forward(ptr(r_out), ptr(v_out), Hm, N, hp["discount"], hp["disclam"], ptr(returns), st))
        cabi.check(lib.bd_sum(ptr(returns), Mi, sc, SLOT_RET, ws, st))
        cabi.check(lib.bd_sum(ptr(ent), Mi, sc, SLOT_ENT, ws, st))
        inv_mi = self.dp.mean_grad_scale(Mi)
        wts = dret = None
        if d.use_discount:
            # Cumulative discount weights of the actor objective and the value loss (src/dreamer.py:323-326,346-351,
            # 374-379): discount * round(sigmoid(discount head)), trajectory 0 forced to 1 at every step (the reference's
            # `discount_arr[:, 0, 0] = 1.0`), cumprod over time.  No gradient flows through them (round; frozen head):
            # a handful of elementwise torch ops on (Hm, N) -- this switch is off by default (conf/config.yaml:76).
            dl, _, _ = self.dense_forward("discount_model", "dsc", "idc", ifeat, F, Mi, 1, sidx=isidx)
            arr = hp["discount"] * torch.round(torch.sigmoid(dl.view(Hm, N)))
            arr[:, 0] = 1.0
            wts = self.buf(ptag + "disc_w", Mi)
            wts.view(Hm, N).copy_(torch.cumprod(arr, 0))
            ew_ = hp["entropy_weight"] if hp["entropy_weight"] != -1 else 0.0
            self.scalars[SLOT_WOBJ] = (wts * (returns + ew_ * ent)).sum()
            dret = self.buf("dret_w", Mi)
            torch.mul(wts, -inv_mi, out=dret)
        if par is not None:
            ev_ret = torch.cuda.Event()
            ev_ret.record(torch.cuda.current_stream())
            with torch.cuda.stream(self._side):
                self._side.wait_event(ev_ret)
                self._critic_phase(ifeat, returns, Mi, F, inv_mi, self.red_ws_side, isidx, wts)
                self._ev_cr_done[par] = torch.cuda.Event()
                self._ev_cr_done[par].record(self._side)
        elif self.overlap_critic:    # fork: critic phase on the side stream, actor backward continues here
            self._side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(self._side):
                self._critic_phase(ifeat, returns, Mi, F, inv_mi, self.red_ws_side, isidx, wts)
        d_r, d_v = self.buf("d_ir_out", Mi), self.buf("d_iv_out", Mi)
        cabi.check(lib.bd_lambda_return_backward(ptr(dret), -inv_mi, Hm, N, hp["discount"], hp["disclam"], ptr(d_r), ptr(d_v), st))
        difeat = self.buf("difeat", Mi, F)
        with self.span("img_heads_bwd"):
            self.mlp_backward(Mi, d_r, 1, r_layers, r_acts + [None], None, din0=difeat, ld0=F, w0=F)
            self.mlp_backward(Mi, d_v, 1, v_layers, v_acts + [None], None, din0=difeat, ld0=F, w0=F, accumulate=True)
        if d.categorical:
            c = cabi.ImagineCatBwdArgs()
            c.N, c.Hm, c.Be, c.D, c.C, c.A, c.Hd = N, Hm, d.Be, d.cat_D, d.cat_C, d.A, d.Hd
            c.wt_p2 = ptr(pk["p2.T"])
            c.prior_logits = ptr(self._buf["iprior_logits"])
        else:
            c = cabi.ImagineBwdArgs()
            c.N, c.Hm, c.Be, c.S, c.A, c.Hd = N, Hm, d.Be, d.S, d.A, d.Hd
            c.wt_p2m, c.wt_p2s = ptr(pk["p2m.T"]), ptr(pk["p2s.T"])
            c.prior_std, c.eps_prior, c.min_std = ptr(self._buf["iprior_std"]), ptr(noise["img_prior"]), hp["min_std_dev"]
        c.wt_embed_s, c.wt_embed_a = ptr(pk["embed_s.T"]), ptr(pk["embed_a.T"])
        c.wt_ir, c.wt_iz, c.wt_in = ptr(pk["ir.T"]), ptr(pk["iz.T"]), ptr(pk["in.T"])
        c.wt_hr, c.wt_hz, c.wt_hn = ptr(pk["hr.T"]), ptr(pk["hz.T"]), ptr(pk["hn.T"])
        c.wt_p1 = ptr(pk["p1.T"])
        for l in range(1, DENSE_LAYERS):
            c.wt_a[l - 1] = ptr(pk[f"a{l}.T"])
        c.wt_a4m, c.wt_a4s = ptr(pk["a4m.T"]), ptr(pk["a4s.T"])
        c.start_feat, c.feat, c.action = ptr(feat), ptr(ifeat), ptr(act)
        c.eps_action = ptr(noise["action"])
        sv_actor = self._buf["sv_actor"]
        c.sv_actor, c.sv_act_stats = ptr(sv_actor), ptr(self._buf["sv_act_stats"])
        c.sv_x, c.sv_gates, c.sv_p = ptr(self._buf["isv_x"]), ptr(self._buf["isv_gates"]), ptr(self._buf["isv_p"])
        c.dfeat = ptr(difeat)
        c.dentropy = -hp["entropy_weight"] * inv_mi if hp["entropy_weight"] != -1 else 0.0
        d_apre, d_aout = self.buf("d_actor_pre", DENSE_LAYERS, Mi, d.Hd), self.buf("d_actor_out", Mi, 2 * d.A)
        # the actor's hidden layers leave the backward scan (their result feeds nothing on the recurrence: detached
        # input) and run as one dense chain over all Hm x N rows from d_actor_out
        actor_chain = os.environ.get("BD_ACTOR_BWD_CHAIN", "1") == "1"
        c.d_actor_pre, c.d_actor_out = (None if actor_chain else ptr(d_apre)), ptr(d_aout)
        c.ent_weight = ptr(wts)
        with self.span("imagine_bwd"):
            cabi.check((lib.bd_imagine_cat_backward if d.categorical else lib.bd_imagine_backward)(C.byref(c), st))
        if actor_chain:
            a_layers = [("a0h", None, d.Hd, F, cabi.ACT_ELU)] + [(f"a{l}", None, d.Hd, d.Hd, cabi.ACT_ELU)
                                                                  for l in range(1, DENSE_LAYERS)] + \
                       [("a4", None, 2 * d.A, d.Hd, cabi.ACT_NONE)]
            with self.span("actor_hidden_bwd"):
                self.mlp_backward(Mi, d_aout, 2 * d.A, a_layers, [sv_actor[l] for l in range(DENSE_LAYERS)] + [None],
                                  [d_apre[l] for l in range(DENSE_LAYERS)] + [None])
        if self.pipeline:       # last reader of the world model in this step
            self._ev_bh_wm_free = torch.cuda.Event()
            self._ev_bh_wm_free.record(torch.cuda.current_stream())
        Ga = lambda n: self.G("actor", n)
        wa = self._wbatch["actor"]
        # layer 0 input = [h_t; s_t]: start features for t = 0 (rows < N), imagined features of step t-1 afterwards
        wa.add(d_apre[0], d.Hd, feat, F, Mi, d.Hd, F, Ga("model.0.weight"), F, Ga("model.0.bias"), act2=ifeat, lda2=F, M1=N)
        for l in range(1, DENSE_LAYERS):
            wa.add(d_apre[l], d.Hd, sv_actor[l - 1], d.Hd, Mi, d.Hd, d.Hd, Ga(f"model.{2 * l}.weight"), d.Hd,
                   Ga(f"model.{2 * l}.bias"))
        wa.add(d_aout, 2 * d.A, sv_actor[DENSE_LAYERS - 1], d.Hd, Mi, 2 * d.A, d.Hd,
               Ga(f"model.{2 * DENSE_LAYERS}.weight"), d.Hd, Ga(f"model.{2 * DENSE_LAYERS}.bias"))
        with self.span("wgrad_actor"):
            wa.run()
        self._optimizer_step_or_defer("opt_actor", "actor", SLOT_GN_ACTOR, hp["actor_learning_rate"], red_ws)

        if par is not None:
            pass
        elif not self.overlap_critic:
            self._critic_phase(ifeat, returns, Mi, F, inv_mi, red_ws, isidx, wts)
        else:
            torch.cuda.current_stream().wait_stream(self._side)     # join before the next step reuses ifeat / returns
        self._counts = dict(N=N, Mi=Mi, S=(d.cat_D if d.categorical else d.S), sum_form=sum_form)

    def _critic_phase(self, ifeat, returns, Mi: int, F: int, inv_mi: float, red_ws: torch.Tensor, isidx=None, wts=None) -> None:
        """Critic update (src/dreamer.py:370-391) on the current stream: forward on the detached imagined features,
        Normal(v, 1) NLL against the detached returns, dgrad chain, grouped weight gradients, clip + Adam, re-pack."""
        d, hp = self.d, self.hp
        st, sc = cabi.stream(), ptr(self.scalars)
        with self.span("critic_fwd_bwd"):
            c_out, c_acts, c_layers = self.dense_forward("critic", "cri", "ic", ifeat, F, Mi, 1, sidx=isidx)
            d_c = self.buf("d_ic_out", Mi, 1)
            cabi.check(lib.bd_normal_nll(ptr(c_out), 1, ptr(returns), 1, Mi, 1, inv_mi, ptr(d_c), 1, sc, SLOT_VAL, ptr(red_ws), st))
            if wts is not None:     # use_discount=True: -(discount * log_prob).mean() (src/dreamer.py:378-379)
                d_c.view(Mi).mul_(wts)
                diff = c_out.view(Mi) - returns
                self.scalars[SLOT_VAL] = (wts * (0.5 * diff * diff + 0.9189385332046727)).sum()
            c_dpre = [self.buf(f"ic_dpre{l}", Mi, d.Hd) for l in range(DENSE_LAYERS)] + [d_c]
            self.mlp_backward(Mi, d_c, 1, c_layers, c_acts + [None], c_dpre[:-1] + [None])
        wc = self._wbatch["critic"]
        self._dense_wgrads(wc, "critic", Mi, c_dpre, ifeat, F, c_acts, [F] + [d.Hd] * DENSE_LAYERS + [1])
        with self.span("wgrad_critic"):
            wc.run()
        self._optimizer_step_or_defer("opt_critic", "critic", SLOT_GN_CRITIC, hp["value_learning_rate"], red_ws)

    def logs(self) -> Dict[str, float]:
        """One D2H copy of the scalar board -> the reference's log dict (src/dreamer.py:293-296,359-360,383).
        Values are this rank's shard means (fp32 arithmetic as in the reference)."""
        self.flush_optimizers()
        self.join()
        if self._obs_ws is not None and self._obs_err_off is not None:
            both = torch.cat([self.scalars, self._cluster_error_word()]).cpu().numpy()
            self._raise_on_cluster_error(int(both[N_SLOTS:].view(np.uint32)[0]))
            s = both[:N_SLOTS].astype(np.float32)
        else:
            s = self.scalars.cpu().numpy().astype(np.float32)
        return self._logs_from(s, self._counts)

    def _logs_from(self, s: np.ndarray, c: dict) -> Dict[str, float]:
        hp = self.hp
        f32 = np.float32
        N, Mi = f32(c["N"]), f32(c["Mi"])
        obs, rew = s[SLOT_OBS] / N, s[SLOT_REW] / N
        if c["sum_form"]:
            kl = s[SLOT_KL] / N
        else:
            mean = s[SLOT_KL] / f32(c["N"] * c["S"] * self.world_size)
            x = np.maximum(mean, f32(hp["free_nats"]))
            kl = f32(hp["kl_balance"]) * x + f32(1 - hp["kl_balance"]) * x
        ew = f32(hp["entropy_weight"]) if hp["entropy_weight"] != -1 else f32(0)
        model_loss = obs + rew + kl * f32(hp["kl_loss_weight"])
        objective = s[SLOT_RET] + ew * s[SLOT_ENT]
        out = {}
        if self.d.use_discount:         # src/dreamer.py:287-290, 346-352
            disc = s[SLOT_DISC] / N
            model_loss = model_loss + disc * f32(hp["discount_weight"])
            out["discount_loss"] = float(disc)
            objective = s[SLOT_WOBJ]
        out.update({
            "observation_loss": float(obs), "reward_loss": float(rew), "kl_loss": float(kl),
            "model_loss": float(model_loss),
            "actor_loss": float(-objective / Mi),
            "policy_entropy": float(s[SLOT_ENT] / Mi),
            "value_loss": float(s[SLOT_VAL] / Mi),
            "grad_norm_model": float(math.sqrt(s[SLOT_GN_MODEL])), "grad_norm_actor": float(math.sqrt(s[SLOT_GN_ACTOR])),
            "grad_norm_critic": float(math.sqrt(s[SLOT_GN_CRITIC])),
        })
        return out
