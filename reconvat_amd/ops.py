"""Host-side operators: thin ``torch.autograd.Function`` wrappers over the C ABI of
libreconvat_hip.so.  PyTorch is used for device memory, streams and the autograd tape only; every
numeric kernel is a hand-written HIP kernel (reconvat_amd/csrc).  There is no CPU fallback.

Internal activation layout is NHWC ([B, H=time, W=bins, C]); a tensor handed to a conv may be a channel
slice (view) of a wider buffer -- the pixel stride is taken from ``stride(2)``.
"""
import ctypes
import os
import sys

import torch
from torch.autograd import Function

from . import _lib, plans
from ._lib import call, invoke, ptr, stream, need_gpu

SLOPE = 0.01            # F.leaky_relu default (model/UNet_onset.py:197-198)
BN_MOMENTUM = 0.1       # model/UNet_onset.py:183
BN_EPS = 1e-5

_EPOCH = [0]
# RV_DETERMINISTIC=1 (or ops.DETERMINISTIC[0] = True): bit-reproducible PARAMETER GRADIENTS.  The data path -- forward passes and
# input-gradient chains, hence every loss term and running statistic -- is free of fp32 atomics in every mode; what is order-dependent by
# default are the folds onto the gradient bucket: the table-driven weight-gradient reductions (several passes of a step target the same
# layer: fp32 atomics), the atomic split-K of the parameter-gradient GEMMs with the bias row sums riding on them, the column-sum bias
# gradients.  In this mode the weight-gradient reductions run per layer in stream order (fixed tree, read-modify-write), the
# parameter-gradient GEMMs use the in-order (park + fold) split-K of the data path, column sums the two-pass ordered form.  Slower (see
# `deterministic_ms_per_step` in bench.py's line); the default stays the atomic folds.
DETERMINISTIC = [os.environ.get('RV_DETERMINISTIC') == '1']
_DIRECT = [False]


class ZeroArena:
    """Bump allocator over ONE device buffer that is cleared once per training step: reduction kernels that need
    zero-initialised scratch (BatchNorm sums) take slices instead of issuing a memset per call.  Armed by
    TrainStep (also under hipGraph capture: the clear is the first node and the slice order is deterministic);
    unarmed, `take` falls back to torch.zeros."""

    def __init__(self):
        self.buf = None
        self.off = 0
        self.armed = False

    def begin_step(self, device, nbytes=8 << 20):
        if self.buf is None or self.buf.device != device or self.buf.numel() * 8 < nbytes:
            self.buf = torch.zeros(nbytes // 8, device=device, dtype=torch.float64)
        else:
            self.buf.zero_()
        self.off = 0
        self.armed = True

    def end_step(self):
        self.armed = False

    def take(self, n_doubles, device):
        if self.armed and self.buf.device == device and self.off + n_doubles <= self.buf.numel():
            out = self.buf[self.off:self.off + n_doubles]
            self.off += (n_doubles + 1) & ~1
            return out
        return torch.zeros(n_doubles, device=device, dtype=torch.float64)


ARENA = ZeroArena()


def bn_ws_doubles(c):
    """fp64 elements of one BatchNorm reduction workspace (rv_bn_workspace_bytes: replicated [2C] sums)."""
    return _lib.load().rv_bn_workspace_bytes(c) // 8

_BN_DEFER = [None]

# Two-stream VAT (model._Base._vat_two_streams).  Off by default: TrainStep switches it on once the weights are packed
# by the one-launch plan and the conv autotuner has run (neither tolerates a concurrent second stream).
DUAL_STREAM = [False]
_SIDE = {}


def side_stream(device, idx=0):
    """The idx-th side stream of `device` (created on first use; idx 0 is the U-Net schedule's, the Onsets&Frames
    schedule also uses idx 1)."""
    key = (device.type, device.index, idx)
    if key not in _SIDE:
        _SIDE[key] = torch.cuda.Stream(device=device)
    return _SIDE[key]


class deferred_bn_updates:
    """Inside this context train-mode BatchNorm forwards use batch statistics but do NOT touch running_mean /
    running_var / num_batches_tracked; the skipped updates are collected and can be replayed (any number of
    times, at the point of the reference's update sequence where they belong) with ``apply()``."""

    def __enter__(self):
        self.prev = _BN_DEFER[0]
        self.items = []
        _BN_DEFER[0] = self.items
        return self

    def __exit__(self, *exc):
        _BN_DEFER[0] = self.prev

    def apply(self):
        for rm, rv, nbt, coef, c in self.items:
            call('rv_bn_running_update', ptr(rm), ptr(rv), ptr(nbt), ptr(coef), c, BN_MOMENTUM, stream())


_REPLAY_KEEP = []          # pinned host tables of captured replays must outlive their hipGraph (captures outside any keep_scope)
_REPLAY_POOL = []          # pre-allocated pinned staging buffers for capture

# Objects a hipGraph capture pins (host tables a captured copy node re-reads at every replay, their device copies) must live exactly
# as long as the graph.  A capture runs inside `keep_scope(owner_list)`: everything pinned during it lands in the OWNER's list
# (TrainStep._keep), which is dropped together with the graph -- so a process may capture and drop any number of steps without
# growing.  Outside a scope the module-level lists above keep the objects for the life of the process (one-off captures in tests).
_KEEP_SCOPE = [None]


class keep_scope:
    def __init__(self, owner_list):
        self.owner = owner_list

    def __enter__(self):
        self.prev = _KEEP_SCOPE[0]
        _KEEP_SCOPE[0] = self.owner
        return self.owner

    def __exit__(self, *exc):
        _KEEP_SCOPE[0] = self.prev


def _keep(obj, fallback):
    (_KEEP_SCOPE[0] if _KEEP_SCOPE[0] is not None else fallback).append(obj)


def prepare_replay_pool(n=4):
    """Pinned staging buffers for replay_bn_updates under hipGraph capture (call outside capture)."""
    while len(_REPLAY_POOL) < n:
        _REPLAY_POOL.append(torch.empty(4096, dtype=torch.int64).pin_memory())


def replay_bn_updates(pendings, device):
    """Apply the deferred BatchNorm updates of several passes -- `pendings`: deferred_bn_updates objects in the order of
    the reference's update sequence (an object may appear twice) -- with ONE launch: per layer the updates are chained
    in that order inside the kernel (rv_bn_running_update_table)."""
    layers, order = {}, []
    for pend in pendings:
        for rm, rv, nbt, coef, c in pend.items:
            key = ptr(rm)
            if key not in layers:
                layers[key] = [ptr(rm), ptr(rv), ptr(nbt) or 0, c, []]
                order.append(key)
            layers[key][4].append(ptr(coef))
    if not order:
        return
    head, coefs = [], []
    for key in order:
        rm, rv, nbt, c, lst = layers[key]
        head += [rm, rv, nbt, c, len(coefs), len(lst)]
        coefs += lst
    words = head + coefs
    # pinned staging buffers are allocated ahead of time (no host allocation may happen under hipGraph capture); a
    # captured copy node re-reads its buffer at every replay, so buffers used under capture are never recycled
    capturing = torch.cuda.is_current_stream_capturing()
    if capturing:
        if not _REPLAY_POOL:
            raise RuntimeError('replay_bn_updates: no pinned staging buffer left for hipGraph capture')
        host = _REPLAY_POOL.pop()
        _keep(host, _REPLAY_KEEP)
        assert len(words) <= host.numel(), 'BatchNorm replay table larger than the staging buffer'
        host[:len(words)] = torch.tensor(words, dtype=torch.int64)
    else:
        prepare_replay_pool()
        host = torch.tensor(words, dtype=torch.int64).pin_memory()      # (the pinned allocator is stream-aware)
    table = host[:len(words)].to(device, non_blocking=True)
    call('rv_bn_running_update_table', ptr(table), len(order), BN_MOMENTUM, stream())
    table.record_stream(torch.cuda.current_stream(device))
    if capturing:
        _keep(table, _REPLAY_KEEP)             # keep the device copy's memory out of the graph pool's reuse



class direct_param_grads:
    """Context manager: while active, conv / BatchNorm backward kernels ACCUMULATE their parameter gradients
    straight into the pre-allocated ``param.grad`` buffers (FlatAdam's flat gradient bucket) and hand autograd
    ``None`` for them -- this removes one add kernel per parameter per graph.  Only for loops that consume
    ``param.grad`` (TrainStep); ``torch.autograd.grad(..., params)`` would not see these gradients."""

    def __enter__(self):
        self.prev = _DIRECT[0]
        _DIRECT[0] = True

    def __exit__(self, *exc):
        _DIRECT[0] = self.prev


# (flat gradient bucket, {side stream: its twin}): backward kernels that run on a side stream accumulate into that stream's
# twin, so concurrent backward chains never read-modify-write the same addresses; FlatAdam.merge_side_grads() adds them in.
SIDE_GRADS = [None]


def _grad_buf(t):
    if not (_DIRECT[0] and t is not None and t.is_leaf and t.requires_grad and t.grad is not None):
        return None
    sg = SIDE_GRADS[0]
    if sg is not None and DUAL_STREAM[0]:
        main, twins = sg                      # twins: {side stream handle: that stream's twin of the flat bucket}
        twin = twins.get(torch.cuda.current_stream(t.device).cuda_stream)
        if twin is not None:
            off = (t.grad.data_ptr() - main.data_ptr()) // 4
            return twin[off:off + t.grad.numel()].view_as(t.grad)
    return t.grad


def drop_packs_of(flat):
    """Forget every packed copy whose source weight lives in the storage of `flat` (FlatAdam's parameter buffer): the cache holds its
    source tensors, so the packs of a model that is gone would otherwise keep its whole parameter buffer alive."""
    sp = flat.untyped_storage().data_ptr()
    for k in [k for k, v in _pack_cache.items() if v[2].untyped_storage().data_ptr() == sp]:
        del _pack_cache[k]


def drop_packs_of_params(params):
    """Forget every packed copy whose source weight shares storage with one of `params` (a model that is going away: _Base.__del__).
    Together with drop_packs_of (FlatAdam.__del__) this is what bounds the cache: entries die with their OWNER, not with their age --
    a live but idle model (an evaluation copy, the frame model while the onset model trains) keeps its packs."""
    sps = set()
    for p in params:
        try:
            sps.add(p.untyped_storage().data_ptr())
        except Exception:  # noqa: BLE001 -- a parameter without storage (meta / already freed)
            pass
    for k in [k for k, v in _pack_cache.items() if v[2].untyped_storage().data_ptr() in sps]:
        del _pack_cache[k]


def invalidate_weight_cache():
    """Call after any out-of-band parameter update (custom optimiser step, load_state_dict)."""
    _EPOCH[0] += 1


# --------------------------------------------------------------------------------------------
# geometry helpers
# --------------------------------------------------------------------------------------------
def _geom(t):
    b, h, w, c = t.shape
    if t.is_contiguous():
        return b, h, w, c, c
    ld = t.stride(2)
    ok = (c == 1 or t.stride(3) == 1) and (h == 1 or t.stride(1) == w * ld) and (b == 1 or t.stride(0) == h * w * ld)
    if not ok or w == 1:
        raise RuntimeError(f'expected an NHWC tensor or a channel slice of one, got shape {tuple(t.shape)} '
                           f'strides {t.stride()}')
    return b, h, w, c, ld


def _new(b, h, w, c, like):
    return torch.empty((b, h, w, c), device=like.device, dtype=torch.float32)


# --------------------------------------------------------------------------------------------
# convolution kinds: how forward / input-gradient / weight-gradient map onto the C ABI
# --------------------------------------------------------------------------------------------
#   kind   PyTorch module                         weight layout
#   c3     Conv2d(k=3, p=1)                       [Cout, Cin, 3, 3]
#   t3     ConvTranspose2d(k=3, p=1)              [Cin, Cout, 3, 3]
#   c1     Conv2d(k=1)                            [Cout, Cin, 1, 1]
#   down   Conv2d(k=2, s=2)                       [Cout, Cin, 2, 2]
#   up     ConvTranspose2d(k=2, s=2, output_size) [Cin, Cout, 2, 2]
def _channels(kind, w):
    if kind in ('t3', 'up'):
        return w.shape[0], w.shape[1]
    return w.shape[1], w.shape[0]


_pack_cache = {}
# Entries die with their OWNER (drop_packs_of / drop_packs_of_params).  Weights packed through the functional API or by modules that
# are not _Base models have no owner: a bounded LRU is the fallback for those (re-inserting a key moves it to the young end; a live
# model's entries are re-inserted by PackPlan.run() every step and by every lazy repack, so they are never the old end for long).
_PACK_CACHE_MAX = int(os.environ.get('RV_PACK_CACHE_MAX', '2048'))


def _pack_put(key, val):
    _pack_cache.pop(key, None)
    _pack_cache[key] = val
    while len(_pack_cache) > _PACK_CACHE_MAX:
        del _pack_cache[next(iter(_pack_cache))]


def _pack(kind, w, which):
    """Packed weight fragments for `which` in {'fwd', 'dgrad'} (cached per weight version)."""
    key = (w.data_ptr(), kind, which)
    tag = (_EPOCH[0], w._version, tuple(w.shape))
    hit = _pack_cache.get(key)
    if hit is not None and hit[0] == tag:
        return hit[1]
    cin, cout = _channels(kind, w)
    scatter = 0
    if kind == 'c3':
        args = (9, cin, cout, 9, cin * 9, 0) if which == 'fwd' else (9, cout, cin, cin * 9, 9, 1)
    elif kind == 't3':
        args = (9, cin, cout, cout * 9, 9, 1) if which == 'fwd' else (9, cout, cin, 9, cout * 9, 0)
    elif kind == 'c1':
        args = (1, cin, cout, 1, cin, 0) if which == 'fwd' else (1, cout, cin, cin, 1, 0)
    elif kind == 'down':
        if which == 'fwd':
            args = (4, cin, cout, 4, cin * 4, 0)
        else:
            args, scatter = (1, cout, 4 * cin, cin * 4, 4, 0), cin
    elif kind == 'up':
        if which == 'fwd':
            args, scatter = (1, cin, 4 * cout, cout * 4, 4, 0), cout
        else:
            args = (4, cout, cin, 4, cout * 4, 0)
    else:
        raise ValueError(kind)
    taps, kdim, ndim, s_k, s_n, flip = args
    plain = 1 if (kdim % 8 != 0 or (ndim <= 2 and not scatter)) else 0
    lib = _lib.load()
    n = lib.rv_packed_weight_floats(taps, kdim, ndim)
    out = torch.empty(n, device=w.device, dtype=torch.float32)
    call('rv_pack_weights', ptr(w), ptr(out), taps, kdim, ndim, s_k, s_n, flip, scatter, plain, stream())
    _pack_put(key, (tag, out, w, (taps, kdim, ndim, s_k, s_n, flip, scatter, plain)))
    return out


def _pack_rel(rel, f):
    """rel^T [31, F] for the attention kernels (cached per weight version; part of the PackPlan table)."""
    key = (rel.data_ptr(), 'rel', 'T')
    tag = (_EPOCH[0], rel._version, tuple(rel.shape))
    hit = _pack_cache.get(key)
    if hit is not None and hit[0] == tag:
        return hit[1]
    args = (1, 31, f, 1, 31, 0, 0, 1)            # out[w*F + c] = rel[c*31 + w]
    out = torch.empty(31 * f, device=rel.device, dtype=torch.float32)
    call('rv_pack_weights', ptr(rel), ptr(out), *args, stream())
    _pack_put(key, (tag, out, rel, args))
    return out


def _lin_t(w2d):
    """Contiguous transposed copy [K, N] of a row-major linear weight view w2d [N, K] (cached per weight version; part of
    the PackPlan table).  The input-gradient GEMM dY @ W then reads BOTH operands along their contiguous dimension
    (rv_gemm's LDS-DMA kernel) instead of gathering W column-wise."""
    n, k = w2d.shape
    assert w2d.stride(1) == 1 and w2d.stride(0) == k
    key = (w2d.data_ptr(), 'lin', 'T', n, k)
    tag = (_EPOCH[0], w2d._version, (n, k))
    hit = _pack_cache.get(key)
    if hit is not None and hit[0] == tag:
        return hit[1].view(k, n)
    args = (1, k, n, 1, k, 0, 0, 1)              # out[kk*n + nn] = w[kk + nn*k]
    out = torch.empty(k * n, device=w2d.device, dtype=torch.float32)
    call('rv_pack_weights', ptr(w2d), ptr(out), *args, stream())
    _pack_put(key, (tag, out, w2d, args))
    return out.view(k, n)


class PackPlan:
    """Every packed weight currently in the cache, repacked by ONE kernel launch (rv_pack_table_run).  Build it after
    a warm-up step has populated the cache, call ``run()`` right after each optimiser step: the cache entries are
    re-tagged as fresh, so the ~100 lazy per-layer pack launches of the next step disappear.  Entries created later
    stay on the lazy path."""

    def __init__(self, device):
        import ctypes
        lib = _lib.load()
        # (the cache is pruned by OWNER -- FlatAdam.__del__ -> drop_packs_of, _Base.__del__ -> drop_packs_of_params --, never by age)
        self.entries = [(k, v) for k, v in _pack_cache.items() if v[2].device == device]
        self.count = len(self.entries)
        if not self.count:
            return
        nbytes = lib.rv_pack_table_entry_bytes() * self.count
        host = (ctypes.c_char * nbytes)()
        total = 0
        for i, (_, (_, out, w, a)) in enumerate(self.entries):
            total = lib.rv_pack_table_fill(host, i, ptr(w), ptr(out), *a)
            if total < 0:
                raise RuntimeError('rv_pack_table_fill: ' + _lib.last_error())
        self.total_blocks = total
        self.table = torch.frombuffer(host, dtype=torch.uint8).clone().to(device)

    def run(self):
        if not self.count:
            return
        call('rv_pack_table_run', ptr(self.table), self.count, self.total_blocks, stream())
        for key, (_, out, w, a) in self.entries:
            _pack_put(key, ((_EPOCH[0], w._version, tuple(w.shape)), out, w, a))


_FWD_MODE = {'c3': 0, 't3': 0, 'c1': 1, 'down': 2, 'up': 3}
_DGRAD_MODE = {'c3': 0, 't3': 0, 'c1': 1, 'down': 3, 'up': 2}


_algo_cache = {}
_tune_us = {}             # ('conv' | 'wgrad', key) -> microseconds the on-line tuner measured for its choice
_tune_top = {}            # ('conv', key) -> [(us, algo)] the three fastest candidates (tools/tune_plans.py --in-situ re-ranks near ties in the step)
_algo_unchecked = set()    # table entries borrowed from another batch size: legality is checked by their first launch
# 'table' (default): the shipped per-shape plan table (reconvat_amd/plans.py, tuned_plans.json) -- what bench.py, the scripts and
# the -m gpu tests all run; True (RV_AUTOTUNE=1): time every legal tile on the first eager call of a shape (how the table is
# made, tools/tune_plans.py); False (RV_AUTOTUNE=0): library default tiles.
AUTOTUNE = plans.default_mode()


# BASELINE config 3 as an OPT-IN experiment (default off; the shipped / headline configuration is fp32 end to end): bf16 operands on
# the matrix pipe (fp32 accumulation, statistics, losses, master weights) for the convolutions of the FINAL graphs -- the convs
# whose weights are live; the no_grad target pass and the detached XI*d pass of the power iteration always stay fp32 (at XI = 1e-6
# the perturbation is below one bf16 ulp).  'fwd': forward convs, 'bwd': input- and weight-gradient convs.  Measured
# (tools/bf16_emulation.py): 'fwd' breaks the 1e-3 forward parity bar (frame2 moves by 18 %), 'bwd' leaves every
# loss term and posteriorgram bit-identical and moves the gradients by 0.6 % -- so only 'bwd' is meant to be used.
BF16 = {'fwd': False, 'bwd': False}
ALGO_BF16 = 1 << 20
# algo families (bits 8..11) that are fp32 Winograd tiles -- conv3x3_wino_k (0x6 / 0xA / 0xC) and the software-pipelined conv3x3_wino2_k
# (0x8 / 0x9 / 0xB / 0xD): no bf16-operand form, a bf16 launch of such a shape runs the library-default direct tile instead
WINOGRAD_FAMILIES = (6, 8, 9, 10, 11, 12, 13, 14)


class bf16_final_graphs:
    def __init__(self, fwd=False, bwd=True):
        self.want = {'fwd': bool(fwd), 'bwd': bool(bwd)}

    def __enter__(self):
        self.prev = dict(BF16)
        BF16.update(self.want)
        return self

    def __exit__(self, *exc):
        BF16.update(self.prev)


def _conv_call(mode, x, ild, bb, h, wd, cin, out, old, ho, wo, cout, wpack, bias, stats=None, bnbwd=None, accumulate=False, bf16=False):
    """rv_conv_fwd with a per-shape choice between the LDS-free and the LDS/DMA-pipelined 3x3 kernel.  The first
    eager call of a shape times both (HIP events on the launch stream) and caches the winner; under hipGraph
    capture an untuned shape uses the library default.  ``stats`` (fp64 [2*cout], zeroed): the conv also leaves the
    BatchNorm batch statistics of its output there (fused epilogue of the persistent kernel, else a statistics pass
    -- the tuner times whichever the candidate implies).  ``bnbwd`` = (z, coef, slope): the call is an input gradient
    and ``stats`` receives the backward reduction of the BatchNorm whose output gradient is being produced."""
    args = (mode, ptr(x), ild, bb, h, wd, cin, ptr(out), old, ho, wo, cout, ptr(wpack), ptr(bias), 1 if accumulate else 0)
    if bnbwd is not None:
        bz, bcoef, bslope = bnbwd
        tail = (ptr(bz), _geom(bz)[4], ptr(bcoef), float(bslope))
    else:
        tail = (None, 0, None, 0.0)
    algo = 0
    if os.environ.get('RV_FORCE_ALGO') and (mode == 0 or os.environ.get('RV_FORCE_ALGO_ALL')):   # kernel experiments
        algo = int(os.environ['RV_FORCE_ALGO'], 0)
    elif AUTOTUNE and cin % 8 == 0 and (cout > 2 or mode == 3):     # (the small-channel VALU kernels have one form)
        base_key = (mode, bb, h, wd, cin, cout, ild, old, stats is not None, bnbwd is not None)
        # the bf16-operand variant of a shape is its own cache entry in every mode: its tile may differ from the fp32 one (the fp32
        # Winograd tiles have no bf16 form), and a bf16 launch must never overwrite what the fp32 launches of the same shape run
        key = base_key + ('bf16',) if bf16 else base_key
        algo = _algo_cache.get(key, -1)
        if algo < 0 and AUTOTUNE == 'table':
            hit = plans.lookup_conv(base_key)
            algo = hit[0] if hit is not None else 0
            if bf16 and (algo >> 8) & 15 in WINOGRAD_FAMILIES:
                algo = 0                        # fp32 Winograd tile: the bf16 launch of this shape runs the library-default direct tile
            _algo_cache[key] = algo
            if hit is not None and not hit[1] and algo != 0:
                _algo_unchecked.add(key)
        if algo < 0:
            if torch.cuda.is_current_stream_capturing():
                algo = 0
            else:
                best, algo = None, 0
                st = torch.cuda.current_stream()
                lib = _lib.load()
                ntile_n = (4 * cout if mode == 3 else cout + 15) // 16
                bfbit = ALGO_BF16 if bf16 else 0
                scratch = ptr(torch.zeros_like(stats)) if stats is not None else None
                targs = args
                if accumulate:          # the timing runs must not touch the buffer that is being accumulated into
                    tmp_out = torch.empty(bb * ho * wo * old, device=out.device, dtype=torch.float32)
                    targs = args[:7] + (ptr(tmp_out),) + args[8:14] + (0,)
                cands = [1, 2] if mode == 0 else [0]
                for nt in (1, 2, 3, 4):
                    if ntile_n % nt:
                        continue
                    cands += [0x100 | nt << 4 | mt for mt in (1, 2, 4)]     # direct kernel (every mode)
                    if mode != 0:                                           # ... with the K loop split over the four waves (deep layers)
                        cands += [0x500 | nt << 4 | mt for mt in (1, 2, 4) if nt * mt <= 4]
                    if mode == 0:                                           # persistent LDS kernel, 4 / 8 / 16 waves
                        cands += [0x200 | nt << 4 | mt for mt in (1, 2, 4, 8)]
                        cands += [0x300 | nt << 4 | mt for mt in (1, 2, 4)]
                        cands += [0x400 | nt << 4 | mt for mt in (1, 2, 4)]
                        cands += [0x700 | nt << 4 | mt for mt in (1, 2, 3, 4, 5, 6)]     # 12 waves: 3 .. 18 tiles per SIMD and band
                if mode == 0:
                    # rows per band: the tile slots of a (waves, MTW) pair hold th_max rows; fewer rows trade padding for a band
                    # count that divides over the 256 workgroup slots (the deep layers have only a few bands per image)
                    extra = []
                    for cand in cands:
                        nwv = {2: 4, 3: 8, 4: 16, 7: 12}.get(cand >> 8)
                        if nwv is None:
                            continue
                        th_max = min(h, (cand & 15) * 16 * nwv // wd)
                        if th_max < 2:
                            continue
                        nb0 = -(-h // th_max)
                        for nb in range(nb0, nb0 + 4):
                            th = -(-h // nb)
                            if 0 < th < th_max and th < 256:
                                extra.append(th << 12 | cand)
                    cands += sorted(set(extra))
                    if cin % 16 == 0 and os.environ.get('RV_TUNE_WINOGRAD', '1') != '0':
                        # Winograd F(2x2,3x3) form (0x6NM: 8 waves; 0xANM / 0xCNM: 8 / 12 waves with the patch read half a chunk at a
                        # time; bands of an even number of rows).  NT = 2 exists in the half-chunk 8-wave form only.
                        wt_ = (wd + 1) // 2
                        for fam, nwv in ((6, 8), (10, 8), (12, 12)):
                            for nt, mt in ((1, 1), (2, 1)):
                                if ntile_n % nt or (nt == 2 and fam != 10):
                                    continue
                                th_max = min(h, 2 * ((nwv * mt * 16) // wt_))
                                ths = [0] + [t for t in range(2, th_max, 2) if -(-h // t) != -(-h // (t + 2))]
                                cands += [t << 12 | fam << 8 | nt << 4 | mt for t in ths]
                        if os.environ.get('RV_TUNE_WINO2', '1') != '0':
                            # ... and its software-pipelined form (conv_wino2.hip, round 5): 0x8NM / 0x9NM = 8 waves with the full / half-chunk patch,
                            # 0xBNM = 4 waves, 0xDNM = 12 waves (half-chunk patch); the taller half of the legal band heights only (short bands lose to their halo)
                            for fam, nwv, tiles in ((8, 8, ((1, 1),)), (9, 8, ((1, 1), (2, 1), (1, 2))), (11, 4, ((1, 2), (2, 1))), (13, 12, ((1, 1),))):
                                for nt, mt in tiles:
                                    if ntile_n % nt:
                                        continue
                                    th_max = min(h, 2 * ((nwv * mt * 16) // wt_))
                                    ths = [0] + [t for t in range(2, th_max, 2) if -(-h // t) != -(-h // (t + 2)) and t >= th_max // 2]
                                    cands += [t << 12 | fam << 8 | nt << 4 | mt for t in ths]
                    fams = os.environ.get('RV_TUNE_FAMILIES')          # experiment: restrict the LDS tile families the tuner may pick
                    if fams:
                        keep = {int(f, 0) for f in fams.split(',')}
                        cands = [c for c in cands if ((c >> 8) & 15) in keep or c in (1, 2) and ((1 if c == 1 else 2) in keep)]
                ranked = []
                for cand in cands:
                    if lib.rv_conv_fwd(*targs, cand | bfbit, scratch, *tail, st.cuda_stream) != 0:
                        continue                                   # tile does not fit this shape
                    t = None
                    for _rep in range(2):                          # best of two bursts of three: less timing noise
                        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                        e0.record(st)
                        for _ in range(3):
                            lib.rv_conv_fwd(*targs, cand | bfbit, scratch, *tail, st.cuda_stream)
                        e1.record(st)
                        e1.synchronize()
                        dt = e0.elapsed_time(e1)
                        t = dt if t is None else min(t, dt)
                    ranked.append((t / 3 * 1e3, cand))
                    if best is None or t < best:
                        best, algo = t, cand
                _algo_cache[key] = algo
                if best is not None:
                    _tune_us[('conv', key)] = best / 3 * 1e3
                    _tune_top[('conv', key)] = sorted(ranked)[:3]
                if os.environ.get('RV_TUNE_LOG') and best is not None:
                    print(f'[tune] conv mode={mode} {cin}->{cout} {h}x{wd} B={bb}: algo={algo:#x} {best / 3 * 1e3:.1f} us', file=sys.stderr)
        if key in _algo_unchecked:
            _algo_unchecked.discard(key)
            if invoke('rv_conv_fwd', *args, algo | (ALGO_BF16 if bf16 else 0), ptr(stats), *tail, stream()) == 0:
                return
            algo = _algo_cache[key] = 0         # the borrowed tile does not fit this batch size: library default
    if bf16 and (algo >> 8) & 15 in WINOGRAD_FAMILIES:
        algo = 0                                # (forced / on-line tuned Winograd tile: no bf16 form -> library-default direct tile)
    if bf16 and algo not in (1,) and (algo >> 8) & 15 != 1:
        algo |= ALGO_BF16                       # (the library ignores the bit outside the persistent 3x3 kernel / 16-channel chunks)
    call('rv_conv_fwd', *args, algo, ptr(stats), *tail, stream())


def conv_forward_into(kind, x, w, b, out, stats=None, bf16=False):
    """out (NHWC view) = conv(x) ; shapes are taken from the views.  stats: see _conv_call."""
    need_gpu(x, w, out)
    bb, h, wd, cin, ild = _geom(x)
    _, ho, wo, cout, old = _geom(out)
    _conv_call(_FWD_MODE[kind], x, ild, bb, h, wd, cin, out, old, ho, wo, cout, _pack(kind, w, 'fwd'), b, stats, bf16=bf16)


def conv_dgrad_into(kind, dy, w, dx, bn_link=None, accumulate=False, bf16=False, sum_ws=None):
    """dx (NHWC view) (+)= input gradient of the conv given dy (NHWC view).  bn_link: the BnLink of the BatchNorm that
    produced the conv's input -- its backward reduction is then computed in this kernel's epilogue."""
    assert not (accumulate and bn_link is not None)
    bb, h, wd, c, ild = _geom(dy)
    _, ho, wo, co, old = _geom(dx)
    stats = bnbwd = None
    if bn_link is not None and bn_link.usable(dx):
        stats, bnbwd = bn_link.ws, (bn_link.z, bn_link.coef, bn_link.slope)
    elif sum_ws is not None:
        stats = sum_ws                           # plain per-channel sums of dx (ColsumLink): the fused statistics epilogue
    _conv_call(_DGRAD_MODE[kind], dy, ild, bb, h, wd, c, dx, old, ho, wo, co, _pack(kind, w, 'dgrad'), None, stats, bnbwd, accumulate,
               bf16=bf16)
    if bnbwd is not None:
        bn_link.ready = True


class GradShare:
    """One activation consumed by several convs (a block's conv1 + skip, the decoder's skip conv): the first
    input-gradient kernel to run writes the shared buffer and hands it to autograd, the others ACCUMULATE into it
    (rv_conv_fwd accumulate=1) and hand autograd nothing -- no add kernel, no extra pass over the tensor.  Create one per
    forward call and pass it to every consumer; all consumers must run on the same stream."""

    def __init__(self):
        self.buf = None

    def dgrad(self, kind, dy, w, shape, bf16=False):
        if self.buf is None:
            self.buf = torch.empty(shape, device=dy.device, dtype=torch.float32)
            conv_dgrad_into(kind, dy, w, self.buf, bf16=bf16)
            return self.buf
        conv_dgrad_into(kind, dy, w, self.buf, accumulate=True, bf16=bf16)
        return None


class ColsumLink:
    """Connects the conv whose INPUT gradient is dY of a 2x2 up-conv (d_block: conv2d consumes the up-conv's output) with that
    up-conv: the column sums of dY -- the up-conv's bias gradient -- are the per-channel sums of what the producer's input-gradient
    kernel stores, which its fused statistics epilogue provides for free (instead of a colsum pass re-reading dY: 20 launches of
    ~17 us per step)."""

    def __init__(self):
        self.sums = None         # fp64 statistics workspace filled by the producer's dgrad
        self.c = 0               # its channel count


class BnLink:
    """Connects a BatchNorm+activation node with the ONE conv that consumes its output, so that the conv's
    input-gradient kernel can produce the BatchNorm's backward reduction in its epilogue (the gradient of the
    BatchNorm output is exactly that conv's dgrad result -- valid only for a single consumer)."""

    def __init__(self):
        self.z = self.coef = self.ws = None
        self.slope = 0.0
        self.ready = False

    def usable(self, dx):
        return self.ws is not None and self.z is not None and tuple(self.z.shape) == tuple(dx.shape)


_wgrad_tuned = set()
_wgrad_plans = {}          # (taps, B, Hv, Wv, Ca, Cb) -> (nw, wgs): what this process pinned in the library (tools/tune_plans.py dumps it)


def _tune_wgrad(lib, mode, taps, u, uld, hu, wu, ca, v, vld, hv, wv, cb, bb, w, s_a, s_b, flip):
    """Per-shape partition of the MFMA weight-gradient kernel (waves per workgroup x workgroups on the chip): the first eager
    call of a shape times the candidates (HIP events on the launch stream, scratch outputs) and pins the winner in the library
    (rv_conv_wgrad_set_plan); under hipGraph capture an untuned shape keeps the library default (8 waves, 256 workgroups)."""
    key = (taps, bb, hv, ca, cb)
    if not AUTOTUNE or key in _wgrad_tuned or ca * cb * taps <= 144 or ca == 1:
        return
    if AUTOTUNE == 'table':
        # host-only (legal under hipGraph capture); runs before the first launch of the shape, i.e. never between a deferred
        # weight-gradient launch of that shape and its table flush
        _wgrad_tuned.add(key)
        plan = plans.lookup_wgrad((taps, bb, hv, wv, ca, cb))
        if plan is not None and lib.rv_conv_wgrad_set_plan(taps, bb, hv, ca, cb, *plan) == 0:
            _wgrad_plans[(taps, bb, hv, wv, ca, cb)] = plan
        return
    if torch.cuda.is_current_stream_capturing():
        return
    _wgrad_tuned.add(key)
    st = torch.cuda.current_stream()
    dw = torch.empty_like(w)
    db = torch.empty(cb, device=w.device, dtype=torch.float32)
    best, choice = None, (0, 0)
    cands = [(8, 256), (8, 512), (4, 256), (4, 512), (8, 128), (8, 1024)]
    if taps == 9 and hv % 2 == 0 and os.environ.get('RV_TUNE_WGRAD_WINO', '1') != '0':
        cands += [(24, 256), (24, 512), (24, 128)]          # nw = 24: eight waves, Winograd F(3x3, 2x2) form (wgrad_wino_k)
    for nw, wgs in cands:
        if lib.rv_conv_wgrad_set_plan(taps, bb, hv, ca, cb, nw, wgs) != 0:
            continue
        nbytes = lib.rv_conv_wgrad_workspace_bytes(taps, bb, hv, ca, cb)
        ws = torch.empty(nbytes // 4, device=w.device, dtype=torch.float32)
        args = (mode, ptr(u), uld, hu, wu, ca, ptr(v), vld, hv, wv, cb, bb, ptr(dw), s_a, s_b, flip, ptr(db), 0, ptr(ws), nbytes,
                st.cuda_stream)
        if lib.rv_conv_wgrad(*args) != 0:
            continue
        t = None
        for _rep in range(2):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(st)
            for _ in range(3):
                lib.rv_conv_wgrad(*args)
            e1.record(st)
            e1.synchronize()
            dt = e0.elapsed_time(e1)
            t = dt if t is None else min(t, dt)
        if best is None or t < best:
            best, choice = t, (nw, wgs)
    lib.rv_conv_wgrad_set_plan(taps, bb, hv, ca, cb, *choice)
    _wgrad_plans[(taps, bb, hv, wv, ca, cb)] = choice
    if best is not None:
        _tune_us[('wgrad', (taps, bb, hv, wv, ca, cb))] = best / 3 * 1e3
    if os.environ.get('RV_TUNE_LOG'):
        print(f'[tune] wgrad taps={taps} {ca}->{cb} {hv}x{wv} B={bb}: nw={choice[0]} wgs={choice[1]} {best / 3 * 1e3:.1f} us', file=sys.stderr)


def _wgrad_geom(kind, x, dy, w, bf16):
    """The pixel-reduction GEMM behind a conv layer's weight gradient: (mode, taps, u, uld, hu, wu, ca, v, vld, hv, wv, cb, bb, s_a, s_b, flip)."""
    bb, h, wd, cin, xld = _geom(x)
    _, ho, wo, cout, yld = _geom(dy)
    if kind == 'up':
        # G[tap][a=co][b=ci] = sum_p dY[2p+tap][co] * X[p][ci]  -> dWt[ci][co][tap]
        taps, mode = 4, 2
        u, uld, hu, wu, ca = dy, yld, ho, wo, cout
        v, vld, hv, wv, cb = x, xld, h, wd, cin
        s_a, s_b, flip = 4, cout * 4, 0
    else:
        mode = {'c3': 0, 't3': 0, 'c1': 1, 'down': 2}[kind]
        taps = {'c3': 9, 't3': 9, 'c1': 1, 'down': 4}[kind]
        u, uld, hu, wu, ca = x, xld, h, wd, cin
        v, vld, hv, wv, cb = dy, yld, ho, wo, cout
        if kind == 't3':
            s_a, s_b, flip = cout * 9, 9, 1
        else:
            s_a, s_b, flip = taps, cin * taps, 0
        if bf16 and taps == 9:
            mode |= 0x100                       # bf16 operands on the matrix pipe (opt-in experiment, see BF16)
    return mode, taps, u, uld, hu, wu, ca, v, vld, hv, wv, cb, bb, s_a, s_b, flip


def conv_wgrad(kind, x, dy, w, want_bias=True, dw_acc=None, db_acc=None, bf16=False, colsum=None):
    """(dw, db) in the PyTorch layouts of `w` / bias; with dw_acc/db_acc the results are ADDED into those buffers."""
    mode, taps, u, uld, hu, wu, ca, v, vld, hv, wv, cb, bb, s_a, s_b, flip = _wgrad_geom(kind, x, dy, w, bf16)
    cout = dy.shape[3]
    acc = 1 if dw_acc is not None else 0
    dw = dw_acc if acc else torch.empty_like(w)
    if not want_bias:
        db = None
    else:
        db = db_acc if acc else torch.empty(cout, device=w.device, dtype=torch.float32)
    bias_ptr = None if kind == 'up' else ptr(db)
    lib = _lib.load()
    merger = _WGRAD_MERGE[0]
    if merger is not None and acc and not DETERMINISTIC[0] and ca > 2 and cb > 2 and \
            merger.submit((_bucket_offset(dw.data_ptr()), mode), (kind, x, dy, w, dw, None if kind == 'up' else db, bf16, torch.cuda.current_stream(w.device))):
        pass                                       # (launched with the same layer's other passes of this step: WgradMerger)
    else:
        _tune_wgrad(lib, mode & 0xff, taps, u, uld, hu, wu, ca, v, vld, hv, wv, cb, bb, w, s_a, s_b, flip)
        nbytes = lib.rv_conv_wgrad_workspace_bytes(taps, bb, hv, ca, cb)
        ws = torch.empty(nbytes // 4, device=w.device, dtype=torch.float32)
        tables = _WGRAD_DEFER[0]
        if tables is not None and acc and not DETERMINISTIC[0] and not (taps == 9 and ca == 1 and cb > 16):
            cur = torch.cuda.current_stream(w.device)
            tab = tables.get(cur.cuda_stream)
            if tab is None:
                tab = tables[cur.cuda_stream] = _WgradTable(w.device, cur)
            rc = invoke('rv_conv_wgrad_deferred', mode, ptr(u), uld, hu, wu, ca, ptr(v), vld, hv, wv, cb, bb, ptr(dw), s_a, s_b, flip,
                        bias_ptr, ptr(ws), nbytes, tab.slot(), cur.cuda_stream)
            if rc <= 0:
                raise RuntimeError(f'rv_conv_wgrad_deferred failed ({rc}): {_lib.last_error()}')
            tab.n += 1
            tab.keep.append(ws)
        else:
            call('rv_conv_wgrad', mode, ptr(u), uld, hu, wu, ca, ptr(v), vld, hv, wv, cb, bb, ptr(dw), s_a, s_b, flip,
                 bias_ptr, acc, ptr(ws), nbytes, stream())
    if kind == 'up' and want_bias:
        if colsum is not None and colsum.sums is not None:
            # dY's column sums came with the kernel that produced dY (ColsumLink): fold the replicas, no pass over dY
            call('rv_sums_fold', ptr(colsum.sums), colsum.c, 0, cout, ptr(db), acc, stream())
        else:
            _colsum_into(dy, dy.stride(2), bb * dy.shape[1] * dy.shape[2], cout, db, acc)
    return (None, None) if acc else (dw, db)


def _bucket_offset(p_):
    """A gradient buffer's identity across the chains' gradient buckets (the side chains add into twins of the flat bucket, SIDE_GRADS): its
    offset inside whichever bucket holds it (no buckets: the pointer itself)."""
    sg = SIDE_GRADS[0] if (_WGRAD_MERGE_ACROSS and _WGRAD_MERGE_HELPER) else None
    if sg is None:
        return p_
    main, twins = sg
    for t in (main, *twins.values()):
        if t.data_ptr() <= p_ < t.data_ptr() + t.numel() * 4:
            return p_ - t.data_ptr()
    return p_


def _in_main_bucket(p_):
    sg = SIDE_GRADS[0]
    return sg is not None and sg[0].data_ptr() <= p_ < sg[0].data_ptr() + sg[0].numel() * 4


def conv_wgrad_merged(items):
    """ONE weight-gradient launch for the same layer's (x, dY) pairs of several backward passes (rv_conv_wgrad_seg: up to four segments of
    identical geometry).  items: [(kind, x, dy, w, dw_acc, db_acc_or_None), ...], all of one layer and one gradient buffer."""
    kind, x0, dy0, w, dw, db, bf16 = items[0][:7]
    geo = [_wgrad_geom(it[0], it[1], it[2], w, bf16) for it in items]
    mode, taps, u0, uld, hu, wu, ca, v0, vld, hv, wv, cb, bb, s_a, s_b, flip = geo[0]
    same = all(g[0] == mode and g[3:7] == geo[0][3:7] and g[8:] == geo[0][8:] for g in geo)
    lib = _lib.load()
    n = len(items)
    if n == 1 or not same:
        return False
    tkey = (taps, n * bb, hv, ca, cb)
    if AUTOTUNE and tkey not in _wgrad_tuned:
        # the partition of n * bb images: the table's entry (or the nearest batch's), else the one this process tuned for one pass -- never the
        # on-line tuner's timing launches (they read n * bb CONTIGUOUS images from the first segment)
        _wgrad_tuned.add(tkey)
        plan = (plans.lookup_wgrad((taps, n * bb, hv, wv, ca, cb)) if AUTOTUNE == 'table' else None) or _wgrad_plans.get((taps, bb, hv, wv, ca, cb))
        if plan is not None and lib.rv_conv_wgrad_set_plan(taps, n * bb, hv, ca, cb, *plan) == 0:
            _wgrad_plans[(taps, n * bb, hv, wv, ca, cb)] = tuple(plan)
    nbytes = lib.rv_conv_wgrad_workspace_bytes(taps, n * bb, hv, ca, cb)
    ws = torch.empty(nbytes // 4, device=w.device, dtype=torch.float32)
    us = (ctypes.c_void_p * n)(*[g[2].data_ptr() for g in geo])
    vs = (ctypes.c_void_p * n)(*[g[7].data_ptr() for g in geo])
    tables = _WGRAD_DEFER[0]
    if tables is not None:
        cur = torch.cuda.current_stream(w.device)
        tab = tables.get(cur.cuda_stream)
        if tab is None:
            tab = tables[cur.cuda_stream] = _WgradTable(w.device, cur)
        rc = invoke('rv_conv_wgrad_deferred_seg', mode, n, ctypes.addressof(us), ctypes.addressof(vs), uld, hu, wu, ca, vld, hv, wv, cb, bb,
                    ptr(dw), s_a, s_b, flip, ptr(db), ptr(ws), nbytes, tab.slot(), cur.cuda_stream)
        if rc <= 0:
            return False                           # (no segmented form for this layer: the caller launches per pass)
        tab.n += 1
        tab.keep.append(ws)
    else:
        rc = invoke('rv_conv_wgrad_seg', mode, n, ctypes.addressof(us), ctypes.addressof(vs), uld, hu, wu, ca, vld, hv, wv, cb, bb,
                    ptr(dw), s_a, s_b, flip, ptr(db), 1, ptr(ws), nbytes, stream())
        if rc != 0:
            return False
    return True


# --------------------------------------------------------------------------------------------
# one weight-gradient launch per layer and STEP (per gradient bucket), not per backward pass
# --------------------------------------------------------------------------------------------
_WGRAD_MERGE = [None]
_WGRAD_MERGE_HELPER = os.environ.get('RV_WGRAD_MERGE_HELPER', '1') != '0'      # the main chain's merged launches run on the (then idle) side stream
_WGRAD_MERGE_ACROSS = os.environ.get('RV_WGRAD_MERGE_ACROSS', '1') != '0'      # the side chain's pass of a shared layer joins the main chain's launch
_WGRAD_MERGE_MAX = min(4, max(2, int(os.environ.get('RV_WGRAD_MERGE_MAX', '4'))))      # passes per launch (rv_conv_wgrad_seg takes up to four)


class WgradMerger:
    """The transcriber is back-propagated four times per training step (model/UNet_onset.py:383,117-146: the two VAT final passes, the
    reconstruction branch's second transcription, the main forward) and every pass launched its own weight-gradient kernel per layer.  A
    weight gradient is a sum over pixels, so the passes of one layer can be ONE launch over their (x, dY) pairs as segments
    (rv_conv_wgrad_seg) -- the fixed cost of such a launch (prologue, accumulator fold, partial sums, reduction entry: ~10 us of 35-60)
    is paid once: 4 x 49.9 -> 155 us for 64 -> 64 @160x57 at B = 8 per pass.
    * A layer is known by the OFFSET of its gradient in the gradient bucket, so the side chain's pass (which adds into a twin of the
      bucket) joins the main chain's; the sum goes to the main bucket.
    * How many passes a layer sees per step is LEARNED from a step that runs unmerged (per mode: one chain or two); afterwards a pass
      only registers its operands and the last one launches.  `finish()` launches whatever is still pending (a step that took another
      path) and re-learns the counts.
    * WHERE it launches decides whether it pays (profiles/r05_wgrad_merge_ab.txt): on the side stream, which is idle while the main chain
      runs the main forward's backward -- merged on the main chain the same launches LOSE 0.14 ms/step, next door they win 0.5.  The
      launching stream first waits for every other stream a segment was produced on.
    * Memory: the parked (x, dY) pairs of up to three passes stay alive until the last pass of a layer arrives: +0.8 GB of peak device memory at
      B = 8 + 8 (bench.py `device_memory_gb.max_allocated`: 6.28 GB merged, 5.48 GB per pass; 7.9 GB reserved either way) -- 0.3 % of the 288 GB of
      an MI355X, so there is no budget gate; an exception inside the step drops the parked references (`abort`)."""

    def __init__(self):
        self.learned = {}          # (mode, key) -> launches per step
        self.mode = None
        self.counts, self.pending = {}, {}
        self.merged_launches = 0   # segmented launches of the current step (0 in a learning step)
        self.seen = set()          # modes whose pass counts have been learned (a finished step of that mode)

    def begin(self, mode):
        self.mode, self.counts, self.pending, self.merged_launches = mode, {}, {}, 0

    def knows(self, mode):
        """True once a step of this mode (one chain / two chains) has run unmerged, i.e. the next step of that mode merges."""
        return mode in self.seen

    def abort(self):
        """An exception left the step: drop the parked (x, dY) references instead of pinning them until the next step."""
        self.pending, self.counts = {}, {}

    def submit(self, key, item):
        self.counts[key] = self.counts.get(key, 0) + 1
        want = self.learned.get((self.mode, key))
        if want is None or want < 2:
            return False                           # learning step / single pass: the caller launches now
        q = self.pending.setdefault(key, [])
        q.append(item)
        if len(q) >= min(want, _WGRAD_MERGE_MAX):
            self._launch(key)
        return True

    def _launch(self, key):
        q = self.pending.pop(key, [])
        if not q:
            return
        q.sort(key=lambda it: not _in_main_bucket(it[4].data_ptr()))       # the sum goes to the main chain's bucket when one of the passes is its
        dev = q[0][3].device
        cur = torch.cuda.current_stream(dev)
        where = self._helper_for(q) or cur
        # the last pass of the main chain (the main forward's backward) runs while the side chain is idle: its input-gradient chain stays
        # there, the merged weight gradients go next door (they ADD into the main bucket through the reduction table's atomics).  Whatever
        # stream launches: it first waits for every OTHER stream a pass of the group was produced on.
        origins = {it[7].cuda_stream: it[7] for it in q}
        with torch.cuda.stream(where):
            for sp, st in origins.items():
                if sp != where.cuda_stream:
                    ev = torch.cuda.Event()
                    ev.record(st)
                    where.wait_event(ev)
            for it in q:
                if it[7].cuda_stream != where.cuda_stream:
                    it[1].record_stream(where)
                    it[2].record_stream(where)
            self._launch_here(q)

    def _helper_for(self, q):
        if not _WGRAD_MERGE_HELPER or len(q) < 2 or not DUAL_STREAM[0] or SIDE_GRADS[0] is None or _WGRAD_DEFER[0] is None:
            return None
        main, twins = SIDE_GRADS[0]
        p_ = q[0][4].data_ptr()
        if not (main.data_ptr() <= p_ < main.data_ptr() + main.numel() * 4) or not twins:
            return None
        dev = q[0][3].device
        helper = side_stream(dev, 0)
        if helper.cuda_stream == torch.cuda.current_stream(dev).cuda_stream:
            return None
        return helper

    def _launch_here(self, q):
        # passes of different geometry (the labelled and the unlabelled batch may differ in size) merge among their likes
        groups = {}
        for it in q:
            groups.setdefault((tuple(it[1].shape), it[1].stride(), tuple(it[2].shape), it[2].stride()), []).append(it)
        dw0, db0 = q[0][4], q[0][5]                # (every launch of the group adds into the first item's buffers: the main chain's when it has a pass)
        for g in groups.values():
            g = [(it[0], it[1], it[2], it[3], dw0, db0 if it[5] is not None else None) + tuple(it[6:]) for it in g]
            if len(g) >= 2 and conv_wgrad_merged(g):
                self.merged_launches += 1
                continue
            merger, _WGRAD_MERGE[0] = _WGRAD_MERGE[0], None
            try:
                for kind, x, dy, w, dw, db, bf16, _st in g:   # a single pass, or no segmented form for this layer: per pass after all
                    conv_wgrad(kind, x, dy, w, db is not None, dw, db, bf16=bf16)
            finally:
                _WGRAD_MERGE[0] = merger

    def finish(self):
        for key in list(self.pending):
            self._launch(key)
        for key, n in self.counts.items():
            self.learned[(self.mode, key)] = n
        self.seen.add(self.mode)


class wgrad_merging:
    def __init__(self, merger, mode):
        self.merger, self.mode = merger, mode

    def __enter__(self):
        self.prev = _WGRAD_MERGE[0]
        _WGRAD_MERGE[0] = self.merger
        if self.merger is not None:
            self.merger.begin(self.mode)
        return self.merger

    def __exit__(self, exc_type, *exc):
        _WGRAD_MERGE[0] = self.prev
        if exc_type is not None and self.merger is not None:
            self.merger.abort()


# --------------------------------------------------------------------------------------------
# deferred weight-gradient reductions: one table-driven launch per stream and backward pass
# --------------------------------------------------------------------------------------------
_WGRAD_DEFER = [None]
_WGRAD_POOL = []           # pre-allocated pinned host tables for hipGraph capture (never recycled once captured)
_WGRAD_KEEP = []
_WGRAD_MAX = 512           # entries per table


def prepare_wgrad_tables(n=3):
    """Pinned host tables for deferred_wgrad_reductions under hipGraph capture (call outside capture)."""
    eb = _lib.load().rv_wgrad_table_entry_bytes()
    while len(_WGRAD_POOL) < n:
        _WGRAD_POOL.append(torch.empty(_WGRAD_MAX * eb, dtype=torch.uint8).pin_memory())


class _WgradTable:
    def __init__(self, device, torch_stream):
        self.eb = _lib.load().rv_wgrad_table_entry_bytes()
        self.device, self.stream, self.n, self.keep = device, torch_stream, 0, []
        if torch.cuda.is_current_stream_capturing():
            if not _WGRAD_POOL:
                raise RuntimeError('deferred_wgrad_reductions: no pinned table left for hipGraph capture')
            self.host = _WGRAD_POOL.pop()
            _keep(self.host, _WGRAD_KEEP)
            self.pinned = True
        else:
            self.host = torch.empty(_WGRAD_MAX * self.eb, dtype=torch.uint8)
            self.pinned = False

    def slot(self):
        if self.n >= _WGRAD_MAX:
            raise RuntimeError('deferred_wgrad_reductions: table full')
        return self.host.data_ptr() + self.n * self.eb

    def flush(self):
        if self.n == 0:
            return
        lib = _lib.load()
        total = lib.rv_wgrad_table_finalize(self.host.data_ptr(), self.n)
        used = self.host[:self.n * self.eb]
        with torch.cuda.stream(self.stream):
            src = used if self.pinned else used.pin_memory()
            table = src.to(self.device, non_blocking=True)
            call('rv_wgrad_reduce_table', ptr(table), self.n, total, self.stream.cuda_stream)
            table.record_stream(self.stream)
            if self.pinned:
                _keep(table, _WGRAD_KEEP)      # captured: keep the device copy's memory out of the graph pool's reuse
        self.n, self.keep = 0, []


class deferred_wgrad_reductions:
    """While active, conv weight-gradient calls that accumulate into param.grad launch only their partial-sum kernel; the
    reductions of ALL layers run as one launch per stream at ``flush()`` (call it after backward, before the gradients are
    read).  One reduction launch per layer is launch-latency bound (161 launches of ~5 us per step)."""

    def __enter__(self):
        self.prev = _WGRAD_DEFER[0]
        self.tables = {}
        _WGRAD_DEFER[0] = self.tables
        return self

    def __exit__(self, *exc):
        _WGRAD_DEFER[0] = self.prev

    def flush(self):
        for t in self.tables.values():
            t.flush()


def _out_hw(kind, h, w, size):
    if kind == 'down':
        return h // 2, w // 2
    if kind == 'up':
        return (2 * h, 2 * w) if size is None else (int(size[0]), int(size[1]))
    return h, w


# A conv layer's weight gradient is launched BEFORE its input gradient in the backward (RV_WGRAD_FIRST=0: the other way round).  The input
# gradient is the one the next layer's backward waits for, so "input gradient first" looks right -- measured in situ it is the slower order
# (tools/knob_ab.sh, round 5: 21.30 vs 21.48 ms/step): both kernels read dY, and the weight-gradient kernel's single pass over X and dY leaves
# dY in the cache hierarchy for the input-gradient kernel that follows, whose output the next layer then finds warm.  Results are unaffected.
WGRAD_FIRST = [os.environ.get('RV_WGRAD_FIRST', '1') != '0']
_WGRAD_FIRST_UPCAT = os.environ.get('RV_WGRAD_FIRST_UPCAT', '1') != '0'      # (the decoder's up-conv + skip-conv pair follows the same order)


class ConvFn(Function):
    """y = conv(x) for any of the five conv kinds (new contiguous NHWC tensor)."""

    @staticmethod
    def forward(ctx, x, w, b, kind, size, stats=None, bn_in=None, share=None, dx_colsum=None, dy_colsum=None):
        """dx_colsum: a ColsumLink this conv's input-gradient kernel fills (column sums of dx); dy_colsum: a filled-in-backward
        ColsumLink holding the column sums of this conv's dY (kind 'up': its bias gradient)."""
        bb, h, wd, cin, _ = _geom(x)
        _, cout = _channels(kind, w)
        ho, wo = _out_hw(kind, h, wd, size)
        y = _new(bb, ho, wo, cout, x)
        ctx.live = bool(ctx.needs_input_grad[1])       # live weights <=> a final graph (bf16_final_graphs applies to those only)
        conv_forward_into(kind, x, w, b, y, stats, bf16=BF16['fwd'] and ctx.live)
        ctx.kind = kind
        ctx.xshape = tuple(x.shape)
        ctx.bn_in = bn_in            # BnLink of the BatchNorm whose output is x (single consumer), or None
        ctx.share = share            # GradShare of x (several conv consumers), or None
        ctx.dx_colsum, ctx.dy_colsum = dx_colsum, dy_colsum
        ctx.save_for_backward(x if ctx.needs_input_grad[1] else None, w)
        ctx.params = (w, b)          # parameter objects (for direct gradient accumulation)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        dy = dy.contiguous()
        dx = dw = db = None
        bf = BF16['bwd'] and ctx.live

        def input_grad():
            nonlocal dx
            if not ctx.needs_input_grad[0]:
                return
            if ctx.share is not None and ctx.bn_in is None:
                dx = ctx.share.dgrad(ctx.kind, dy, w, ctx.xshape, bf16=bf)
            else:
                dx = torch.empty(ctx.xshape, device=dy.device, dtype=torch.float32)
                sum_ws = None
                if ctx.dx_colsum is not None and ctx.live and ctx.bn_in is None:
                    sum_ws = ARENA.take(bn_ws_doubles(ctx.xshape[3]), dy.device)
                conv_dgrad_into(ctx.kind, dy, w, dx, ctx.bn_in, bf16=bf, sum_ws=sum_ws)
                if sum_ws is not None:
                    ctx.dx_colsum.sums, ctx.dx_colsum.c = sum_ws, ctx.xshape[3]

        def weight_grad():
            nonlocal dw, db
            if not ctx.needs_input_grad[1]:
                return
            pw, pb = ctx.params
            gw, gb = _grad_buf(pw), _grad_buf(pb)
            if gw is not None and (gb is not None or not ctx.needs_input_grad[2]):
                conv_wgrad(ctx.kind, x, dy, w, ctx.needs_input_grad[2], gw, gb, bf16=bf, colsum=ctx.dy_colsum)
            else:
                dw, db = conv_wgrad(ctx.kind, x, dy, w, ctx.needs_input_grad[2], bf16=bf, colsum=ctx.dy_colsum)
        # launch order: see WGRAD_FIRST
        for part in ((weight_grad, input_grad) if WGRAD_FIRST[0] else (input_grad, weight_grad)):
            part()
        if ctx.dy_colsum is not None:
            ctx.dy_colsum.sums = None            # drop the reference
        return dx, dw, db, None, None, None, None, None, None, None


class UpCatFn(Function):
    """cat = concat(up(x, output_size), conv3x3(skip_src)) along channels, written in place into one NHWC
    buffer (d_block.forward's `self.us(x, output_size=size)` + `torch.cat((x, skip), 1)`,
    model/UNet_onset.py:219-220, with the encoder's extra skip conv, :244-246, computed here)."""

    @staticmethod
    def forward(ctx, x, w_up, b_up, s, w_skip, b_skip, size, share=None, dy_colsum=None):
        bb, h, wd, cin, _ = _geom(x)
        cu = w_up.shape[1]
        cs = w_skip.shape[0]
        ho, wo = int(size[0]), int(size[1])
        cat = _new(bb, ho, wo, cu + cs, x)
        ctx.live = bool(ctx.needs_input_grad[1] or ctx.needs_input_grad[4])
        conv_forward_into('up', x, w_up, b_up, cat[..., :cu])
        conv_forward_into('c3', s, w_skip, b_skip, cat[..., cu:], bf16=BF16['fwd'] and ctx.live)
        ctx.cu = cu
        ctx.dy_colsum = dy_colsum    # ColsumLink: column sums of dcat from the kernel that produces it
        ctx.share = share            # GradShare of s (it also feeds the next encoder block)
        ctx.shapes = (tuple(x.shape), tuple(s.shape))
        need_w = ctx.needs_input_grad[1] or ctx.needs_input_grad[4]
        ctx.save_for_backward(x if need_w else None, s if need_w else None, w_up, w_skip)
        ctx.params = (w_up, b_up, w_skip, b_skip)
        return cat

    @staticmethod
    def backward(ctx, dcat):
        x, s, w_up, w_skip = ctx.saved_tensors
        dcat = dcat.contiguous()
        cu = ctx.cu
        d_up, d_sk = dcat[..., :cu], dcat[..., cu:]
        dx = ds = dwu = dbu = dws = dbs = None
        pwu, pbu, pws, pbs = ctx.params

        def input_grads():
            nonlocal dx, ds
            if ctx.needs_input_grad[0]:
                dx = torch.empty(ctx.shapes[0], device=dcat.device, dtype=torch.float32)
                conv_dgrad_into('up', d_up, w_up, dx)
            if ctx.needs_input_grad[3]:
                bf = BF16['bwd'] and ctx.live
                if ctx.share is not None:
                    ds = ctx.share.dgrad('c3', d_sk, w_skip, ctx.shapes[1], bf16=bf)
                else:
                    ds = torch.empty(ctx.shapes[1], device=dcat.device, dtype=torch.float32)
                    conv_dgrad_into('c3', d_sk, w_skip, ds, bf16=bf)

        def weight_grads():
            nonlocal dwu, dbu, dws, dbs
            if ctx.needs_input_grad[1]:
                gw, gb = _grad_buf(pwu), _grad_buf(pbu)
                if gw is not None and gb is not None:
                    conv_wgrad('up', x, d_up, w_up, True, gw, gb, colsum=ctx.dy_colsum)
                else:
                    dwu, dbu = conv_wgrad('up', x, d_up, w_up, True, colsum=ctx.dy_colsum)
            if ctx.needs_input_grad[4]:
                gw, gb = _grad_buf(pws), _grad_buf(pbs)
                bfw = BF16['bwd'] and ctx.live
                if gw is not None and gb is not None:
                    conv_wgrad('c3', s, d_sk, w_skip, True, gw, gb, bf16=bfw)
                else:
                    dws, dbs = conv_wgrad('c3', s, d_sk, w_skip, True, bf16=bfw)
        for part in ((weight_grads, input_grads) if (WGRAD_FIRST[0] and _WGRAD_FIRST_UPCAT) else (input_grads, weight_grads)):
            part()
        if ctx.dy_colsum is not None:
            ctx.dy_colsum.sums = None
        return dx, dwu, dbu, ds, dws, dbs, None, None, None


# --------------------------------------------------------------------------------------------
# BatchNorm (train / eval) + leaky ReLU (+ residual)
# --------------------------------------------------------------------------------------------
def skip_conv_ksplit(x, cout):
    """Which arithmetic the SEPARATE launch of the 1x1 skip conv x -> cout channels would use (the fused form inside the BatchNorm apply kernel must reproduce it bit
    for bit): False = one fmaf chain over the input channels (conv_small_k for one input channel; conv_mfma_k's plain tiles), True = the K-split tile family 0x5NM
    (four partial chains added in wave order), None = unknown -- on-line tuning, a forced algo, a table entry borrowed from another batch size, or an input that the
    fused kernel does not take -- in which case the caller keeps the separate launch."""
    bb, h, wd, cin, ild = _geom(x)
    if cin == 1:
        return False
    if cin not in (16, 32, 64) or ild % 4 or x.data_ptr() % 16 or os.environ.get('RV_FORCE_ALGO_ALL'):
        return None
    if AUTOTUNE is False:
        return False                         # library default tile: choose_tiles, never the K-split family
    if AUTOTUNE != 'table':
        return None
    hit = plans.lookup_conv((1, bb, h, wd, cin, cout, ild, cout, False, False))
    if hit is None:
        return False                         # not in the table: library default tile
    if not hit[1]:
        return None                          # borrowed from another batch size: its legality is only known after the first launch
    return ((hit[0] >> 8) & 15) == 5


class BnActFn(Function):
    """y = leaky_relu(batch_norm(z)) (+ res).  Training mode updates running_mean / running_var /
    num_batches_tracked in place exactly like nn.BatchNorm2d(momentum=0.1)."""

    @staticmethod
    def forward(ctx, z, gamma, beta, running_mean, running_var, nbt, res, training, slope, stats=None, link=None,
                r1_x=None, r1_w=None, r1_b=None, r1_share=None):
        """r1_x / r1_w / r1_b (round 6): the residual as a RANK-1 term -- skip(x) of the first encoder block, a 1 -> C 1x1 conv of the single-channel
        input (model/UNet_onset.py:191,198) -- evaluated inside the apply kernel (rv_bn_lrelu_fwd_skip; 16 / 32 / 64-channel inputs: the dense form, opt-in) instead of a conv launch writing it and this kernel
        reading it back; the backward launches that conv's input / weight gradients from dy exactly as ConvFn.backward would (r1_share: the GradShare of x)."""
        need_gpu(z, gamma)
        bb, h, wd, c, zld = _geom(z)
        p = bb * h * wd
        y = torch.empty_like(z, memory_format=torch.contiguous_format)
        coef = torch.empty(5 * c, device=z.device, dtype=torch.float32)
        mode = 1 if training else 0
        if training and _BN_DEFER[0] is not None:
            mode = 2
            _BN_DEFER[0].append((running_mean, running_var, nbt, coef, c))
        # zero-initialised fp64 sums: one slice for the forward statistics, one for the backward reduction
        # (``stats``: the producing conv already left the forward sums in its slice -- bn_stats_slot / ConvFn)
        ready = training and stats is not None
        ws = (stats if ready else ARENA.take(bn_ws_doubles(c), z.device),
              ARENA.take(bn_ws_doubles(c), z.device) if (ctx.needs_input_grad[0] or ctx.needs_input_grad[1]) else None)
        rld = _geom(res)[4] if res is not None else 0
        ctx.r1 = r1_x is not None
        if ctx.r1:
            cin1 = r1_x.shape[-1]
            assert res is None and tuple(r1_x.shape[:3]) == (bb, h, wd) and r1_w.shape[0] == c and r1_w.numel() == c * cin1 and r1_w.is_contiguous()
            need_gpu(r1_x, r1_w)
            ks = skip_conv_ksplit(r1_x, c)
            assert ks is not None, 'fused skip conv: the tile of the separate launch is not known (ops.fusable_skip decides)'
            call('rv_bn_lrelu_fwd_skip', ptr(z), zld, p, c, ptr(gamma), ptr(beta), ptr(running_mean), ptr(running_var), ptr(nbt),
                 BN_MOMENTUM, BN_EPS, mode, slope, ptr(r1_x), _geom(r1_x)[4], cin1, ptr(r1_w), ptr(r1_b), 1 if ks else 0, ptr(y), c, ptr(coef), ptr(ws[0]),
                 1 if ready else 0, stream())
            ctx.r1_share, ctx.r1_xshape, ctx.r1_params = r1_share, tuple(r1_x.shape), (r1_w, r1_b)
            ctx.r1_live = bool(ctx.needs_input_grad[12])
        else:
            call('rv_bn_lrelu_fwd', ptr(z), zld, p, c, ptr(gamma), ptr(beta), ptr(running_mean), ptr(running_var), ptr(nbt),
                 BN_MOMENTUM, BN_EPS, mode, slope, ptr(res), rld, ptr(y), c, ptr(coef), ptr(ws[0]), 1 if ready else 0, stream())
        ctx.ws = ws
        ctx.link = None
        if link is not None and training and ws[1] is not None and res is None and not ctx.r1:
            link.z, link.coef, link.ws, link.slope, link.ready = z, coef, ws[1], slope, False
            ctx.link = link
        ctx.training = training
        ctx.slope = slope
        ctx.params = (gamma, beta)
        if ctx.r1:
            ctx.save_for_backward(z, coef, r1_x if ctx.needs_input_grad[12] else None, r1_w)
        else:
            ctx.save_for_backward(z, coef)
        return y

    @staticmethod
    def backward(ctx, dy):
        if ctx.r1:
            z, coef, r1_x, r1_w = ctx.saved_tensors
        else:
            z, coef = ctx.saved_tensors
        dy = dy.contiguous()
        dx1 = dw1 = db1 = None
        if ctx.r1:
            # the skip conv's own backward (ConvFn.backward of a 'c1' conv with input x, output gradient dy), launched here: it used to be
            # the first node behind this one on the residual path, so the order of the adds into x's shared gradient is unchanged
            def r1_input_grad():
                nonlocal dx1
                if not ctx.needs_input_grad[11]:
                    return
                if ctx.r1_share is not None:
                    dx1 = ctx.r1_share.dgrad('c1', dy, r1_w, ctx.r1_xshape)
                else:
                    dx1 = torch.empty(ctx.r1_xshape, device=dy.device, dtype=torch.float32)
                    conv_dgrad_into('c1', dy, r1_w, dx1)

            def r1_weight_grad():
                nonlocal dw1, db1
                if not ctx.needs_input_grad[12]:
                    return
                pw1, pb1 = ctx.r1_params
                gw1, gb1 = _grad_buf(pw1), _grad_buf(pb1)
                if gw1 is not None and (gb1 is not None or not ctx.needs_input_grad[13]):
                    conv_wgrad('c1', r1_x, dy, r1_w, ctx.needs_input_grad[13], gw1, gb1)
                else:
                    dw1, db1 = conv_wgrad('c1', r1_x, dy, r1_w, ctx.needs_input_grad[13])
            for part in ((r1_weight_grad, r1_input_grad) if WGRAD_FIRST[0] else (r1_input_grad, r1_weight_grad)):
                part()
        bb, h, wd, c, zld = _geom(z)
        p = bb * h * wd
        need_w = ctx.needs_input_grad[1] or ctx.needs_input_grad[2]
        dz = torch.empty_like(z, memory_format=torch.contiguous_format)
        pg, pb = ctx.params
        gg, gb = (_grad_buf(pg), _grad_buf(pb)) if need_w else (None, None)
        direct = gg is not None and gb is not None
        if direct:
            dg, db = gg, gb
        else:
            dg = torch.empty(c, device=z.device, dtype=torch.float32) if need_w else None
            db = torch.empty(c, device=z.device, dtype=torch.float32) if need_w else None
        ready = ctx.link is not None and ctx.link.ready      # the consumer conv's dgrad already reduced (dy, z)
        call('rv_bn_lrelu_bwd', ptr(dy), c, ptr(z), zld, p, c, ptr(coef), ctx.slope, 0 if ctx.training else 1,
             ptr(dz), c, ptr(dg), ptr(db), 1 if direct else 0,
             ptr(ctx.ws[1] if ctx.ws[1] is not None else ARENA.take(bn_ws_doubles(c), z.device)), 1 if ready else 0, stream())
        if ctx.link is not None:
            ctx.link.z = ctx.link.coef = ctx.link.ws = None   # drop the references
        if direct:
            dg = db = None
        dres = dy if ctx.needs_input_grad[6] else None
        return (dz if ctx.needs_input_grad[0] else None), dg, db, None, None, None, dres, None, None, None, None, dx1, dw1, db1, None


# --------------------------------------------------------------------------------------------
# linear layers
# --------------------------------------------------------------------------------------------
def _mat(t):
    """(ptr, rows, cols, row stride, col stride) of a 2-D view."""
    assert t.dim() == 2
    return ptr(t), t.shape[0], t.shape[1], t.stride(0), t.stride(1)


_gemm_splitk = {}          # (M, N, K, batch, A k-fast, B k-fast, act, accumulate) -> split-K factor in use


def _splitk_for(m_out, n_out, k):
    """Library-default split-K heuristic (shapes outside the plan table): enough workgroups to cover the chip twice."""
    blocks = ((m_out + 63) // 64) * ((n_out + 63) // 64)
    if blocks >= 256 or k < 512:
        return 1
    return max(1, min(16, 512 // blocks, k // 128))


def _tune_gemm(key, launch, m, n, k, cands=(1, 2, 3, 4, 6, 8, 12, 16, 24, 32)):
    """Time the split-K candidates of one GEMM shape (HIP events on the launch stream, scratch output) and keep the fastest.
    Split-K is deterministic for a given factor (in-order fold), so the choice only fixes the summation grouping."""
    st = torch.cuda.current_stream()
    blocks = ((m + 63) // 64) * ((n + 63) // 64)
    best, choice = None, 1
    for s in cands:
        if s > 1 and (k // s < 64 or blocks * s > 4096):
            continue
        launch(s)
        t = None
        for _rep in range(2):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(st)
            for _ in range(3):
                launch(s)
            e1.record(st)
            e1.synchronize()
            dt = e0.elapsed_time(e1)
            t = dt if t is None else min(t, dt)
        if best is None or t < best * 0.97:               # a larger factor must win by 3 %: ties go to fewer slices
            best, choice = t, s
    _tune_us[('gemm', key)] = best / 3 * 1e3
    if os.environ.get('RV_TUNE_LOG'):
        print(f'[tune] gemm {key}: splitk={choice} {best / 3 * 1e3:.1f} us', file=sys.stderr)
    return choice


def gemm(a, b_kn, c, bias=None, act=0, accumulate=False, splitk=None, c2=None, batch=1, bstrides=(0, 0, 0), a_rowsum=None,
         deterministic=True, defer_ok=False):
    """c[m,n] (+)= act(sum_k a[m,k] * b_kn[k,n] + bias[n]) on arbitrary 2-D views.  batch > 1: `batch` problems of this
    shape, problem z offset by z * bstrides (elements) from a / b_kn / c.  splitk None: the shipped plan table's factor for this
    shape (ops.AUTOTUNE == 'table'), the on-line tuner's (True), else the library heuristic; the reduction over k slices is
    folded in order inside the kernel (deterministic=True: forward / input-gradient chains, whose results feed the chaotic VAT
    direction and must be reproducible) or accumulated by fp32 atomics (deterministic=False: parameter gradients; faster)."""
    need_gpu(a, b_kn, c)
    pa, m, k, sam, sak = _mat(a)
    pb, k2, n, sbk, sbn = _mat(b_kn)
    pc, m2, n2, scm, scn = _mat(c)
    assert k == k2 and m == m2 and n == n2, (a.shape, b_kn.shape, c.shape)
    if c2 is not None:
        pc2, _, _, s2m, s2n = _mat(c2)
    else:
        pc2, s2m, s2n = None, 0, 0
    lib = _lib.load()

    def launch(s, pc_=pc, rs=ptr(a_rowsum), raise_=True):
        ws = tk = None
        if s > 1 and deterministic:
            # deterministic split-K: partial tiles are parked in `ws`, a second launch folds them in k order and runs the epilogue
            ws = torch.empty(lib.rv_gemm_splitk_workspace_bytes(m, n, s, batch) // 4, device=c.device, dtype=torch.float32)
        call('rv_gemm', pa, sam, sak, pb, sbk, sbn, pc_, scm, scn, pc2, s2m, s2n, ptr(bias), m, n, k, act,
             1 if accumulate else 0, s, batch, bstrides[0], bstrides[1], bstrides[2], rs, ptr(ws), ptr(tk), stream())

    deferrable = defer_ok and accumulate and not deterministic and bias is None and act == 0 and c2 is None and _GEMM_DEFER[0] is not None
    if splitk is None:
        key = (m, n, k, batch, int(sak <= sam), int(sbk <= sbn), act, int(bool(accumulate)) + 2 * int(bool(deterministic)))
        splitk = _gemm_splitk.get(key)
        if splitk is None:
            if AUTOTUNE == 'table':
                splitk = plans.lookup_gemm(key)
            elif AUTOTUNE and c2 is None and not torch.cuda.is_current_stream_capturing():
                # time on scratch outputs: the real C may be a gradient that is being accumulated into
                extent = (m - 1) * scm + (n - 1) * scn + 1 + (batch - 1) * bstrides[2]
                tmp = torch.zeros(extent, device=c.device, dtype=torch.float32)
                rs_tmp = torch.zeros(m, device=c.device, dtype=torch.float32) if a_rowsum is not None else None
                splitk = _tune_gemm(key, lambda s: launch(s, ptr(tmp), ptr(rs_tmp)), m, n, k,
                                    (1, 2, 3, 4, 6, 8, 12, 16, 24, 32) if (deterministic or act == 0) else (1,))
            if splitk is None:
                splitk = _splitk_for(m, n, k)
            _gemm_splitk[key] = splitk
    if deferrable and _defer_gemm(a, b_kn, c, splitk, batch, bstrides, a_rowsum):
        return
    launch(splitk)


# --------------------------------------------------------------------------------------------
# deferred parameter-gradient GEMMs: one grouped launch per stream and backward pass
# --------------------------------------------------------------------------------------------
_GEMM_DEFER = [None]
_GEMM_POOL = []            # pre-allocated pinned host tables for hipGraph capture (never recycled once captured)
_GEMM_KEEP = []
_GEMM_MAX = 64             # entries per table (rv_gemm_table_run's scan limit)
KEEP_TABLES = [False]      # measurement mode (bench.py): flushed device tables and their operands stay alive for re-launching
GEMM_TABLE_WORK = {}       # device table pointer -> (flops, algorithmic bytes) of the grouped launch (filled in measurement mode)


def prepare_gemm_tables(n=3):
    """Pinned host tables for deferred_param_gemms under hipGraph capture (call outside capture)."""
    eb = _lib.load().rv_gemm_table_entry_bytes()
    while len(_GEMM_POOL) < n:
        _GEMM_POOL.append(torch.empty(_GEMM_MAX * eb, dtype=torch.uint8).pin_memory())


class _GemmTable:
    """The pending accumulating GEMMs of one (stream, operand orientation)."""

    def __init__(self, device, torch_stream, orientation):
        self.eb = _lib.load().rv_gemm_table_entry_bytes()
        self.device, self.stream, self.orientation, self.n, self.keep = device, torch_stream, orientation, 0, []
        self.flops = self.bytes = 0.0
        if torch.cuda.is_current_stream_capturing():
            if not _GEMM_POOL:
                raise RuntimeError('deferred_param_gemms: no pinned table left for hipGraph capture')
            self.host = _GEMM_POOL.pop()
            _keep(self.host, _GEMM_KEEP)
            self.pinned = True
        else:
            self.host = torch.empty(_GEMM_MAX * self.eb, dtype=torch.uint8)
            self.pinned = False

    def flush(self):
        if self.n == 0:
            return
        fold = ctypes.c_long(0)
        total = _lib.load().rv_gemm_table_finalize(self.host.data_ptr(), self.n, ctypes.addressof(fold))
        used = self.host[:self.n * self.eb]
        with torch.cuda.stream(self.stream):
            src = used if self.pinned else used.pin_memory()
            table = src.to(self.device, non_blocking=True)
            call('rv_gemm_table_run', ptr(table), self.n, total, fold.value, self.orientation, self.stream.cuda_stream)
            table.record_stream(self.stream)
            if self.pinned:
                _keep(table, _GEMM_KEEP)       # captured: keep the device copy's memory out of the graph pool's reuse
            if KEEP_TABLES[0]:
                _GEMM_KEEP.append((table, self.keep))
                GEMM_TABLE_WORK[table.data_ptr()] = (self.flops, self.bytes)
        self.n, self.keep = 0, []
        self.flops = self.bytes = 0.0


class deferred_param_gemms:
    """While active, linear-layer / attention parameter-gradient GEMMs that accumulate into param.grad are only REGISTERED; all of
    them run as one grouped launch per stream at ``flush()`` (after backward, before the gradients are read).  Alone each of them
    is a few dozen workgroups with a 5 120-long reduction: latency-bound, 25-90 us; side by side they fill the chip."""

    def __enter__(self):
        self.prev = _GEMM_DEFER[0]
        self.tables = {}
        _GEMM_DEFER[0] = self.tables
        return self

    def __exit__(self, *exc):
        _GEMM_DEFER[0] = self.prev

    def flush(self):
        for t in self.tables.values():
            t.flush()


_TABLE_PARK = os.environ.get('RV_GEMM_TABLE_PARK', '1') != '0'      # (0: every k slice of a grouped GEMM adds atomically, the form until round 5)


def _defer_gemm(a, b_kn, c, splitk, batch, bstrides, a_rowsum):
    """Register c += a @ b_kn (atomic split-K) with the active deferred_param_gemms context; False if none is active."""
    tables = _GEMM_DEFER[0]
    if tables is None:
        return False
    pa, m, k, sam, sak = _mat(a)
    pb, _, n, sbk, sbn = _mat(b_kn)
    pc, _, _, scm, scn = _mat(c)
    cur = torch.cuda.current_stream(c.device)
    orient = (1 if sak <= sam else 0) | (2 if sbk <= sbn else 0)
    key = (cur.cuda_stream, orient)
    tab = tables.get(key)
    if tab is None:
        tab = tables[key] = _GemmTable(c.device, cur, orient)
    if tab.n >= _GEMM_MAX:
        return False
    lib = _lib.load()
    # split-K inside the grouped launch: slices parked in a per-entry workspace, folded by the table's second launch (one atomic per element)
    ws = torch.empty(lib.rv_gemm_splitk_workspace_bytes(m, n, splitk, batch) // 4, device=c.device, dtype=torch.float32) \
        if (splitk > 1 and _TABLE_PARK) else None
    rc = lib.rv_gemm_table_fill(tab.host.data_ptr() + tab.n * tab.eb, pa, sam, sak, pb, sbk, sbn, pc, scm, scn, None, m, n, k,
                                splitk, batch, bstrides[0], bstrides[1], bstrides[2], ptr(a_rowsum), ptr(ws))
    if rc < 0:
        raise RuntimeError(f'rv_gemm_table_fill failed ({rc}): {_lib.last_error()}')
    assert rc == orient
    tab.n += 1
    tab.flops += 2.0 * m * n * k * batch
    tab.bytes += 4.0 * batch * (m * k + k * n + m * n)
    tab.keep += [a, b_kn, c, a_rowsum, ws]      # operands stay alive (and their memory un-reused) until the grouped launch
    return True


def colsum(x2d, out=None, accumulate=False):
    if out is None:
        out = torch.empty(x2d.shape[1], device=x2d.device, dtype=torch.float32)
    assert x2d.stride(1) == 1
    _colsum_into(x2d, x2d.stride(0), x2d.shape[0], x2d.shape[1], out, accumulate)
    return out


def _colsum_into(x, ld, m, n, out, accumulate):
    if DETERMINISTIC[0]:
        ws = torch.empty(_lib.load().rv_colsum_ordered_workspace_bytes(m, n) // 4, device=out.device, dtype=torch.float32)
        call('rv_colsum_ordered', ptr(x), ld, m, n, ptr(out), 1 if accumulate else 0, ptr(ws), stream())
    else:
        call('rv_colsum', ptr(x), ld, m, n, ptr(out), 1 if accumulate else 0, stream())


def _param_wgrad(a_t, b, param, splitk=None, bias_param=None):
    """d(param) = a_t @ b.  Under direct_param_grads() the product is accumulated straight into param.grad (split-K
    atomics add onto it: no zero fill, no temporary, no autograd add) and None is returned.  bias_param: the layer's
    bias -- its gradient (the row sums of a_t = dY^T) then rides on the same GEMM; returns (None, True) in that case."""
    g = _grad_buf(param)
    if bias_param is not None:
        gb = _grad_buf(bias_param)
        if g is not None and gb is not None:
            gemm(a_t, b, g.view(a_t.shape[0], b.shape[1]), accumulate=True, splitk=splitk, a_rowsum=gb, deterministic=DETERMINISTIC[0], defer_ok=True)
            return None, True
        return _param_wgrad(a_t, b, param, splitk), False
    if g is not None:
        gemm(a_t, b, g.view(a_t.shape[0], b.shape[1]), accumulate=True, splitk=splitk, deterministic=DETERMINISTIC[0], defer_ok=True)
        return None
    dw = torch.empty((a_t.shape[0], b.shape[1]), device=b.device, dtype=torch.float32)
    gemm(a_t, b, dw, splitk=splitk, deterministic=DETERMINISTIC[0])
    return dw.view_as(param)


def _param_bgrad(dz2d, param):
    g = _grad_buf(param)
    if g is not None:
        colsum(dz2d, g, accumulate=True)
        return None
    return colsum(dz2d)


class LinearFn(Function):
    """y = act(x @ w.T + b) for a 2-D (possibly strided) x; act in {0: none, 1: sigmoid}."""

    @staticmethod
    def forward(ctx, x, w, b, act):
        m, k = x.shape
        n = w.shape[0]
        y = torch.empty((m, n), device=x.device, dtype=torch.float32)
        gemm(x, w.t(), y, b, act)
        ctx.act = act
        ctx.params = (w, b)
        ctx.save_for_backward(x, w, y if act == 1 else None)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w, y = ctx.saved_tensors
        dy = dy.contiguous()
        m, k = x.shape
        n = w.shape[0]
        if ctx.act == 1:
            dz = torch.empty_like(dy)
            call('rv_sigmoid_bwd', ptr(dy), n, None, 0, ptr(y), n, ptr(dz), n, m, n, stream())
        else:
            dz = dy
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty((m, k), device=dy.device, dtype=torch.float32)
            gemm(dz, _lin_t(w).t(), dx)
        bias_done = False
        if ctx.needs_input_grad[1]:
            if ctx.needs_input_grad[2]:
                dw, bias_done = _param_wgrad(dz.t(), x, ctx.params[0], None, ctx.params[1])
            else:
                dw = _param_wgrad(dz.t(), x, ctx.params[0], None)
        if ctx.needs_input_grad[2] and not bias_done:
            db = _param_bgrad(dz, ctx.params[1])
        return dx, dw, db, None


class OnsetHeadsFn(Function):
    """Spec2Roll's two heads + concat (model/UNet_onset.py:307-312):
         onset = sigmoid(linear_onset(y[..., 0])); feat = linear_feature(y[..., 1]); cat = [onset | feat]
       y: decoder output NHWC [B, T, 229, 2].  Returns (cat [B*T, 176], onset [B*T, 88])."""

    @staticmethod
    def forward(ctx, y, wo, bo, wf, bf):
        bb, t, nb, c = y.shape
        assert c == 2 and y.is_contiguous()
        m = bb * t
        y2 = y.view(m, nb, 2)
        cat = torch.empty((m, 176), device=y.device, dtype=torch.float32)
        onset = torch.empty((m, 88), device=y.device, dtype=torch.float32)
        gemm(y2[..., 0], wo.t(), cat[:, :88], bo, act=1, c2=onset)
        gemm(y2[..., 1], wf.t(), cat[:, 88:], bf, act=0)
        ctx.params = (wo, bo, wf, bf)
        ctx.save_for_backward(y, wo, wf, onset)
        ctx.set_materialize_grads(False)          # donset is None (not a zero tensor) when the onset output is unused
        return cat, onset

    @staticmethod
    def backward(ctx, dcat, donset):
        y, wo, wf, onset = ctx.saved_tensors
        bb, t, nb, _ = y.shape
        m = bb * t
        y2 = y.view(m, nb, 2)
        dcat = dcat.contiguous()
        dzo = torch.empty((m, 88), device=y.device, dtype=torch.float32)
        call('rv_sigmoid_bwd', ptr(dcat), 176, ptr(donset.contiguous()) if donset is not None else None, 88,
             ptr(onset), 88, ptr(dzo), 88, m, 88, stream())
        dzf = dcat[:, 88:]
        dy = dwo = dbo = dwf = dbf = None
        if ctx.needs_input_grad[0]:
            dy = torch.empty_like(y)
            d2 = dy.view(m, nb, 2)
            gemm(dzo, _lin_t(wo).t(), d2[..., 0])
            gemm(dzf, _lin_t(wf).t(), d2[..., 1])
        pwo, pbo, pwf, pbf = ctx.params
        if ctx.needs_input_grad[1]:
            dwo, done = _param_wgrad(dzo.t(), y2[..., 0], pwo, None, pbo)
            if not done:
                dbo = _param_bgrad(dzo, pbo)
        if ctx.needs_input_grad[3]:
            dwf, done = _param_wgrad(dzf.t(), y2[..., 1], pwf, None, pbf)
            if not done:
                dbf = _param_bgrad(dzf, pbf)
        return dy, dwo, dbo, dwf, dbf


# --------------------------------------------------------------------------------------------
# local attention
# --------------------------------------------------------------------------------------------
def _adjacent(*ts):
    """True when the tensors are contiguous and laid out back to back in memory (flat parameter / gradient buffers)."""
    for a, b in zip(ts, ts[1:]):
        if a is None or b is None or not (a.is_contiguous() and b.is_contiguous()):
            return False
        if a.data_ptr() + a.numel() * a.element_size() != b.data_ptr():
            return False
        if a.untyped_storage().data_ptr() != b.untyped_storage().data_ptr():
            return False                     # neighbours by accident of the allocator, not views of one buffer
    return True


class LocalAttnFn(Function):
    """MutliHeadAttention1D.forward (model/UNet_onset.py:56-91) on x [B, L, Fin] (contiguous):
       returns (out [B, L, F], attention [B, L, G, 31]).

    The three projections write column slices [k | q | v] of ONE [B*L, 3F] buffer (the attention kernels take a row
    stride).  When W_k, W_q, W_v sit back to back in memory -- they do in FlatAdam's flat parameter buffer, in
    registration order -- the projection, its input gradient and its weight gradient are one GEMM each."""

    @staticmethod
    def forward(ctx, x, wq, wk, wv, rel, groups):
        need_gpu(x, wq)
        bb, l, fin = x.shape
        f = wq.shape[0]
        m = bb * l
        x2 = x.reshape(m, fin)
        qkv = torch.empty((m, 3 * f), device=x.device, dtype=torch.float32)
        k, q, v = qkv[:, :f], qkv[:, f:2 * f], qkv[:, 2 * f:]
        fused = _adjacent(wk, wq, wv)
        if fused:
            gemm(x2, torch.as_strided(wk, (3 * f, fin), (fin, 1)).t(), qkv)
        else:
            gemm(x2, wk.t(), k)
            gemm(x2, wq.t(), q)
            gemm(x2, wv.t(), v)
        out = torch.empty((bb, l, f), device=x.device, dtype=torch.float32)
        att = torch.empty((bb, l, groups, 31), device=x.device, dtype=torch.float32)
        call('rv_local_attn_fwd', ptr(q), ptr(k), ptr(v), 3 * f, ptr(_pack_rel(rel, f)), ptr(out), ptr(att), bb, l, groups,
             f // groups, stream())
        ctx.groups = groups
        ctx.fused = fused
        ctx.params = (wq, wk, wv, rel)
        ctx.save_for_backward(x2, wq, wk, wv, rel, qkv, att)
        ctx.mark_non_differentiable(att)
        ctx.set_materialize_grads(False)          # no zero-filled gradient tensor for the attention map
        return out, att

    @staticmethod
    def backward(ctx, dout, _datt):
        x2, wq, wk, wv, rel, qkv, att = ctx.saved_tensors
        g = ctx.groups
        bb, l = att.shape[0], att.shape[1]
        f = wq.shape[0]
        fin = x2.shape[1]
        dh = f // g
        m = bb * l
        k, q, v = qkv[:, :f], qkv[:, f:2 * f], qkv[:, 2 * f:]
        dout = dout.contiguous()
        dqkv = torch.empty_like(qkv)
        dk, dq, dv = dqkv[:, :f], dqkv[:, f:2 * f], dqkv[:, 2 * f:]
        de = torch.empty_like(att)
        call('rv_local_attn_bwd', ptr(dout), ptr(q), ptr(k), ptr(v), 3 * f, ptr(_pack_rel(rel, f)), ptr(att), ptr(dq), ptr(dk),
             ptr(dv), 3 * f, ptr(de), bb, l, g, dh, stream())
        dx = dwq = dwk = dwv = drel = None
        fused = ctx.fused
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x2)
            if fused:
                gemm(dqkv, _lin_t(torch.as_strided(wk, (3 * f, fin), (fin, 1))).t(), dx)
            else:
                gemm(dq, _lin_t(wq).t(), dx)
                gemm(dk, _lin_t(wk).t(), dx, accumulate=True)
                gemm(dv, _lin_t(wv).t(), dx, accumulate=True)
            dx = dx.view(bb, l, -1)
        pwq, pwk, pwv, prel = ctx.params
        if ctx.needs_input_grad[1]:
            gk, gq, gv = _grad_buf(pwk), _grad_buf(pwq), _grad_buf(pwv)
            if fused and _adjacent(gk, gq, gv):
                gemm(dqkv.t(), x2, torch.as_strided(gk, (3 * f, fin), (fin, 1)), accumulate=True, deterministic=DETERMINISTIC[0], defer_ok=True)
            elif fused:
                dw3 = torch.empty((3 * f, fin), device=x2.device, dtype=torch.float32)
                gemm(dqkv.t(), x2, dw3, deterministic=DETERMINISTIC[0])
                dwk, dwq, dwv = dw3[:f], dw3[f:2 * f], dw3[2 * f:]
            else:
                sk = None
                dwq = _param_wgrad(dq.t(), x2, pwq, sk)
                dwk = _param_wgrad(dk.t(), x2, pwk, sk)
                dwv = _param_wgrad(dv.t(), x2, pwv, sk)
        if ctx.needs_input_grad[4]:
            # drel[g*dh+f, w] = sum_{b,t} q[b,t,g,f] * de[b,t,g,w]
            grel = _grad_buf(prel)
            direct = grel is not None
            drel = grel.view(f, 31) if direct else torch.zeros((f, 31), device=q.device, dtype=torch.float32)
            de2 = de.view(m, g, 31)
            # one batched split-K launch over the heads: head h reads q[:, h*dh:], de[:, h, :], writes drel[h*dh:]
            gemm(q[:, :dh].t(), de2[:, 0, :], drel[:dh], accumulate=True, deterministic=DETERMINISTIC[0], batch=g, bstrides=(dh, 31, dh * 31),
                 defer_ok=direct)       # (a zero-filled temporary handed back to autograd is filled NOW, never by a deferred launch)
            drel = None if direct else drel.view_as(rel)
        return dx, dwq, dwk, dwv, drel, None


# --------------------------------------------------------------------------------------------
# losses and VAT primitives
# --------------------------------------------------------------------------------------------
_LOSS_TICKET = os.environ.get('RV_LOSS_TICKET') == '1'


def _reduce(kind, p, t):
    need_gpu(p, t)
    n = p.numel()
    out = torch.empty((), device=p.device, dtype=torch.float32)
    ws = torch.empty((n + 2047) // 2048, device=p.device, dtype=torch.float32)
    # two launches (partials, then the fp64 fold): the single-launch form (RV_LOSS_TICKET=1: the last workgroup to arrive folds) needs
    # device-scope fences, which cost more than the second launch on this multi-XCD part (21.02 vs 21.13 ms/step in situ, round 5)
    ticket = ARENA.take(1, p.device) if _LOSS_TICKET else None
    call('rv_reduce_mean', kind, ptr(p), ptr(t), n, ptr(out), ptr(ws), ptr(ticket), stream())
    return out


class _MeanLoss(Function):
    kind = 0

    @classmethod
    def _fwd(cls, ctx, p, t):
        p = p.contiguous()
        t = t.contiguous()
        if p.shape != t.shape:
            raise ValueError(f'target size {tuple(t.shape)} must match input size {tuple(p.shape)}')
        ctx.save_for_backward(p, t)
        return _reduce(cls.kind, p, t)

    @classmethod
    def _bwd(cls, ctx, gout):
        p, t = ctx.saved_tensors
        gp = torch.empty_like(p)
        call('rv_loss_bwd', cls.kind, ptr(p), ptr(t), p.numel(), ptr(gout.contiguous()), ptr(gp), stream())
        return gp, None


class BceMeanFn(_MeanLoss):
    """F.binary_cross_entropy(p, t) (mean; soft targets allowed; log clamped at -100)."""
    kind = 0

    @staticmethod
    def forward(ctx, p, t):
        return BceMeanFn._fwd(ctx, p, t)

    @staticmethod
    def backward(ctx, gout):
        return BceMeanFn._bwd(ctx, gout)


class MseMeanFn(_MeanLoss):
    """F.mse_loss(p, t) (mean)."""
    kind = 1

    @staticmethod
    def forward(ctx, p, t):
        return MseMeanFn._fwd(ctx, p, t)

    @staticmethod
    def backward(ctx, gout):
        return MseMeanFn._bwd(ctx, gout)


def bce_mean(p, t):
    return BceMeanFn.apply(p, t.detach())


def mse_mean(p, t):
    return MseMeanFn.apply(p, t.detach())


def abs_mean(x):
    return _reduce(2, x.detach().contiguous(), None)


def l2_norm(x):
    return _reduce(3, x.detach().contiguous(), None)


class VatPerturbFn(Function):
    """x_adv = clamp(x + scale * d / ||d||_2(last dim), 0, 1) with the gradient wrt d
    (model/UNet_onset.py:130-131 through _l2_normalize :165-171)."""

    @staticmethod
    def forward(ctx, x, d, scale):
        x = x.contiguous()
        d = d.contiguous()
        n = x.shape[-1]
        rows = x.numel() // n
        x_adv = torch.empty_like(x)
        call('rv_vat_perturb_fwd', ptr(x), ptr(d), rows, n, 1.0, scale, ptr(x_adv), None, None, None, stream())
        ctx.scale = scale
        ctx.save_for_backward(x, d)
        return x_adv

    @staticmethod
    def backward(ctx, g):
        x, d = ctx.saved_tensors
        n = x.shape[-1]
        gd = torch.empty_like(d)
        call('rv_vat_perturb_bwd', ptr(g.contiguous()), ptr(x), ptr(d), x.numel() // n, n, 1.0, ctx.scale, ptr(gd),
             stream())
        return None, gd, None


def vat_adversarial(x, g, prescale, eps, nan_flag=None):
    """d = g*prescale; r_adv = eps*d/||d||; x_adv = clamp(x + r_adv, 0, 1); also returns d/||d||
    (model/UNet_onset.py:141-151,162).  No autograd."""
    x = x.contiguous()
    g = g.contiguous()
    n = x.shape[-1]
    x_adv, r_adv, dn = torch.empty_like(x), torch.empty_like(x), torch.empty_like(x)
    call('rv_vat_perturb_fwd', ptr(x), ptr(g), x.numel() // n, n, prescale, eps, ptr(x_adv), ptr(r_adv), ptr(dn),
         ptr(nan_flag), stream())
    return x_adv, r_adv, dn


# --------------------------------------------------------------------------------------------
# front-end
# --------------------------------------------------------------------------------------------
def melspec(audio, tables, do_log, normalise, hop=512):
    """audio [B, nsamp] -> [B, T, n_mels] (time-major) log-mel / mel power; see rv_melspec_lognorm_fwd."""
    need_gpu(audio)
    audio = audio.float()
    if audio.stride(-1) != 1:
        audio = audio.contiguous()
    bb, nsamp = audio.shape
    t = 1 + nsamp // hop
    n_mels = tables['mel_start'].numel()
    out = torch.empty((bb, t, n_mels), device=audio.device, dtype=torch.float32)
    ws = torch.empty(2 * bb, device=audio.device, dtype=torch.int32)
    call('rv_melspec_lognorm_fwd', ptr(audio), audio.stride(0), bb, nsamp, ptr(tables['window']), ptr(tables['twiddle']),
         ptr(tables['mel_start']), ptr(tables['mel_len']), ptr(tables['mel_w']), tables['mel_w'].shape[1], n_mels, hop,
         1 if do_log else 0, 1 if normalise else 0, ptr(out), t, ptr(ws), stream())
    return out


# --------------------------------------------------------------------------------------------
# Onsets&Frames baseline pieces: BiLSTM, MaxPool2d((1,2)) + Dropout, Dropout
# --------------------------------------------------------------------------------------------
_SEED = [0x1234567]


def next_seed():
    """Counter-hash seed of the next dropout draw (deterministic per process; ``seed_dropout`` resets it)."""
    _SEED[0] = (_SEED[0] * 1664525 + 1013904223) & 0xFFFFFFFF
    return _SEED[0]


def seed_dropout(seed):
    _SEED[0] = int(seed) & 0xFFFFFFFF


_DROP_EPOCH = {}


def drop_epoch(device):
    """Device-resident step counter mixed into every dropout draw: TrainStep bumps it inside the (possibly graph-captured)
    step, so replays of a captured dropout launch draw new masks."""
    key = (device.type, device.index)
    t = _DROP_EPOCH.get(key)
    if t is None:
        t = _DROP_EPOCH[key] = torch.zeros(1, dtype=torch.int64, device=device)
    return t


def bump_drop_epoch(device):
    call('rv_counter_add', ptr(drop_epoch(device)), 1, None, stream())


_LSTM_ERR = {}


def _lstm_err(device):
    key = (device.type, device.index)
    t = _LSTM_ERR.get(key)
    if t is None:
        t = _LSTM_ERR[key] = torch.zeros(1, dtype=torch.int32, device=device)
    return t


def step_error_word(device):
    """The persistent per-device "this step is invalid" word: kernels atomicOr into it (BiLSTM time-outs), rv_adam_step skips
    the update while it is set, lstm_check() reads and clears it."""
    return _lstm_err(torch.device(device))


def lstm_check(device):
    """Raise if any BiLSTM launch since the last call gave up waiting for a neighbour workgroup (its workgroups were
    not co-resident): the results of that launch are invalid (and FlatAdam did not apply them).  Synchronises; call it at
    a point where the host waits for the step anyway (the per-step loss read-back), not per op."""
    t = _lstm_err(torch.device(device))
    if int(t.item()) != 0:
        t.zero_()
        raise RuntimeError('rv_lstm: a workgroup timed out waiting for its neighbours (launch not co-resident); '
                           'the BiLSTM outputs of this step are invalid')


class BiLstmFn(Function):
    """nn.LSTM(I, H, batch_first=True, bidirectional=True)(x)[0] with zero initial state
    (model/onset_frame_VAT.py:614; Onset_Stack.forward_LSTM :370-381, Combine_Stack.forward_LSTM :401-410).
    Parameter order: forward direction (w_ih [4H,I], w_hh [4H,H], b_ih, b_hh) then the ``_reverse`` four."""

    @staticmethod
    def forward(ctx, x, w_ih, w_hh, b_ih, b_hh, w_ih_r, w_hh_r, b_ih_r, b_hh_r):
        need_gpu(x, w_ih)
        bb, t, i = x.shape
        h = w_hh.shape[1]
        x2 = x.reshape(bb * t, i)
        xg = torch.empty((bb * t, 2, 4 * h), device=x.device, dtype=torch.float32)
        gemm(x2, w_ih.t(), xg[:, 0, :], b_ih + b_hh)
        gemm(x2, w_ih_r.t(), xg[:, 1, :], b_ih_r + b_hh_r)
        out = torch.empty((bb, t, 2 * h), device=x.device, dtype=torch.float32)
        train = any(ctx.needs_input_grad)
        gates = torch.empty((bb, t, 2, 4, h), device=x.device, dtype=torch.float32) if train else None
        cs = torch.empty((bb, t, 2, h), device=x.device, dtype=torch.float32) if train else None
        flags = torch.empty(_lib.load().rv_lstm_flag_bytes(h) // 4, device=x.device, dtype=torch.int32)
        w_hh, w_hh_r = w_hh.contiguous(), w_hh_r.contiguous()
        # a time-out is atomicOr'ed by the kernel itself into the persistent per-device word (ops.lstm_check / FlatAdam skip)
        call('rv_lstm_fwd', ptr(xg), ptr(w_hh), ptr(w_hh_r), ptr(out), ptr(gates), ptr(cs), ptr(flags), ptr(_lstm_err(x.device)),
             bb, t, h, stream())
        ctx.params = (w_ih, w_hh, b_ih, b_hh, w_ih_r, w_hh_r, b_ih_r, b_hh_r)
        ctx.save_for_backward(x2, out, gates, cs, w_ih, w_hh, w_ih_r, w_hh_r)
        ctx.dims = (bb, t, i, h)
        ctx.flags = flags
        return out

    @staticmethod
    def backward(ctx, dout):
        x2, out, gates, cs, w_ih, w_hh, w_ih_r, w_hh_r = ctx.saved_tensors
        bb, t, i, h = ctx.dims
        dout = dout.contiguous()
        dxg = torch.empty((bb * t, 2, 4 * h), device=dout.device, dtype=torch.float32)
        call('rv_lstm_bwd', ptr(dout), ptr(w_hh), ptr(w_hh_r), ptr(gates), ptr(cs), ptr(dxg), ptr(ctx.flags),
             ptr(_lstm_err(dout.device)), bb, t, h, stream())
        # h_{prev} of every step: the output shifted by one step along each direction's own time arrow
        hprev = torch.zeros((bb, t, 2, h), device=dout.device, dtype=torch.float32)
        o4 = out.view(bb, t, 2, h)
        hprev[:, 1:, 0] = o4[:, :-1, 0]
        hprev[:, :-1, 1] = o4[:, 1:, 1]
        hprev = hprev.view(bb * t, 2, h)
        grads = [None] * 9
        if ctx.needs_input_grad[0]:
            dx = torch.empty((bb * t, i), device=dout.device, dtype=torch.float32)
            gemm(dxg[:, 0, :], w_ih, dx)
            gemm(dxg[:, 1, :], w_ih_r, dx, accumulate=True)
            grads[0] = dx.view(bb, t, i)
        for d, base in ((0, 1), (1, 5)):
            dz = dxg[:, d, :]
            p_ih, p_hh, p_bi, p_bh = ctx.params[base - 1:base + 3]
            ih_bias = hh_bias = False
            if ctx.needs_input_grad[base]:
                if ctx.needs_input_grad[base + 2]:
                    grads[base], ih_bias = _param_wgrad(dz.t(), x2, p_ih, None, p_bi)
                else:
                    grads[base] = _param_wgrad(dz.t(), x2, p_ih, None)
            if ctx.needs_input_grad[base + 1]:
                if ctx.needs_input_grad[base + 3]:
                    grads[base + 1], hh_bias = _param_wgrad(dz.t(), hprev[:, d, :], p_hh, None, p_bh)
                else:
                    grads[base + 1] = _param_wgrad(dz.t(), hprev[:, d, :], p_hh, None)
            if ctx.needs_input_grad[base + 2] and not ih_bias:
                grads[base + 2] = _param_bgrad(dz, p_bi)
            if ctx.needs_input_grad[base + 3] and not hh_bias:
                grads[base + 3] = _param_bgrad(dz, p_bh)
        return tuple(grads)


class PoolDropFn(Function):
    """nn.MaxPool2d((1,2)) then nn.Dropout(p) on an NHWC tensor (model/onset_frame_VAT.py:336-343)."""

    @staticmethod
    def forward(ctx, x, p, training):
        need_gpu(x)
        x = x.contiguous()
        bb, hh, w, c = x.shape
        y = torch.empty((bb, hh, w // 2, c), device=x.device, dtype=torch.float32)
        code = torch.empty((bb, hh, w // 2, c), device=x.device, dtype=torch.uint8)
        p = float(p) if training else 0.0
        call('rv_maxpool_w2_dropout_fwd', ptr(x), ptr(y), ptr(code), bb * hh, w, c, p, next_seed(), ptr(drop_epoch(x.device)), stream())
        ctx.p, ctx.xshape = p, tuple(x.shape)
        ctx.save_for_backward(code)
        return y

    @staticmethod
    def backward(ctx, dy):
        code, = ctx.saved_tensors
        bb, hh, w, c = ctx.xshape
        dx = torch.empty(ctx.xshape, device=dy.device, dtype=torch.float32)
        call('rv_maxpool_w2_dropout_bwd', ptr(dy.contiguous()), ptr(code), ptr(dx), bb * hh, w, c, ctx.p, stream())
        return dx, None, None


class DropoutFn(Function):
    """nn.Dropout(p) in training mode (model/onset_frame_VAT.py:346-348)."""

    @staticmethod
    def forward(ctx, x, p):
        need_gpu(x)
        x = x.contiguous()
        y = torch.empty_like(x)
        code = torch.empty(x.shape, device=x.device, dtype=torch.uint8)
        call('rv_dropout', ptr(x), ptr(y), ptr(code), None, x.numel(), float(p), next_seed(), ptr(drop_epoch(x.device)), stream())
        ctx.p = float(p)
        ctx.save_for_backward(code)
        return y

    @staticmethod
    def backward(ctx, dy):
        code, = ctx.saved_tensors
        dy = dy.contiguous()
        dx = torch.empty_like(dy)
        call('rv_dropout', ptr(dy), ptr(dx), None, ptr(code), dy.numel(), ctx.p, 0, None, stream())
        return dx, None


def dropout(x, p, training):
    return DropoutFn.apply(x, p) if (training and p > 0.0) else x
