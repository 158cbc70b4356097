"""Training harness for the hot path.

``train_VAT_model`` keeps the reference's signature and loop semantics
(model/helper_functions.py:570-615).  ``FlatAdam`` is torch.optim.Adam + StepLR fused into one HIP kernel
over flat parameter / gradient buffers (the same flat gradient buffer is what gets all-reduced over RCCL
in data-parallel runs).  ``TrainStep`` runs one optimiser step -- optionally captured ONCE into a
hipGraph and replayed, which removes the ~1.5-2 k kernel launches per step from the host critical path.
"""
import contextlib
import os

import torch
import torch.distributed as dist

from . import dp, ops
from ._lib import call, ptr, stream


def cycle(iterable):
    while True:
        for item in iterable:
            yield item


_WEIGHTS = {}


def weighted_loss(losses, alpha):
    """model/helper_functions.py:589-595: every 'loss/train_LDS*' key weighs alpha/2, all others 1.  Device-resident
    terms are summed as ONE stacked dot product (a handful of launches instead of ~30 scalar kernels forward + backward)."""
    weight = lambda key: alpha / 2 if key.startswith('loss/train_LDS') else 1.0
    dev = [(k, v) for k, v in losses.items() if torch.is_tensor(v) and v.is_cuda]
    loss = 0
    for key, v in losses.items():
        if not (torch.is_tensor(v) and v.is_cuda):
            loss = loss + weight(key) * v if key.startswith('loss/train_LDS') else loss + v
    if len(dev) == 1:
        key, v = dev[0]
        loss = loss + (alpha * v / 2 if key.startswith('loss/train_LDS') else v)
    elif dev:
        ws = tuple(float(weight(k)) for k, _ in dev)
        ck = (ws, dev[0][1].device)
        w = _WEIGHTS.get(ck)
        if w is None:
            w = _WEIGHTS[ck] = torch.tensor(ws, dtype=torch.float32, device=dev[0][1].device)
        loss = loss + (torch.stack([v.reshape(()) for _, v in dev]) * w).sum()
    return loss


def train_VAT_model(model, iteration, ep, l_loader, ul_loader, optimizer, scheduler, clip_gradient_norm, alpha,
                    VAT=False, VAT_start=0):
    """Drop-in for model/helper_functions.py:570-615 (same arguments, same return value).  Works with any
    torch optimiser/scheduler pair, or with ``FlatAdam`` (pass ``scheduler=None``: StepLR is built in)."""
    model.train()
    batch_size = l_loader.batch_size
    total_loss = 0
    l_loader = cycle(l_loader)
    if ul_loader:
        ul_loader = cycle(ul_loader)
    for i in range(iteration):
        optimizer.zero_grad()
        batch_l = next(l_loader)
        if (ep < VAT_start) or (VAT is False):
            predictions, losses, _ = model.run_on_batch(batch_l, None, False)
        else:
            batch_ul = next(ul_loader)
            predictions, losses, _ = model.run_on_batch(batch_l, batch_ul, VAT)
        loss = weighted_loss(losses, alpha)
        loss.backward()
        total_loss += loss.item()
        optimizer.step()
        if scheduler is not None:
            scheduler.step()
        if clip_gradient_norm:
            # the reference clips AFTER the step (no effect on the update); kept for .grad parity
            if isinstance(optimizer, FlatAdam):
                optimizer.clip_grad_norm_(clip_gradient_norm)
            else:
                torch.nn.utils.clip_grad_norm_(model.parameters(), clip_gradient_norm)
        print(f'Train Epoch: {ep} [{i * batch_size}/{iteration * batch_size}'
              f'({100. * i / iteration:.0f}%)]'
              f"\tMain Loss: {sum(losses.values()):.6f}\t", end='\r')
    print(' ' * 100, end='\r')
    print(f'Train Epoch: {ep}\tLoss: {total_loss / iteration:.6f}')
    return predictions, losses, optimizer


def eval_model(model, ep, loader, VAT_start=0, VAT=False):
    """Drop-in for model/helper_functions.py:667-687: eval-mode run_on_batch over a loader, every loss key collected."""
    from collections import defaultdict
    model.eval()
    batch_size = loader.batch_size
    metrics = defaultdict(list)
    for i, batch in enumerate(loader):
        use_vat = not (ep < VAT_start or VAT is False)
        predictions, losses, _ = model.run_on_batch(batch, None, use_vat)
        for key, loss in losses.items():
            metrics[key].append(loss.item())
        print(f'Eval Epoch: {ep} [{i * batch_size}/{len(loader) * batch_size}({100. * i / len(loader):.0f}%)]'
              f"\tMain Loss: {sum(losses.values()):.6f}", end='\r')
    print(' ' * 100, end='\r')
    return metrics


class FlatAdam:
    """Adam(lr, betas=(0.9,0.999), eps=1e-8) + StepLR(step_size, gamma) on ONE flat fp32 buffer.

    Parameters are re-pointed to views of ``flat_param`` and their ``.grad`` to views of ``flat_grad``,
    so autograd accumulates straight into the bucket that data-parallel training all-reduces.
    Parameters that never receive a gradient keep a zero gradient and zero moments -> they do not move,
    which is what torch.optim.Adam does by skipping them (6 such tensors in UNet_Onset)."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, step_size=1000, gamma=0.98, data_parallel=None,
                 sync_error_word=False):
        self.params = [p for p in params if p.requires_grad]
        assert self.params, 'no trainable parameters'
        dev = self.params[0].device
        self._require_hip(dev)
        # every parameter starts on a 16-byte boundary of the flat buffers: the GEMM / conv kernels take 16-byte (LDS-DMA,
        # dwordx4) loads only from aligned rows, and a weight that follows a 229- or 88-element bias would otherwise sit
        # at an odd offset and fall back to scalar loads.  The pad elements stay zero (zero gradient, zero moments).
        self.offsets, n = [], 0
        for p in self.params:
            self.offsets.append(n)
            n += (p.numel() + 3) & ~3
        self.flat_param = torch.zeros(n, device=dev, dtype=torch.float32)
        self.flat_grad = torch.zeros(n, device=dev, dtype=torch.float32)
        self.flat_grad_side = None            # twin bucket of the side stream (TrainStep two-stream mode)
        self.flat_grad_sides = []             # ... all twins (one per side stream in use)
        self.exp_avg = torch.zeros(n, device=dev, dtype=torch.float32)
        self.exp_avg_sq = torch.zeros(n, device=dev, dtype=torch.float32)
        self.step_count = torch.zeros((), device=dev, dtype=torch.int64)
        self.norm_buf = torch.zeros((), device=dev, dtype=torch.float32)
        for p, off in zip(self.params, self.offsets):
            k = p.numel()
            self.flat_param[off:off + k].copy_(p.data.reshape(-1))
            p.data = self.flat_param[off:off + k].view_as(p.data)
            p.grad = self.flat_grad[off:off + k].view_as(p.data)
        self.n = n
        self.lr, self.betas, self.eps, self.step_size, self.gamma = lr, betas, eps, step_size, gamma
        self.grad_scale = 1.0
        ops.step_error_word(dev)                # create the per-device error word outside any graph capture
        # the ONE all-reduce call site of a step is step(); None = whenever a process group with world > 1 is up
        self.data_parallel = data_parallel
        # models with a recurrence (BiLSTM time-out flag, `model.has_recurrence`): data-parallel ranks MAX the per-device error word
        # along with the gradients so that all of them skip the same steps (see allreduce_gradients).  Pass it here for every
        # data-parallel loop (eager train_VAT_model included); TrainStep also switches it on from the model.
        self.sync_error_word = bool(sync_error_word)
        ops.invalidate_weight_cache()

    def __del__(self):
        try:
            ops.drop_packs_of(self.flat_param)     # the packed-weight cache holds views of this buffer
        except Exception:  # noqa: BLE001 -- interpreter shutdown
            pass

    @staticmethod
    def _require_hip(dev):
        if dev.type != 'cuda':
            raise RuntimeError('FlatAdam runs on a HIP device only (no CPU fallback)')

    def enable_side_bucket(self, n=1):
        """n twin buckets (one per side stream); returns them as a list."""
        while len(self.flat_grad_sides) < n:
            self.flat_grad_sides.append(torch.zeros_like(self.flat_grad))
        self.flat_grad_side = self.flat_grad_sides[0]
        return self.flat_grad_sides[:n]

    def merge_side_grads(self):
        """flat_grad += side-stream twins (call on the main stream after joining the side streams)."""
        for twin in self.flat_grad_sides:
            self.flat_grad.add_(twin)

    def zero_grad(self, set_to_none=False):
        self.flat_grad.zero_()
        for twin in self.flat_grad_sides:
            twin.zero_()
        for p, off in zip(self.params, self.offsets):      # re-attach if something replaced .grad
            if p.grad is None or p.grad.data_ptr() != self.flat_grad.data_ptr() + 4 * off:
                p.grad = self.flat_grad[off:off + p.numel()].view_as(p.data)

    def step(self):
        if self.data_parallel or self.data_parallel is None:
            allreduce_gradients(self)              # the ONE collective of a step (no-op without a process group)
        call('rv_adam_step', ptr(self.flat_param), ptr(self.flat_grad), ptr(self.exp_avg), ptr(self.exp_avg_sq), self.n,
             ptr(self.step_count), self.lr, self.step_size, self.gamma, self.betas[0], self.betas[1], self.eps,
             self.grad_scale, ptr(ops.step_error_word(self.flat_grad.device)), stream())
        # (same skip word: a step whose update was skipped advances neither StepLR nor the bias correction -- `step_count`, and with
        # it the `step` of state_dict(), counts APPLIED updates, not calls; the word stays set until check() reports it, so every
        # step between a fault and the next check() is skipped on all ranks alike)
        call('rv_counter_add', ptr(self.step_count), 1, ptr(ops.step_error_word(self.flat_grad.device)), stream())
        ops.invalidate_weight_cache()

    def clip_grad_norm_(self, max_norm):
        ws = torch.empty((self.n + 2047) // 2048, device=self.flat_grad.device, dtype=torch.float32)
        call('rv_reduce_mean', 3, ptr(self.flat_grad), None, self.n, ptr(self.norm_buf), ptr(ws), None, stream())
        call('rv_clip_scale', ptr(self.flat_grad), self.n, ptr(self.norm_buf), float(max_norm), stream())
        return self.norm_buf

    def current_lr(self):
        return self.lr * self.gamma ** (int(self.step_count.item()) // self.step_size)

    def state_dict(self):
        """The `torch.optim.Adam.state_dict()` layout (what the reference writes to last-optimizer-state.pt,
        train_UNet_Onset_VAT.py:152): per-parameter `step` / `exp_avg` / `exp_avg_sq` keyed by parameter index, plus one
        param group.  Checkpoints therefore interchange with the reference and with `fused_optimizer=False`.  Parameters
        that never received a gradient have no state entry, exactly like torch (Adam skips `grad is None`)."""
        steps = int(self.step_count.item())
        touched = [bool(x) for x in torch.stack([self.exp_avg_sq[o:o + p.numel()].any() for p, o in zip(self.params, self.offsets)]).tolist()]
        state = {}
        for i, (p, off) in enumerate(zip(self.params, self.offsets)):
            if steps and touched[i]:
                k = p.numel()
                state[i] = {'step': torch.tensor(float(steps)), 'exp_avg': self.exp_avg[off:off + k].view_as(p).clone(),
                            'exp_avg_sq': self.exp_avg_sq[off:off + k].view_as(p).clone()}
        group = {'lr': self.current_lr(), 'betas': tuple(self.betas), 'eps': self.eps, 'weight_decay': 0, 'amsgrad': False,
                 'maximize': False, 'foreach': None, 'capturable': False, 'differentiable': False, 'fused': None,
                 'initial_lr': self.lr, 'params': list(range(len(self.params)))}
        return {'state': state, 'param_groups': [group]}

    def load_state_dict(self, sd):
        """Accepts the torch.optim.Adam layout (from this class, from torch.optim.Adam or from a reference run)."""
        groups = sd['param_groups']
        index = [i for g in groups for i in g['params']]
        if len(index) != len(self.params):
            raise ValueError(f'optimizer state has {len(index)} parameters, the model has {len(self.params)}')
        self.exp_avg.zero_()
        self.exp_avg_sq.zero_()
        steps = 0
        for slot, key in enumerate(index):
            st = sd['state'].get(key)
            if not st:
                continue
            p, off = self.params[slot], self.offsets[slot]
            if tuple(st['exp_avg'].shape) != tuple(p.shape):
                raise ValueError(f'optimizer state {key}: shape {tuple(st["exp_avg"].shape)} vs parameter {tuple(p.shape)}')
            self.exp_avg[off:off + p.numel()].copy_(st['exp_avg'].reshape(-1))
            self.exp_avg_sq[off:off + p.numel()].copy_(st['exp_avg_sq'].reshape(-1))
            steps = max(steps, int(float(st['step'])))
        self.step_count.fill_(steps)
        g0 = groups[0]
        self.lr = float(g0.get('initial_lr', g0['lr']))
        self.betas, self.eps = tuple(g0['betas']), float(g0['eps'])


def allreduce_gradients(opt):
    """ONE all-reduce (sum) of the flat gradient bucket per optimiser step -- RCCL over xGMI, or gloo through a pinned host
    buffer (reconvat_amd/dp.py) -- after the final backward and never inside the VAT power iteration; the 1/world_size mean is
    folded into the Adam kernel."""
    # (RV_DP_FORCE_ALLREDUCE=1: also with a single-rank group -- lets a one-GPU box execute the RCCL call in place, tests/test_cli_gpu.py)
    min_world = 1 if os.environ.get('RV_DP_FORCE_ALLREDUCE') == '1' else 2
    if dp.active() and dist.get_world_size() >= min_world:
        hook = getattr(opt, 'dp_hook', None)           # test instrumentation: sees the bucket before and after the collective
        if hook is not None:
            hook('pre', opt)
        dp.all_reduce(opt.flat_grad, dist.ReduceOp.SUM)
        opt.grad_scale = 1.0 / dist.get_world_size()
        opt.allreduce_calls = getattr(opt, 'allreduce_calls', 0) + 1
        if getattr(opt, 'sync_error_word', False):
            # models with a recurrence (BiLSTM time-out flag): every rank must skip the same steps, or the replicas diverge and the
            # rank that raises in check() leaves its peers blocked in the next collective -- MAX the flag along with the gradients
            dp.all_reduce(ops.step_error_word(opt.flat_grad.device), dist.ReduceOp.MAX)
        if hook is not None:
            hook('post', opt)


class TrainStep:
    """One optimiser step of the reference loop body (zero_grad, run_on_batch, weighted sum, backward,
    [all-reduce], Adam+StepLR) on static device buffers.  With ``graph=True`` the forward+backward is
    captured once into a hipGraph and replayed; the all-reduce and the optimiser kernel stay outside the
    graph so that the collective is an ordinary RCCL call."""

    def __init__(self, model, opt, batch, batch_ul, alpha=1.0, VAT=True, clip=3.0, graph=True, dual_stream=True, bf16_backward=False):
        self.model, self.opt, self.alpha, self.VAT, self.clip = model, opt, alpha, VAT, clip
        self.batch = {k: v.clone() for k, v in batch.items() if torch.is_tensor(v)}
        self.batch_ul = {k: v.clone() for k, v in batch_ul.items() if torch.is_tensor(v)} if batch_ul else None
        self.graph = None
        self._keep = []                       # what the captured graph pins (ops.keep_scope): dropped with the graph in release()
        self.losses = None
        self.loss = None
        self.use_graph = graph
        self.pack_plan = None
        # opt-in experiment (BASELINE config 3): bf16 operands on the matrix pipe for the BACKWARD 3x3 convs of the final graphs; every
        # forward pass and the whole power iteration stay fp32, so all loss terms / posteriorgrams are unchanged (ops.BF16)
        self.bf16_backward = bool(bf16_backward)
        if self.bf16_backward and os.environ.get('RV_BF16_LOG'):
            import sys
            print('[bf16] backward convs of the final graphs use bf16 operands', file=sys.stderr)
        self.dual_stream = dual_stream        # the two VAT chains on two HIP streams (model._vat_two_streams)
        self._merger = ops.WgradMerger()       # one weight-gradient launch per layer and gradient bucket, not per backward pass
        self._dual_ready = False              # ... from the second step on: the first one packs weights and autotunes
        if getattr(model, 'has_recurrence', False):
            opt.sync_error_word = True
        self._top_up_pools()

    def _top_up_pools(self):
        """Pinned host tables a hipGraph capture pops from the process-wide pools (one per stream and table kind; they are freed with
        the graph, never returned): topped up to what ONE capture of this step needs, here and again at the top of every capture() --
        so re-capturing the same object, or building several steps before any of them captures, never runs a pool dry."""
        if not self.batch['audio'].is_cuda:
            return
        streams = 1 + (getattr(self.model, 'side_streams', 1) if self.dual_stream else 0)
        if self.dual_stream:
            ops.prepare_replay_pool()
        if self.use_graph:
            ops.prepare_wgrad_tables(max(4, streams))          # one table per stream
            ops.prepare_gemm_tables(max(8, 2 * streams))       # one per stream and operand orientation

    def _dual_now(self):
        """Whether the next _fwd_bwd() runs the two-chain schedule."""
        return bool(self.dual_stream and self._dual_ready and self.batch_ul is not None and self.VAT)

    def _merging(self):
        """Whether the next _fwd_bwd() hands its weight gradients to the merger (one launch per layer and step)."""
        return bool(os.environ.get('RV_WGRAD_MERGE', '1') != '0' and getattr(self.model, 'defer_wgrad_reductions', True)
                    and not ops.DETERMINISTIC[0] and not self.bf16_backward)

    def _fwd_bwd(self):
        self.opt.zero_grad()
        ops.ARENA.begin_step(self.opt.flat_grad.device)
        ops.bump_drop_epoch(self.opt.flat_grad.device)
        prev_dual, prev_side = ops.DUAL_STREAM[0], ops.SIDE_GRADS[0]
        dual = self._dual_now()
        ops.DUAL_STREAM[0] = dual
        if dual:
            dev = self.opt.flat_grad.device
            n_side = getattr(self.model, 'side_streams', 1)
            twins = self.opt.enable_side_bucket(n_side)
            ops.SIDE_GRADS[0] = (self.opt.flat_grad, {ops.side_stream(dev, i).cuda_stream: twins[i] for i in range(n_side)})
        try:
            # (RV_DETERMINISTIC=1: per-layer reductions in stream order instead of the table launch with its fp32 atomics, ops.DETERMINISTIC)
            defer = os.environ.get('RV_DEFER_WGRAD', '1') != '0' and getattr(self.model, 'defer_wgrad_reductions', True) and not ops.DETERMINISTIC[0]
            defer_g = os.environ.get('RV_DEFER_GEMM', '1') != '0' and getattr(self.model, 'defer_param_gemms', True)
            # (the merged launches do not need the reduction table; without it they stay on the launching chain)
            merge = self._merger if self._merging() else None
            with ops.bf16_final_graphs(fwd=False, bwd=self.bf16_backward), ops.direct_param_grads(), \
                    (ops.deferred_wgrad_reductions() if defer else contextlib.nullcontext()) as pending, \
                    (ops.deferred_param_gemms() if defer_g else contextlib.nullcontext()) as pending_g, \
                    ops.wgrad_merging(merge, bool(dual)):
                # conv grads accumulate straight into the flat bucket; their partial sums are folded by ONE launch per stream, and the
                # linear / attention parameter-gradient GEMMs run as ONE grouped launch per stream
                _, losses, _ = self.model.run_on_batch(self.batch, self.batch_ul, self.VAT)
                loss = weighted_loss(losses, self.alpha)
                loss.backward()
                if merge is not None:
                    merge.finish()                 # (a layer whose passes did not all arrive: launched now, merged as far as they came)
                if pending is not None:
                    pending.flush()
                if pending_g is not None:
                    pending_g.flush()
            if dual:
                # the side stream ran the reconstruction branch's backward into its own bucket: join, then fold
                for i in range(n_side):
                    torch.cuda.current_stream().wait_stream(ops.side_stream(dev, i))
                self.opt.merge_side_grads()
        finally:
            ops.DUAL_STREAM[0], ops.SIDE_GRADS[0] = prev_dual, prev_side
        ops.ARENA.end_step()
        self.losses = {k: v.detach() for k, v in losses.items()}
        self.loss = loss.detach()

    def load(self, batch, batch_ul=None):
        for k, v in self.batch.items():
            v.copy_(batch[k], non_blocking=True)
        if self.batch_ul is not None and batch_ul is not None:
            for k, v in self.batch_ul.items():
                v.copy_(batch_ul[k], non_blocking=True)

    def _bn_state(self):
        """BatchNorm running statistics / counters (NOT the 17.7 MB spectrogram tables) and the dropout epoch."""
        bufs = [b for n, b in self.model.named_buffers() if n.rsplit('.', 1)[-1] in ('running_mean', 'running_var', 'num_batches_tracked')]
        return bufs + [ops.drop_epoch(self.opt.flat_grad.device)]

    def capture(self, warmup=2):
        """Warm up (weight packing, conv autotune, allocator) and capture the step.  The warm-up passes run the real step
        body on the first batch but must leave no trace in the training trajectory: they apply no optimiser step, and the
        BatchNorm running statistics / `num_batches_tracked` / dropout epoch they advance are restored afterwards, so the
        first replay sees exactly the state the reference loop would (checkpointed buffers stay comparable)."""
        self.model.train()
        self._top_up_pools()
        state = self._bn_state()
        saved = [b.clone() for b in state]
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            ops.invalidate_weight_cache()
            for w in range(warmup):
                self._fwd_bwd()
                # after the first step every weight is packed (nothing changes them here) and every conv shape is
                # tuned: from now on the step may use the second stream
                self._dual_ready = True
            # the weight-gradient merger learns the pass counts of a mode (one chain / two chains) from one unmerged step of that mode:
            # keep warming up until the mode about to be captured has been seen, or the graph would replay unmerged forever
            while self._merging() and not self._merger.knows(self._dual_now()):
                self._fwd_bwd()
        torch.cuda.current_stream().wait_stream(s)
        for b, v in zip(state, saved):
            b.copy_(v)
        torch.cuda.synchronize()
        # all weights are repacked by ONE launch after every optimiser step (outside the graph); the graph itself is
        # captured with a fresh cache and therefore holds no packing kernels
        self.pack_plan = ops.PackPlan(self.opt.flat_grad.device)
        self.pack_plan.run()
        self.graph = torch.cuda.CUDAGraph()
        with ops.keep_scope(self._keep), torch.cuda.graph(self.graph):
            self._fwd_bwd()

    def release(self):
        """Drop the captured graph and everything it pinned (pinned host tables, their device copies, the static losses).  Called
        by __del__; call it explicitly to re-capture (e.g. another batch shape) without waiting for the garbage collector."""
        self.graph = None
        self.losses = self.loss = None
        self._keep.clear()

    def __del__(self):
        try:
            self.release()
        except Exception:  # noqa: BLE001 -- interpreter shutdown
            pass

    def __call__(self):
        if self.use_graph:
            if self.graph is None:
                self.capture()
            self.graph.replay()
        else:
            self.model.train()
            self._fwd_bwd()
            if self.pack_plan is None:
                self.pack_plan = ops.PackPlan(self.opt.flat_grad.device)
            self._dual_ready = True
        self.opt.step()                        # [RCCL all-reduce of the flat bucket] + Adam + StepLR
        if self.clip:
            self.opt.clip_grad_norm_(self.clip)
        self.pack_plan.run()                   # the next step's forward finds every packed weight fresh
        return self.loss

    def check(self):
        """Host-side health check of the steps since the last call (synchronises: call it where the host reads the loss
        anyway).  Graph replays cannot assert on the device, so the reference's VAT NaN assert (model/UNet_onset.py:146-147)
        and the BiLSTM time-out are device flags read here."""
        vat = getattr(self.model, 'vat_loss', None)
        flag = getattr(vat, 'nan_flag', None)
        if flag is not None and int(flag.item()) != 0:
            flag.zero_()
            raise AssertionError('r_adv has nan, please debug tune down the XI for VAT')
        ops.lstm_check(self.opt.flat_grad.device)
