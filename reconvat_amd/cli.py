"""Shared body of the drop-in training entry points (train_UNet_Onset_VAT.py / train_UNet_VAT.py /
train_baseline_onset_frame_VAT.py).

Keeps the reference CLI (`python train_UNet_Onset_VAT.py with key=value ...`, keys and defaults of
train_UNet_Onset_VAT.py:28-78 / train_UNet_VAT.py:26-79) and loop semantics (:128-154): epochs of
`iteration` optimiser steps through train_VAT_model, a checkpoint every `saving_freq` epochs
(model-{ep}.pt + last-optimizer-state.pt), scalar logging of every loss key per epoch.
MI355X-side additions: one process per GPU under torch.distributed.run (data-parallel, ONE flat RCCL
gradient all-reduce per step), the fused FlatAdam, optional whole-step hipGraph replay (`graph=True`) and
`train_on=Synthetic`.
"""
import json
import os
import time
from datetime import datetime

import numpy as np
import torch
from torch.utils.data import DataLoader

from . import UNet_Onset, UNet, dp
from .dataset import prepare_VAT_dataset
from .train import FlatAdam, TrainStep, train_VAT_model, cycle

ds_ksize, ds_stride = (2, 2), (2, 2)
mode = 'imagewise'
logging_freq = 100
saving_freq = 200


def base_config(o, onset_script):
    """Config scope shared by both scripts; `o` holds CLI overrides (needed for the derived entries)."""
    c = dict(
        root='runs', device='cuda:0', log=True, w_size=31, spec='Mel', resume_iteration=None,
        train_on='MAPS' if onset_script else 'Wind', n_heads=4, position=True, iteration=10, VAT_start=0, alpha=1,
        VAT=True, XI=1e-6, eps=2, small=False, supersmall=False, onset_stack=True,
        KL_Div=False, reconstruction=False, batch_size=8,
        train_batch_size=8 if onset_script else 1, sequence_length=327680, epoches=20000,
        step_size_up=100, max_lr=1e-4, learning_rate=1e-3, learning_rate_decay_steps=1000,
        learning_rate_decay_rate=0.98, leave_one_out=None, clip_gradient_norm=3, refresh=False,
        # MI355X-side extras (not in the reference)
        graph=True, fused_optimizer=True, saving_freq=saving_freq, logging_freq=logging_freq, device_feed=True,
        dtype='fp32',          # 'bf16': opt-in experiment -- bf16-operand backward convs of the final graphs (forward stays fp32)
    )
    c.update(o)
    if torch.cuda.is_available() and torch.cuda.get_device_properties(torch.cuda.current_device()).total_memory < 10e9:
        if 'batch_size' not in o:
            c['batch_size'] //= 2
        if 'sequence_length' not in o:
            c['sequence_length'] //= 2
        print(f"Reducing batch size to {c['batch_size']} and sequence_length to {c['sequence_length']} to save memory")
    c.setdefault('validation_length', c['sequence_length'])
    if 'validation_length' not in o:
        c['validation_length'] = c['sequence_length']
    stamp = datetime.now().strftime('%y%m%d-%H%M%S')
    if 'logdir' not in o:
        if onset_script:
            c['logdir'] = (f"{c['root']}/Unet_Onset-recons={c['reconstruction']}-XI={c['XI']}-eps={c['eps']}-alpha={c['alpha']}"
                           f"-train_on=small_{c['small']}_{c['train_on']}-w_size={c['w_size']}-n_heads={c['n_heads']}"
                           f"-lr={c['learning_rate']}-" + stamp)
        else:
            c['logdir'] = (f"{c['root']}/Unet-recons={c['reconstruction']}-XI={c['XI']}-eps={c['eps']}-alpha={c['alpha']}"
                           f"-train_on=small_{c['small']}_{c['train_on']}-w_size={c['w_size']}-n_heads={c['n_heads']}"
                           f"-lr={c['learning_rate']}-" + stamp)
    return c


def baseline_config(o):
    """Config scope of train_baseline_onset_frame_VAT.py:25-73 (model_name='onset_frame' is the one on the MI355X path)."""
    c = dict(
        root='runs', onset_stack=True, device='cuda:0', log=True, w_size=31, model_complexity=48, spec='Mel',
        resume_iteration=None, train_on='String', iteration=10, alpha=1, VAT=False, XI=1e-6, eps=1e-1, VAT_mode='all',
        model_name='onset_frame', VAT_start=0, small=True, supersmall=False, batch_size=8, train_batch_size=8,
        sequence_length=327680, epoches=20000, learning_rate=5e-4, learning_rate_decay_steps=10000,
        learning_rate_decay_rate=0.98, leave_one_out=None, clip_gradient_norm=3, refresh=False, reconstruction=False,
        graph=True, fused_optimizer=True, saving_freq=saving_freq, logging_freq=logging_freq, device_feed=True,
    )
    c.update(o)
    if c['model_name'] not in ('onset_frame', 'frame', 'onset'):
        raise NotImplementedError("model_name must be 'onset_frame', 'frame' or 'onset' (the 'attention' variant is not built)")
    if 'validation_length' not in o:
        c['validation_length'] = c['sequence_length']
    if 'logdir' not in o:
        c['logdir'] = f"{c['root']}/baseline_Onset_Frame-" + datetime.now().strftime('%y%m%d-%H%M%S')
    return c


class ScalarLog:
    """TensorBoard SummaryWriter when available, else one JSON line per (tag, step)."""

    def __init__(self, logdir):
        os.makedirs(logdir, exist_ok=True)
        try:
            from torch.utils.tensorboard import SummaryWriter
            self.tb = SummaryWriter(logdir)
        except Exception:  # noqa: BLE001 -- tensorboard is optional
            self.tb = None
        self.f = open(os.path.join(logdir, 'scalars.jsonl'), 'a')

    def add_scalar(self, tag, value, global_step):
        if self.tb is not None:
            self.tb.add_scalar(tag, value, global_step=global_step)
        self.f.write(json.dumps({'tag': tag, 'value': value, 'step': global_step}) + '\n')
        self.f.flush()


def log_validation(model, val_set, l_loader, ep, writer, reconstruction, onset_script, VAT, VAT_start):
    """The scalar half of tensorboard_log (model/helper_functions.py:120-141): `evaluate_wo_velocity(validation_dataset, model,
    reconstruction=reconstruction, VAT=False)` -- precision / recall / f1 of every non-chroma metric logged under its key -- then
    `eval_model(model, ep, supervised_loader, VAT_start, VAT)` -- the mean of every eval-mode loss key."""
    from .evaluate import evaluate_wo_velocity
    from .train import eval_model
    was_training = model.training
    model.eval()
    with torch.no_grad():
        metrics = evaluate_wo_velocity(val_set, model, reconstruction=reconstruction, VAT=False)
    for key, values in metrics.items():
        if key.startswith('metric/') and values:
            _, category, name = key.split('/')
            print(f'{category:>32} {name:25}: {np.mean(values):.3f} \u00b1 {np.std(values):.3f}')
            if ('precision' in name or 'recall' in name or 'f1' in name) and 'chroma' not in name:
                writer.add_scalar(key, float(np.mean(values)), ep)
    test_losses = eval_model(model, ep, l_loader, VAT_start, VAT)
    for key, values in test_losses.items():
        if key.startswith('loss/') and values:
            writer.add_scalar(key, float(np.mean(values)), ep)
    model.train(was_training)


def run_training(onset_script, spec, resume_iteration, train_on, batch_size, sequence_length, small, supersmall,
                 train_batch_size, learning_rate, learning_rate_decay_steps, learning_rate_decay_rate, alpha,
                 clip_gradient_norm, validation_length, refresh, device, epoches, logdir, log, iteration, VAT_start, VAT,
                 XI, eps, reconstruction, graph, fused_optimizer, saving_freq, device_feed=True, model_complexity=48, model_name='onset_frame', VAT_mode='all',
                 logging_freq=logging_freq, dtype='fp32', **_unused):
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    if not str(device).startswith('cuda') or not torch.cuda.is_available():
        raise SystemExit(f"device={device!r}: this build runs the training path on an MI355X only (hand-written HIP kernels, no CPU "
                         "fallback); use device=cuda:0.  The reference's CPU plumbing run maps to the same command with device=cuda:0.")
    if world > 1:
        # one process per GPU (RV_DP_BACKEND=nccl: RCCL; gloo + RV_DP_SAME_GPU=1: every rank on cuda:0, for one-GPU boxes).  The
        # training group keeps the default collective timeout; rank 0 alone runs the validation passes (whole songs can take
        # minutes) while its peers wait in dp.wait_for_rank0() on a separate long-timeout gloo group
        dev_ = dp.local_device()
        device = f'cuda:{dev_.index}'
        torch.cuda.set_device(dev_)
        dp.init(dev_, long_wait_group=True)
    elif str(device).startswith('cuda'):
        torch.cuda.set_device(torch.device(device))

    l_set, ul_set, val_set, full_validation = prepare_VAT_dataset(sequence_length=sequence_length, validation_length=validation_length,
                                                        refresh=refresh, device=device, small=small, supersmall=supersmall,
                                                        dataset=train_on, rank=rank)
    if device_feed and hasattr(l_set, 'data') and str(device).startswith('cuda'):
        # corpus resident in HBM, batches cropped/decoded by rv_crop_segments (reconvat_amd/feed.py)
        from .feed import device_loader
        l_loader = device_loader(l_set, train_batch_size, device, rank=rank, world=world, seed=42 + rank)
        ul_loader = device_loader(ul_set, batch_size, device, rank=rank, world=world, seed=43 + rank) if VAT else None
    else:
        ul_loader = DataLoader(ul_set, batch_size, shuffle=True, drop_last=True) if VAT else None
        l_loader = DataLoader(l_set, train_batch_size, shuffle=True, drop_last=True)

    # which recordings this rank trains on (labelled / unlabelled corpus and this rank's shard of each)
    os.makedirs(logdir, exist_ok=True)
    with open(os.path.join(logdir, f'corpus_rank{rank}.json'), 'w') as fh:
        paths = lambda ds: [str(d['path']) for d in getattr(ds, 'data', [])]
        json.dump({'train_on': train_on, 'rank': rank, 'world': world, 'labelled': paths(l_set), 'unlabelled': paths(ul_set),
                   'labelled_shard': list(getattr(l_loader, 'paths', [])), 'unlabelled_shard': list(getattr(ul_loader, 'paths', [])),
                   'validation': paths(val_set), 'full_validation': paths(full_validation)}, fh, indent=0)

    torch.manual_seed(0)                                   # identical initial weights on every rank
    if onset_script == 'baseline':
        from . import onset_frames as onf
        from .constants import N_BINS, MAX_MIDI, MIN_MIDI
        cls = {'onset_frame': onf.OnsetsAndFrames_VAT_full, 'frame': onf.Frame_stack_VAT, 'onset': onf.Onset_stack_VAT}[model_name]
        model = cls(N_BINS, MAX_MIDI - MIN_MIDI + 1, model_complexity=model_complexity, log=log, mode=mode, spec=spec, XI=XI, eps=eps,
                    VAT_mode=VAT_mode)
    else:
        cls = UNet_Onset if onset_script else UNet
        model = cls(ds_ksize, ds_stride, log=log, reconstruction=reconstruction, mode=mode, spec=spec, device=device, XI=XI, eps=eps)
    if resume_iteration is not None:
        sd = torch.load(os.path.join(logdir, f'model-{resume_iteration}.pt'), map_location='cpu')
        model.load_state_dict(sd)
    model.to(device)
    torch.manual_seed(1 + rank)
    scheduler = None
    if fused_optimizer:
        optimizer = FlatAdam(model.parameters(), lr=learning_rate, step_size=learning_rate_decay_steps,
                             gamma=learning_rate_decay_rate, data_parallel=world > 1,
                             sync_error_word=bool(getattr(model, 'has_recurrence', False)))
    else:
        optimizer = torch.optim.Adam(model.parameters(), learning_rate)
        scheduler = torch.optim.lr_scheduler.StepLR(optimizer, step_size=learning_rate_decay_steps, gamma=learning_rate_decay_rate)
    first_epoch = 1
    if resume_iteration is not None:
        optimizer.load_state_dict(torch.load(os.path.join(logdir, 'last-optimizer-state.pt'), map_location=device))
        first_epoch = int(resume_iteration) + 1
        if scheduler is not None:                          # StepLR is not checkpointed by the reference: rebuild its position
            steps_done = max((int(float(st['step'])) for st in optimizer.state_dict()['state'].values()), default=0)
            scheduler.last_epoch = steps_done
        if rank == 0:
            if fused_optimizer:
                print(f'Resumed from model-{resume_iteration}.pt: optimiser step {int(optimizer.step_count.item())}, '
                      f'lr {optimizer.current_lr():.6e}')
            else:
                print(f'Resumed from model-{resume_iteration}.pt')
    n_params = sum(p.numel() for p in model.parameters())
    if rank == 0:
        print(f'{cls.__name__}: {n_params} parameters, world size {world}, device {device}')
    writer = ScalarLog(logdir) if rank == 0 else None

    step_runner = None
    for ep in range(first_epoch, epoches + 1):
        use_vat = VAT and ep >= VAT_start
        if graph and fused_optimizer:
            # whole-step hipGraph replay on static buffers (same loop semantics as train_VAT_model)
            model.train()
            li, ui = cycle(l_loader), (cycle(ul_loader) if use_vat else None)
            total = 0.0
            t_epoch = time.perf_counter()
            for _ in range(iteration):
                bl = next(li)
                bul = next(ui) if use_vat else None
                if step_runner is None or step_runner.VAT != use_vat:
                    step_runner = TrainStep(model, optimizer, bl, bul, alpha=alpha, VAT=use_vat, clip=clip_gradient_norm, graph=True,
                                            bf16_backward=str(dtype).lower() in ('bf16', 'bfloat16'))
                else:
                    step_runner.load(bl, bul)
                total += float(step_runner())
                step_runner.check()                # VAT NaN assert + BiLSTM time-out flag (the loss read-back just synchronised)
            losses = step_runner.losses
            if rank == 0:
                print(f'Train Epoch: {ep}\tLoss: {total / iteration:.6f}\t({(time.perf_counter() - t_epoch) / iteration * 1e3:.1f} ms/step)')
        else:
            _, losses, optimizer = train_VAT_model(model, iteration, ep, l_loader, ul_loader if VAT else None, optimizer,
                                                   scheduler, clip_gradient_norm, alpha, VAT, VAT_start)
        if onset_script == 'baseline' and str(device).startswith('cuda'):
            from . import ops
            ops.lstm_check(torch.device(device))          # eager loop: a timed-out recurrence launch fails the epoch loudly
        if rank == 0:
            # the scalar part of the reference's tensorboard_log (model/helper_functions.py:120-141; the figures are out of scope):
            # every `logging_freq` epochs (and after the first) note / frame metrics on the validation segments and the eval-mode
            # loss terms over the labelled loader
            if ep % logging_freq == 0 or ep == 1:
                log_validation(model, val_set, l_loader, ep, writer, reconstruction, onset_script, VAT and ep >= VAT_start, VAT_start)
            for key, value in losses.items():
                writer.add_scalar(key, float(value), ep)
            if ep % saving_freq == 0:
                torch.save(model.state_dict(), os.path.join(logdir, f'model-{ep}.pt'))
                torch.save(optimizer.state_dict(), os.path.join(logdir, 'last-optimizer-state.pt'))
        if world > 1 and (ep % logging_freq == 0 or ep == 1):
            dp.wait_for_rank0()                            # the other ranks wait for rank 0's validation pass
    if rank == 0:
        torch.save(model.state_dict(), os.path.join(logdir, 'model-final.pt'))
        # final evaluation exactly as the reference scripts end (train_UNet_Onset_VAT.py:156-170): WHOLE songs of the test split
        # (`full_validation`, sequence_length=None), transcriptions written to <logdir>/MIDI_results, metrics pickled to
        # <logdir>/result_dict
        import pickle
        from .evaluate import evaluate_wo_velocity
        print('Training finished, now evaluating on the test split (full songs)')
        model.eval()
        with torch.no_grad():
            metrics = evaluate_wo_velocity(full_validation, model, reconstruction=False, save_path=os.path.join(logdir, 'MIDI_results'))
        for key, values in metrics.items():
            if key.startswith('metric/'):
                _, category, name = key.split('/')
                print(f'{category:>32} {name:25}: {np.mean(values):.3f} \u00b1 {np.std(values):.3f}')
                writer.add_scalar('validation/' + key, float(np.mean(values)), epoches)
        with open(os.path.join(logdir, 'result_dict'), 'wb') as fh:
            pickle.dump(dict(metrics), fh)
        f1 = float(np.mean(metrics['metric/frame/f1'])) if metrics['metric/frame/f1'] else float('nan')
        print(f'Training finished.  validation frame F1 {f1:.4f}, note F1 '
              f"{float(np.mean(metrics['metric/note/f1'])) if metrics['metric/note/f1'] else float('nan'):.4f}")
    if world > 1:
        dp.wait_for_rank0()                                # nobody tears the process group down while rank 0 still evaluates
        dp.shutdown()
    return model
