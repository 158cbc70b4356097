#!/usr/bin/env python
"""Drop-in entry point: `python train_UNet_Onset_VAT.py with key=value ...` (same keys/defaults as the
reference script of this name).  One process per GPU: `python -m torch.distributed.run --nproc-per-node 8
--master-addr 127.0.0.1 train_UNet_Onset_VAT.py with reconstruction=True` trains data-parallel."""
from reconvat_amd.cli import base_config, run_training
from reconvat_amd.sacred_lite import Experiment

ex = Experiment('train_original')


@ex.config
def config(overrides):
    return base_config(overrides, onset_script=True)


@ex.automain
def train(spec, resume_iteration, train_on, batch_size, sequence_length, small, supersmall, train_batch_size, learning_rate,
          learning_rate_decay_steps, learning_rate_decay_rate, alpha, clip_gradient_norm, validation_length, refresh, device,
          epoches, logdir, log, iteration, VAT_start, VAT, XI, eps, reconstruction, graph, fused_optimizer, saving_freq,
          device_feed, logging_freq, dtype):
    return run_training(True, **locals())
