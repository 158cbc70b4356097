#!/usr/bin/env python
"""Drop-in entry point: `python train_baseline_onset_frame_VAT.py with key=value ...` (keys/defaults of the reference
script of this name; model_name in onset_frame | frame | onset: the Onsets&Frames BiLSTM baseline with stepwise VAT and its
single-stack variants).  One process per
GPU under torch.distributed.run trains data-parallel, as for the U-Net scripts."""
from reconvat_amd.cli import baseline_config, run_training
from reconvat_amd.sacred_lite import Experiment

ex = Experiment('train_original')


@ex.config
def config(overrides):
    return baseline_config(overrides)


@ex.automain
def train(spec, resume_iteration, train_on, batch_size, sequence_length, small, supersmall, train_batch_size, learning_rate,
          learning_rate_decay_steps, learning_rate_decay_rate, alpha, clip_gradient_norm, validation_length, refresh, device,
          epoches, logdir, log, iteration, VAT_start, VAT, XI, eps, reconstruction, graph, fused_optimizer, saving_freq,
          device_feed, model_complexity, model_name, VAT_mode, logging_freq):
    return run_training('baseline', **locals())
