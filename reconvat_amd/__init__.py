"""reconvat_amd -- MI355X-native (gfx950) implementation of the ReconVAT per-segment training hot path.

Public surface mirrors the reference's ``model`` package for this path:
``UNet_Onset``, ``UNet``, ``OnsetsAndFrames_VAT_full`` (the BiLSTM baseline), ``train_VAT_model``, the constants, plus the MI355X-side additions
(``FlatAdam``, ``TrainStep`` with whole-step hipGraph capture, data-parallel helpers).
"""
from .constants import *  # noqa: F401,F403
from .model import UNet_Onset, UNet, UNet_VAT, MutliHeadAttention1D, Spec2Roll, Roll2Spec, Encoder, Decoder  # noqa: F401
from .onset_frames import (OnsetsAndFrames_VAT_full, Frame_stack_VAT, Onset_stack_VAT, stepwise_VAT,  # noqa: F401
                           stepwise_VAT_frame_stack, ConvStack, Onset_Stack, Combine_Stack)
from .frontend import MelSpectrogram, Normalization  # noqa: F401
from .train import train_VAT_model, eval_model, FlatAdam, TrainStep, weighted_loss, cycle  # noqa: F401
