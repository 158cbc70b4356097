"""Drop-in surface of the reference's Onsets&Frames BiLSTM baseline with stepwise VAT on MI355X.

Same class names, constructor arguments, ``run_on_batch`` contract, loss / prediction keys and ``state_dict`` keys as

    ConvStack / Onset_Stack / Combine_Stack ......... model/onset_frame_VAT.py:321-415
    stepwise_VAT .................................... model/onset_frame_VAT.py:158-207
    OnsetsAndFrames_VAT_full ........................ model/onset_frame_VAT.py:603-721

As in ``reconvat_amd.model`` the ``nn`` modules are parameter containers only; every numeric step goes through
``reconvat_amd.ops``: the three ConvStack convolutions on the MFMA / small-channel conv kernels with BatchNorm statistics
fused into their epilogues, ReLU as the slope-0 case of the BatchNorm+activation kernel, MaxPool(1,2)+Dropout as one kernel,
the LSTM input projections as one MFMA GEMM per direction and the recurrence as one persistent launch per BiLSTM pass.
Activations are NHWC ([B, T, bins, C]); the ConvStack's Linear sees them flattened bin-major, so its weight is read
through a (bin, channel)-permuted copy.
"""
import torch
import torch.nn as nn

from . import ops
from .constants import N_BINS
from .model import _Base, _p
from .ops import ARENA, BiLstmFn, BnActFn, ConvFn, LinearFn, PoolDropFn, VatPerturbFn, abs_mean, bce_mean


def _conv_bn_relu(conv, bn, x, detach):
    stats = ARENA.take(ops.bn_ws_doubles(bn.num_features), x.device) if bn.training else None
    z = ConvFn.apply(x, _p(conv.weight, detach), _p(conv.bias, detach), 'c3', None, stats, None)
    return BnActFn.apply(z, _p(bn.weight, detach), _p(bn.bias, detach), bn.running_mean, bn.running_var,
                         bn.num_batches_tracked, None, bn.training, 0.0, stats, None)


def _linear(m, x2, act, detach):
    return LinearFn.apply(x2, _p(m.weight, detach), _p(m.bias, detach), act)


def _bilstm(m, x, detach):
    ps = [getattr(m, n + s) for s in ('', '_reverse') for n in ('weight_ih_l0', 'weight_hh_l0', 'bias_ih_l0', 'bias_hh_l0')]
    return BiLstmFn.apply(x, *[_p(p, detach) for p in ps])


class ConvStack(nn.Module):
    """model/onset_frame_VAT.py:321-355 (same Sequential indices, hence the same state_dict keys)."""

    def __init__(self, input_features, output_features):
        super().__init__()
        self.cnn = nn.Sequential(
            nn.Conv2d(1, output_features // 16, (3, 3), padding=1), nn.BatchNorm2d(output_features // 16), nn.ReLU(),
            nn.Conv2d(output_features // 16, output_features // 16, (3, 3), padding=1), nn.BatchNorm2d(output_features // 16),
            nn.ReLU(),
            nn.MaxPool2d((1, 2)), nn.Dropout(0.25),
            nn.Conv2d(output_features // 16, output_features // 8, (3, 3), padding=1), nn.BatchNorm2d(output_features // 8),
            nn.ReLU(),
            nn.MaxPool2d((1, 2)), nn.Dropout(0.25),
        )
        self.fc = nn.Sequential(nn.Linear((output_features // 8) * (input_features // 4), output_features), nn.Dropout(0.5))

    def forward(self, spec, detach=False):
        """spec [B, T, bins] -> [B*T, output_features]."""
        c = self.cnn
        b, t, nb = spec.shape
        x = spec.reshape(b, t, nb, 1)
        x = _conv_bn_relu(c[0], c[1], x, detach)
        x = _conv_bn_relu(c[3], c[4], x, detach)
        x = PoolDropFn.apply(x, c[7].p, self.training)
        x = _conv_bn_relu(c[8], c[9], x, detach)
        x = PoolDropFn.apply(x, c[12].p, self.training)
        _, _, wq, ch = x.shape
        fc = self.fc[0]
        # reference flattens NCHW as (channel, bin); NHWC activations are (bin, channel): permute the weight's columns
        w = _p(fc.weight, detach).view(-1, ch, wq).permute(0, 2, 1).reshape(-1, wq * ch)
        y = LinearFn.apply(x.view(b * t, wq * ch), w, _p(fc.bias, detach), 0)
        return ops.dropout(y, self.fc[1].p, self.training)


class Onset_Stack(nn.Module):
    """model/onset_frame_VAT.py:357-388."""

    def __init__(self, input_features, model_size, output_features, sequence_model):
        super().__init__()
        self.convstack = ConvStack(input_features, model_size)
        self.sequence_model = sequence_model
        self.linear = nn.Linear(model_size, output_features)

    def forward(self, x, detach=False):
        b, t, _ = x.shape
        y = self.convstack(x, detach)
        if self.sequence_model:
            y = _bilstm(self.sequence_model, y.view(b, t, -1), detach).view(b * t, -1)
        return _linear(self.linear, y, 1, detach)


class Combine_Stack(nn.Module):
    """model/onset_frame_VAT.py:390-415."""

    def __init__(self, model_size, output_features, sequence_model):
        super().__init__()
        self.sequence_model = sequence_model
        self.linear = nn.Linear(model_size if sequence_model else output_features, output_features)

    def forward(self, x, detach=False):
        b, t, _ = x.shape
        y = x.reshape(b * t, -1)
        if self.sequence_model:
            y = _bilstm(self.sequence_model, x, detach).view(b * t, -1)
        return _linear(self.linear, y, 1, detach)


class stepwise_VAT(nn.Module):
    """model/onset_frame_VAT.py:158-207: one power iteration, BCE distance on the frame posteriorgram.  As in
    reconvat_amd.model.UNet_VAT the weights are detached for the power-iteration pass (the reference discards those
    gradients with model.zero_grad()), so only the input-gradient chain runs."""

    def __init__(self, XI, epsilon, n_power, KL_Div):
        super().__init__()
        if KL_Div:
            raise NotImplementedError('only the BCE distance of stepwise_VAT is on the MI355X path (the scripts pass KL_Div=False)')
        self.n_power, self.XI, self.epsilon, self.KL_Div, self.binwise = n_power, XI, epsilon, KL_Div, False
        self.nan_flag = None
        self.noise = None          # optional callable(x) -> d0 (tests inject deterministic noise)

    def forward(self, model, x, check=True):
        if self.nan_flag is None or self.nan_flag.device != x.device:
            self.nan_flag = torch.zeros(1, dtype=torch.int32, device=x.device)
        with torch.no_grad():
            frame_ref = model(x)[2]
        d = (self.noise(x) if self.noise is not None else torch.randn_like(x)).requires_grad_(True)
        g = None
        for it in range(self.n_power):
            if it > 0:
                d = (g * 1e10).requires_grad_(True)
            x_adv = VatPerturbFn.apply(x, d, float(self.XI))
            loss = bce_mean(model(x_adv, detach=True)[2], frame_ref)
            g, = torch.autograd.grad(loss, d)
            g = g.detach()
        x_adv, r_adv, d_norm = ops.vat_adversarial(x, g, 1e10, float(self.epsilon), self.nan_flag)
        if check:
            self.check_nan()
        return bce_mean(model(x_adv)[2], frame_ref), r_adv, d_norm

    def check_nan(self):
        if not torch.cuda.is_current_stream_capturing():
            assert int(self.nan_flag.item()) == 0, 'r_adv contains nan'


class OnsetsAndFrames_VAT_full(_Base):
    """model/onset_frame_VAT.py:603-721."""

    def __init__(self, input_features, output_features, model_complexity=48, log=True, mode='imagewise', spec='Mel', XI=1e-5,
                 eps=10, VAT_mode='all'):
        super().__init__(log, False, mode, spec, XI, eps)
        model_size = model_complexity * 16

        def sequence_model(input_size, output_size):
            return nn.LSTM(input_size, output_size // 2, batch_first=True, bidirectional=True)
        self.vat_loss = stepwise_VAT(XI, eps, 1, False)
        self.onset_stack = Onset_Stack(input_features, model_size, output_features, sequence_model(model_size, model_size))
        self.combined_stack = Combine_Stack(model_size, output_features, sequence_model(output_features * 2, model_size))
        self.frame_stack = nn.Sequential(ConvStack(input_features, model_size), nn.Linear(model_size, output_features), nn.Sigmoid())

    def forward(self, spec, detach=False):
        """spec [B, T, 229] -> (onset, activation, frame), each [B, T, 88]."""
        b, t, _ = spec.shape
        onset = self.onset_stack(spec, detach)
        act = _linear(self.frame_stack[1], self.frame_stack[0](spec, detach), 1, detach)
        combined = torch.cat([onset.detach(), act], dim=-1).view(b, t, -1)
        frame = self.combined_stack(combined, detach)
        return onset.view(b, t, -1), act.view(b, t, -1), frame.view(b, t, -1)

    def _spec(self, audio, ref_len):
        return self._front(audio, ref_len).squeeze(1)

    side_streams = 2           # TrainStep: twin gradient buckets to provide
    has_recurrence = True      # BiLSTM time-out flag: data-parallel ranks agree on skipped steps (train.allreduce_gradients)
    defer_wgrad_reductions = False   # few conv layers, three chains: one reduction launch per chain end measured slower (91.5 vs 87.5 ms)

    def _two_streams(self, audio_ul, audio_l, VAT):
        """The step as THREE concurrent kernel chains: the unlabelled VAT on side stream 0, the main forward on side stream 1,
        the labelled VAT on this stream; autograd then runs the three backward chains on the same streams.  The BiLSTM
        recurrences are latency-bound launches on 48 of the 256 CUs, so the chains overlap almost completely and the step
        takes about as long as its longest chain (3 forward + 2 backward passes instead of 7 + 5).  Shared state as in
        model._Base._vat_two_streams: parameter gradients of each side chain go to that stream's twin of the flat gradient
        bucket (ops.SIDE_GRADS), BatchNorm running-statistic updates are deferred and replayed in the reference's order
        (UL target, UL xi*d, UL final, L target, L xi*d, L final, main)."""
        cur = torch.cuda.current_stream()
        dev = audio_l.device
        side_ul, side_main = ops.side_stream(dev, 0), ops.side_stream(dev, 1)
        ref_len = audio_l.shape[-1]
        side_ul.wait_stream(cur)
        with torch.cuda.stream(side_ul):
            spec_ul = self._spec(audio_ul, ref_len)
            with ops.deferred_bn_updates() as pend_ul:
                lds_ul, _, dn_ul = self.vat_loss(self, spec_ul, check=False)
                r_norm_ul = abs_mean(dn_ul)
        spec = self._spec(audio_l, ref_len)
        if VAT:
            spec_ready = torch.cuda.Event()
            spec_ready.record(cur)
            with torch.cuda.stream(side_main):
                side_main.wait_event(spec_ready)
                spec.record_stream(side_main)
                with ops.deferred_bn_updates() as pend_main:
                    onset_pred, _, frame_pred = self(spec)
            with ops.deferred_bn_updates() as pend_l:
                lds_l, r_adv, dn_l = self.vat_loss(self, spec, check=False)
                r_norm_l = abs_mean(dn_l)
            cur.wait_stream(side_main)
            for t in (onset_pred, frame_pred):
                t.record_stream(cur)
            seq = [pend_ul, pend_l, pend_main]
        else:
            r_adv, lds_l, r_norm_l = None, torch.tensor(0.), torch.tensor(0.)
            with ops.deferred_bn_updates() as pend_main:
                onset_pred, _, frame_pred = self(spec)
            seq = [pend_ul, pend_main]
        cur.wait_stream(side_ul)
        for t in (lds_ul, r_norm_ul):
            t.record_stream(cur)
        ops.replay_bn_updates(seq, dev)
        self.vat_loss.check_nan()
        return spec, onset_pred, frame_pred, lds_ul, r_norm_ul, lds_l, r_adv, r_norm_l

    def run_on_batch(self, batch, batch_ul=None, VAT=False):
        audio_label = batch['audio']
        onset_label = batch['onset']
        frame_label = batch['frame']
        if batch_ul and self.training and ops.DUAL_STREAM[0] and audio_label.is_cuda:
            spec, onset_pred, frame_pred, lds_ul, r_norm_ul, lds_l, r_adv, r_norm_l = self._two_streams(
                batch_ul['audio'], audio_label, VAT)
            return self._pack(spec, onset_pred, frame_pred, onset_label, frame_label, lds_ul, r_norm_ul, lds_l, r_adv, r_norm_l)
        if batch_ul:
            spec = self._spec(batch_ul['audio'], audio_label.shape[-1])
            lds_ul, _, r_norm_ul = self.vat_loss(self, spec)
            r_norm_ul = abs_mean(r_norm_ul)
        else:
            lds_ul, r_norm_ul = torch.tensor(0.), torch.tensor(0.)
        spec = self._spec(audio_label, audio_label.shape[-1])
        if VAT:
            lds_l, r_adv, r_norm_l = self.vat_loss(self, spec)
            r_norm_l = abs_mean(r_norm_l)
        else:
            r_adv, lds_l, r_norm_l = None, torch.tensor(0.), torch.tensor(0.)
        onset_pred, _, frame_pred = self(spec)
        return self._pack(spec, onset_pred, frame_pred, onset_label, frame_label, lds_ul, r_norm_ul, lds_l, r_adv, r_norm_l)

    def _pack(self, spec, onset_pred, frame_pred, onset_label, frame_label, lds_ul, r_norm_ul, lds_l, r_adv, r_norm_l):
        predictions = {'onset': onset_pred.reshape(*frame_label.shape), 'frame': frame_pred.reshape(*frame_label.shape),
                       'r_adv': r_adv}
        tag = 'train' if self.training else 'test'
        losses = {f'loss/{tag}_frame': bce_mean(predictions['frame'], frame_label),
                  f'loss/{tag}_onset': bce_mean(predictions['onset'], onset_label),
                  f'loss/{tag}_LDS_l': lds_l}
        if self.training:
            losses['loss/train_LDS_ul'] = lds_ul
        losses[f'loss/{tag}_r_norm_l'] = r_norm_l
        if self.training:
            losses['loss/train_r_norm_ul'] = r_norm_ul
        return predictions, losses, spec


class stepwise_VAT_frame_stack(nn.Module):
    """model/onset_frame_VAT.py:209-269: distance = BCE(frame) and / or MSE(activation) by `VAT_mode`, d = d.grad * 1e20;
    returns (vat_loss, r_adv).  Weights detached for the power-iteration pass (see stepwise_VAT)."""

    def __init__(self, XI, epsilon, n_power, VAT_mode):
        super().__init__()
        if VAT_mode not in ('activation', 'frame', 'all'):
            raise ValueError(f'VAT_mode {VAT_mode!r}')
        self.n_power, self.XI, self.epsilon, self.VAT_mode = n_power, XI, epsilon, VAT_mode
        self.nan_flag = None
        self.noise = None

    def _dist(self, act, frame, act_ref, frame_ref):
        if self.VAT_mode == 'activation':
            return ops.mse_mean(act, act_ref)
        if self.VAT_mode == 'frame':
            return bce_mean(frame, frame_ref)
        return bce_mean(frame, frame_ref) + ops.mse_mean(act, act_ref)

    def forward(self, model, x):
        if self.nan_flag is None or self.nan_flag.device != x.device:
            self.nan_flag = torch.zeros(1, dtype=torch.int32, device=x.device)
        with torch.no_grad():
            act_ref, frame_ref = model(x)
        d = (self.noise(x) if self.noise is not None else torch.randn_like(x)).requires_grad_(True)
        g = None
        for it in range(self.n_power):
            if it > 0:
                d = (g * 1e20).requires_grad_(True)
            act, frame = model(VatPerturbFn.apply(x, d, float(self.XI)), detach=True)
            g, = torch.autograd.grad(self._dist(act, frame, act_ref, frame_ref), d)
            g = g.detach()
        x_adv, r_adv, _ = ops.vat_adversarial(x, g, 1e20, float(self.epsilon), self.nan_flag)
        if not torch.cuda.is_current_stream_capturing():
            assert int(self.nan_flag.item()) == 0, 'r_adv exploded, please debug tune down the XI for VAT'
        act, frame = model(x_adv)
        return self._dist(act, frame, act_ref, frame_ref), r_adv


class Frame_stack_VAT(_Base):
    """model/onset_frame_VAT.py:417-514 (`model_name='frame'` of the baseline script): ConvStack -> Linear -> sigmoid ->
    BiLSTM(88 -> 768) -> Linear -> sigmoid."""
    has_recurrence = True

    def __init__(self, input_features, output_features, model_complexity=48, log=True, mode='imagewise', spec='Mel', XI=1e-5,
                 eps=10, VAT_mode='all'):
        super().__init__(log, False, mode, spec, XI, eps)
        model_size = model_complexity * 16
        self.vat_loss = stepwise_VAT_frame_stack(XI, eps, 1, VAT_mode)
        self.combined_stack = Combine_Stack(model_size, output_features,
                                            nn.LSTM(output_features, model_size // 2, batch_first=True, bidirectional=True))
        self.frame_stack = nn.Sequential(ConvStack(input_features, model_size), nn.Linear(model_size, output_features), nn.Sigmoid())

    def forward(self, spec, detach=False):
        b, t, _ = spec.shape
        act = _linear(self.frame_stack[1], self.frame_stack[0](spec, detach), 1, detach).view(b, t, -1)
        return act, self.combined_stack(act, detach).view(b, t, -1)

    def run_on_batch(self, batch, batch_ul=None, VAT=False):
        audio_label, frame_label = batch['audio'], batch['frame']
        spec = self._front(audio_label, audio_label.shape[-1]).squeeze(1)
        if batch_ul and VAT:
            # the reference transposes the LABELLED spectrogram a second time here (model/onset_frame_VAT.py:466) and fails
            # in the ConvStack's Linear (640 "bins"); there is no behaviour to reproduce
            raise RuntimeError('Frame_stack_VAT.run_on_batch: the unlabelled VAT branch of the reference is not executable '
                               '(shape error at model/onset_frame_VAT.py:466-467); pass batch_ul=None')
        lds_ul = torch.tensor(0.)
        if VAT:
            lds_l, r_adv = self.vat_loss(self, spec)
        else:
            r_adv, lds_l = None, torch.tensor(0.)
        _, frame_pred = self(spec)
        predictions = {'onset': frame_pred, 'frame': frame_pred.reshape(*frame_label.shape), 'r_adv': r_adv}
        if self.training:
            losses = {'loss/train_frame': bce_mean(predictions['frame'], frame_label),
                      'loss/train_LDS': (lds_ul.to(lds_l.device) + lds_l) / 2}
        else:
            losses = {'loss/test_frame': bce_mean(predictions['frame'], frame_label), 'loss/test_LDS': lds_l}
        return predictions, losses, spec


class Onset_stack_VAT(_Base):
    """model/onset_frame_VAT.py:516-601 (`model_name='onset'`): the onset stack alone.  Its VAT branch references undefined
    names in the reference (stepwise_VAT_onset_stack, :305-306), so only VAT=False exists."""
    has_recurrence = True

    def __init__(self, input_features, output_features, model_complexity=48, log=True, mode='imagewise', spec='Mel', XI=1e-5,
                 eps=10, VAT_mode='all'):
        super().__init__(log, False, mode, spec, XI, eps)
        model_size = model_complexity * 16
        self.vat_loss = None
        self.onset_stack = Onset_Stack(input_features, model_size, output_features,
                                       nn.LSTM(model_size, model_size // 2, batch_first=True, bidirectional=True))

    def forward(self, spec, detach=False):
        b, t, _ = spec.shape
        return self.onset_stack(spec, detach).view(b, t, -1)

    def run_on_batch(self, batch, batch_ul=None, VAT=False):
        if VAT:
            raise NotImplementedError("Onset_stack_VAT: the reference's VAT branch raises NameError "
                                      '(model/onset_frame_VAT.py:305-306); only VAT=False is defined')
        audio_label, onset_label = batch['audio'], batch['onset']
        spec = self._front(audio_label, audio_label.shape[-1]).squeeze(1)
        onset_pred = self(spec)
        accuracy = (onset_label == (onset_pred.detach() > 0.5)).float().sum() / onset_label.flatten(0).shape[0]
        tag = 'train' if self.training else 'test'
        lds = torch.tensor(0.)
        losses = {f'loss/{tag}_onset': bce_mean(onset_pred.reshape(*onset_label.shape), onset_label),
                  f'metric/{tag}_accuracy': accuracy,
                  f'loss/{tag}_LDS': torch.mean(torch.stack((lds, lds)), dim=0) if self.training else lds}
        return {'onset': onset_pred, 'r_adv': None}, losses, spec
