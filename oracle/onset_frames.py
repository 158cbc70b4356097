"""Oracle: the Onsets&Frames BiLSTM baseline with stepwise VAT.  TEST INFRASTRUCTURE ONLY.

Functional CPU / fp32 restatement over a flat ``{state_dict key: tensor}`` dictionary with the reference's own key
names.  The LSTM is written out step by step (no ``nn.LSTM``) so the recurrence the HIP kernels implement is visible.

Reference anchors (all under /root/reference):
  ConvStack ................................ model/onset_frame_VAT.py:321-355
  Onset_Stack.forward_LSTM ................. model/onset_frame_VAT.py:357-381
  Combine_Stack.forward_LSTM ............... model/onset_frame_VAT.py:390-410
  stepwise_VAT + _l2_normalize ............. model/onset_frame_VAT.py:158-207,313-319
  OnsetsAndFrames_VAT_full ................. model/onset_frame_VAT.py:603-704
  nn.LSTM gate order (i, f, g, o) .......... torch.nn.LSTM documentation

Dropout: the reference's ConvStack holds Dropout(0.25) x2 and Dropout(0.5); they draw fresh random masks in training
mode, so value parity is defined with the drop probabilities set to 0 (the golden vectors were produced by running the
reference with ``module.p = 0`` on its Dropout instances) -- the HIP dropout kernels are tested on their own.

Parity status: PINNED by ``tests/golden/onset_frames.npz`` (outputs of the reference itself, see
``tests/golden/make_golden.py:g_onset_frames``).
"""
import numpy as np
import torch
import torch.nn.functional as F

from . import fixture as fx
from . import frontend as fe
from .model import Net

N_BINS = 229
N_KEYS = 88


# ---------------------------------------------------------------------------------------------
# parameters
# ---------------------------------------------------------------------------------------------
def _convstack_shapes(s, pre, model_size):
    c1, c2 = model_size // 16, model_size // 8
    for idx, (ci, co) in ((0, (1, c1)), (3, (c1, c1)), (8, (c1, c2))):
        s[f'{pre}.cnn.{idx}.weight'] = (co, ci, 3, 3)
        s[f'{pre}.cnn.{idx}.bias'] = (co,)
        b = f'{pre}.cnn.{idx + 1}'
        s[b + '.weight'] = (co,); s[b + '.bias'] = (co,)
        s[b + '.running_mean'] = (co,); s[b + '.running_var'] = (co,); s[b + '.num_batches_tracked'] = ()
    s[f'{pre}.fc.0.weight'] = (model_size, c2 * (N_BINS // 4))
    s[f'{pre}.fc.0.bias'] = (model_size,)


def _lstm_shapes(s, pre, inp, hidden):
    for suffix in ('', '_reverse'):
        s[f'{pre}.weight_ih_l0{suffix}'] = (4 * hidden, inp)
        s[f'{pre}.weight_hh_l0{suffix}'] = (4 * hidden, hidden)
        s[f'{pre}.bias_ih_l0{suffix}'] = (4 * hidden,)
        s[f'{pre}.bias_hh_l0{suffix}'] = (4 * hidden,)


def param_shapes(model_complexity=48):
    """Ordered {state_dict key: shape} of OnsetsAndFrames_VAT_full without the ``spectrogram.*`` buffers
    (module order of model/onset_frame_VAT.py:603-625)."""
    ms = model_complexity * 16
    s = {}
    _convstack_shapes(s, 'onset_stack.convstack', ms)
    _lstm_shapes(s, 'onset_stack.sequence_model', ms, ms // 2)
    s['onset_stack.linear.weight'] = (N_KEYS, ms); s['onset_stack.linear.bias'] = (N_KEYS,)
    _lstm_shapes(s, 'combined_stack.sequence_model', 2 * N_KEYS, ms // 2)
    s['combined_stack.linear.weight'] = (N_KEYS, ms); s['combined_stack.linear.bias'] = (N_KEYS,)
    _convstack_shapes(s, 'frame_stack.0', ms)
    s['frame_stack.1.weight'] = (N_KEYS, ms); s['frame_stack.1.bias'] = (N_KEYS,)
    return s


def param_shapes_frame(model_complexity=48):
    """Frame_stack_VAT (model/onset_frame_VAT.py:417-443): module order combined_stack, frame_stack."""
    ms = model_complexity * 16
    s = {}
    _lstm_shapes(s, 'combined_stack.sequence_model', N_KEYS, ms // 2)
    s['combined_stack.linear.weight'] = (N_KEYS, ms); s['combined_stack.linear.bias'] = (N_KEYS,)
    _convstack_shapes(s, 'frame_stack.0', ms)
    s['frame_stack.1.weight'] = (N_KEYS, ms); s['frame_stack.1.bias'] = (N_KEYS,)
    return s


def param_shapes_onset(model_complexity=48):
    """Onset_stack_VAT (model/onset_frame_VAT.py:516-532)."""
    ms = model_complexity * 16
    s = {}
    _convstack_shapes(s, 'onset_stack.convstack', ms)
    _lstm_shapes(s, 'onset_stack.sequence_model', ms, ms // 2)
    s['onset_stack.linear.weight'] = (N_KEYS, ms); s['onset_stack.linear.bias'] = (N_KEYS,)
    return s


def fixture_params(model_complexity=48, with_frontend=True, tag='onf:', kind='onset_frame'):
    """Deterministic parameters: conv / linear ~ U(+-sqrt(3/fan_in)), LSTM ~ U(+-1/sqrt(H)) (PyTorch's own range),
    BatchNorm affine near (1, 0), running stats at their defaults."""
    out = {}
    shapes = {'onset_frame': param_shapes, 'frame': param_shapes_frame, 'onset': param_shapes_onset}[kind](model_complexity)
    for k, shp in shapes.items():
        name = tag + k
        if k.endswith('num_batches_tracked'):
            out[k] = torch.zeros((), dtype=torch.long)
        elif k.endswith('running_mean'):
            out[k] = torch.zeros(shp)
        elif k.endswith('running_var'):
            out[k] = torch.ones(shp)
        elif 'sequence_model' in k:
            hidden = shp[0] // 4
            out[k] = fx.hashed(name, shp, float(1.0 / np.sqrt(hidden)))
        elif len(shp) == 1 and '.cnn.' in k and int(k.split('.cnn.')[1].split('.')[0]) in (1, 4, 9):
            out[k] = fx.hashed(name, shp, 0.2, 1.0) if k.endswith('.weight') else fx.hashed(name, shp, 0.1)
        elif k.endswith('.bias'):
            out[k] = fx.hashed(name, shp, 0.1)
        else:
            fan = shp[1] * (9 if len(shp) == 4 else 1)
            out[k] = fx.hashed(name, shp, float(np.sqrt(3.0 / fan)))
    if with_frontend:
        out.update(fe.frontend_buffers())
    return out


# ---------------------------------------------------------------------------------------------
# network
# ---------------------------------------------------------------------------------------------
def conv_stack(net, spec, pre):
    """ConvStack.forward (model/onset_frame_VAT.py:350-355) with the Dropout probabilities at 0.  spec [B,T,229]."""
    x = spec.view(spec.size(0), 1, spec.size(1), spec.size(2))
    x = F.relu(net.bn(net.conv(x, f'{pre}.cnn.0', padding=1), f'{pre}.cnn.1'))
    x = F.relu(net.bn(net.conv(x, f'{pre}.cnn.3', padding=1), f'{pre}.cnn.4'))
    x = F.max_pool2d(x, (1, 2))
    x = F.relu(net.bn(net.conv(x, f'{pre}.cnn.8', padding=1), f'{pre}.cnn.9'))
    x = F.max_pool2d(x, (1, 2))
    x = x.transpose(1, 2).flatten(-2)
    return net.linear(x, f'{pre}.fc.0')


def lstm_direction(x, w_ih, w_hh, b_ih, b_hh, reverse):
    """One direction of nn.LSTM(batch_first=True), zero initial state: gates = W_ih x_t + b_ih + W_hh h_{t-1} + b_hh,
    split (i, f, g, o); c_t = sig(f) c_{t-1} + sig(i) tanh(g); h_t = sig(o) tanh(c_t)."""
    b, t, _ = x.shape
    hidden = w_hh.shape[1]
    xg = F.linear(x, w_ih, b_ih + b_hh)
    h = x.new_zeros(b, hidden)
    c = x.new_zeros(b, hidden)
    outs = [None] * t
    for step in (range(t - 1, -1, -1) if reverse else range(t)):
        g = xg[:, step] + F.linear(h, w_hh)
        gi, gf, gg, go = g.chunk(4, dim=-1)
        c = torch.sigmoid(gf) * c + torch.sigmoid(gi) * torch.tanh(gg)
        h = torch.sigmoid(go) * torch.tanh(c)
        outs[step] = h
    return torch.stack(outs, dim=1)


def bilstm(net, x, pre):
    """nn.LSTM(..., bidirectional=True)(x)[0]: [forward | reverse] halves (model/onset_frame_VAT.py:614)."""
    halves = []
    for suffix, rev in (('', False), ('_reverse', True)):
        halves.append(lstm_direction(x, net.p(f'{pre}.weight_ih_l0{suffix}'), net.p(f'{pre}.weight_hh_l0{suffix}'),
                                     net.p(f'{pre}.bias_ih_l0{suffix}'), net.p(f'{pre}.bias_hh_l0{suffix}'), rev))
    return torch.cat(halves, dim=-1)


def forward(params, training, spec, detach=False):
    """OnsetsAndFrames_VAT_full.forward (model/onset_frame_VAT.py:627-635) -> (onset, activation, frame)."""
    net = Net(params, training, detach)
    x = conv_stack(net, spec, 'onset_stack.convstack')
    x = bilstm(net, x, 'onset_stack.sequence_model')
    onset = torch.sigmoid(net.linear(x, 'onset_stack.linear'))
    act = torch.sigmoid(net.linear(conv_stack(net, spec, 'frame_stack.0'), 'frame_stack.1'))
    comb = torch.cat([onset.detach(), act], dim=-1)
    frame = torch.sigmoid(net.linear(bilstm(net, comb, 'combined_stack.sequence_model'), 'combined_stack.linear'))
    return onset, act, frame


def l2_normalise(d):
    return d / torch.norm(d, dim=-1, keepdim=True)


def vat(params, training, x, xi, eps, d0):
    """stepwise_VAT.forward, n_power = 1, BCE distance on the frame output (model/onset_frame_VAT.py:175-207).
    Returns (vat_loss, r_adv, d_normalised, d.grad).  The reference's model.zero_grad() after the power iteration is
    expressed by detaching the weights for that pass."""
    with torch.no_grad():
        _, _, frame_ref = forward(params, training, x)
    d = d0.clone().requires_grad_(True)
    r = xi * l2_normalise(d)
    _, _, frame_pred = forward(params, training, (x + r).clamp(0, 1), detach=True)
    loss = F.binary_cross_entropy(frame_pred, frame_ref)
    g, = torch.autograd.grad(loss, d)
    d = g.detach() * 1e10
    r_adv = eps * l2_normalise(d)
    assert not torch.isnan(r_adv).any() and not torch.isinf(r_adv).any(), 'r_adv contains nan'
    _, _, frame_pred = forward(params, training, (x + r_adv).clamp(0, 1))
    return F.binary_cross_entropy(frame_pred, frame_ref), r_adv, l2_normalise(d * 1e8), g


def _spec(params, audio, log=True):
    """model/onset_frame_VAT.py:662-668: same front-end as the U-Net models, without the channel dimension."""
    return fe.frontend(audio.reshape(-1, audio.shape[-1])[:, :-1], params, log).squeeze(1)


def run_on_batch(params, training, batch, batch_ul=None, VAT=False, xi=1e-5, eps=10.0, d0_l=None, d0_ul=None):
    """OnsetsAndFrames_VAT_full.run_on_batch (model/onset_frame_VAT.py:637-704): same key names and order."""
    frame_label, onset_label = batch['frame'], batch['onset']
    if batch_ul:
        spec = _spec(params, batch_ul['audio'].reshape(-1, batch['audio'].shape[-1]))
        lds_ul, _, r_norm_ul = vat(params, training, spec, xi, eps, d0_ul)[:3]
    else:
        lds_ul, r_norm_ul = torch.tensor(0.), torch.tensor(0.)
    spec = _spec(params, batch['audio'])
    if VAT:
        lds_l, r_adv, r_norm_l = vat(params, training, spec, xi, eps, d0_l)[:3]
    else:
        r_adv, lds_l, r_norm_l = None, torch.tensor(0.), torch.tensor(0.)
    onset, _, frame = forward(params, training, spec)
    predictions = {'onset': onset.reshape(*frame_label.shape), 'frame': frame.reshape(*frame_label.shape), 'r_adv': r_adv}
    tag = 'train' if training else 'test'
    losses = {f'loss/{tag}_frame': F.binary_cross_entropy(predictions['frame'], frame_label),
              f'loss/{tag}_onset': F.binary_cross_entropy(predictions['onset'], onset_label),
              f'loss/{tag}_LDS_l': lds_l}
    if training:
        losses['loss/train_LDS_ul'] = lds_ul
    losses[f'loss/{tag}_r_norm_l'] = r_norm_l.abs().mean()
    if training:
        losses['loss/train_r_norm_ul'] = r_norm_ul.abs().mean()
    return predictions, losses, spec


# ---------------------------------------------------------------------------------------------
# the two single-stack variants of the baseline script (model_name = 'frame' / 'onset')
# ---------------------------------------------------------------------------------------------
def forward_frame_stack(params, training, spec, detach=False):
    """Frame_stack_VAT.forward (model/onset_frame_VAT.py:445-450) -> (activation, frame)."""
    net = Net(params, training, detach)
    act = torch.sigmoid(net.linear(conv_stack(net, spec, 'frame_stack.0'), 'frame_stack.1'))
    frame = torch.sigmoid(net.linear(bilstm(net, act, 'combined_stack.sequence_model'), 'combined_stack.linear'))
    return act, frame


def vat_frame_stack(params, training, x, xi, eps, d0, mode='all'):
    """stepwise_VAT_frame_stack.forward (model/onset_frame_VAT.py:221-269): distance = BCE(frame) [+ MSE(activation)],
    d = d.grad * 1e20.  Returns (vat_loss, r_adv, d.grad)."""
    def dist(act, frame, act_ref, frame_ref):
        terms = {'activation': F.mse_loss(act, act_ref), 'frame': F.binary_cross_entropy(frame, frame_ref)}
        return terms[mode] if mode != 'all' else terms['frame'] + terms['activation']
    with torch.no_grad():
        act_ref, frame_ref = forward_frame_stack(params, training, x)
    d = d0.clone().requires_grad_(True)
    act, frame = forward_frame_stack(params, training, (x + xi * l2_normalise(d)).clamp(0, 1), detach=True)
    g, = torch.autograd.grad(dist(act, frame, act_ref, frame_ref), d)
    r_adv = eps * l2_normalise(g.detach() * 1e20)
    act, frame = forward_frame_stack(params, training, (x + r_adv).clamp(0, 1))
    return dist(act, frame, act_ref, frame_ref), r_adv, g


def run_on_batch_frame_stack(params, training, batch, VAT=False, xi=1e-5, eps=10.0, d0_l=None, mode='all'):
    """Frame_stack_VAT.run_on_batch (model/onset_frame_VAT.py:453-503) without an unlabelled batch (with one AND VAT=True
    the reference feeds a [B, 229, 640] tensor to the ConvStack and fails in its Linear, :466)."""
    spec = _spec(params, batch['audio'])
    if VAT:
        lds_l, r_adv = vat_frame_stack(params, training, spec, xi, eps, d0_l, mode)[:2]
    else:
        r_adv, lds_l = None, torch.tensor(0.)
    _, frame = forward_frame_stack(params, training, spec)
    predictions = {'onset': frame, 'frame': frame.reshape(*batch['frame'].shape), 'r_adv': r_adv}
    if training:
        losses = {'loss/train_frame': F.binary_cross_entropy(predictions['frame'], batch['frame']),
                  'loss/train_LDS': (torch.tensor(0.) + lds_l) / 2}
    else:
        losses = {'loss/test_frame': F.binary_cross_entropy(predictions['frame'], batch['frame']), 'loss/test_LDS': lds_l}
    return predictions, losses, spec


def run_on_batch_onset_stack(params, training, batch):
    """Onset_stack_VAT.run_on_batch with VAT=False (model/onset_frame_VAT.py:539-589; VAT=True hits undefined names at
    :305-306)."""
    spec = _spec(params, batch['audio'])
    net = Net(params, training)
    x = bilstm(net, conv_stack(net, spec, 'onset_stack.convstack'), 'onset_stack.sequence_model')
    onset = torch.sigmoid(net.linear(x, 'onset_stack.linear'))
    label = batch['onset']
    accuracy = (label == (onset > 0.5)).float().sum() / label.flatten(0).shape[0]
    tag = 'train' if training else 'test'
    lds = torch.tensor(0.)
    losses = {f'loss/{tag}_onset': F.binary_cross_entropy(onset, label), f'metric/{tag}_accuracy': accuracy,
              f'loss/{tag}_LDS': torch.mean(torch.stack((lds, lds)), dim=0) if training else lds}
    return {'onset': onset, 'r_adv': None}, losses, spec
