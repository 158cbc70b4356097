"""Oracle: U-Net transcriber / reconstructor, local attention, VAT, losses and
one optimiser step.  TEST INFRASTRUCTURE ONLY.

Functional restatement over a flat ``{state_dict key: tensor}`` parameter
dictionary (the reference's own key names, so a reference checkpoint is a valid
parameter set).  CPU, fp32, plain ``torch.nn.functional`` calls.

Reference anchors (all under /root/reference):
  block / d_block / Encoder / Decoder ....... model/UNet_onset.py:186-268
  MutliHeadAttention1D ....................... model/UNet_onset.py:22-98
  Stack / Spec2Roll / Roll2Spec .............. model/UNet_onset.py:270-339
  no-onset Spec2Roll ......................... model/self_attention_VAT.py:929-945
  UNet_VAT + _l2_normalize ................... model/UNet_onset.py:101-171
  UNet_Onset.forward / run_on_batch .......... model/UNet_onset.py:380-542
  UNet.forward / run_on_batch ................ model/self_attention_VAT.py:1064-1203
  train_VAT_model ............................ model/helper_functions.py:570-615
"""
import torch
import torch.nn.functional as F

from . import frontend as fe

BN_MOMENTUM = 0.1      # model/UNet_onset.py:183
BN_EPS = 1e-5          # nn.BatchNorm2d default
WINDOW = 31            # model/UNet_onset.py:289,301,323


class Net:
    """Parameter view: ``Net(params, training)`` then ``net.p('a.b.weight')``."""

    def __init__(self, params, training=True, detach=False):
        self.params = params
        self.training = training
        self.detach = detach

    def p(self, name):
        t = self.params[name]
        return t.detach() if self.detach else t

    def conv(self, x, name, **kw):
        return F.conv2d(x, self.p(name + '.weight'), self.p(name + '.bias'), **kw)

    def convT(self, x, name, **kw):
        return F.conv_transpose2d(x, self.p(name + '.weight'), self.p(name + '.bias'), **kw)

    def bn(self, x, name):
        # nn.BatchNorm2d(momentum=0.1) train/eval semantics incl. running-stat update
        rm = self.params[name + '.running_mean']
        rv = self.params[name + '.running_var']
        if self.training:
            self.params[name + '.num_batches_tracked'] += 1
        return F.batch_norm(x, rm, rv, self.p(name + '.weight'), self.p(name + '.bias'),
                            self.training, BN_MOMENTUM, BN_EPS)

    def linear(self, x, name, bias=True):
        return F.linear(x, self.p(name + '.weight'), self.p(name + '.bias') if bias else None)


def enc_block(net, x, name):
    """block.forward, model/UNet_onset.py:196-201 -> (downsampled, size before ds)."""
    a = F.leaky_relu(net.bn(net.conv(x, name + '.conv1', padding=1), name + '.bn1'))
    b = F.leaky_relu(net.bn(net.conv(a, name + '.conv2', padding=1), name + '.bn2'))
    b = b + net.conv(x, name + '.skip')
    return net.conv(b, name + '.ds', stride=2), b.shape


def dec_block(net, x, name, size, last, skip):
    """d_block.forward, model/UNet_onset.py:216-224.  ``us`` is a 2x2/s2
    ConvTranspose2d called with output_size -> output_padding = size - 2*in."""
    op = (size[2] - 2 * x.shape[2], size[3] - 2 * x.shape[3])
    x = net.convT(x, name + '.us', stride=2, output_padding=op)
    if not last:
        x = torch.cat((x, skip), 1)
    x = F.leaky_relu(net.bn(net.convT(x, name + '.conv2d', padding=1), name + '.bn2d'))
    x = net.convT(x, name + '.conv1d', padding=1)
    if not last:
        x = F.leaky_relu(net.bn(x, name + '.bn1d'))
    return x


def unet(net, x, enc, dec):
    """Encoder.forward + Decoder.forward, model/UNet_onset.py:239-268."""
    x1, s1 = enc_block(net, x, enc + '.block1')
    x2, s2 = enc_block(net, x1, enc + '.block2')
    x3, s3 = enc_block(net, x2, enc + '.block3')
    x4, s4 = enc_block(net, x3, enc + '.block4')
    c1 = net.conv(x3, enc + '.conv1', padding=1)
    c2 = net.conv(x2, enc + '.conv2', padding=1)
    c3 = net.conv(x1, enc + '.conv3', padding=1)
    y = dec_block(net, x4, dec + '.d_block1', s4, False, c1)
    y = dec_block(net, y, dec + '.d_block2', s3, False, c2)
    y = dec_block(net, y, dec + '.d_block3', s2, False, c3)
    y = dec_block(net, y, dec + '.d_block4', s1, True, None)
    return y


def local_attention(net, x, name, groups):
    """MutliHeadAttention1D.forward, model/UNet_onset.py:56-91.
    x [B, L, Fin] -> (out [B, L, F], attention [B, L, groups, 31])."""
    b, l, _ = x.shape
    pad = (WINDOW - 1) // 2
    xp = F.pad(x, [0, 0, pad, pad])
    q = net.linear(x, name + '.W_q', bias=False)
    k = net.linear(xp, name + '.W_k', bias=False).unfold(1, WINDOW, 1)
    v = net.linear(xp, name + '.W_v', bias=False).unfold(1, WINDOW, 1)
    k = k + net.p(name + '.rel')
    f = q.shape[-1]
    k = k.contiguous().view(b, l, groups, f // groups, WINDOW)
    v = v.contiguous().view(b, l, groups, f // groups, WINDOW)
    q = q.view(b, l, groups, f // groups, 1)
    energy = (q * k).sum(-2, keepdim=True)
    att = F.softmax(energy, dim=-1)
    out = (att * v).sum(-1).flatten(2)
    return out, att.squeeze(3)


def spec2roll_onset(net, x, pre='transcriber'):
    """Spec2Roll.forward (onset variant), model/UNet_onset.py:303-315."""
    y = unet(net, x, pre + '.Unet1_encoder', pre + '.Unet1_decoder')
    onset = torch.sigmoid(net.linear(y[:, 0], pre + '.linear_onset'))
    feat = net.linear(y[:, 1], pre + '.linear_feature')
    z = torch.cat((onset, feat), -1)
    z, a = local_attention(net, z, pre + '.combine_stack.attention', 6)
    z = net.linear(z, pre + '.combine_stack.linear')     # Dropout(0) is the identity
    return torch.sigmoid(z), onset, a


def spec2roll_frame(net, x, pre='transcriber'):
    """Spec2Roll.forward (no-onset variant), model/self_attention_VAT.py:938-945."""
    y = unet(net, x, pre + '.Unet1_encoder', pre + '.Unet1_decoder')
    z, a = local_attention(net, y.squeeze(1), pre + '.lstm1', 4)
    return torch.sigmoid(net.linear(z, pre + '.linear1')), a


def roll2spec(net, roll, pre='reconstructor'):
    """Roll2Spec.forward, model/UNet_onset.py:326-339."""
    z, a = local_attention(net, roll, pre + '.lstm2', 4)
    z = torch.sigmoid(net.linear(z, pre + '.linear2'))
    return unet(net, z.unsqueeze(1), pre + '.Unet2_encoder', pre + '.Unet2_decoder'), a


def l2_normalise(d):
    """_l2_normalize(binwise=False), model/UNet_onset.py:165-171."""
    return d / torch.norm(d, dim=-1, keepdim=True)


def vat_onset(params, training, x, xi, eps, d0=None, n_power=1):
    """UNet_VAT.forward for the onset model, model/UNet_onset.py:116-162
    (n_power in {0, 1}, KL_Div=False).  ``d0`` injects the initial noise (the reference
    draws torch.randn_like(x)); with n_power = 0 the loop at :129-142 never runs and d0 itself is normalised into r_adv."""
    with torch.no_grad():
        frame_ref, onset_ref, _ = spec2roll_onset(Net(params, training), x)
    d = (torch.randn_like(x) if d0 is None else d0.clone()).requires_grad_(True)
    g = None
    if n_power:
        x_adv = (x + xi * l2_normalise(d)).clamp(0, 1)
        fp, op, _ = spec2roll_onset(Net(params, training), x_adv)
        loss = F.binary_cross_entropy(fp, frame_ref) + F.binary_cross_entropy(op, onset_ref)
        # the reference backpropagates into the weights too, then model.zero_grad()s
        g, = torch.autograd.grad(loss, d)
        d = g.detach() * 1e10
    else:
        d = d.detach()
    r_adv = eps * l2_normalise(d)
    assert not torch.isnan(r_adv).any(), "r_adv has nan, please debug tune down the XI for VAT"
    x_adv = (x + r_adv).clamp(0, 1)
    fp, op, _ = spec2roll_onset(Net(params, training), x_adv)
    lds = {'frame': F.binary_cross_entropy(fp, frame_ref),
           'onset': F.binary_cross_entropy(op, onset_ref)}
    return lds, r_adv, l2_normalise(d), g


def vat_frame(params, training, x, xi, eps, d0=None, n_power=1):
    """UNet_VAT.forward for the no-onset model, model/self_attention_VAT.py:162-202 (n_power in {0, 1})."""
    with torch.no_grad():
        y_ref, _ = spec2roll_frame(Net(params, training), x)
    d = (torch.randn_like(x) if d0 is None else d0.clone()).requires_grad_(True)
    g = None
    if n_power:
        x_adv = (x + xi * l2_normalise(d)).clamp(0, 1)
        yp, _ = spec2roll_frame(Net(params, training), x_adv)
        g, = torch.autograd.grad(F.binary_cross_entropy(yp, y_ref), d)
        d = g.detach() * 1e10
    else:
        d = d.detach()
    r_adv = eps * l2_normalise(d)
    assert not torch.isnan(r_adv).any(), "r_adv has nan, please debug tune down the XI for VAT"
    yp, _ = spec2roll_frame(Net(params, training), (x + r_adv).clamp(0, 1))
    return F.binary_cross_entropy(yp, y_ref), r_adv, l2_normalise(d), g


def forward_onset(params, training, spec, reconstruction):
    """UNet_Onset.forward, model/UNet_onset.py:380-405."""
    net = Net(params, training)
    roll, onset, a = spec2roll_onset(net, spec)
    if not reconstruction:
        return roll, onset, a
    rec, _ = roll2spec(net, roll)
    roll2, onset2, _ = spec2roll_onset(net, rec)
    return rec, roll, onset, roll2, onset2, a


def forward_frame(params, training, spec, reconstruction):
    """UNet.forward, model/self_attention_VAT.py:1064-1086."""
    net = Net(params, training)
    roll, a = spec2roll_frame(net, spec)
    if not reconstruction:
        return roll, a
    rec, _ = roll2spec(net, roll)
    roll2, _ = spec2roll_frame(net, rec)
    return rec, roll, roll2, a


def _spec(params, audio, log=True):
    return fe.frontend(audio.reshape(-1, audio.shape[-1])[:, :-1], params, log)


def run_on_batch_onset(params, training, batch, batch_ul=None, VAT=False, reconstruction=True,
                       xi=1e-6, eps=2.0, d0_l=None, d0_ul=None, log=True, n_power=1):
    """UNet_Onset.run_on_batch, model/UNet_onset.py:409-542."""
    audio, onset_label, frame_label = batch['audio'], batch['onset'], batch['frame']
    if frame_label.dim() == 2:
        frame_label = frame_label.unsqueeze(0)
    if onset_label.dim() == 2:
        onset_label = onset_label.unsqueeze(0)
    zero = torch.tensor(0.)
    if batch_ul:
        spec_ul = _spec(params, batch_ul['audio'].reshape(-1, audio.shape[-1]), log)
        lds_ul, _, r_norm_ul, _ = vat_onset(params, training, spec_ul, xi, eps, d0_ul, n_power)
    else:
        lds_ul, r_norm_ul = {'frame': zero, 'onset': zero}, zero
    spec = _spec(params, audio, log)
    if VAT:
        lds_l, r_adv, r_norm_l, _ = vat_onset(params, training, spec, xi, eps, d0_l, n_power)
        r_adv = r_adv.squeeze(1)
    else:
        r_adv, lds_l, r_norm_l = None, {'frame': zero, 'onset': zero}, zero
    tag = 'train' if training else 'test'
    bce = F.binary_cross_entropy
    if reconstruction:
        rec, roll, onset, roll2, onset2, a = forward_onset(params, training, spec, True)
        pred = {'frame': roll, 'onset': onset, 'frame2': roll2, 'onset2': onset2,
                'attention': a, 'r_adv': r_adv, 'reconstruction': rec}
        losses = {
            f'loss/{tag}_reconstruction': F.mse_loss(rec.squeeze(1), spec.squeeze(1).detach()),
            f'loss/{tag}_frame': bce(roll, frame_label),
            f'loss/{tag}_frame2': bce(roll2, frame_label),
            f'loss/{tag}_onset': bce(onset, onset_label),
            f'loss/{tag}_onset2': bce(onset2, onset_label),
        }
    else:
        roll, onset, a = forward_onset(params, training, spec, False)
        pred = {'onset': onset, 'frame': roll, 'r_adv': r_adv, 'attention': a}
        losses = {f'loss/{tag}_frame': bce(roll, frame_label),
                  f'loss/{tag}_onset': bce(onset, onset_label)}
    losses[f'loss/{tag}_LDS_l_frame'] = lds_l['frame']
    losses[f'loss/{tag}_LDS_l_onset'] = lds_l['onset']
    if training:
        losses['loss/train_LDS_ul_frame'] = lds_ul['frame']
        losses['loss/train_LDS_ul_onset'] = lds_ul['onset']
    losses[f'loss/{tag}_r_norm_l'] = r_norm_l.abs().mean()
    if training:
        losses['loss/train_r_norm_ul'] = r_norm_ul.abs().mean()
    return pred, losses, spec.squeeze(1)


def run_on_batch_frame(params, training, batch, batch_ul=None, VAT=False, reconstruction=True,
                       xi=1e-6, eps=2.0, d0_l=None, d0_ul=None, log=True, n_power=1):
    """UNet.run_on_batch, model/self_attention_VAT.py:1090-1203."""
    audio, frame_label = batch['audio'], batch['frame']
    if frame_label.dim() == 2:
        frame_label = frame_label.unsqueeze(0)
    zero = torch.tensor(0.)
    if batch_ul:
        spec_ul = _spec(params, batch_ul['audio'].reshape(-1, audio.shape[-1]), log)
        lds_ul, _, r_norm_ul, _ = vat_frame(params, training, spec_ul, xi, eps, d0_ul, n_power)
    else:
        lds_ul, r_norm_ul = zero, zero
    spec = _spec(params, audio, log)
    if VAT:
        lds_l, r_adv, r_norm_l, _ = vat_frame(params, training, spec, xi, eps, d0_l, n_power)
        r_adv = r_adv.squeeze(1)
    else:
        r_adv, lds_l, r_norm_l = None, zero, zero
    tag = 'train' if training else 'test'
    bce = F.binary_cross_entropy
    if reconstruction:
        rec, roll, roll2, a = forward_frame(params, training, spec, True)
        pred = {'onset': roll, 'frame': roll, 'frame2': roll2, 'onset2': roll2,
                'attention': a, 'r_adv': r_adv, 'reconstruction': rec}
        losses = {
            f'loss/{tag}_reconstruction': F.mse_loss(rec.squeeze(1), spec.squeeze(1).detach()),
            f'loss/{tag}_frame': bce(roll, frame_label),
            f'loss/{tag}_frame2': bce(roll2, frame_label),
        }
    else:
        roll, a = forward_frame(params, training, spec, False)
        pred = {'onset': roll, 'frame': roll, 'r_adv': r_adv, 'attention': a}
        losses = {f'loss/{tag}_frame': bce(roll, frame_label)}
    losses[f'loss/{tag}_LDS_l'] = lds_l
    if training:
        losses['loss/train_LDS_ul'] = lds_ul
    losses[f'loss/{tag}_r_norm_l'] = r_norm_l.abs().mean()
    if training:
        losses['loss/train_r_norm_ul'] = r_norm_ul.abs().mean()
    return pred, losses, spec.squeeze(1)


def run_on_batch_application(params, training, batch, batch_ul, VAT=False, xi=1e-6, eps=2.0, d0_l=None, d0_ul=None, log=True, n_power=1):
    """UNet.run_on_batch_application, model/self_attention_VAT.py:1205-1291 (reconstruction=True model; the reference needs
    `batch_ul`: without it `spec` is read before assignment, :1225).  Adds the unlabelled consistency term
    `loss/ul_consistency_wrt1` = BCE(ul_frame2, ul_frame.detach())."""
    audio, frame_label = batch['audio'], batch['frame']
    if frame_label.dim() == 2:
        frame_label = frame_label.unsqueeze(0)
    zero = torch.tensor(0.)
    spec_ul = _spec(params, batch_ul['audio'].reshape(-1, audio.shape[-1]), log)
    lds_ul, _, r_norm_ul, _ = vat_frame(params, training, spec_ul, xi, eps, d0_ul, n_power)
    _, ul_roll, ul_roll2, _ = forward_frame(params, training, spec_ul, True)
    spec = _spec(params, audio, log)
    if VAT:
        lds_l, r_adv, r_norm_l, _ = vat_frame(params, training, spec, xi, eps, d0_l, n_power)
        r_adv = r_adv.squeeze(1)
    else:
        r_adv, lds_l, r_norm_l = None, zero, zero
    rec, roll, roll2, a = forward_frame(params, training, spec, True)
    bce = F.binary_cross_entropy
    if training:
        pred = {'onset': roll, 'frame': roll, 'frame2': roll2, 'onset2': roll2, 'ul_frame': ul_roll, 'ul_frame2': ul_roll2,
                'attention': a, 'r_adv': r_adv, 'reconstruction': rec}
        losses = {
            'loss/train_reconstruction': F.mse_loss(rec.squeeze(1), spec.squeeze(1).detach()),
            'loss/train_frame': bce(roll, frame_label),
            'loss/train_frame2': bce(roll2, frame_label),
            'loss/ul_consistency_wrt1': bce(ul_roll2, ul_roll.detach()),
            'loss/train_LDS_l': lds_l,
            'loss/train_LDS_ul': lds_ul,
            'loss/train_r_norm_l': r_norm_l.abs().mean(),
            'loss/train_r_norm_ul': r_norm_ul.abs().mean(),
        }
    else:
        roll, roll2 = roll.reshape(*frame_label.shape), roll2.reshape(*frame_label.shape)
        pred = {'onset': roll, 'frame': roll, 'frame2': roll2, 'onset2': roll2, 'attention': a, 'r_adv': r_adv,
                'reconstruction': rec}
        losses = {
            'loss/test_reconstruction': F.mse_loss(rec.squeeze(1), spec.squeeze(1).detach()),
            'loss/test_frame': bce(roll, frame_label),
            'loss/test_frame2': bce(roll2, frame_label),
            'loss/test_LDS_l': lds_l,
            'loss/test_r_norm_l': r_norm_l.abs().mean(),
        }
    return pred, losses, spec.squeeze(1)


def weighted_loss(losses, alpha=1.0):
    """model/helper_functions.py:589-595: LDS keys weigh alpha/2, the rest 1."""
    total = 0
    for key, val in losses.items():
        total = total + (alpha * val / 2 if key.startswith('loss/train_LDS') else val)
    return total


def trainable_keys(params):
    return [k for k, v in params.items()
            if v.is_floating_point() and not k.startswith('spectrogram.')
            and not k.endswith(('running_mean', 'running_var'))]


def train_step(params, adam_state, step_index, batch, batch_ul, run_fn, alpha=1.0, lr0=1e-3,
               decay_steps=1000, decay_rate=0.98, clip=3.0, **kw):
    """One iteration of train_VAT_model (model/helper_functions.py:577-607) with
    torch.optim.Adam(lr) + StepLR(decay_steps, decay_rate) restated in closed
    form: zero_grad, run_on_batch, weighted sum, backward, Adam, StepLR.
    The post-step clip_grad_norm_ (:606-607) has no effect on the update; it is
    applied to the ``.grad`` tensors afterwards exactly as the reference leaves them."""
    keys = trainable_keys(params)
    for k in keys:
        params[k].requires_grad_(True)
        params[k].grad = None
    pred, losses, spec = run_fn(params, True, batch, batch_ul, **kw)
    loss = weighted_loss(losses, alpha)
    loss.backward()
    lr = lr0 * decay_rate ** (step_index // decay_steps)
    b1, b2, eps = 0.9, 0.999, 1e-8
    t = step_index + 1
    with torch.no_grad():
        for k in keys:
            g = params[k].grad
            if g is None:       # Adam skips parameters that never received a gradient
                continue
            m, v = adam_state.setdefault(k, (torch.zeros_like(g), torch.zeros_like(g)))
            m.mul_(b1).add_(g, alpha=1 - b1)
            v.mul_(b2).addcmul_(g, g, value=1 - b2)
            denom = (v.sqrt() / (1 - b2 ** t) ** 0.5).add_(eps)
            params[k].addcdiv_(m, denom, value=-lr / (1 - b1 ** t))
        if clip:
            grads = [params[k].grad for k in keys if params[k].grad is not None]
            total = torch.norm(torch.stack([torch.norm(g, 2.0) for g in grads]), 2.0)
            coef = torch.clamp(clip / (total + 1e-6), max=1.0)
            for g in grads:
                g.mul_(coef)
    return pred, losses, loss.detach()
