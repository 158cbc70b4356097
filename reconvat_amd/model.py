"""Drop-in model surface of the ReconVAT hot path on MI355X.

Same class names, constructor arguments, attribute names, ``run_on_batch`` contract, loss / prediction
dictionary keys and ``state_dict`` keys as the reference:

    UNet_Onset ................ model/UNet_onset.py:341-553
    UNet ....................... model/self_attention_VAT.py:1014-1325
    UNet_VAT ................... model/UNet_onset.py:101-162, model/self_attention_VAT.py:147-202
    MutliHeadAttention1D ....... model/UNet_onset.py:22-98

The torch ``nn.Conv2d`` / ``nn.ConvTranspose2d`` / ``nn.BatchNorm2d`` / ``nn.Linear`` objects below are
PARAMETER CONTAINERS ONLY (they give the reference's parameter names, shapes and default initialisation);
their ``forward`` is never called -- every numeric step runs through ``reconvat_amd.ops`` (hand-written
HIP kernels).  Activations are NHWC internally; public tensors keep the reference's shapes.
"""
import os

import torch
import torch.nn as nn
import torch.nn.init as init

from . import ops
from .constants import N_BINS
from .frontend import MelSpectrogram, Normalization
from .ops import (ARENA, BnLink, ColsumLink, GradShare, ConvFn, UpCatFn, BnActFn, LinearFn, OnsetHeadsFn, LocalAttnFn, VatPerturbFn, SLOPE, bce_mean,
                  mse_mean, abs_mean)

batchNorm_momentum = 0.1


def _p(t, detach):
    return t.detach() if (detach and t is not None) else t


def _conv(m, x, kind, detach, size=None, share=None):
    return ConvFn.apply(x, _p(m.weight, detach), _p(m.bias, detach), kind, size, None, None, share)


# encoder blocks: `x12 += skip(x)` with the 1x1 skip conv evaluated inside the BatchNorm apply kernel (ops.BnActFn r1_*, rv_bn_lrelu_fwd_skip) instead of a conv
# launch writing it and the apply kernel reading it back; results are bit-identical either way.  RV_FUSE_SKIP = 0: never; 1 (default): the first block only (one
# input channel: a rank-1 term -- 20.36 vs 20.58 ms/step); 2: every block whose separate launch's arithmetic is known (ops.skip_conv_ksplit) -- correct and SLOWER
# (20.81 ms: the 16 .. 64-term fmaf chains and their LDS weight reads cost the apply kernel more than the conv launch did; profiles/r06_fused_skip_ab.txt)
FUSE_SKIP = [int(os.environ.get('RV_FUSE_SKIP', '1'))]


def _conv_bn(conv, bn, x, kind, res, detach, bn_in=None, link=None, share=None, dx_colsum=None, r1=None):
    """lrelu(bn(conv(x))) (+ res).  In training mode the conv leaves the batch statistics of its output in a
    zeroed fp64 slice (fused epilogue) and the BatchNorm skips its own statistics pass.  ``link`` (a fresh
    ops.BnLink) is handed to the ONE conv that consumes the result as ``bn_in``: that conv's input-gradient kernel
    then also produces this BatchNorm's backward reduction."""
    stats = ARENA.take(ops.bn_ws_doubles(bn.num_features), x.device) if bn.training else None
    z = ConvFn.apply(x, _p(conv.weight, detach), _p(conv.bias, detach), kind, None, stats, bn_in, share, dx_colsum)
    if r1 is not None:                 # (x1, skip conv, GradShare of x1): the residual is skip(x1), evaluated inside the apply kernel
        x1, sk, sh1 = r1
        return BnActFn.apply(z, _p(bn.weight, detach), _p(bn.bias, detach), bn.running_mean, bn.running_var,
                             bn.num_batches_tracked, None, bn.training, SLOPE, stats, link, x1, _p(sk.weight, detach), _p(sk.bias, detach), sh1)
    return BnActFn.apply(z, _p(bn.weight, detach), _p(bn.bias, detach), bn.running_mean, bn.running_var,
                         bn.num_batches_tracked, res, bn.training, SLOPE, stats, link)


class block(nn.Module):
    """Encoder stage (model/UNet_onset.py:186-201)."""

    def __init__(self, inp, out, ksize, pad, ds_ksize, ds_stride):
        super().__init__()
        self.conv1 = nn.Conv2d(inp, out, kernel_size=ksize, padding=pad)
        self.bn1 = nn.BatchNorm2d(out, momentum=batchNorm_momentum)
        self.conv2 = nn.Conv2d(out, out, kernel_size=ksize, padding=pad)
        self.bn2 = nn.BatchNorm2d(out, momentum=batchNorm_momentum)
        self.skip = nn.Conv2d(inp, out, kernel_size=1, padding=0)
        self.ds = nn.Conv2d(out, out, kernel_size=ds_ksize, stride=ds_stride, padding=0)

    def forward(self, x, detach=False, share=None):
        """share: the GradShare of x (x feeds conv1 and skip here, and -- for the inner blocks -- a decoder skip conv)."""
        share = share if share is not None else GradShare()
        l1 = BnLink()                                                          # a1 feeds conv2 only
        a1 = _conv_bn(self.conv1, self.bn1, x, 'c3', None, detach, link=l1, share=share)
        cin = self.skip.in_channels
        if (FUSE_SKIP[0] >= 1 and cin == 1) or (FUSE_SKIP[0] >= 2 and ops.skip_conv_ksplit(x, self.skip.out_channels) is not None):
            a2 = _conv_bn(self.conv2, self.bn2, a1, 'c3', None, detach, bn_in=l1, r1=(x, self.skip, share))      # lrelu(bn2(.)) + skip(x), skip evaluated in the apply kernel
        else:
            sk = _conv(self.skip, x, 'c1', detach, share=share)
            a2 = _conv_bn(self.conv2, self.bn2, a1, 'c3', sk, detach, bn_in=l1)     # lrelu(bn2(.)) + skip(x)
        xp = _conv(self.ds, a2, 'down', detach)
        return xp, (a2.shape[1], a2.shape[2])


class d_block(nn.Module):
    """Decoder stage (model/UNet_onset.py:203-224)."""

    def __init__(self, inp, out, isLast, ksize, pad, ds_ksize, ds_stride):
        super().__init__()
        self.conv2d = nn.ConvTranspose2d(inp, int(inp / 2), kernel_size=ksize, padding=pad)
        self.bn2d = nn.BatchNorm2d(int(inp / 2), momentum=batchNorm_momentum)
        self.conv1d = nn.ConvTranspose2d(int(inp / 2), out, kernel_size=ksize, padding=pad)
        if not isLast:
            self.bn1d = nn.BatchNorm2d(out, momentum=batchNorm_momentum)
            self.us = nn.ConvTranspose2d(inp - out, inp - out, kernel_size=ds_ksize, stride=ds_stride)
        else:
            self.us = nn.ConvTranspose2d(inp, inp, kernel_size=ds_ksize, stride=ds_stride)
        self.isLast = isLast

    def forward(self, x, size, skip_src=None, skip_conv=None, detach=False, share=None):
        cs = ColsumLink()            # conv2d's input-gradient kernel also leaves the column sums of dY(us): the up-conv's bias gradient
        if self.isLast:
            x = ConvFn.apply(x, _p(self.us.weight, detach), _p(self.us.bias, detach), 'up', size, None, None, None, None, cs)
        else:
            x = UpCatFn.apply(x, _p(self.us.weight, detach), _p(self.us.bias, detach), skip_src,
                              _p(skip_conv.weight, detach), _p(skip_conv.bias, detach), size, share, cs)
        l2 = BnLink()                                                          # the bn2d output feeds conv1d only
        x = _conv_bn(self.conv2d, self.bn2d, x, 't3', None, detach, link=l2, dx_colsum=cs)
        if self.isLast:
            return ConvFn.apply(x, _p(self.conv1d.weight, detach), _p(self.conv1d.bias, detach), 't3', None, None, l2)
        return _conv_bn(self.conv1d, self.bn1d, x, 't3', None, detach, bn_in=l2)


class Encoder(nn.Module):
    def __init__(self, ds_ksize, ds_stride):
        super().__init__()
        self.block1 = block(1, 16, (3, 3), (1, 1), ds_ksize, ds_stride)
        self.block2 = block(16, 32, (3, 3), (1, 1), ds_ksize, ds_stride)
        self.block3 = block(32, 64, (3, 3), (1, 1), ds_ksize, ds_stride)
        self.block4 = block(64, 128, (3, 3), (1, 1), ds_ksize, ds_stride)
        self.conv1 = nn.Conv2d(64, 64, kernel_size=(3, 3), padding=(1, 1))
        self.conv2 = nn.Conv2d(32, 32, kernel_size=(3, 3), padding=(1, 1))
        self.conv3 = nn.Conv2d(16, 16, kernel_size=(3, 3), padding=(1, 1))

    def forward(self, x, detach=False):
        """x: NHWC [B, T, bins, 1].  Returns (x4, sizes, skip sources); the skip convs conv1..3
        (model/UNet_onset.py:244-246) are evaluated by the decoder straight into its concat buffers."""
        g1, g2, g3 = GradShare(), GradShare(), GradShare()      # x1..x3 feed the next block (conv1 + skip) and a decoder skip conv
        x1, s1 = self.block1(x, detach)
        x2, s2 = self.block2(x1, detach, g1)
        x3, s3 = self.block3(x2, detach, g2)
        x4, s4 = self.block4(x3, detach, g3)
        return x4, [s1, s2, s3, s4], [(x3, self.conv1, g3), (x2, self.conv2, g2), (x1, self.conv3, g1)]


class Decoder(nn.Module):
    def __init__(self, ds_ksize, ds_stride, num_instruments=1):
        super().__init__()
        self.d_block1 = d_block(192, 64, False, (3, 3), (1, 1), ds_ksize, ds_stride)
        self.d_block2 = d_block(96, 32, False, (3, 3), (1, 1), ds_ksize, ds_stride)
        self.d_block3 = d_block(48, 16, False, (3, 3), (1, 1), ds_ksize, ds_stride)
        self.d_block4 = d_block(16, num_instruments, True, (3, 3), (1, 1), ds_ksize, ds_stride)

    def forward(self, x, s, c, detach=False):
        x = self.d_block1(x, s[3], c[0][0], c[0][1], detach, c[0][2])
        x = self.d_block2(x, s[2], c[1][0], c[1][1], detach, c[1][2])
        x = self.d_block3(x, s[1], c[2][0], c[2][1], detach, c[2][2])
        return self.d_block4(x, s[0], None, None, detach)


def _unet(enc, dec, x_nchw1, detach):
    """[B, 1, T, bins] -> NHWC [B, T, bins, C_out]."""
    b, c, t, f = x_nchw1.shape
    assert c == 1
    x = x_nchw1.contiguous().view(b, t, f, 1)
    x4, s, skips = enc(x, detach)
    return dec(x4, s, skips, detach)


class MutliHeadAttention1D(nn.Module):
    def __init__(self, in_features, out_features, kernel_size, stride=1, groups=1, position=True, bias=False):
        super().__init__()
        assert kernel_size == 31 and stride == 1 and position and not bias, \
            'the fused kernel implements the configuration the reference uses (window 31, relative position, no bias)'
        assert out_features % groups == 0, \
            f"out_channels should be divided by groups. Now out_channels={out_features}, groups={groups}"
        self.out_features = out_features
        self.kernel_size = kernel_size
        self.stride = stride
        self.position = position
        self.padding = (kernel_size - 1) // 2
        self.groups = groups
        self.rel = nn.Parameter(torch.randn(1, out_features, kernel_size), requires_grad=True)
        self.W_k = nn.Linear(in_features, out_features, bias=bias)
        self.W_q = nn.Linear(in_features, out_features, bias=bias)
        self.W_v = nn.Linear(in_features, out_features, bias=bias)
        self.reset_parameters()

    def forward(self, x, detach=False):
        return LocalAttnFn.apply(x.contiguous(), _p(self.W_q.weight, detach), _p(self.W_k.weight, detach),
                                 _p(self.W_v.weight, detach), _p(self.rel, detach), self.groups)

    def reset_parameters(self):
        init.kaiming_normal_(self.W_k.weight, mode='fan_out', nonlinearity='relu')
        init.kaiming_normal_(self.W_v.weight, mode='fan_out', nonlinearity='relu')
        init.kaiming_normal_(self.W_q.weight, mode='fan_out', nonlinearity='relu')
        if self.position:
            init.normal_(self.rel, 0, 1)


def _linear(m, x3, act, detach):
    b, l, k = x3.shape
    y = LinearFn.apply(x3.reshape(b * l, k), _p(m.weight, detach), _p(m.bias, detach), act)
    return y.view(b, l, -1)


class Stack(nn.Module):
    def __init__(self, input_size, hidden_dim, attn_size=31, attn_group=4, output_dim=88, dropout=0.5):
        super().__init__()
        assert dropout == 0, 'the reference only instantiates Stack with dropout=0 (model/UNet_onset.py:301)'
        self.attention = MutliHeadAttention1D(input_size, hidden_dim, attn_size, position=True, groups=attn_group)
        self.linear = nn.Linear(hidden_dim, output_dim)
        self.dropout = nn.Dropout(dropout)

    def forward(self, x, act=0, detach=False):
        x, a = self.attention(x, detach)
        return _linear(self.linear, x, act, detach), a


class Spec2Roll(nn.Module):
    """Transcriber.  onset=True: model/UNet_onset.py:284-315; onset=False: model/self_attention_VAT.py:929-945."""

    def __init__(self, ds_ksize, ds_stride, complexity=4, onset=True):
        super().__init__()
        self.onset = onset
        self.Unet1_encoder = Encoder(ds_ksize, ds_stride)
        self.Unet1_decoder = Decoder(ds_ksize, ds_stride, 2 if onset else 1)
        if onset:
            self.lstm1 = MutliHeadAttention1D(N_BINS + 88, N_BINS * complexity, 31, position=True, groups=complexity)
            self.linear1 = nn.Linear(N_BINS * complexity, 88)
            self.linear_onset = nn.Linear(N_BINS, 88)
            self.linear_feature = nn.Linear(N_BINS, 88)
            self.dropout_layer = nn.Dropout(0.5)
            self.combine_stack = Stack(input_size=88 * 2, hidden_dim=768, attn_size=31, attn_group=6, output_dim=88,
                                       dropout=0)
        else:
            self.lstm1 = MutliHeadAttention1D(N_BINS, N_BINS * complexity, 31, position=True, groups=complexity)
            self.linear1 = nn.Linear(N_BINS * complexity, 88)

    def forward(self, x, detach=False):
        b, _, t, f = x.shape
        y = _unet(self.Unet1_encoder, self.Unet1_decoder, x, detach)
        if self.onset:
            cat, onset = OnsetHeadsFn.apply(y, _p(self.linear_onset.weight, detach), _p(self.linear_onset.bias, detach),
                                            _p(self.linear_feature.weight, detach), _p(self.linear_feature.bias, detach))
            roll, a = self.combine_stack(cat.view(b, t, 176), act=1, detach=detach)
            return roll, onset.view(b, t, 88), a
        z, a = self.lstm1(y.view(b, t, f), detach)
        return _linear(self.linear1, z, 1, detach), a


class Roll2Spec(nn.Module):
    """Reconstructor (model/UNet_onset.py:317-339)."""

    def __init__(self, ds_ksize, ds_stride, complexity=4):
        super().__init__()
        self.Unet2_encoder = Encoder(ds_ksize, ds_stride)
        self.Unet2_decoder = Decoder(ds_ksize, ds_stride, 1)
        self.lstm2 = MutliHeadAttention1D(88, N_BINS * complexity, 31, position=True, groups=4)
        self.linear2 = nn.Linear(N_BINS * complexity, N_BINS)

    def forward(self, x, detach=False):
        b, t, _ = x.shape
        z, a = self.lstm2(x, detach)
        s = _linear(self.linear2, z, 1, detach)                    # sigmoid(linear2(.)) [B, T, 229]
        y = _unet(self.Unet2_encoder, self.Unet2_decoder, s.unsqueeze(1), detach)
        return y.view(b, 1, t, y.shape[2]), a


def _l2_normalize(d, binwise=False):
    """model/UNet_onset.py:165-171 (host-visible helper; the hot path uses the fused kernels)."""
    if binwise:
        return d / (torch.abs(d) + 1e-8)
    return d / torch.norm(d, dim=-1, keepdim=True)


class UNet_VAT(nn.Module):
    """Virtual adversarial perturbation by one power iteration (model/UNet_onset.py:101-162).

    The reference back-propagates the power-iteration loss into the weights and then discards those
    gradients with ``model.zero_grad()``; here the weights are detached for that pass, so only the
    input-gradient chain runs (identical ``d.grad``)."""

    def __init__(self, XI, epsilon, n_power, KL_Div, reconstruction=False):
        super().__init__()
        if KL_Div:
            raise NotImplementedError('KL_Div=True references undefined names in the reference '
                                      '(model/UNet_onset.py:133-134); only the BCE distance exists')
        self.n_power = n_power
        self.XI = XI
        self.epsilon = epsilon
        self.KL_Div = KL_Div
        self.binwise = False
        self.reconstruction = reconstruction
        self.nan_flag = None
        self.noise = None          # optional callable(x) -> d0 (tests inject deterministic noise)

    @staticmethod
    def _outputs(model, x, detach=False):
        out = model.transcriber(x, detach) if detach else model.transcriber(x)
        return out[:-1]            # (frame[, onset]) without the attention map

    def power_iteration(self, model, x, refs=None):
        """Everything of the VAT call that needs no weight gradients: the target predictions (`refs`: the
        transcriber's outputs on `x` if the caller already has them -- run_on_batch reuses its main forward pass,
        see _Base._vat_reusing_forward), the power iteration (forward on x + XI*d, input-gradient backward) and the
        adversarial input.  Returns (x_adv, r_adv, d_normalised, refs)."""
        if self.nan_flag is None or self.nan_flag.device != x.device:
            self.nan_flag = torch.zeros(1, dtype=torch.int32, device=x.device)
        if refs is None:
            with torch.no_grad():
                refs = self._outputs(model, x)
        else:
            refs = tuple(r.detach() for r in refs)
        d = (self.noise(x) if self.noise is not None else torch.randn_like(x)).requires_grad_(True)
        g = None
        for it in range(self.n_power):
            if it > 0:
                d = (g * 1e10).requires_grad_(True)
            x_adv = VatPerturbFn.apply(x, d, float(self.XI))
            preds = self._outputs(model, x_adv, detach=True)
            loss = None
            for p, r in zip(preds, refs):
                term = bce_mean(p, r)
                loss = term if loss is None else loss + term
            g, = torch.autograd.grad(loss, d)
            g = g.detach()
        # d = d.grad * 1e10 ; r_adv = eps * d / ||d||   (model/UNet_onset.py:141-151).  n_power = 0: the loop body never runs in the
        # reference either, d stays the random draw and r_adv = eps * d / ||d|| (a RANDOM, not adversarial, perturbation)
        if g is None:
            x_adv, r_adv, d_norm = ops.vat_adversarial(x, d.detach(), 1.0, float(self.epsilon), self.nan_flag)
        else:
            x_adv, r_adv, d_norm = ops.vat_adversarial(x, g, 1e10, float(self.epsilon), self.nan_flag)
        return x_adv, r_adv, d_norm, refs

    def check_nan(self):
        if not torch.cuda.is_current_stream_capturing():
            assert int(self.nan_flag.item()) == 0, \
                "r_adv has nan, please debug tune down the XI for VAT"

    def final_loss(self, model, x_adv, refs):
        """The grad-enabled pass on the adversarial input and its distance to the target predictions."""
        preds = self._outputs(model, x_adv)
        losses = [bce_mean(p, r) for p, r in zip(preds, refs)]
        if len(losses) == 2:
            return {'frame': losses[0], 'onset': losses[1]}
        return losses[0]

    def forward(self, model, x, refs=None):
        x_adv, r_adv, d_norm, refs = self.power_iteration(model, x, refs)
        self.check_nan()
        return self.final_loss(model, x_adv, refs), r_adv, d_norm


class _Base(nn.Module):
    def __init__(self, log, reconstruction, mode, spec, XI, eps):
        super().__init__()
        if spec != 'Mel':
            raise NotImplementedError("only spec='Mel' is on the MI355X hot path (the reference scripts' default)")
        self.spectrogram = MelSpectrogram()
        self.log = log
        self.normalize = Normalization(mode)
        self.reconstruction = reconstruction
        self.vat_loss = UNet_VAT(XI, eps, 1, False)

    def __del__(self):
        try:
            ops.drop_packs_of_params(list(self.parameters()))     # the packed-weight cache holds its source weights: they go with the model
        except Exception:  # noqa: BLE001 -- interpreter shutdown
            pass

    def _front(self, audio, ref_len):
        audio = audio.reshape(-1, ref_len)[:, :-1]
        if self.normalize.mode == 'imagewise':
            return self.spectrogram.lognorm(audio, log=self.log, normalise=True)
        spec = self.spectrogram(audio)
        if self.log:
            spec = torch.log(spec + 1e-5)
        return self.normalize.transform(spec).transpose(-1, -2).unsqueeze(1).contiguous()

    def _vat_reusing_forward(self, spec):
        """Labelled-branch VAT + the main transcriber forward with ONE transcriber pass less than the reference.

        The reference evaluates transcriber(spec) twice with identical weights: once under no_grad as the VAT
        target (model/UNet_onset.py:117-118) and once as the main forward (:383).  Train-mode BatchNorm makes both
        passes bit-identical functions of the batch, so the main (grad-enabled) pass is run first and its
        detached outputs serve as the VAT target.  The only state the skipped pass would have touched are the
        BatchNorm running statistics: the main pass runs with deferred updates, which are replayed at the two
        positions of the reference's update sequence (before and after the two perturbed VAT passes)."""
        if self.training:
            with ops.deferred_bn_updates() as pending:
                out = self.transcriber(spec)
            pending.apply()                                   # position of the no_grad reference pass
        else:
            pending = None
            out = self.transcriber(spec)
        lds, r_adv, r_norm = self.vat_loss(self, spec, refs=out[:-1])
        if pending is not None:
            pending.apply()                                   # position of the main forward pass
        return out, lds, r_adv, r_norm

    def _vat_two_streams(self, audio_ul, audio_l, recon_fn=None):
        """A training step as TWO concurrent kernel chains of equal length (two independent chains hide each other's
        launch tails and latency-bound kernels; measured 17 % on the overlapped region):

            side stream : front-end(ul), UL target pass, UL power iteration | reconstruction branch R -> T(recon) + its losses
            this stream : front-end(l), main forward T(x), L power iteration | UL final pass, L final pass

        and, because autograd runs every node's backward on its forward stream, the backward splits the same way
        (side: T(recon), R; here: UL final, L final, then T(x) which joins both).  Shared state is kept race-free:
        * the power iterations produce no parameter gradients (detached weights); the reconstruction branch's parameter
          gradients go to the side stream's twin of the flat gradient bucket (ops.SIDE_GRADS, folded in by TrainStep);
        * every BatchNorm running-statistic update of the concurrent passes is deferred and replayed on this stream in
          the reference's order: UL target, UL xi*d, UL final, L target, L xi*d, L final, main forward, R, T(recon).
        `recon_fn(first, spec)` runs the reconstruction branch (None: model without reconstructor).
        Returns (spec_l, main outputs, lds_ul, d_ul, lds_l, r_adv_l, d_l, recon_fn's result)."""
        cur = torch.cuda.current_stream()
        side = ops.side_stream(audio_l.device)
        ref_len = audio_l.shape[-1]
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            spec_ul = self._front(audio_ul, ref_len)
            with ops.deferred_bn_updates() as pend_ul:
                xa_ul, r_ul, dn_ul, refs_ul = self.vat_loss.power_iteration(self, spec_ul)
            ul_done = torch.cuda.Event()
            ul_done.record(side)
        spec = self._front(audio_l, ref_len)
        with ops.deferred_bn_updates() as pend_main:
            out = self.transcriber(spec)
        pack, pend_r = None, None
        if recon_fn is not None:
            main_done = torch.cuda.Event()
            main_done.record(cur)
            with torch.cuda.stream(side):
                side.wait_event(main_done)
                for t in (spec,) + tuple(out):
                    t.record_stream(side)
                with ops.deferred_bn_updates() as pend_r:
                    pack = recon_fn(out, spec)
        with ops.deferred_bn_updates() as pend_l:
            xa_l, r_l, dn_l, refs_l = self.vat_loss.power_iteration(self, spec, refs=out[:-1])
        cur.wait_event(ul_done)
        for t in (spec_ul, xa_ul, r_ul, dn_ul) + tuple(refs_ul):
            t.record_stream(cur)                              # produced on the side stream, consumed here from now on
        with ops.deferred_bn_updates() as pend_ulf:
            lds_ul = self.vat_loss.final_loss(self, xa_ul, refs_ul)
        with ops.deferred_bn_updates() as pend_lf:
            lds_l = self.vat_loss.final_loss(self, xa_l, refs_l)
        cur.wait_stream(side)
        if pack is not None:
            for t in pack.values():
                t.record_stream(cur)
        # every running-statistic update of the step, in the reference's order, in ONE launch (pend_main twice: the
        # labelled no_grad target pass and the main forward pass are the same computation)
        seq = [pend_ul, pend_ulf, pend_main, pend_l, pend_lf, pend_main] + ([pend_r] if pend_r is not None else [])
        ops.replay_bn_updates(seq, audio_l.device)
        self.vat_loss.check_nan()
        return spec, out, lds_ul, dn_ul, lds_l, r_l, dn_l, pack

    def load_my_state_dict(self, state_dict):
        own_state = self.state_dict()
        for name, param in state_dict.items():
            if name not in own_state:
                continue
            if isinstance(param, nn.Parameter):
                param = param.data
            own_state[name].copy_(param)
        ops.invalidate_weight_cache()

    def _load_from_state_dict(self, *args, **kwargs):
        super()._load_from_state_dict(*args, **kwargs)
        ops.invalidate_weight_cache()


class UNet_Onset(_Base):
    def __init__(self, ds_ksize, ds_stride, log=True, reconstruction=True, mode='imagewise', spec='CQT', device='cpu',
                 XI=1e-6, eps=1e-2):
        super().__init__(log, reconstruction, mode, spec, XI, eps)
        self.transcriber = Spec2Roll(ds_ksize, ds_stride, onset=True)
        if reconstruction:
            self.reconstructor = Roll2Spec(ds_ksize, ds_stride)

    def forward(self, x, _first=None):
        pianoroll, onset, a = self.transcriber(x) if _first is None else _first
        if self.reconstruction:
            reconstruction, _ = self.reconstructor(pianoroll)
            pianoroll2, onset2, _ = self.transcriber(reconstruction)
            return reconstruction, pianoroll, onset, pianoroll2, onset2, a
        return pianoroll, onset, a

    def run_on_batch(self, batch, batch_ul=None, VAT=False):
        audio_label = batch['audio']
        onset_label = batch['onset']
        frame_label = batch['frame']
        if frame_label.dim() == 2:
            frame_label = frame_label.unsqueeze(0)
        if onset_label.dim() == 2:
            onset_label = onset_label.unsqueeze(0)
        dual = bool(batch_ul) and VAT and self.training and ops.DUAL_STREAM[0] and audio_label.is_cuda
        pack = None
        if dual:
            def recon_fn(first_, spec_):
                rec_, _, _, roll2_, onset2_, _ = self(spec_, first_)
                return {'rec': rec_, 'frame2': roll2_, 'onset2': onset2_,
                        'l_rec': mse_mean(rec_.squeeze(1), spec_.squeeze(1)),
                        'l_frame2': bce_mean(roll2_, frame_label), 'l_onset2': bce_mean(onset2_, onset_label)}
            spec, first, lds_ul, r_norm_ul, lds_l, r_adv, r_norm_l, pack = self._vat_two_streams(
                batch_ul['audio'], audio_label, recon_fn if self.reconstruction else None)
            r_norm_ul, r_norm_l, r_adv = abs_mean(r_norm_ul), abs_mean(r_norm_l), r_adv.squeeze(1)
        elif batch_ul:
            spec = self._front(batch_ul['audio'], audio_label.shape[-1])
            lds_ul, _, r_norm_ul = self.vat_loss(self, spec)
            r_norm_ul = abs_mean(r_norm_ul)
        else:
            lds_ul = {'frame': torch.tensor(0.), 'onset': torch.tensor(0.)}
            r_norm_ul = torch.tensor(0.)
        if not dual:
            spec = self._front(audio_label, audio_label.shape[-1])
            first = None
        if dual:
            pass
        elif VAT:
            first, lds_l, r_adv, r_norm_l = self._vat_reusing_forward(spec)
            r_adv = r_adv.squeeze(1)
            r_norm_l = abs_mean(r_norm_l)
        else:
            r_adv = None
            lds_l = {'frame': torch.tensor(0.), 'onset': torch.tensor(0.)}
            r_norm_l = torch.tensor(0.)
        tag = 'train' if self.training else 'test'
        if self.reconstruction and pack is not None:
            pianoroll, onset, a = first
            predictions = {'frame': pianoroll, 'onset': onset, 'frame2': pack['frame2'], 'onset2': pack['onset2'], 'attention': a,
                           'r_adv': r_adv, 'reconstruction': pack['rec']}
            losses = {
                f'loss/{tag}_reconstruction': pack['l_rec'],
                f'loss/{tag}_frame': bce_mean(pianoroll, frame_label),
                f'loss/{tag}_frame2': pack['l_frame2'],
                f'loss/{tag}_onset': bce_mean(onset, onset_label),
                f'loss/{tag}_onset2': pack['l_onset2'],
            }
        elif self.reconstruction:
            reconstrut, pianoroll, onset, pianoroll2, onset2, a = self(spec, first)
            predictions = {'frame': pianoroll, 'onset': onset, 'frame2': pianoroll2, 'onset2': onset2, 'attention': a,
                           'r_adv': r_adv, 'reconstruction': reconstrut}
            if not self.training:
                predictions['frame'] = pianoroll.reshape(*frame_label.shape)
                predictions['frame2'] = pianoroll2.reshape(*frame_label.shape)
            losses = {
                f'loss/{tag}_reconstruction': mse_mean(reconstrut.squeeze(1), spec.squeeze(1)),
                f'loss/{tag}_frame': bce_mean(predictions['frame'], frame_label),
                f'loss/{tag}_frame2': bce_mean(predictions['frame2'], frame_label),
                f'loss/{tag}_onset': bce_mean(predictions['onset'], onset_label),
                f'loss/{tag}_onset2': bce_mean(predictions['onset2'], onset_label),
            }
        else:
            frame_pred, onset, a = self(spec, first)
            predictions = {'onset': onset, 'frame': frame_pred, 'r_adv': r_adv, 'attention': a}
            if not self.training:
                predictions['frame'] = frame_pred.reshape(*frame_label.shape)
            losses = {f'loss/{tag}_frame': bce_mean(predictions['frame'], frame_label),
                      f'loss/{tag}_onset': bce_mean(predictions['onset'], onset_label)}
        losses[f'loss/{tag}_LDS_l_frame'] = lds_l['frame']
        losses[f'loss/{tag}_LDS_l_onset'] = lds_l['onset']
        if self.training:
            losses['loss/train_LDS_ul_frame'] = lds_ul['frame']
            losses['loss/train_LDS_ul_onset'] = lds_ul['onset']
        losses[f'loss/{tag}_r_norm_l'] = r_norm_l
        if self.training:
            losses['loss/train_r_norm_ul'] = r_norm_ul
        return predictions, losses, spec.squeeze(1)


class UNet(_Base):
    def __init__(self, ds_ksize, ds_stride, log=True, reconstruction=True, mode='imagewise', spec='CQT', device='cpu',
                 XI=1e-6, eps=1e-2):
        super().__init__(log, reconstruction, mode, spec, XI, eps)
        self.transcriber = Spec2Roll(ds_ksize, ds_stride, onset=False)
        if reconstruction:
            self.reconstructor = Roll2Spec(ds_ksize, ds_stride)

    def forward(self, x, _first=None):
        pianoroll, a = self.transcriber(x) if _first is None else _first
        if self.reconstruction:
            reconstruction, _ = self.reconstructor(pianoroll)
            pianoroll2, _ = self.transcriber(reconstruction)
            return reconstruction, pianoroll, pianoroll2, a
        return pianoroll, a

    def run_on_batch(self, batch, batch_ul=None, VAT=False):
        audio_label = batch['audio']
        frame_label = batch['frame']
        if frame_label.dim() == 2:
            frame_label = frame_label.unsqueeze(0)
        dual = bool(batch_ul) and VAT and self.training and ops.DUAL_STREAM[0] and audio_label.is_cuda
        if dual:
            spec, first, lds_ul, r_norm_ul, lds_l, r_adv, r_norm_l, _ = self._vat_two_streams(batch_ul['audio'], audio_label)
            r_norm_ul, r_norm_l, r_adv = abs_mean(r_norm_ul), abs_mean(r_norm_l), r_adv.squeeze(1)
        elif batch_ul:
            spec = self._front(batch_ul['audio'], audio_label.shape[-1])
            lds_ul, _, r_norm_ul = self.vat_loss(self, spec)
            r_norm_ul = abs_mean(r_norm_ul)
        else:
            lds_ul = torch.tensor(0.)
            r_norm_ul = torch.tensor(0.)
        if not dual:
            spec = self._front(audio_label, audio_label.shape[-1])
            first = None
        if dual:
            pass
        elif VAT:
            first, lds_l, r_adv, r_norm_l = self._vat_reusing_forward(spec)
            r_adv = r_adv.squeeze(1)
            r_norm_l = abs_mean(r_norm_l)
        else:
            r_adv = None
            lds_l = torch.tensor(0.)
            r_norm_l = torch.tensor(0.)
        tag = 'train' if self.training else 'test'
        if self.reconstruction:
            reconstrut, pianoroll, pianoroll2, a = self(spec, first)
            if not self.training:
                pianoroll = pianoroll.reshape(*frame_label.shape)
                pianoroll2 = pianoroll2.reshape(*frame_label.shape)
            predictions = {'onset': pianoroll, 'frame': pianoroll, 'frame2': pianoroll2, 'onset2': pianoroll2,
                           'attention': a, 'r_adv': r_adv, 'reconstruction': reconstrut}
            losses = {
                f'loss/{tag}_reconstruction': mse_mean(reconstrut.squeeze(1), spec.squeeze(1)),
                f'loss/{tag}_frame': bce_mean(predictions['frame'], frame_label),
                f'loss/{tag}_frame2': bce_mean(predictions['frame2'], frame_label),
            }
        else:
            frame_pred, a = self(spec, first)
            if not self.training:
                frame_pred = frame_pred.reshape(*frame_label.shape)
            predictions = {'onset': frame_pred, 'frame': frame_pred, 'r_adv': r_adv, 'attention': a}
            losses = {f'loss/{tag}_frame': bce_mean(predictions['frame'], frame_label)}
        losses[f'loss/{tag}_LDS_l'] = lds_l
        if self.training:
            losses['loss/train_LDS_ul'] = lds_ul
        losses[f'loss/{tag}_r_norm_l'] = r_norm_l
        if self.training:
            losses['loss/train_r_norm_ul'] = r_norm_ul
        return predictions, losses, spec.squeeze(1)

    def run_on_batch_application(self, batch, batch_ul=None, VAT=False):
        """model/self_attention_VAT.py:1205-1291: the fine-tuning-on-unlabelled-recordings variant of run_on_batch --
        the unlabelled batch additionally goes through transcriber -> reconstructor -> transcriber and contributes the
        consistency term `loss/ul_consistency_wrt1` = BCE(ul_frame2, ul_frame.detach()).  Same contract as the reference:
        needs the reconstruction model and an unlabelled batch (the reference reads `spec` before assignment without one)."""
        if not self.reconstruction:
            raise ValueError('run_on_batch_application needs reconstruction=True (the reference unpacks four forward outputs)')
        if not batch_ul:
            raise UnboundLocalError("run_on_batch_application needs batch_ul (model/self_attention_VAT.py:1225 reads `spec` "
                                    'before assignment without it)')
        audio_label = batch['audio']
        frame_label = batch['frame']
        if frame_label.dim() == 2:
            frame_label = frame_label.unsqueeze(0)
        spec_ul = self._front(batch_ul['audio'], audio_label.shape[-1])
        lds_ul, _, r_norm_ul = self.vat_loss(self, spec_ul)
        _, ul_pianoroll, ul_pianoroll2, _ = self(spec_ul)
        spec = self._front(audio_label, audio_label.shape[-1])
        if VAT:
            first, lds_l, r_adv, r_norm_l = self._vat_reusing_forward(spec)
            r_adv = r_adv.squeeze(1)
            r_norm_l = abs_mean(r_norm_l)
        else:
            first, r_adv, lds_l, r_norm_l = None, None, torch.tensor(0.), torch.tensor(0.)
        reconstrut, pianoroll, pianoroll2, a = self(spec, first)
        if self.training:
            predictions = {'onset': pianoroll, 'frame': pianoroll, 'frame2': pianoroll2, 'onset2': pianoroll2,
                           'ul_frame': ul_pianoroll, 'ul_frame2': ul_pianoroll2, 'attention': a, 'r_adv': r_adv,
                           'reconstruction': reconstrut}
            losses = {
                'loss/train_reconstruction': mse_mean(reconstrut.squeeze(1), spec.squeeze(1)),
                'loss/train_frame': bce_mean(pianoroll, frame_label),
                'loss/train_frame2': bce_mean(pianoroll2, frame_label),
                'loss/ul_consistency_wrt1': bce_mean(ul_pianoroll2, ul_pianoroll.detach()),
                'loss/train_LDS_l': lds_l,
                'loss/train_LDS_ul': lds_ul,
                'loss/train_r_norm_l': r_norm_l,
                'loss/train_r_norm_ul': abs_mean(r_norm_ul),
            }
        else:
            pianoroll = pianoroll.reshape(*frame_label.shape)
            pianoroll2 = pianoroll2.reshape(*frame_label.shape)
            predictions = {'onset': pianoroll, 'frame': pianoroll, 'frame2': pianoroll2, 'onset2': pianoroll2, 'attention': a,
                           'r_adv': r_adv, 'reconstruction': reconstrut}
            losses = {
                'loss/test_reconstruction': mse_mean(reconstrut.squeeze(1), spec.squeeze(1)),
                'loss/test_frame': bce_mean(pianoroll, frame_label),
                'loss/test_frame2': bce_mean(pianoroll2, frame_label),
                'loss/test_LDS_l': lds_l,
                'loss/test_r_norm_l': r_norm_l,
            }
        return predictions, losses, spec.squeeze(1)

    def transcribe(self, batch):
        """model/self_attention_VAT.py:1293-1314."""
        audio_label = batch['audio']
        spec = self._front(audio_label, audio_label.shape[-1])
        out = self(spec)
        pianoroll = out[1] if self.reconstruction else out[0]
        return {'onset': pianoroll, 'frame': pianoroll}
