"""Oracle fixtures: closed-form deterministic weights and inputs.  TEST
INFRASTRUCTURE ONLY.

Every value is a pure function of (tensor name, flat index) through an integer
hash (splitmix64), so the same parameter set / input can be regenerated on any
machine without shipping megabytes of weights: the golden generator (run in the
build container against /root/reference), the CPU oracle tests and the GPU
parity tests all call these functions.

The shape table restates the reference constructors:
  Encoder/Decoder/block/d_block ... model/UNet_onset.py:173-258
  Spec2Roll (onset) ................ model/UNet_onset.py:284-301
  Spec2Roll (no onset) ............. model/self_attention_VAT.py:929-936
  Roll2Spec ........................ model/UNet_onset.py:317-324
"""
import zlib

import numpy as np
import torch

from . import frontend as fe

N_BINS = 229


def _hash_uniform(name, n, lo=-1.0, hi=1.0):
    """n floats in [lo, hi): splitmix64 of (crc32(name), index), top 24 bits."""
    seed = np.uint64(zlib.crc32(name.encode()))
    with np.errstate(over='ignore'):
        z = (np.arange(1, n + 1, dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15)
             + seed * np.uint64(0xBF58476D1CE4E5B9))
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    u = (z >> np.uint64(40)).astype(np.float64) / float(1 << 24)
    return (lo + (hi - lo) * u).astype(np.float32)


def hashed(name, shape, scale=1.0, offset=0.0):
    n = int(np.prod(shape))
    return torch.from_numpy(_hash_uniform(name, n) * np.float32(scale) + np.float32(offset)).reshape(shape)


def hashed_normalish(name, shape, scale=1.0):
    """Sum of three uniforms: bell-shaped, std = scale, bounded."""
    n = int(np.prod(shape))
    s = _hash_uniform(name + '#a', n) + _hash_uniform(name + '#b', n) + _hash_uniform(name + '#c', n)
    return torch.from_numpy(s.astype(np.float32) * np.float32(scale)).reshape(shape)


def _unet_shapes(enc, dec, out_ch):
    s = {}
    chans = [(1, 16), (16, 32), (32, 64), (64, 128)]
    for i, (ci, co) in enumerate(chans, 1):
        b = f'{enc}.block{i}'
        s[b + '.conv1.weight'] = (co, ci, 3, 3)
        s[b + '.conv2.weight'] = (co, co, 3, 3)
        s[b + '.skip.weight'] = (co, ci, 1, 1)
        s[b + '.ds.weight'] = (co, co, 2, 2)
        for c in ('conv1', 'conv2', 'skip', 'ds'):
            s[f'{b}.{c}.bias'] = (co,)
        for bn in ('bn1', 'bn2'):
            _bn(s, f'{b}.{bn}', co)
    for name, c in (('conv1', 64), ('conv2', 32), ('conv3', 16)):
        s[f'{enc}.{name}.weight'] = (c, c, 3, 3)
        s[f'{enc}.{name}.bias'] = (c,)
    for i, (inp, out, last) in enumerate([(192, 64, False), (96, 32, False), (48, 16, False),
                                          (16, out_ch, True)], 1):
        d = f'{dec}.d_block{i}'
        s[d + '.conv2d.weight'] = (inp, inp // 2, 3, 3)      # ConvTranspose2d: [Cin, Cout, kh, kw]
        s[d + '.conv2d.bias'] = (inp // 2,)
        _bn(s, d + '.bn2d', inp // 2)
        s[d + '.conv1d.weight'] = (inp // 2, out, 3, 3)
        s[d + '.conv1d.bias'] = (out,)
        if not last:
            _bn(s, d + '.bn1d', out)
            u = inp - out
        else:
            u = inp
        s[d + '.us.weight'] = (u, u, 2, 2)
        s[d + '.us.bias'] = (u,)
    return s


def _bn(s, name, c):
    s[name + '.weight'] = (c,)
    s[name + '.bias'] = (c,)
    s[name + '.running_mean'] = (c,)
    s[name + '.running_var'] = (c,)
    s[name + '.num_batches_tracked'] = ()


def _attn(s, name, fin, fout):
    s[name + '.rel'] = (1, fout, 31)
    for w in ('W_k', 'W_q', 'W_v'):
        s[f'{name}.{w}.weight'] = (fout, fin)


def param_shapes(model='onset', reconstruction=True):
    """Ordered {state_dict key: shape} for UNet_Onset ('onset') or UNet ('frame'),
    excluding the four ``spectrogram.*`` buffers."""
    s = {}
    t = 'transcriber'
    if model == 'onset':
        s.update(_unet_shapes(f'{t}.Unet1_encoder', f'{t}.Unet1_decoder', 2))
        _attn(s, f'{t}.lstm1', N_BINS + 88, N_BINS * 4)          # constructed, never used
        s[f'{t}.linear1.weight'] = (88, N_BINS * 4)
        s[f'{t}.linear1.bias'] = (88,)
        for lin in ('linear_onset', 'linear_feature'):
            s[f'{t}.{lin}.weight'] = (88, N_BINS)
            s[f'{t}.{lin}.bias'] = (88,)
        _attn(s, f'{t}.combine_stack.attention', 176, 768)
        s[f'{t}.combine_stack.linear.weight'] = (88, 768)
        s[f'{t}.combine_stack.linear.bias'] = (88,)
    else:
        s.update(_unet_shapes(f'{t}.Unet1_encoder', f'{t}.Unet1_decoder', 1))
        _attn(s, f'{t}.lstm1', N_BINS, N_BINS * 4)
        s[f'{t}.linear1.weight'] = (88, N_BINS * 4)
        s[f'{t}.linear1.bias'] = (88,)
    if reconstruction:
        r = 'reconstructor'
        s.update(_unet_shapes(f'{r}.Unet2_encoder', f'{r}.Unet2_decoder', 1))
        _attn(s, f'{r}.lstm2', 88, N_BINS * 4)
        s[f'{r}.linear2.weight'] = (N_BINS, N_BINS * 4)
        s[f'{r}.linear2.bias'] = (N_BINS,)
    return s


def fixture_params(model='onset', reconstruction=True, with_frontend=True, tag=''):
    """Deterministic parameter set.  Conv/linear weights ~ U(-a, a) with
    a = sqrt(3/fan_in) (unit-gain), biases small, BN affine near (1, 0), BN
    running stats at their constructor defaults, attention ``rel`` bell-shaped
    with std 0.5."""
    out = {}
    for k, shp in param_shapes(model, reconstruction).items():
        name = tag + k
        if k.endswith('num_batches_tracked'):
            out[k] = torch.zeros((), dtype=torch.long)
        elif k.endswith('running_mean'):
            out[k] = torch.zeros(shp)
        elif k.endswith('running_var'):
            out[k] = torch.ones(shp)
        elif k.endswith('.rel'):
            out[k] = hashed_normalish(name, shp, 0.5)
        elif '.bn' in k and k.endswith('.weight'):
            out[k] = hashed(name, shp, 0.2, 1.0)
        elif k.endswith('.bias'):
            out[k] = hashed(name, shp, 0.1)
        else:
            w = shp
            if len(w) == 4:
                # Conv2d [Cout,Cin,kh,kw]: fan_in = Cin*kh*kw ; ConvTranspose2d [Cin,Cout,kh,kw]:
                # each output sums Cin*kh*kw/stride^2 terms -- use Cin*kh*kw for 3x3, Cin for 2x2/s2
                is_t = ('d_block' in k)
                cin = w[0] if is_t else w[1]
                taps = w[2] * w[3] if w[2] == 3 or not is_t else 1
                fan = cin * taps
            else:
                fan = w[1]
            out[k] = hashed(name, shp, float(np.sqrt(3.0 / fan)))
    if with_frontend:
        out.update(fe.frontend_buffers())
    return out


def clone_params(params):
    return {k: v.clone() for k, v in params.items()}


def fixture_audio(b, n, tag='audio'):
    """[b, n] waveform in [-0.5, 0.5): a few decaying partials + noise so the
    log-mel has structure (plain noise gives a nearly flat image)."""
    t = np.arange(n, dtype=np.float64) / 16000.0
    out = np.zeros((b, n), dtype=np.float64)
    for i in range(b):
        for j, f0 in enumerate((110.0, 261.6, 392.0, 1046.5)):
            f = f0 * (1.0 + 0.07 * i)
            out[i] += 0.08 / (j + 1) * np.sin(2 * np.pi * f * t) * np.exp(-1.5 * (t % 0.7))
    noise = _hash_uniform(tag, b * n).reshape(b, n).astype(np.float64) * 0.05
    return torch.from_numpy((out + noise).astype(np.float32))


def fixture_spec(b, t, tag='spec'):
    """[b, 1, t, 229] image in [0,1] that touches both ends like a min-max
    normalised log-mel (smooth ridge pattern + hash noise)."""
    tt = np.arange(t, dtype=np.float64)[:, None]
    ff = np.arange(N_BINS, dtype=np.float64)[None, :]
    out = np.zeros((b, t, N_BINS), dtype=np.float64)
    for i in range(b):
        base = 0.5 + 0.3 * np.sin(0.11 * tt + 0.05 * ff * (i + 1)) * np.cos(0.031 * ff - 0.02 * tt)
        noise = _hash_uniform(f'{tag}{i}', t * N_BINS).reshape(t, N_BINS) * 0.2
        img = base + noise
        img = (img - img.min()) / (img.max() - img.min())
        out[i] = img
    return torch.from_numpy(out.astype(np.float32)).unsqueeze(1)


def fixture_labels(b, t, tag='lab'):
    """(onset, frame) float masks [b, t, 88]: frame ~5 % dense, onset subset."""
    u = _hash_uniform(tag, b * t * 88, 0.0, 1.0).reshape(b, t, 88)
    frame = torch.from_numpy((u > 0.95).astype(np.float32))
    onset = torch.from_numpy((u > 0.99).astype(np.float32))
    return onset, frame


def fixture_noise(shape, tag='d0'):
    """Bell-shaped stand-in for torch.randn_like (std 1, bounded)."""
    return hashed_normalish(tag, shape, 1.0)


# ---- the multi-step trajectory fixture (tests/golden/trajectory.npz, generator g_trajectory of tests/golden/make_golden.py) ----------
# K iterations of the reference's train_VAT_model (model/helper_functions.py:570-615): Adam(lr), StepLR(step_size, gamma), post-step
# clip, `n_l` labelled / `n_ul` unlabelled batches cycled out of phase, one injected VAT noise pair per iteration (mode `radv`).
TRAJ = dict(K=6, B=2, T=64, N=128, lr=1e-3, step_size=2, gamma=0.7, clip=3.0, n_l=3, n_ul=2)


def trajectory_inputs(mode=None):
    """The closed-form inputs of the trajectory fixture (shared by the generator, the CPU oracle test and the GPU test)."""
    c = TRAJ

    def batch(tag):
        onset, frame = fixture_labels(c['B'], c['T'], tag)
        return {'audio': fixture_audio(c['B'], c['T'] * 512, tag), 'onset': onset, 'frame': frame}
    lbs = [batch(f'traj_L{i}') for i in range(c['n_l'])]
    ubs = [batch(f'traj_UL{i}') for i in range(c['n_ul'])]
    noises = [(fixture_noise((c['B'], 1, c['T'], 229), f'traj_ul_{i}'), fixture_noise((c['B'], 1, c['T'], 229), f'traj_l_{i}'))
              for i in range(c['K'])]
    return lbs, ubs, noises
