"""The SHIPPED kernel configuration (reconvat_amd/tuned_plans.json) under test, at the shipped size.

* every conv entry of the table -- (launch shape -> tile) -- is launched exactly as keyed (B = 8, the BASELINE layer shapes, the
  keyed pixel strides, with / without the fused BatchNorm statistics or backward reduction) and compared with a plain PyTorch
  fp32 CPU convolution of the same operands; every weight-gradient entry -- (shape -> waves x workgroups) -- likewise against
  torch's weight gradient;
* one full BASELINE-size training step (UNet_Onset, VAT + reconstruction, B_l = B_ul = 8) must launch ONLY tiles that are
  table entries (exact key hits): the tiles behind the bench line are the tiles verified above.
"""
import pytest
import torch
import torch.nn.functional as F

from conftest import rel_err

pytestmark = pytest.mark.gpu
UPW = {14: 28, 28: 57, 57: 114, 114: 229, 7: 14}
SLOPE = 0.01


def _rand(*shape, seed):
    g = torch.Generator(device='cpu').manual_seed(seed)
    return torch.rand(*shape, generator=g) * 2 - 1


def _conv_cases():
    from reconvat_amd import plans
    return sorted(plans.conv_entries().items())


def _wgrad_cases():
    from reconvat_amd import plans
    return sorted(plans.wgrad_entries().items())


def _ids(cases):
    return ['-'.join(str(int(x)) for x in k) for k, _ in cases]


def test_table_is_loaded_and_default(dev):
    from reconvat_amd import ops, plans
    assert ops.AUTOTUNE == 'table'
    assert plans.digest() is not None and len(plans.conv_entries()) >= 40 and len(plans.wgrad_entries()) >= 20


@pytest.mark.parametrize('key,algo', _conv_cases(), ids=_ids(_conv_cases()))
def test_table_conv_entry_vs_torch(dev, key, algo):
    from reconvat_amd import ops, plans
    mode, bb, h, w, cin, cout, ild, old, stats, bnbwd = key
    if mode in (0, 1):
        ho, wo = h, w
    elif mode == 2:
        ho, wo = h // 2, w // 2
    else:
        ho, wo = 2 * h, UPW[w]
    xs = _rand(bb, h, w, cin, seed=1)
    kind = {0: 'c3', 1: 'c1', 2: 'down', 3: 'up'}[mode]
    wshape = {'c3': (cout, cin, 3, 3), 'c1': (cout, cin, 1, 1), 'down': (cout, cin, 2, 2), 'up': (cin, cout, 2, 2)}[kind]
    wt = _rand(*wshape, seed=2) * (1.0 / (cin * wshape[2] * wshape[3]) ** 0.5)
    bias = _rand(cout, seed=3)
    xn = xs.permute(0, 3, 1, 2)
    if kind == 'c3':
        ref = F.conv2d(xn, wt, bias, padding=1)
    elif kind == 'c1':
        ref = F.conv2d(xn, wt, bias)
    elif kind == 'down':
        ref = F.conv2d(xn, wt, bias, stride=2)
    else:
        ref = F.conv_transpose2d(xn, wt, bias, stride=2, output_padding=(ho - 2 * h, wo - 2 * w))
    ref = ref.permute(0, 2, 3, 1).contiguous()
    assert tuple(ref.shape) == (bb, ho, wo, cout)
    xbuf = torch.zeros(bb, h, w, ild, device=dev)
    xbuf[..., :cin] = xs.to(dev)
    obuf = torch.full((bb, ho, wo, old), 7.0, device=dev)
    x, out = xbuf[..., :cin], obuf[..., :cout]
    wd, bd = wt.to(dev), bias.to(dev)
    ws = torch.zeros(ops.bn_ws_doubles(cout), device=dev, dtype=torch.float64) if stats else None
    link = None
    if bnbwd:
        z = _rand(bb, ho, wo, cout, seed=4).to(dev)
        mean, invstd = _rand(cout, seed=5) * 0.1, _rand(cout, seed=6).abs() + 0.5
        scale, shift = _rand(cout, seed=7), _rand(cout, seed=8) * 0.2
        coef = torch.cat([mean, invstd, scale, shift, torch.ones(cout)]).to(dev)
        link = (z, coef, SLOPE)
    assert ops.AUTOTUNE == 'table'
    ops._algo_cache.clear()
    ops._conv_call(mode, x, ild, bb, h, w, cin, out, old, ho, wo, cout, ops._pack(kind, wd, 'fwd'), bd, ws, link)
    torch.cuda.synchronize()
    assert key in plans.HITS['conv'], 'the launch did not resolve to this table entry'
    ckey = (mode, bb, h, w, cin, cout, ild, old, bool(stats), bool(bnbwd))
    assert ops._algo_cache.get(ckey) == algo, (ops._algo_cache, algo)
    assert rel_err(out, ref) < 1e-4
    if old > cout:
        assert float((obuf[..., cout:] - 7.0).abs().max()) == 0.0          # neighbouring channels of the wider buffer untouched
    if stats:
        fold = ws.view(-1, 2, cout).sum(0).cpu()
        p = bb * ho * wo
        if bnbwd:
            zc, rd = z.cpu().double(), ref.double()
            dd = rd * torch.where(zc * scale.double() + shift.double() > 0, 1.0, SLOPE)
            want0 = dd.reshape(-1, cout).sum(0)
            want1 = (dd * (zc - mean.double()) * invstd.double()).reshape(-1, cout).sum(0)
            scale_ = p * float(dd.pow(2).mean().sqrt())
            assert float((fold[0] - want0).abs().max()) < 2e-5 * scale_ and float((fold[1] - want1).abs().max()) < 2e-5 * scale_
        else:
            rd = ref.double().reshape(-1, cout)
            rms = float(rd.pow(2).mean().sqrt())
            assert float((fold[0] - rd.sum(0)).abs().max()) < 1e-5 * p * rms
            assert float((fold[1] - rd.pow(2).sum(0)).abs().max()) < 1e-5 * p * rms * rms


@pytest.mark.parametrize('key,plan', _wgrad_cases(), ids=_ids(_wgrad_cases()))
def test_table_wgrad_entry_vs_torch(dev, key, plan):
    from reconvat_amd import ops, plans
    taps, bb, hv, wv, ca, cb = key
    kind = {9: 'c3', 1: 'c1', 4: 'down'}[taps]
    if kind == 'down':
        hu, wu = 2 * hv, UPW[wv]
    else:
        hu, wu = hv, wv
    xs, dys = _rand(bb, hu, wu, ca, seed=11), _rand(bb, hv, wv, cb, seed=12)
    k = 3 if taps == 9 else (2 if taps == 4 else 1)
    wt = torch.zeros(cb, ca, k, k, requires_grad=True)
    xn = xs.permute(0, 3, 1, 2)
    y = F.conv2d(xn, wt, torch.zeros(cb), padding=1) if kind == 'c3' else F.conv2d(xn, wt, None, stride=2 if kind == 'down' else 1)
    (y * dys.permute(0, 3, 1, 2)).sum().backward()
    ops._wgrad_tuned.clear()
    assert ops.AUTOTUNE == 'table'
    dw, db = ops.conv_wgrad(kind, xs.to(dev), dys.to(dev), wt.detach().to(dev), True)
    torch.cuda.synchronize()
    assert key in plans.HITS['wgrad'] and ops._wgrad_plans.get(key) == plan
    assert rel_err(dw, wt.grad) < 2e-4
    assert rel_err(db, dys.reshape(-1, cb).double().sum(0).float()) < 1e-4


@pytest.mark.parametrize('cls', ['UNet_Onset', 'UNet'])
def test_baseline_step_runs_only_table_tiles(dev, cls):
    """One BASELINE-size optimiser step: every tunable conv launch and every tunable weight gradient resolves to an EXACT entry
    of the shipped table (no library-default fall-back, no borrowed entry), so bench.py runs exactly the tiles tested above."""
    import reconvat_amd as ra
    from reconvat_amd import ops, plans
    g = torch.Generator().manual_seed(1)

    def batch():
        u = torch.rand(8, 640, 88, generator=g)
        return {'audio': (torch.rand(8, 327680, generator=g) * 0.2 - 0.1).to(dev), 'frame': (u > 0.95).float().to(dev),
                'onset': (u > 0.99).float().to(dev)}
    torch.manual_seed(5)
    m = getattr(ra, cls)((2, 2), (2, 2), log=True, reconstruction=True, mode='imagewise', spec='Mel', XI=1e-6, eps=2).to(dev)
    opt = ra.FlatAdam(m.parameters(), lr=1e-3)
    ops._algo_cache.clear()
    ops._wgrad_tuned.clear()
    ops._wgrad_plans.clear()
    step = ra.TrainStep(m, opt, batch(), batch(), graph=False, dual_stream=False)
    step()
    torch.cuda.synchronize()
    step.check()
    conv, wgrad = plans.conv_entries(), plans.wgrad_entries()
    assert len(ops._algo_cache) >= 30
    missing = [k for k in ops._algo_cache if tuple(int(x) for x in k) not in conv]
    assert not missing, missing
    assert all(ops._algo_cache[k] == conv[tuple(int(x) for x in k)] for k in ops._algo_cache)
    assert len(ops._wgrad_plans) >= 15 and all(k in wgrad for k in ops._wgrad_plans)
    # ... and every tunable weight-gradient shape of the step did find a plan
    assert len(ops._wgrad_plans) == len(ops._wgrad_tuned), (len(ops._wgrad_plans), len(ops._wgrad_tuned))
    assert torch.isfinite(step.loss).item()
