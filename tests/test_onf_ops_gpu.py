"""GPU parity of the Onsets&Frames baseline operators (BiLSTM recurrence, MaxPool(1,2)+Dropout, Dropout, the 1->48 /
48->96 ConvStack convolutions) against plain PyTorch fp32 CPU references of the same ops
(model/onset_frame_VAT.py:321-415,614)."""
import pytest
import torch
import torch.nn.functional as F

from conftest import rel_err

pytestmark = pytest.mark.gpu

TOL = 3e-5
TOL_G = 2e-4


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed + sum(shape))
    return (torch.rand(*shape, generator=g) * 2 - 1) * scale


def lstm_params(i, h, seed):
    k = 1.0 / h ** 0.5
    shapes = [(4 * h, i), (4 * h, h), (4 * h,), (4 * h,)] * 2
    return [rnd(*s, seed=seed + n, scale=k) for n, s in enumerate(shapes)]


@pytest.mark.parametrize('b,t,i,h', [(2, 5, 24, 32), (3, 17, 40, 32), (8, 33, 176, 384), (1, 1, 16, 32), (8, 640, 768, 384), (16, 9, 24, 32), (13, 21, 176, 384)])
def test_bilstm_fwd_bwd(dev, b, t, i, h):
    from reconvat_amd import ops
    x = rnd(b, t, i, seed=1)
    ps = lstm_params(i, h, 7)
    ref = torch.nn.LSTM(i, h, batch_first=True, bidirectional=True)
    names = ['weight_ih_l0', 'weight_hh_l0', 'bias_ih_l0', 'bias_hh_l0']
    with torch.no_grad():
        for n, p in zip(names + [s + '_reverse' for s in names], ps):
            getattr(ref, n).copy_(p)
    xr = x.clone().requires_grad_(True)
    yr, _ = ref(xr)
    gy = rnd(b, t, 2 * h, seed=3)
    yr.backward(gy)

    xg = x.to(dev).requires_grad_(True)
    pg = [p.to(dev).requires_grad_(True) for p in ps]
    y = ops.BiLstmFn.apply(xg, *pg)
    torch.cuda.synchronize()
    assert rel_err(y, yr) < TOL
    y.backward(gy.to(dev))
    torch.cuda.synchronize()
    assert rel_err(xg.grad, xr.grad) < TOL_G
    for n, p in zip(names + [s + '_reverse' for s in names], pg):
        assert rel_err(p.grad, getattr(ref, n).grad) < TOL_G, n
    # inference path (no saved state) gives the same output
    with torch.no_grad():
        y2 = ops.BiLstmFn.apply(xg.detach(), *[p.detach() for p in pg])
    assert torch.equal(y2, y.detach())
    ops.lstm_check(dev)          # no workgroup timed out


def test_bilstm_rejects_unsupported(dev):
    from reconvat_amd import ops
    ps = [p.to(dev) for p in lstm_params(8, 32, 1)]
    with pytest.raises(RuntimeError, match='batch'):
        ops.BiLstmFn.apply(torch.zeros(17, 4, 8, device=dev), *ps)
    ps = [p.to(dev) for p in lstm_params(8, 48, 1)]
    with pytest.raises(RuntimeError, match='hidden size'):
        ops.BiLstmFn.apply(torch.zeros(2, 4, 8, device=dev), *ps)


@pytest.mark.parametrize('shape', [(2, 7, 229, 48), (1, 3, 114, 96), (2, 4, 2, 5)])
def test_maxpool_w2(dev, shape):
    from reconvat_amd import ops
    x = rnd(*shape, seed=2)
    xr = x.clone().requires_grad_(True)
    yr = F.max_pool2d(xr.permute(0, 3, 1, 2), (1, 2)).permute(0, 2, 3, 1)
    gy = rnd(*yr.shape, seed=5)
    yr.backward(gy)
    xg = x.to(dev).requires_grad_(True)
    y = ops.PoolDropFn.apply(xg, 0.25, False)
    assert torch.equal(y.cpu(), yr.detach().contiguous())
    y.backward(gy.to(dev))
    assert torch.equal(xg.grad.cpu(), xr.grad)


def test_pool_dropout_statistics(dev):
    from reconvat_amd import ops
    ops.seed_dropout(11)
    x = (torch.rand(4, 64, 229, 48, device=dev) + 0.5).requires_grad_(True)
    y = ops.PoolDropFn.apply(x, 0.25, True)
    pooled = F.max_pool2d(x.detach().permute(0, 3, 1, 2), (1, 2)).permute(0, 2, 3, 1)
    kept = y != 0
    frac = kept.float().mean().item()
    assert abs(frac - 0.75) < 2e-3
    assert torch.allclose(y[kept], pooled[kept] / 0.75, rtol=1e-6)
    y.backward(torch.ones_like(y))
    # every kept output sends 1/(1-p) to exactly one of its two inputs, dropped ones send nothing
    assert abs(x.grad.sum().item() - kept.sum().item() / 0.75) < 1e-3 * kept.sum().item()
    # a second draw uses a different mask
    y2 = ops.PoolDropFn.apply(x, 0.25, True)
    assert (y2 != 0).ne(kept).any()
    # no spatial structure: per-channel and per-column keep rates are all near 0.75
    assert (kept.float().mean(dim=(0, 1, 2)) - 0.75).abs().max().item() < 0.02
    assert (kept.float().mean(dim=(0, 1, 3)) - 0.75).abs().max().item() < 0.02


def test_dropout(dev):
    from reconvat_amd import ops
    ops.seed_dropout(5)
    x = (torch.rand(4096, 768, device=dev) + 0.5).requires_grad_(True)
    y = ops.dropout(x, 0.5, True)
    kept = y != 0
    assert abs(kept.float().mean().item() - 0.5) < 2e-3
    assert torch.allclose(y[kept], x.detach()[kept] * 2.0)
    g = torch.rand_like(y)
    y.backward(g)
    assert torch.equal(x.grad, torch.where(kept, g * 2.0, torch.zeros_like(g)))
    assert ops.dropout(x, 0.5, False) is x


@pytest.mark.parametrize('cin,cout,h,w', [(1, 48, 9, 229), (48, 48, 9, 229), (48, 96, 6, 114)])
def test_convstack_convs(dev, cin, cout, h, w):
    from reconvat_amd import ops
    x = rnd(2, cin, h, w, seed=1)
    wt = rnd(cout, cin, 3, 3, seed=2, scale=0.2)
    bs = rnd(cout, seed=3)
    xr, wr, br = x.clone().requires_grad_(True), wt.clone().requires_grad_(True), bs.clone().requires_grad_(True)
    yr = F.conv2d(xr, wr, br, padding=1)
    gy = rnd(*yr.shape, seed=4)
    yr.backward(gy)
    xg = x.permute(0, 2, 3, 1).contiguous().to(dev).requires_grad_(True)
    wg, bg = wt.to(dev).requires_grad_(True), bs.to(dev).requires_grad_(True)
    y = ops.ConvFn.apply(xg, wg, bg, 'c3', None)
    assert rel_err(y.permute(0, 3, 1, 2), yr) < TOL
    y.backward(gy.permute(0, 2, 3, 1).contiguous().to(dev))
    assert rel_err(xg.grad.permute(0, 3, 1, 2), xr.grad) < TOL_G
    assert rel_err(wg.grad, wr.grad) < TOL_G
    assert rel_err(bg.grad, br.grad) < TOL_G
