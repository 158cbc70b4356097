"""List every rv_gemm launch of one training step (shape, strides class, split-K) with its re-timed duration."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import reconvat_amd as ra
from reconvat_amd import _lib, ops
import bench

dev = torch.device('cuda:0')
torch.manual_seed(0)
model = ra.UNet_Onset((2, 2), (2, 2), log=True, reconstruction=True, mode='imagewise', spec='Mel', device=str(dev),
                      XI=1e-6, eps=2).to(dev)
opt = ra.FlatAdam(model.parameters(), lr=1e-3, step_size=1000, gamma=0.98)
gen = torch.Generator().manual_seed(1)
b, bu = bench.synthetic_batch(8, gen, dev), bench.synthetic_batch(8, gen, dev)
step = ra.TrainStep(model, opt, b, bu, alpha=1.0, VAT=True, clip=3.0, graph=False)
step(); step()
recs = []
real = _lib.call


def spy(name, *a):
    if name == 'rv_gemm':
        recs.append(a)
    return real(name, *a)


ops.call = spy
step()
torch.cuda.synchronize()
ops.call = real
lib = _lib.load()
groups = {}
for a in recs:
    # (A, sam, sak, B, sbk, sbn, C, scm, scn, C2, s2m, s2n, bias, M, N, K, act, acc, splitk, batch, bsa, bsb, bsc, stream)
    sig = (a[13], a[14], a[15], 'Ak' if a[2] <= a[1] else 'Am', 'Bk' if a[4] <= a[5] else 'Bn', a[16], a[17], a[18], a[1], a[2], a[4], a[5], a[7], a[8], a[19])
    g = groups.setdefault(sig, [0, a])
    g[0] += 1
rows = []
cur = torch.cuda.current_stream().cuda_stream
for sig, (cnt, a) in groups.items():
    a = list(a)
    a[-1] = cur                # re-time on the stream the events are recorded on (the step runs some GEMMs on a side stream)
    for _ in range(2):
        lib.rv_gemm(*a)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        lib.rv_gemm(*a)
    e1.record()
    e1.synchronize()
    ms = e0.elapsed_time(e1) / 5
    fl = 2.0 * sig[0] * sig[1] * sig[2] * sig[-1]
    rows.append((ms * cnt, cnt, ms, fl / ms / 1e9, sig))
rows.sort(reverse=True)
tot = sum(r[0] for r in rows)
print(f'{len(recs)} gemm launches/step, {tot:.3f} ms/step')
for r in rows:
    print(f'{r[0]:7.3f} ms/step x{r[1]:3d} {r[2] * 1e3:7.1f} us {r[3]:6.1f} TF/s  M,N,K={r[4][:3]} {r[4][3]}{r[4][4]} act={r[4][5]} acc={r[4][6]} splitk={r[4][7]} batch={r[4][-1]} strides={r[4][8:-1]}')
