"""Does running two independent transcriber chains on two HIP streams beat running them back to back?
(forward + input-gradient backward of the VAT power iteration, B=8 each, captured in hipGraphs)
Round 5: ... and what would ONE chain on the concatenated batch of 16 cost (`merged16`: the unlabelled and the labelled VAT branch of a step
in lock step -- the COST of that schedule; its BatchNorm statistics would have to stay per group of 8, which this probe does not do)?
`FULL=1`: the grad-enabled final pass with parameter gradients instead of the detached power-iteration pass."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import reconvat_amd as ra
from reconvat_amd import ops

dev = torch.device('cuda:0')
torch.manual_seed(0)
m = ra.UNet_Onset((2, 2), (2, 2), log=True, reconstruction=True, mode='imagewise', spec='Mel', device=str(dev), XI=1e-6, eps=2).to(dev)
m.train()
NCH = int(sys.argv[1]) if len(sys.argv) > 1 else 2
xs = [torch.rand(8, 1, 640, 229, device=dev) for _ in range(NCH)]


def chain(x):
    with ops.deferred_bn_updates():
        with torch.no_grad():
            m.transcriber(x)
        d = torch.randn_like(x).requires_grad_(True)
        xa = ops.VatPerturbFn.apply(x, d, 1e-6)
        roll, onset, _ = m.transcriber(xa, True)
        g, = torch.autograd.grad(roll.sum() + onset.sum(), d)
    return g


FULL = os.environ.get('FULL') == '1'
if FULL:
    opt = ra.FlatAdam(m.parameters(), lr=0.0)

    def chain(x):                                   # the final VAT pass: grad-enabled forward, full backward into the flat bucket
        with ops.deferred_bn_updates(), ops.direct_param_grads():
            roll, onset, _ = m.transcriber(x)
            (roll.sum() + onset.sum()).backward()
        return opt.flat_grad


def merged16():
    return [chain(torch.cat(xs, 0))] if not FULL else [chain(x16)]


x16 = torch.cat(xs, 0)


def both_serial():
    return [chain(x) for x in xs]


sides = [torch.cuda.Stream() for _ in range(NCH - 1)]


def both_parallel():
    cur = torch.cuda.current_stream()
    outs = []
    for sd, x in zip(sides, xs[1:]):
        sd.wait_stream(cur)
        with torch.cuda.stream(sd):
            outs.append(chain(x))
    outs.append(chain(xs[0]))
    for sd in sides:
        cur.wait_stream(sd)
    return outs


for fn in (both_serial, both_parallel, merged16):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = fn()
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        g.replay()
    e1.record(); e1.synchronize()
    print(f'{fn.__name__}: {e0.elapsed_time(e1) / 10:.3f} ms')

# ---- round 6: the same two chains as TWO single-branch graphs replayed concurrently on two streams (instead of one forked graph).
# tools/pair_probe.py found kernel pairs that hide each other completely when launched on two streams but not at all as the two branches of a
# forked hipGraph; this is the same question for whole chains.  TWO_GRAPHS=0 skips it.
if os.environ.get('TWO_GRAPHS', '1') != '0' and NCH == 2:
    graphs = []
    for x in xs:
        for _ in range(2):
            chain(x)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            out = chain(x)
        graphs.append((g, out))
    for trial in range(3):
        s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
        ts = []
        for rep in range(12):
            torch.cuda.synchronize()
            e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
            e0.record(torch.cuda.current_stream())
            s1.wait_event(e0); s2.wait_event(e0)
            with torch.cuda.stream(s1):
                graphs[0][0].replay(); e1.record(s1)
            with torch.cuda.stream(s2):
                graphs[1][0].replay(); e2.record(s2)
            torch.cuda.synchronize()
            ts.append(max(e0.elapsed_time(e1), e0.elapsed_time(e2)))
        ts = sorted(ts[2:])
        print(f'two graphs on two streams (stream pair {trial}): median {ts[len(ts) // 2]:.3f} ms  min {ts[0]:.3f}')
