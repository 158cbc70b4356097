"""Race hunt for the shipped step schedule.

The whole step -- front-end, both VAT power iterations (forward + input-gradient chain), the five final graphs, every loss term,
every BatchNorm running-statistic update -- is a DETERMINISTIC function of (weights, batch, injected noise) in this build: no
fp32 atomics anywhere in the forward / input-gradient chains (split-K GEMMs fold their slices in a fixed order, loss reductions
fold in a fixed order; the fp64 BatchNorm sums are the one order-dependent accumulation and are exact to ~1e-16).  So the
two-stream hipGraph schedule -- side stream, twin gradient bucket, arena slices, deferred BatchNorm table, captured pinned
tables -- can be checked for races EXACTLY: N replays with frozen weights must reproduce, bit for bit, the loss terms (all
eleven, the chaos-amplifying VAT terms included) and the running statistics of N single-stream eager steps.  A stale buffer, a
missing stream dependency or a reused arena slice shows up as a mismatch in at least one replay."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _mk(dev, b, t, tag):
    from oracle import fixture as fx
    onset, frame = fx.fixture_labels(b, t, tag)
    return {'audio': fx.fixture_audio(b, t * 512, tag).to(dev), 'onset': onset.to(dev), 'frame': frame.to(dev)}


def _run(dev, kind, t, graph, dual, steps, jitter=0):
    import reconvat_amd as ra
    from oracle import fixture as fx
    from test_model_gpu import build
    bl, bul = _mk(dev, 2, t, 'L'), _mk(dev, 2, t, 'UL')
    noise = [fx.fixture_noise((2, 1, t, 229), 'd0_ul').to(dev), fx.fixture_noise((2, 1, t, 229), 'd0_l').to(dev)]
    m = build(kind, True, dev)
    opt = ra.FlatAdam(m.parameters(), lr=0.0)                # frozen weights
    state = {'i': 0}

    def draw(x):
        state['i'] += 1
        return noise[(state['i'] - 1) % 2].clone()
    m.vat_loss.noise = draw
    step = ra.TrainStep(m, opt, bl, bul, alpha=1.0, VAT=True, clip=3.0, graph=graph, dual_stream=dual)
    restore = _install_jitter(jitter) if jitter else (lambda: None)
    if not graph:
        # eager: the first call packs weights single-stream; make the noise parity the same as for the captured step (whose
        # warm-up passes leave no trace): run it on a throw-away copy of the BatchNorm state
        saved = [b_.clone() for b_ in step._bn_state()]
        step._fwd_bwd()
        step._dual_ready = True
        for b_, v in zip(step._bn_state(), saved):
            b_.copy_(v)
        state['i'] = 0
    out = []
    for _ in range(steps):
        if graph and step.graph is None:
            step.capture()
            state['i'] = 0
        step()
        torch.cuda.synchronize()
        stats = torch.cat([v.detach().float().flatten() for k, v in sorted(m.state_dict().items())
                           if k.endswith(('running_mean', 'running_var', 'num_batches_tracked'))])
        out.append(({k: float(v) for k, v in step.losses.items()}, stats.clone(), opt.flat_grad.clone()))
    step.check()
    restore()
    return out


def _install_jitter(seed):
    """Perturb the interleaving of the two chains: a spin kernel of pseudo-random length (0 .. ~150 us) is queued on whichever
    stream is current at the start of every transcriber / reconstructor pass section (ops.deferred_bn_updates.__enter__ is
    entered once per pass, on that pass's stream) -- eager launches and captured graph nodes alike.  A dependency that only
    holds by timing luck breaks under some seed."""
    import random
    from reconvat_amd import ops
    rng = random.Random(seed)
    real = ops.deferred_bn_updates.__enter__

    def enter(self):
        torch.cuda._sleep(rng.randrange(0, 300000))
        return real(self)
    ops.deferred_bn_updates.__enter__ = enter

    def restore():
        ops.deferred_bn_updates.__enter__ = real
    return restore


@pytest.mark.parametrize('kind,t,steps', [('onset', 64, 50), ('onset', 640, 12), ('frame', 640, 6)])
def test_two_stream_graph_replays_bit_identical_to_single_stream_eager(dev, kind, t, steps):
    ref = _run(dev, kind, t, graph=False, dual=False, steps=steps)
    got = _run(dev, kind, t, graph=True, dual=True, steps=steps)
    eag = _run(dev, kind, t, graph=False, dual=True, steps=min(steps, 6))
    for name, runs in (('two-stream hipGraph', got), ('two-stream eager', eag)):
        for i, ((lr, sr, gr), (lg, sg, gg)) in enumerate(zip(ref, runs)):
            bad = {k: (lr[k], lg[k]) for k in lr if lr[k] != lg[k]}
            assert not bad, (name, 'step', i, bad)
            assert torch.equal(sr, sg), (name, 'running statistics differ at step', i, float((sr - sg).abs().max()))
            # parameter gradients: the per-layer weight-gradient partial sums are folded by fp32 atomics (several passes add to
            # the same layer), so these agree to summation-order noise only -- a stale twin bucket would be O(1) off
            den = float(gr.abs().max())
            assert float((gr - gg).abs().max()) <= 2e-4 * den, (name, i)
    # and the steps differ from each other only through the running statistics: loss terms are the same every step
    assert all(ref[0][0] == r[0] for r in ref[1:])


@pytest.mark.parametrize('seed', [1, 2, 3])
def test_two_stream_schedule_is_timing_independent(dev, seed):
    """The same exact comparison with the relative timing of the two chains perturbed by random spin kernels (captured into the
    graph as nodes, so every replay runs the perturbed interleaving)."""
    ref = _run(dev, 'onset', 64, graph=False, dual=False, steps=4)
    for graph in (False, True):
        got = _run(dev, 'onset', 64, graph=graph, dual=True, steps=4, jitter=seed)
        for i, ((lr, sr, _), (lg, sg, _)) in enumerate(zip(ref, got)):
            assert lr == lg, (graph, seed, i, {k: (lr[k], lg[k]) for k in lr if lr[k] != lg[k]})
            assert torch.equal(sr, sg), (graph, seed, i)
