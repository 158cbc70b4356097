"""CPU tests of the host-side logic around the hot path: CLI parsing (sacred-compatible `with k=v`), config
scope, loss weighting, dataset crop rule, state_dict surface, data-parallel bucket (gloo, world size 2)."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_cli_parsing_and_config_scope():
    from reconvat_amd.sacred_lite import parse_cli, Experiment, ConfigError
    from reconvat_amd.cli import base_config
    o = parse_cli(['with', 'train_on=MAPS', 'small=True', 'supersmall=True', 'VAT=False', 'reconstruction=False',
                   'device=cpu', 'XI=1e-6', 'eps=2', 'root=my_runs'])
    assert o == {'train_on': 'MAPS', 'small': True, 'supersmall': True, 'VAT': False, 'reconstruction': False,
                 'device': 'cpu', 'XI': 1e-6, 'eps': 2, 'root': 'my_runs'}
    for onset in (True, False):                      # BASELINE.json config 1 uses supersmall on train_UNet_VAT.py
        ex = Experiment('t')
        ex.config(lambda ov, onset=onset: base_config(ov, onset))
        cfg = ex.build_config(o)
        assert cfg['batch_size'] == 8 and cfg['sequence_length'] == 327680 and cfg['learning_rate'] == 1e-3
        assert cfg['train_batch_size'] == (8 if onset else 1)
        assert cfg['logdir'].startswith('my_runs/Unet')          # derived entry sees the override
        assert cfg['validation_length'] == cfg['sequence_length']
        with pytest.raises(ConfigError):
            ex.build_config({'no_such_key': 1})
    assert base_config({}, True)['train_on'] == 'MAPS' and base_config({}, False)['train_on'] == 'Wind'


def test_weighted_loss_rule():
    from reconvat_amd.train import weighted_loss
    losses = {'loss/train_frame': torch.tensor(1.0), 'loss/train_LDS_l_frame': torch.tensor(2.0),
              'loss/train_LDS_ul_onset': torch.tensor(4.0), 'loss/train_r_norm_l': torch.tensor(0.5)}
    assert float(weighted_loss(losses, alpha=1)) == 1.0 + 1.0 + 2.0 + 0.5
    assert float(weighted_loss(losses, alpha=3)) == 1.0 + 3.0 + 6.0 + 0.5


def test_dataset_crop_rule(tmp_path):
    """PianoRollAudioDataset.__getitem__ (reference model/dataset.py:35-69): RandomState(42) crop, label decode."""
    from reconvat_amd.dataset import CachedFolder
    rng = np.random.RandomState(0)
    audio = torch.from_numpy(rng.randint(-2000, 2000, size=60000).astype(np.int16))
    label = torch.from_numpy(rng.randint(0, 4, size=(60000 // 512 + 1, 88)).astype(np.uint8))
    os.makedirs(tmp_path / 'g')
    torch.save(dict(path='x.flac', audio=audio, label=label, velocity=label.clone()), tmp_path / 'g' / 'x.pt')
    ds = CachedFolder(str(tmp_path), ('g',), sequence_length=16384, seed=42)
    item = ds[0]
    step_begin = np.random.RandomState(42).randint(60000 - 16384) // 512
    assert item['start_idx'] == step_begin * 512
    assert item['audio'].shape == (16384,) and item['frame'].shape == (32, 88)
    assert torch.equal(item['audio'], audio[step_begin * 512: step_begin * 512 + 16384].float() / 32768.0)
    lab = label[step_begin:step_begin + 32]
    assert torch.equal(item['onset'], (lab == 3).float()) and torch.equal(item['frame'], (lab > 1).float())
    assert torch.equal(item['offset'], (lab == 1).float())


def test_state_dict_surface_matches_reference_keys():
    import reconvat_amd as ra
    from oracle import fixture as fx
    for kind, cls in (('onset', ra.UNet_Onset), ('frame', ra.UNet)):
        for recon in (True, False):
            m = cls((2, 2), (2, 2), log=True, reconstruction=recon, mode='imagewise', spec='Mel', XI=1e-6, eps=2)
            sd = {k: tuple(v.shape) for k, v in m.state_dict().items()}
            want = dict(fx.param_shapes(kind, recon))
            want.update({'spectrogram.mel_basis': (229, 1025), 'spectrogram.stft.wsin': (1025, 1, 2048),
                         'spectrogram.stft.wcos': (1025, 1, 2048), 'spectrogram.stft.window_mask': (1, 2048, 1)})
            assert sd == want
            m.load_state_dict(fx.fixture_params(kind, recon))          # strict
    m = ra.UNet_Onset((2, 2), (2, 2), log=True, reconstruction=True, mode='imagewise', spec='Mel')
    assert sum(p.numel() for p in m.parameters()) == 3639256         # flat gradient bucket size


def test_baseline_script_config_and_state_dict_surface():
    """train_baseline_onset_frame_VAT.py: config scope of the reference script (:25-73) and the state_dict surface of
    OnsetsAndFrames_VAT_full (constructed on CPU; its forward is GPU-only)."""
    from reconvat_amd.sacred_lite import parse_cli, Experiment
    from reconvat_amd.cli import baseline_config
    import reconvat_amd as ra
    from oracle import onset_frames as oo
    ex = Experiment('t')
    ex.config(baseline_config)
    cfg = ex.build_config(parse_cli(['with', 'VAT=True', 'root=my_runs']))
    assert cfg['learning_rate'] == 5e-4 and cfg['learning_rate_decay_steps'] == 10000 and cfg['XI'] == 1e-6 and cfg['eps'] == 1e-1
    assert cfg['model_complexity'] == 48 and cfg['train_on'] == 'String' and cfg['small'] is True and cfg['VAT'] is True
    assert cfg['batch_size'] == cfg['train_batch_size'] == 8 and cfg['sequence_length'] == cfg['validation_length'] == 327680
    assert cfg['logdir'].startswith('my_runs/baseline_Onset_Frame-')
    with pytest.raises(NotImplementedError):
        baseline_config({'model_name': 'attention'})
    m = ra.OnsetsAndFrames_VAT_full(229, 88, model_complexity=48, log=True, mode='imagewise', spec='Mel', XI=1e-6, eps=1e-1)
    sd = {k: tuple(v.shape) for k, v in m.state_dict().items()}
    want = dict(oo.param_shapes())
    want.update({'spectrogram.mel_basis': (229, 1025), 'spectrogram.stft.wsin': (1025, 1, 2048),
                 'spectrogram.stft.wcos': (1025, 1, 2048), 'spectrogram.stft.window_mask': (1, 2048, 1)})
    assert sd == want
    m.load_state_dict(oo.fixture_params())                             # strict
    with pytest.raises(RuntimeError, match='HIP device only'):
        m(torch.zeros(1, 4, 229))                                      # no CPU fallback


DP_WORKER = r'''
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1])
from reconvat_amd.train import allreduce_gradients
dist.init_process_group('gloo', init_method='tcp://127.0.0.1:' + sys.argv[2], rank=int(sys.argv[3]), world_size=2)
class Bucket:                      # the flat-bucket contract FlatAdam exposes
    pass
b = Bucket(); rank = dist.get_rank()
b.flat_grad = torch.arange(10, dtype=torch.float32) * (rank + 1)
b.grad_scale = 1.0
allreduce_gradients(b)
assert torch.equal(b.flat_grad, torch.arange(10, dtype=torch.float32) * 3), b.flat_grad
assert b.grad_scale == 0.5
dist.barrier(); dist.destroy_process_group()
print('ok', rank)
'''


def test_data_parallel_bucket_gloo(tmp_path):
    """World size 2 over gloo on CPU: ONE all-reduce(sum) of the flat bucket, mean folded into grad_scale."""
    script = tmp_path / 'w.py'
    script.write_text(DP_WORKER)
    port = str(29500 + os.getpid() % 2000)
    procs = [subprocess.Popen([sys.executable, str(script), ROOT, port, str(r)], stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT) for r in range(2)]
    outs = [p.communicate(timeout=180)[0].decode() for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o


DP_ADAM_WORKER = r'''
import math, os, sys, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1])
import reconvat_amd.train as tr
from reconvat_amd import ops
rank = int(sys.argv[3])
dist.init_process_group('gloo', init_method='tcp://127.0.0.1:' + sys.argv[2], rank=rank, world_size=2)

# fake backend for the two device entry points FlatAdam.step() launches (the product has no CPU path: this is the test's
# stand-in for libreconvat_hip.so so that the HOST logic -- flat views, twin-bucket fold, the one all-reduce, grad_scale,
# step counter, replica equality -- runs under gloo)
def fake_call(name, *a):
    if name == 'rv_adam_step':
        p, g, m, v, n, step, lr0, decay_steps, gamma, b1, b2, eps, gscale, skip, _st = a
        assert int(skip.item()) == 0
        t = int(step.item())
        lr = lr0 * gamma ** (t // decay_steps)
        gg = g * gscale
        m.mul_(b1).add_(gg, alpha=1 - b1)
        v.mul_(b2).addcmul_(gg, gg, value=1 - b2)
        bc1, bc2 = 1 - b1 ** (t + 1), 1 - b2 ** (t + 1)
        p.sub_((lr / bc1) * m / (v.sqrt() / math.sqrt(bc2) + eps))
    elif name == 'rv_counter_add':
        if a[2] is None or int(a[2].item()) == 0:
            a[0].add_(a[1])
    else:
        raise AssertionError(name)
tr.call, tr.ptr, tr.stream = fake_call, (lambda t: t), (lambda: None)
tr.FlatAdam._require_hip = staticmethod(lambda dev: None)
err = torch.zeros(1, dtype=torch.int32)
ops.step_error_word = lambda dev: err

torch.manual_seed(0)                                   # identical initial weights on every rank
net = torch.nn.Sequential(torch.nn.Linear(5, 7), torch.nn.Linear(7, 3))
ref = torch.nn.Sequential(torch.nn.Linear(5, 7), torch.nn.Linear(7, 3))
ref.load_state_dict(net.state_dict())
ropt = torch.optim.Adam(ref.parameters(), lr=1e-2)
rsch = torch.optim.lr_scheduler.StepLR(ropt, step_size=2, gamma=0.5)
opt = tr.FlatAdam(net.parameters(), lr=1e-2, step_size=2, gamma=0.5)        # data_parallel=None: auto
twin, = opt.enable_side_bucket(1)
calls = {'n': 0}
real_allreduce = dist.all_reduce
def counting(t, *a, **k):
    calls['n'] += 1
    return real_allreduce(t, *a, **k)
dist.all_reduce = counting
for it in range(5):
    opt.zero_grad()
    ropt.zero_grad()
    gens = [torch.Generator().manual_seed(100 * it + r) for r in range(2)]
    mine = None
    per_rank = []
    for r in range(2):
        main = [torch.randn(p.shape, generator=gens[r]) for p in net.parameters()]
        side = [torch.randn(p.shape, generator=gens[r]) for p in net.parameters()]
        per_rank.append([a + b for a, b in zip(main, side)])
        if r == rank:
            mine = (main, side)
    for p, g in zip(net.parameters(), mine[0]):
        p.grad.add_(g)                                 # autograd accumulates into views of the flat bucket
        assert p.grad.data_ptr() >= opt.flat_grad.data_ptr()
    for p, g, off in zip(net.parameters(), mine[1], opt.offsets):   # the side stream's backward lands in the twin bucket
        twin[off:off + p.numel()].add_(g.reshape(-1))
    opt.merge_side_grads()                             # fold BEFORE the collective
    before = calls['n']
    opt.step()
    assert calls['n'] == before + 1, 'exactly ONE all-reduce per optimiser step'
    for p, a, b in zip(ref.parameters(), *per_rank):
        p.grad = (a + b) / 2                           # what DDP would hand torch.optim.Adam
    ropt.step(); rsch.step()
    for p, q in zip(net.parameters(), ref.parameters()):
        assert torch.allclose(p, q, rtol=1e-5, atol=1e-7), (it, (p - q).abs().max())
    assert abs(opt.current_lr() - ropt.param_groups[0]['lr']) < 1e-12
# replicas are BIT-identical (what bench.py reports as replicas_equal)
chk = opt.flat_param.view(torch.int32).sum(dtype=torch.int64).reshape(1)
hi, lo = chk.clone(), chk.clone()
real_allreduce(hi, op=dist.ReduceOp.MAX); real_allreduce(lo, op=dist.ReduceOp.MIN)
assert int(hi - lo) == 0
# checkpoint layout == torch.optim.Adam's, and it round-trips
sd = opt.state_dict()
rsd = ropt.state_dict()
assert set(sd['state']) == set(rsd['state']) and sd['param_groups'][0]['params'] == rsd['param_groups'][0]['params']
for k in rsd['state']:
    assert torch.allclose(sd['state'][k]['exp_avg'], rsd['state'][k]['exp_avg'], rtol=1e-5, atol=1e-8)
    assert float(sd['state'][k]['step']) == float(rsd['state'][k]['step'])
opt2 = tr.FlatAdam(torch.nn.Sequential(torch.nn.Linear(5, 7), torch.nn.Linear(7, 3)).parameters(), lr=1.0)
opt2.load_state_dict(rsd)                              # a torch / reference optimizer checkpoint loads
assert int(opt2.step_count) == 5 and abs(opt2.lr - 1e-2) < 1e-12 and torch.allclose(opt2.exp_avg, opt.exp_avg, rtol=1e-5, atol=1e-8)
# a poisoned step is not applied
err.fill_(1)
try:
    opt.step()
    raise SystemExit('poisoned step was applied')
except AssertionError:
    pass
dist.barrier(); dist.destroy_process_group()
print('ok', rank)
'''


def test_flat_adam_two_ranks_gloo(tmp_path):
    """World size 2 over gloo: FlatAdam's host logic end to end on a fake device backend -- gradients accumulate into the
    flat bucket, the side-stream twin is folded in BEFORE the one all-reduce, the mean is folded into the Adam launch,
    StepLR follows, replicas stay bit-identical and match torch.optim.Adam on the averaged gradients; the checkpoint has
    torch.optim.Adam's layout."""
    script = tmp_path / 'w.py'
    script.write_text(DP_ADAM_WORKER)
    port = str(31500 + os.getpid() % 2000)
    procs = [subprocess.Popen([sys.executable, str(script), ROOT, port, str(r)], stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT) for r in range(2)]
    outs = [p.communicate(timeout=180)[0].decode() for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o


def test_every_run_training_option_is_injected_by_the_scripts():
    """sacred injects config entries into the main function BY PARAMETER NAME: an option of cli.run_training that a script's `train(...)`
    does not list silently falls back to its default (round 3: `dtype` and `logging_freq` did)."""
    import importlib.util
    import inspect
    from reconvat_amd import cli
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    options = set(inspect.signature(cli.run_training).parameters) - {'onset_script', '_unused'}
    for script in ('train_UNet_Onset_VAT.py', 'train_UNet_VAT.py', 'train_baseline_onset_frame_VAT.py'):
        spec = importlib.util.spec_from_file_location('script_' + script[:-3], os.path.join(root, script))
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)                    # (automain only runs under __main__)
        injected, cfg = set(inspect.signature(mod.train).parameters), set(mod.config({}))
        assert not (options & cfg) - injected, (script, sorted((options & cfg) - injected))
        assert not injected - cfg, (script, sorted(injected - cfg))


def test_launcher_counts_gpus_without_the_hip_runtime(monkeypatch):
    """bench.py's self-launcher decides how many ranks fit WITHOUT initialising the runtime (round 5): the *_VISIBLE_DEVICES lists first,
    else the KFD topology, else None (the ranks then fail on their own)."""
    import bench
    for var in ('ROCR_VISIBLE_DEVICES', 'HIP_VISIBLE_DEVICES', 'CUDA_VISIBLE_DEVICES'):
        monkeypatch.delenv(var, raising=False)
    monkeypatch.setenv('HIP_VISIBLE_DEVICES', '0,1,2')
    assert bench.visible_gpus() == 3
    monkeypatch.setenv('ROCR_VISIBLE_DEVICES', '4')              # takes precedence
    assert bench.visible_gpus() == 1
    monkeypatch.setenv('ROCR_VISIBLE_DEVICES', '')
    assert bench.visible_gpus() == 0
    for var in ('ROCR_VISIBLE_DEVICES', 'HIP_VISIBLE_DEVICES'):
        monkeypatch.delenv(var, raising=False)
    n = bench.visible_gpus()
    assert n is None or n >= 0                                    # KFD topology (0 compute nodes in the build container) or unreadable


def test_deterministic_mode_switch_is_read_from_the_environment():
    import subprocess
    import sys
    code = "from reconvat_amd import ops; print(ops.DETERMINISTIC[0])"
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for val, want in (('1', 'True'), ('0', 'False'), (None, 'False')):
        env = dict(os.environ, PYTHONPATH=root)
        env.pop('RV_DETERMINISTIC', None)
        if val is not None:
            env['RV_DETERMINISTIC'] = val
        out = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, env=env, cwd=root)
        assert out.returncode == 0 and out.stdout.strip() == want, (val, out.stdout, out.stderr[-500:])


def test_numerics_emulation_helpers():
    """The two emulation scripts behind DESIGN section 7 (split-bf16, Winograd F(4x4)): the 3-way bf16 split reproduces an fp32 value exactly,
    the 2-way split to 2^-16, and the fp32 Winograd forms agree with a direct convolution to their known error levels."""
    import torch
    import torch.nn.functional as F
    sys_path = os.path.dirname(os.path.abspath(__file__))
    import sys
    sys.path.insert(0, sys_path)
    import emulate_bf16_split as eb
    import emulate_winograd_f4 as ew
    g = torch.Generator().manual_seed(0)
    x = (torch.rand(4096, generator=g) - 0.5) * 37.0
    h, m, l = eb.split(x, 3)
    assert torch.equal(h + m + l, x)
    h2, l2 = eb.split(x, 2)
    assert float(((h2 + l2) - x).abs().max() / x.abs().max()) < 2.0 ** -15
    xi = torch.rand(2, 16, 19, 30, generator=g) - 0.5
    w = (torch.rand(24, 16, 3, 3, generator=g) - 0.5) * 0.3
    ref = F.conv2d(xi.double(), w.double(), padding=1)
    e2 = float((ew.winograd_conv(xi, w, 2).double() - ref).abs().max() / ref.abs().max())
    e4 = float((ew.winograd_conv(xi, w, 4).double() - ref).abs().max() / ref.abs().max())
    assert e2 < 2e-6 and e4 < 5e-5 and e4 > e2
    op = lambda a, b: F.conv2d(a, b, None, padding=1)
    e6 = float((eb.emulated(op, xi, w, 'bf16x6').double() - ref).abs().max() / ref.abs().max())
    e3 = float((eb.emulated(op, xi, w, 'bf16x3').double() - ref).abs().max() / ref.abs().max())
    e1 = float((eb.emulated(op, xi, w, 'bf16').double() - ref).abs().max() / ref.abs().max())
    assert e6 < 2e-6 and e6 < e3 < e1 and e1 > 1e-3


def test_wgrad_merger_bookkeeping(monkeypatch):
    """ops.WgradMerger without a device: what it launches when (the launch itself is stubbed).  A step that runs unmerged teaches it the number of passes per
    gradient buffer and mode; afterwards passes register and the LAST one launches; finish() launches what is left; counts are re-learned every step."""
    from reconvat_amd import ops
    m = ops.WgradMerger()
    launched = []
    monkeypatch.setattr(m, '_launch', lambda key: launched.append((key, len(m.pending.pop(key)))))
    # step 1 (mode two-chain): nothing known -> every pass is launched by the caller (submit returns False)
    m.begin(True)
    assert [m.submit('a', 1), m.submit('a', 2), m.submit('b', 1), m.submit('a', 3)] == [False] * 4
    m.finish()
    assert launched == [] and m.learned == {(True, 'a'): 3, (True, 'b'): 1}
    # step 2: 'a' merges at its third pass, 'b' (a single pass) stays with the caller
    m.begin(True)
    assert m.submit('a', 1) and m.submit('a', 2) and launched == []
    assert m.submit('b', 1) is False
    assert m.submit('a', 3) and launched == [('a', 3)]
    m.finish()
    # another mode (one chain) has its own counts: learning again
    m.begin(False)
    assert m.submit('a', 1) is False
    m.finish()
    assert m.learned[(False, 'a')] == 1 and m.learned[(True, 'a')] == 3
    # a step that comes up one pass short: finish() launches the two that arrived, and the count is re-learned
    launched.clear()
    m.begin(True)
    assert m.submit('a', 1) and m.submit('a', 2) and launched == []
    m.finish()
    assert launched == [('a', 2)] and m.learned[(True, 'a')] == 2
    # more passes than one launch takes (rv_conv_wgrad_seg: four segments): a launch every fourth pass, the rest at finish()
    m.learned[(True, 'c')] = 6
    launched.clear()
    m.begin(True)
    for i in range(6):
        assert m.submit('c', i)
    assert launched == [('c', 4)]
    m.finish()
    assert launched == [('c', 4), ('c', 2)]


def test_rank_cpu_affinity_slices():
    """dp.pin_rank_cpus (round 6): rank r of n pins itself to the r-th of n equal slices of the launcher's CPU set -- checked in child processes (affinity is
    process state); a rank count larger than the CPU set leaves the set alone."""
    import subprocess
    import sys
    code = (
        "import os, sys\n"
        f"sys.path.insert(0, {ROOT!r})\n"
        "from reconvat_amd import dp\n"
        "have = sorted(os.sched_getaffinity(0))\n"
        "r, n = int(sys.argv[1]), int(sys.argv[2])\n"
        "mine = dp.pin_rank_cpus(r, n)\n"
        "k = len(have) // n\n"
        "want = have[r * k:(r + 1) * k] if (n > 1 and k >= 1) else have\n"
        "assert mine == want == sorted(os.sched_getaffinity(0)), (mine, want)\n"
        "print(len(mine))\n")
    ncpu = len(os.sched_getaffinity(0))
    for r, n in ((0, 2), (1, 2), (0, 1), (0, 4 * ncpu)):
        p = subprocess.run([sys.executable, '-c', code, str(r), str(n)], capture_output=True, text=True)
        assert p.returncode == 0, p.stderr[-1500:]
