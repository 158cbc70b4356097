"""Rank-correctness of the data-parallel step WITH THE REAL KERNELS on a one-GPU box.

RCCL refuses two ranks on one device; gloo does not.  `RV_DP_BACKEND=gloo RV_DP_SAME_GPU=1 python bench.py --gpus 2` starts two
fresh rank processes on `cuda:0`, each running the real two-stream hipGraph `TrainStep` + `FlatAdam` on its own shard, with the
ONE gradient all-reduce per optimiser step going through `reconvat_amd/dp.py` (the same call site RCCL uses, staged through pinned
host memory).  Checked here:

* the line: two ranks, replicas bit-identical after all steps, exactly one gradient all-reduce per rank per optimiser step;
* the ranks really differ before the collective (data shard, VAT noise stream, gradient bucket) and agree bit for bit after it;
* the collective is the SUM: post-bucket == pre-bucket(rank 0) + pre-bucket(rank 1), bit for bit, on both ranks;
* rank r computes what a SINGLE process computes on shard r: the eleven loss terms of its first step AND its whole gradient bucket bit
  for bit (the ranks and the rebuilt step run with RV_DETERMINISTIC=1: parameter gradients folded in a fixed order, no fp32 atomics);
* rank 0's parameters after its first optimiser step are, bit for bit, those of a single-process Adam step on the MEAN of the two
  buckets (sum, 1/world folded into the kernel -- model/helper_functions.py:577-607 is one such step at world = 1).
"""
import json
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STEPS, WARMUP, BATCH = 3, 1, 2


@pytest.fixture(scope='module')
def two_ranks(tmp_path_factory):
    out = str(tmp_path_factory.mktemp('dp'))
    # (RV_DETERMINISTIC=1: parameter gradients folded in a fixed order -- a rank's bucket can then be compared bit for bit with a single process)
    env = dict(os.environ, PYTHONPATH=ROOT, RV_DP_BACKEND='gloo', RV_DP_SAME_GPU='1', HSA_ENABLE_IPC_MODE_LEGACY='0', RV_DETERMINISTIC='1')
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_PORT'):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', str(STEPS), '--warmup', str(WARMUP),
                        '--batch', str(BATCH), '--dp-dump', out], capture_output=True, text=True, cwd=ROOT, env=env, timeout=1200)
    assert p.returncode == 0, p.stderr[-4000:]
    line = json.loads(p.stdout.strip().splitlines()[-1])
    ranks = [torch.load(os.path.join(out, f'rank{r}.pt'), map_location='cpu') for r in range(2)]
    return line, ranks


def test_two_replicas_line(dev, two_ranks):
    line, ranks = two_ranks
    assert line['n_gpus'] == 2 and line['dp_ranks'] == 2 and line['dp_backend'] == 'gloo'
    assert line['replicas_equal'] is True
    assert line['dp_allreduce_calls'] == STEPS + WARMUP == line['optimizer_steps']      # ONE collective per optimiser step
    for r in ranks:
        assert r['allreduce_calls'] == STEPS + WARMUP == r['optimizer_steps'] and r['world'] == 2
    assert line['config']['vat_nan_flag'] == 0
    # whole-job aggregate over both ranks: 2 ranks x (2 + 2) segments x 20.48 s per step
    assert abs(line['value'] - 2 * 2 * BATCH * 20.48 / (line['mean_ms_per_step'] * 1e-3)) <= 1e-3 * line['value']


def test_ranks_differ_before_and_agree_after_the_collective(dev, two_ranks):
    _, (r0, r1) = two_ranks
    assert r0['audio_checksum_l'] != r1['audio_checksum_l'] and r0['audio_checksum_ul'] != r1['audio_checksum_ul']   # distinct shards
    assert r0['cuda_seed'] != r1['cuda_seed']                                                                       # distinct VAT noise
    assert r0['losses_step1'] != r1['losses_step1']
    assert not torch.equal(r0['pre_bucket'], r1['pre_bucket'])
    total = r0['pre_bucket'] + r1['pre_bucket']                   # fp32 addition of two operands: exact and commutative
    assert torch.equal(r0['post_bucket'], total) and torch.equal(r1['post_bucket'], total)
    assert torch.equal(r0['params_after_step1'], r1['params_after_step1'])
    assert torch.equal(r0['params_final'], r1['params_final'])
    assert not torch.equal(r0['params_final'], r0['params_after_step1'])


@pytest.mark.parametrize('rank', [0, 1])
def test_rank_equals_single_process_on_its_shard(dev, two_ranks, rank, monkeypatch):
    """The first step of rank `rank`, rebuilt in THIS process (no process group): same seeds -> same shard, same noise stream; in the
    deterministic reduction mode the whole gradient bucket is bit-identical ("replicas bit-identical")."""
    import bench
    from reconvat_amd import ops
    monkeypatch.setattr(ops, 'DETERMINISTIC', [True])
    _, ranks = two_ranks
    ref = ranks[rank]
    model, opt, batch, batch_ul, step = bench.make_rank_step('onset', BATCH, BATCH, rank, dev)
    assert float(batch['audio'].double().sum()) == ref['audio_checksum_l']
    step.capture()
    step.graph.replay()                                           # forward + backward of step 1; no optimiser step yet
    torch.cuda.synchronize()
    for k, v in step.losses.items():
        assert float(v) == ref['losses_step1'][k], (k, float(v), ref['losses_step1'][k])       # deterministic data path: bit for bit
    assert torch.equal(opt.flat_grad.cpu(), ref['pre_bucket']), float((opt.flat_grad.cpu() - ref['pre_bucket']).abs().max())


def test_rank0_step_equals_single_process_adam_on_the_mean_bucket(dev, two_ranks):
    import bench
    _, (r0, r1) = two_ranks
    model, opt, batch, batch_ul, step = bench.make_rank_step('onset', BATCH, BATCH, 0, dev, graph=False)
    assert opt.flat_grad.numel() == r0['pre_bucket'].numel()
    opt.flat_grad.copy_((r0['pre_bucket'] + r1['pre_bucket']).to(dev))        # the summed bucket ...
    opt.grad_scale = 0.5                                                      # ... with the 1/world mean folded into the Adam kernel
    opt.step()                                                                # (no process group here: no collective)
    torch.cuda.synchronize()
    assert torch.equal(opt.flat_param.cpu(), r0['params_after_step1'])


# ---- the launcher at the REAL world size, on one GPU: eight rank processes over gloo, all on cuda:0 ----------------
def _launch_eight(tmp_path, extra_env, steps=2, warmup=1, timeout=1500):
    out = str(tmp_path / 'dp8')
    env = dict(os.environ, PYTHONPATH=ROOT, RV_DP_BACKEND='gloo', RV_DP_SAME_GPU='1', HSA_ENABLE_IPC_MODE_LEGACY='0')
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_PORT', 'OMP_NUM_THREADS'):
        env.pop(k, None)
    env.update(extra_env)
    import time
    t0 = time.time()
    p = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '8', '--steps', str(steps), '--warmup', str(warmup),
                        '--batch', '1', '--dp-dump', out], capture_output=True, text=True, cwd=ROOT, env=env, timeout=timeout)
    return p, out, time.time() - t0


def test_eight_rank_rehearsal_on_one_gpu(dev, tmp_path):
    """`python bench.py --gpus 8` exactly as the driver's 8-GPU node will run it -- self-launch, rendezvous on 127.0.0.1, one process per
    rank, per-rank data shard and VAT noise stream, OMP_NUM_THREADS split, ONE all-reduce of the flat bucket per optimiser step, replica
    check -- with every rank on cuda:0 over gloo (B = 1 + 1 per rank).  What this cannot show is RCCL's own ring on eight devices."""
    p, out, _ = _launch_eight(tmp_path, {})
    assert p.returncode == 0, p.stderr[-4000:]
    line = json.loads(p.stdout.strip().splitlines()[-1])
    assert line['n_gpus'] == 8 and line['dp_ranks'] == 8 and line['dp_backend'] == 'gloo' and line['replicas_equal'] is True
    assert line['dp_allreduce_calls'] == 3 == line['optimizer_steps']
    assert abs(line['value'] - 8 * 2 * 20.48 / (line['mean_ms_per_step'] * 1e-3)) <= 1e-3 * line['value']        # whole-job aggregate over 8 ranks
    ranks = [torch.load(os.path.join(out, f'rank{r}.pt'), map_location='cpu') for r in range(8)]
    assert [r['rank'] for r in ranks] == list(range(8)) and all(r['world'] == 8 for r in ranks)
    assert len({r['pid'] for r in ranks}) == 8                                                # eight processes
    assert len({r['audio_checksum_l'] for r in ranks}) == 8 and len({r['audio_checksum_ul'] for r in ranks}) == 8     # eight distinct shards
    assert len({r['cuda_seed'] for r in ranks}) == 8                                          # eight VAT noise streams
    want_threads = str(max(1, (os.cpu_count() or 8) // 8))
    assert all(r['omp_num_threads'] == want_threads for r in ranks), [r['omp_num_threads'] for r in ranks]
    # per-rank CPU affinity (reconvat_amd.dp.pin_rank_cpus, set before the rank touches the GPU): eight disjoint, equally sized slices
    # of the launcher's own CPU set
    have = sorted(os.sched_getaffinity(0))
    k = len(have) // 8
    if k >= 1:
        for r in ranks:
            assert r['cpu_affinity'] == have[r['rank'] * k:(r['rank'] + 1) * k] == r['rank_cpus'], (r['rank'], r['cpu_affinity'])
        assert len({c for r in ranks for c in r['cpu_affinity']}) == 8 * k
    total = sum(r['pre_bucket'].double() for r in ranks)
    for r in ranks:
        assert r['allreduce_calls'] == 3 and r['optimizer_steps'] == 3
        assert (r['post_bucket'].double() - total).abs().max().item() <= 1e-5 * total.abs().max().item()   # the collective is the sum of all eight
        assert torch.equal(r['post_bucket'], ranks[0]['post_bucket'])                          # ... and bit-identical on every rank
        assert torch.equal(r['params_final'], ranks[0]['params_final'])


def test_launcher_fails_fast_when_one_of_eight_ranks_dies(dev, tmp_path):
    """Rank 5 dies in the middle of the timed loop (its peers are blocked in the next all-reduce): the launcher notices the exit code,
    terminates the other seven and exits non-zero -- within seconds of the death, far inside the collective timeout."""
    p, _, wall = _launch_eight(tmp_path, {'RV_TEST_FAIL_RANK': '5', 'RV_TEST_FAIL_AT': '1'}, steps=4)
    assert p.returncode != 0
    assert 'rank(s) failed' in (p.stderr + p.stdout) and '(5, 17)' in (p.stderr + p.stdout), p.stderr[-2000:]
    assert wall < 600, wall
