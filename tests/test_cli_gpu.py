"""The drop-in command lines on the GPU, as fresh child processes (reference train_UNet_Onset_VAT.py:82-170 /
train_UNet_VAT.py:26-95): epochs of `iteration` steps, a checkpoint every `saving_freq` epochs (model-{ep}.pt +
last-optimizer-state.pt), scalar logging of every loss key, resume, the final validation; BASELINE config 1 (the plumbing
run) on device=cuda:0 and its device=cpu form failing with one clear sentence; `bench.py --gpus N` launching itself."""
import json
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SMALL = ['train_on=Synthetic', 'small=True', 'supersmall=True', 'sequence_length=32768', 'batch_size=2', 'train_batch_size=2',
         'iteration=2']


def run(script, *args, expect_ok=True, timeout=900, cwd=ROOT):
    env = dict(os.environ, PYTHONPATH=ROOT)
    p = subprocess.run([sys.executable, os.path.join(ROOT, script), 'with', *args], capture_output=True, text=True, cwd=cwd,
                       env=env, timeout=timeout)
    if expect_ok:
        assert p.returncode == 0, p.stdout[-3000:] + '\n---\n' + p.stderr[-3000:]
    return p


def scalar_tags(logdir):
    with open(os.path.join(logdir, 'scalars.jsonl')) as fh:
        rows = [json.loads(line) for line in fh]
    return rows


def test_onset_script_checkpoint_resume_final_eval(dev, tmp_path):
    from oracle import fixture as fx
    logdir = str(tmp_path / 'run')
    p = run('train_UNet_Onset_VAT.py', *SMALL, 'reconstruction=True', 'epoches=2', 'saving_freq=1', f'logdir={logdir}')
    assert 'Training finished.' in p.stdout and 'validation frame F1' in p.stdout
    for f in ('model-1.pt', 'model-2.pt', 'model-final.pt', 'last-optimizer-state.pt'):
        assert os.path.exists(os.path.join(logdir, f)), f
    # checkpoint = the reference's state_dict surface (keys and shapes), loadable by a fresh model
    sd = torch.load(os.path.join(logdir, 'model-2.pt'), map_location='cpu')
    want = dict(fx.param_shapes('onset', True))
    want.update({'spectrogram.mel_basis': (229, 1025), 'spectrogram.stft.wsin': (1025, 1, 2048),
                 'spectrogram.stft.wcos': (1025, 1, 2048), 'spectrogram.stft.window_mask': (1, 2048, 1)})
    assert {k: tuple(v.shape) for k, v in sd.items()} == want
    # 8 transcriber forwards per step x 2 steps x 2 epochs; the capture warm-up must leave no trace
    assert int(sd['transcriber.Unet1_encoder.block1.bn1.num_batches_tracked']) == 32
    # optimizer checkpoint has torch.optim.Adam's layout: 4 steps taken
    osd = torch.load(os.path.join(logdir, 'last-optimizer-state.pt'), map_location='cpu')
    assert set(osd) == {'state', 'param_groups'} and osd['param_groups'][0]['initial_lr'] == 1e-3
    assert {float(st['step']) for st in osd['state'].values()} == {4.0}
    reference_adam = torch.optim.Adam([torch.nn.Parameter(torch.zeros(s)) for s in
                                       (tuple(v['exp_avg'].shape) for _, v in sorted(osd['state'].items()))])
    reference_adam.load_state_dict({'state': {i: v for i, (_, v) in enumerate(sorted(osd['state'].items()))},
                                    'param_groups': [dict(osd['param_groups'][0], params=list(range(len(osd['state']))))]})
    # every loss key of the VAT + reconstruction step is logged each epoch
    rows = scalar_tags(logdir)
    keys = {'loss/train_reconstruction', 'loss/train_frame', 'loss/train_frame2', 'loss/train_onset', 'loss/train_onset2',
            'loss/train_LDS_l_frame', 'loss/train_LDS_l_onset', 'loss/train_LDS_ul_frame', 'loss/train_LDS_ul_onset',
            'loss/train_r_norm_l', 'loss/train_r_norm_ul'}
    for ep in (1, 2):
        assert {r['tag'] for r in rows if r['step'] == ep and r['tag'].startswith('loss/train_')} == keys
    assert any(r['tag'] == 'validation/metric/frame/f1' for r in rows)
    _check_evaluation_contract(logdir, rows, n_songs=4, first_epoch_logged=True)
    # resume from epoch 1: runs epoch 2 only, continues the optimiser (step 2 -> 4) and the StepLR position
    p = run('train_UNet_Onset_VAT.py', *SMALL, 'reconstruction=True', 'epoches=2', 'saving_freq=1', f'logdir={logdir}',
            'resume_iteration=2', 'epoches=3')
    assert 'Resumed from model-2.pt: optimiser step 4, lr 1.000000e-03' in p.stdout
    assert 'Train Epoch: 3' in p.stdout and 'Train Epoch: 2\t' not in p.stdout and 'Train Epoch: 1\t' not in p.stdout
    osd = torch.load(os.path.join(logdir, 'last-optimizer-state.pt'), map_location='cpu')
    assert {float(st['step']) for st in osd['state'].values()} == {6.0}


METRIC_KEYS = {f'metric/{c}/{n}' for c in ('note', 'note-with-offsets') for n in ('precision', 'recall', 'f1', 'overlap')} | \
    {'metric/frame/f1', 'metric/MusicNet/micro_avg_P', 'metric/frame/precision', 'metric/frame/recall', 'metric/frame/accuracy',
     'metric/frame/substitution_error', 'metric/frame/miss_error', 'metric/frame/false_alarm_error', 'metric/frame/total_error',
     'metric/frame/chroma_precision', 'metric/frame/chroma_recall', 'metric/frame/chroma_accuracy',
     'metric/frame/chroma_substitution_error', 'metric/frame/chroma_miss_error', 'metric/frame/chroma_false_alarm_error',
     'metric/frame/chroma_total_error'}


def _check_evaluation_contract(logdir, rows, n_songs, first_epoch_logged):
    """What the reference scripts leave behind (train_UNet_Onset_VAT.py:136-170, model/helper_functions.py:120-141):
    <logdir>/result_dict = pickle of the whole-song metrics dict, <logdir>/MIDI_results/<song>.pred.mid (+ piano-roll PNGs),
    and -- every logging_freq epochs and after epoch 1 -- the validation precision / recall / f1 scalars plus the eval-mode
    loss terms of eval_model."""
    import pickle
    from reconvat_amd.midi import parse_midi
    with open(os.path.join(logdir, 'result_dict'), 'rb') as fh:
        metrics = pickle.load(fh)
    assert METRIC_KEYS <= set(metrics), METRIC_KEYS - set(metrics)
    assert all(len(metrics[k]) == n_songs for k in METRIC_KEYS)
    assert any(k.startswith('loss/test_') for k in metrics)
    mids = sorted(f for f in os.listdir(os.path.join(logdir, 'MIDI_results')) if f.endswith('.pred.mid'))
    assert len(mids) == n_songs, mids
    parse_midi(os.path.join(logdir, 'MIDI_results', mids[0]))                  # a well-formed standard MIDI file
    pngs = [f for f in os.listdir(os.path.join(logdir, 'MIDI_results')) if f.endswith('.png')]
    assert len(pngs) in (0, 2 * n_songs)                                         # label + prediction rolls when PIL is present
    if first_epoch_logged:
        ep1 = {r['tag'] for r in rows if r['step'] == 1}
        assert {'metric/note/f1', 'metric/frame/f1', 'metric/note-with-offsets/precision'} <= ep1, ep1
        assert not any('chroma' in t for t in ep1 if t.startswith('metric/'))
        assert any(t.startswith('loss/test_') for t in ep1), ep1              # eval_model over the labelled loader


def test_config5_string_corpus_on_device(dev, tmp_path):
    """BASELINE config 5 end to end on the device: `train_UNet_Onset_VAT.py with train_on=String small=True reconstruction=True`
    over a MusicNet-layout corpus (labelled = first recording of each string ensemble, unlabelled = the rest,
    model/dataset.py:238-342) ingested from wav + tsv, resident in HBM (DeviceCorpus), trained for two steps, evaluated on the
    four whole test_violin recordings.  The corpus split must equal what the REFERENCE's MusicNet class produced on the same
    files (tests/golden/ingest.npz), and data-parallel rank shards must be disjoint and complete."""
    import numpy as np
    from oracle import dataset as od
    corpus = str(tmp_path / 'corpus')
    od.ingest_corpus(corpus)
    logdir = str(tmp_path / 'string')
    p = run('train_UNet_Onset_VAT.py', 'train_on=String', 'small=True', 'reconstruction=True', 'epoches=1', 'iteration=2',
            'sequence_length=8192', 'batch_size=2', 'train_batch_size=2', 'saving_freq=1', f'logdir={logdir}', cwd=corpus)
    assert 'Training finished.' in p.stdout and 'Train Epoch: 1' in p.stdout
    g = np.load(os.path.join(ROOT, 'tests', 'golden', 'ingest.npz'))
    with open(os.path.join(logdir, 'corpus_rank0.json')) as fh:
        info = json.load(fh)
    rel = lambda paths: [os.path.splitext(os.path.relpath(os.path.join(corpus, q) if not os.path.isabs(q) else q, corpus))[0] for q in paths]
    assert rel(info['labelled']) == list(g['mn_train_string_l_paths'])
    assert rel(info['unlabelled']) == list(g['mn_train_string_ul_paths'])
    assert rel(info['full_validation']) == list(g['mn_test_violin_paths'])
    assert sorted(info['labelled_shard']) == sorted(info['labelled'])             # world 1: the whole corpus on this rank
    rows = scalar_tags(logdir)
    _check_evaluation_contract(logdir, rows, n_songs=4, first_epoch_logged=True)
    # rank shards of the device feed (what `torch.distributed.run --nproc-per-node 2` hands each rank): disjoint, complete
    from reconvat_amd.dataset import MusicNet
    from reconvat_amd.feed import device_loader
    for group, bs in (('train_string_l', 2), ('train_string_ul', 2)):
        ds = MusicNet(os.path.join(corpus, 'MusicNet'), [group], sequence_length=8192)
        shards = [device_loader(ds, bs, dev, rank=r, world=2, seed=42 + r).paths for r in range(2)]
        assert not set(shards[0]) & set(shards[1])
        assert sorted(shards[0] + shards[1]) == sorted(d['path'] for d in ds.data)
        batch = next(iter(device_loader(ds, bs, dev, rank=1, world=2, seed=43)))
        assert batch['audio'].shape == (bs, 8192) and batch['audio'].is_cuda and set(batch['path']) <= set(shards[1])


def test_plumbing_config_on_gpu_and_cpu_device_message(dev, tmp_path):
    """BASELINE config 1: `train_UNet_VAT.py with ... VAT=False reconstruction=False` -- runs on device=cuda:0; device=cpu is
    refused with one clear sentence (there is no CPU product path)."""
    logdir = str(tmp_path / 'plumb')
    p = run('train_UNet_VAT.py', *SMALL, 'VAT=False', 'reconstruction=False', 'epoches=1', 'saving_freq=1', 'device=cuda:0',
            f'logdir={logdir}')
    assert 'Training finished.' in p.stdout
    rows = scalar_tags(logdir)
    assert {r['tag'] for r in rows if r['tag'].startswith('loss/train_')} == {'loss/train_frame', 'loss/train_LDS_l', 'loss/train_LDS_ul',
                                                                           'loss/train_r_norm_l', 'loss/train_r_norm_ul'}
    p = run('train_UNet_VAT.py', *SMALL, 'VAT=False', 'reconstruction=False', 'epoches=1', 'device=cpu', expect_ok=False)
    assert p.returncode != 0 and 'MI355X only' in (p.stderr + p.stdout) and 'Traceback' not in p.stderr


def test_bf16_backward_and_tuning_modes_from_the_command_line(dev, tmp_path):
    """`dtype=bf16` (opt-in experiment: bf16-operand backward convs, forward fp32) trains through the script; RV_AUTOTUNE=0 (library
    default tiles) and RV_AUTOTUNE=1 (on-line tuner) run the same command line as the default plan-table mode."""
    logdir = str(tmp_path / 'bf16')
    env = dict(os.environ, PYTHONPATH=ROOT, RV_BF16_LOG='1')
    p = subprocess.run([sys.executable, os.path.join(ROOT, 'train_UNet_Onset_VAT.py'), 'with', *SMALL, 'reconstruction=True', 'epoches=3',
                        'saving_freq=1', 'logging_freq=2', 'dtype=bf16', f'logdir={logdir}'], capture_output=True, text=True, cwd=ROOT,
                       env=env, timeout=900)
    assert p.returncode == 0 and 'Training finished.' in p.stdout, p.stderr[-2000:]
    assert '[bf16] backward convs of the final graphs use bf16 operands' in p.stderr          # the option reached TrainStep
    rows = scalar_tags(logdir)
    assert any(r['tag'] == 'loss/train_frame' for r in rows)
    # logging_freq=2 reached the loop: validation scalars after epochs 1 and 2, not after 3
    assert {r['step'] for r in rows if r['tag'] == 'metric/frame/f1'} == {1, 2}
    for mode in ('0', '1'):
        env = dict(os.environ, PYTHONPATH=ROOT, RV_AUTOTUNE=mode)
        q = subprocess.run([sys.executable, os.path.join(ROOT, 'train_UNet_VAT.py'), 'with', *SMALL, 'VAT=True', 'reconstruction=False',
                            'epoches=1', f'logdir={tmp_path / ("tune" + mode)}'], capture_output=True, text=True, cwd=ROOT, env=env,
                           timeout=900)
        assert q.returncode == 0 and 'Training finished.' in q.stdout, q.stderr[-2000:]


def test_baseline_script_end_to_end(dev, tmp_path):
    """`train_baseline_onset_frame_VAT.py with ... VAT=True` (Onsets&Frames BiLSTM baseline, train_baseline_onset_frame_VAT.py:25-170 of
    the reference): two epochs through the three-chain captured step, checkpoint with the reference's state_dict keys, the
    evaluation contract at the end."""
    logdir = str(tmp_path / 'onf')
    p = run('train_baseline_onset_frame_VAT.py', *SMALL, 'VAT=True', 'epoches=2', 'saving_freq=1', f'logdir={logdir}')
    assert 'Training finished.' in p.stdout and 'Train Epoch: 2' in p.stdout
    sd = torch.load(os.path.join(logdir, 'model-2.pt'), map_location='cpu')
    assert 'combined_stack.1.weight' in sd or any(k.startswith('combined_stack') for k in sd), list(sd)[:8]
    assert any(k.startswith('onset_stack') for k in sd) and any(k.startswith('frame_stack') for k in sd)
    rows = scalar_tags(logdir)
    assert any(r['tag'].startswith('loss/train_') for r in rows)
    _check_evaluation_contract(logdir, rows, n_songs=4, first_epoch_logged=False)


def test_eager_torch_optimizer_path(dev, tmp_path):
    """graph=False fused_optimizer=False: the reference loop verbatim (train_VAT_model + torch.optim.Adam + StepLR)."""
    logdir = str(tmp_path / 'eager')
    p = run('train_UNet_Onset_VAT.py', *SMALL, 'reconstruction=True', 'epoches=1', 'saving_freq=1', 'graph=False',
            'fused_optimizer=False', f'logdir={logdir}')
    assert 'Training finished.' in p.stdout
    osd = torch.load(os.path.join(logdir, 'last-optimizer-state.pt'), map_location='cpu')
    assert set(osd) == {'state', 'param_groups'}


def test_bench_self_launch_two_ranks(dev):
    """`python bench.py --gpus 2` with no launcher: two fresh rank processes over RCCL, replicas bit-identical."""
    if torch.cuda.device_count() < 2:
        pytest.skip('needs 2 GPUs')
    p = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '3', '--warmup', '1',
                        '--batch', '2'], capture_output=True, text=True, cwd=ROOT, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    line = json.loads(p.stdout.strip().splitlines()[-1])
    assert line['n_gpus'] == 2 and line['dp_ranks'] == 2 and line['dp_backend'] == 'nccl' and line['replicas_equal'] is True


def test_rccl_allreduce_executes_on_one_gpu(dev):
    """The builder's lease is ONE GPU and RCCL refuses two ranks on one device ("Duplicate GPU detected"), so the two-rank test
    above skips there.  This one runs the data-parallel code path for real with a SINGLE-rank RCCL group (RV_DP_FORCE_ALLREDUCE=1):
    `init_process_group('nccl')`, the barrier, FlatAdam's one `dist.all_reduce` of the 14.6 MB flat bucket per step between the
    hipGraph replay and the Adam kernel, the MAX / MIN reductions of the replica check, the teardown -- and the step must still
    train (same loss as without the group; an all-reduce over one rank is the identity)."""
    outs = []
    for force in ('0', '1'):
        env = dict(os.environ, PYTHONPATH=ROOT, RV_DP_FORCE_ALLREDUCE=force, HSA_ENABLE_IPC_MODE_LEGACY='0', MASTER_PORT='29547')
        p = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--steps', '3', '--warmup', '1', '--no-cpu-baseline',
                            '--no-roofline', '--no-parity'], capture_output=True, text=True, cwd=ROOT, env=env, timeout=600)
        assert p.returncode == 0, p.stderr[-3000:]
        lines = [ln for ln in p.stdout.splitlines() if ln.startswith('{"metric"')]
        assert len(lines) == 1, p.stdout[-2000:]
        assert p.stdout.strip().splitlines()[-1] == lines[0], 'the JSON line must be the LAST line of stdout: ' + p.stdout[-1500:]
        outs.append(json.loads(lines[0]))
    plain, rccl = outs
    # the line's own contract: `ms_per_step` = the median of the per-step device times, `value` = whole-job throughput over the
    # barrier-to-barrier wall clock (`mean_ms_per_step`), and the library's source digest equals the digest of the sources next to it
    for ln in outs:
        assert ln['min_ms_per_step'] <= ln['ms_per_step'] <= ln['max_ms_per_step'] and ln['steps'] == 3 and ln['warmup'] == 1 and ln['n_gpus'] == 1
        assert abs(ln['value'] - 16 * 20.48 / (ln['mean_ms_per_step'] * 1e-3)) <= 1e-3 * ln['value']
        assert ln['higher_is_better'] is True and ln['scaling'] == 'weak' and ln['dtype'] == 'f32' and ln['vs_baseline'] is None and ln['data'] == 'synthetic'
        assert ln['config']['source_digest'] == ln['config']['source_digest_of_tree'] and len(ln['config']['source_digest']) == 16
    assert plain['dp_allreduce_calls'] == 0 and rccl['dp_allreduce_calls'] == 4 and rccl['dp_ranks'] == 1 and rccl['dp_backend'] == 'nccl'
    assert rccl['replicas_equal'] is True
    assert abs(rccl['config']['final_loss'] - plain['config']['final_loss']) <= 2e-3 * abs(plain['config']['final_loss'])


def test_bench_refuses_more_gpus_than_present(dev):
    n = torch.cuda.device_count() + 1
    p = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', str(n), '--steps', '1', '--warmup', '0'],
                       capture_output=True, text=True, cwd=ROOT, timeout=300)
    assert p.returncode != 0 and 'exposes' in p.stderr
