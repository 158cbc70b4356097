"""Evaluation path: ``evaluate_wo_velocity`` with the reference's metric keys
(model/evaluate_functions.py:20-127).

The reference delegates the metrics to ``mir_eval`` (multipitch.evaluate, transcription.precision_recall_f1_overlap),
which is neither vendored in the reference nor installed here.  They are restated below from their published
definitions -- PARITY UNPINNED for these two functions (no reference output available to pin them); their call sites,
argument conventions and the metric keys follow the reference, and tests/test_decoding.py checks them on cases with
known answers.  ``average_precision_score`` is scikit-learn's, as in the reference.
"""
import os
import sys
from collections import defaultdict

import numpy as np
import torch
from scipy.sparse import csr_matrix
from scipy.sparse.csgraph import maximum_bipartite_matching
from scipy.stats import hmean

from .constants import HOP_LENGTH, SAMPLE_RATE, MIN_MIDI
from .decoding import extract_notes_wo_velocity, notes_to_frames
from .midi import save_midi

eps = sys.float_info.epsilon
N_DECIMALS = 4          # mir_eval rounds time differences to 0.1 ms before comparing with a tolerance


def midi_to_hz(midi):
    return 440.0 * (2.0 ** ((np.asarray(midi, dtype=np.float64) - 69.0) / 12.0))


def _hz_to_midi(hz):
    return 12.0 * (np.log2(np.asarray(hz, dtype=np.float64)) - np.log2(440.0)) + 69.0


# ---------------------------------------------------------------------------------------------
# frame metrics (mir_eval.multipitch.evaluate: Poliner & Ellis 2007 error decomposition)
# ---------------------------------------------------------------------------------------------
def _count_matches(ref_midi, est_midi, window=0.5):
    """Maximum number of one-to-one pairs with |ref - est| < window semitones (greedy on sorted lists is optimal in 1-D)."""
    r, e = np.sort(ref_midi), np.sort(est_midi)
    i = j = n = 0
    while i < len(r) and j < len(e):
        d = r[i] - e[j]
        if abs(d) < window:
            n += 1; i += 1; j += 1
        elif d < 0:
            i += 1
        else:
            j += 1
    return n


def evaluate_frames(ref_time, ref_freqs, est_time, est_freqs, window=0.5):
    """Frame-level Precision / Recall / Accuracy and the substitution / miss / false-alarm / total errors, plus their
    chroma (octave-folded) variants.  Both sequences must be on the same time base (they are: same hop)."""
    ref_time, est_time = np.asarray(ref_time, dtype=np.float64), np.asarray(est_time, dtype=np.float64)
    if len(ref_time) != len(est_time) or not np.allclose(ref_time, est_time):
        raise ValueError('evaluate_frames: reference and estimate must share one time base')
    out = {}
    for chroma in (False, True):
        tp = n_ref = n_est = sub = miss = fa = tot = 0
        for rf, ef in zip(ref_freqs, est_freqs):
            rm = _hz_to_midi(rf) if len(rf) else np.array([])
            em = _hz_to_midi(ef) if len(ef) else np.array([])
            if chroma:
                rm, em = np.mod(rm, 12), np.mod(em, 12)
                # circular distance: try both unwrapped copies
                c = max(_count_matches(rm, em, window), _count_matches(rm, np.concatenate([em, em + 12, em - 12]), window)
                        if len(em) else 0)
                c = min(c, len(rm), len(em))
            else:
                c = _count_matches(rm, em, window)
            nr, ne = len(rm), len(em)
            tp += c; n_ref += nr; n_est += ne
            sub += min(nr, ne) - c
            miss += max(0, nr - ne)
            fa += max(0, ne - nr)
            tot += max(nr, ne) - c
        pre = 'Chroma ' if chroma else ''
        out[pre + 'Precision'] = tp / n_est if n_est else 0.0
        out[pre + 'Recall'] = tp / n_ref if n_ref else 0.0
        out[pre + 'Accuracy'] = tp / (n_est + n_ref - tp) if (n_est + n_ref - tp) else 0.0
        out[pre + 'Substitution Error'] = sub / n_ref if n_ref else 0.0
        out[pre + 'Miss Error'] = miss / n_ref if n_ref else 0.0
        out[pre + 'False Alarm Error'] = fa / n_ref if n_ref else 0.0
        out[pre + 'Total Error'] = tot / n_ref if n_ref else 0.0
    return out


# ---------------------------------------------------------------------------------------------
# note metrics (mir_eval.transcription.precision_recall_f1_overlap)
# ---------------------------------------------------------------------------------------------
def match_notes(ref_intervals, ref_pitches, est_intervals, est_pitches, onset_tolerance=0.05, pitch_tolerance=50.0,
                offset_ratio=0.2, offset_min_tolerance=0.05):
    """Maximum one-to-one matching of notes that agree in onset (+-50 ms), pitch (+-50 cents) and -- unless
    offset_ratio is None -- offset (+-max(20 % of the reference duration, 50 ms)).  Returns [(ref_i, est_j), ...]."""
    ref_intervals = np.asarray(ref_intervals, dtype=np.float64).reshape(-1, 2)
    est_intervals = np.asarray(est_intervals, dtype=np.float64).reshape(-1, 2)
    if len(ref_intervals) == 0 or len(est_intervals) == 0:
        return []
    onset_d = np.around(np.abs(np.subtract.outer(ref_intervals[:, 0], est_intervals[:, 0])), N_DECIMALS)
    hit = onset_d <= onset_tolerance
    pitch_d = np.abs(1200.0 * np.subtract.outer(np.log2(ref_pitches), np.log2(est_pitches)))
    hit &= pitch_d <= pitch_tolerance
    if offset_ratio is not None:
        offset_d = np.around(np.abs(np.subtract.outer(ref_intervals[:, 1], est_intervals[:, 1])), N_DECIMALS)
        tol = offset_ratio * (ref_intervals[:, 1] - ref_intervals[:, 0])
        tol[tol <= offset_min_tolerance] = offset_min_tolerance
        hit &= offset_d <= tol.reshape(-1, 1)
    match = maximum_bipartite_matching(csr_matrix(hit), perm_type='column')
    return [(i, int(j)) for i, j in enumerate(match) if j >= 0]


def evaluate_notes(ref_intervals, ref_pitches, est_intervals, est_pitches, onset_tolerance=0.05, pitch_tolerance=50.0,
                   offset_ratio=0.2, offset_min_tolerance=0.05, beta=1.0):
    """(precision, recall, f-measure, average overlap ratio of the matched pairs)."""
    ref_intervals = np.asarray(ref_intervals, dtype=np.float64).reshape(-1, 2)
    est_intervals = np.asarray(est_intervals, dtype=np.float64).reshape(-1, 2)
    if len(ref_pitches) == 0 or len(est_pitches) == 0:
        return 0.0, 0.0, 0.0, 0.0
    m = match_notes(ref_intervals, ref_pitches, est_intervals, est_pitches, onset_tolerance, pitch_tolerance, offset_ratio,
                    offset_min_tolerance)
    p, r = len(m) / len(est_pitches), len(m) / len(ref_pitches)
    f = (1 + beta ** 2) * p * r / (beta ** 2 * p + r) if (p + r) > 0 else 0.0
    ratios = []
    for i, j in m:
        (rs, re_), (es, ee) = ref_intervals[i], est_intervals[j]
        ratios.append((min(re_, ee) - max(rs, es)) / (max(re_, ee) - min(rs, es)))
    return p, r, f, float(np.mean(ratios)) if ratios else 0.0


# ---------------------------------------------------------------------------------------------
# the reference's evaluation loop
# ---------------------------------------------------------------------------------------------
def _to_eval_units(pitches, intervals):
    scaling = HOP_LENGTH / SAMPLE_RATE
    i = (np.asarray(intervals) * scaling).reshape(-1, 2)
    p = np.array([midi_to_hz(MIN_MIDI + midi) for midi in pitches])
    return p, i


def _frames_to_eval_units(t, freqs):
    scaling = HOP_LENGTH / SAMPLE_RATE
    return t.astype(np.float64) * scaling, [np.array([midi_to_hz(MIN_MIDI + midi) for midi in f]) for f in freqs]


def evaluate_wo_velocity(data, model, onset_threshold=0.5, frame_threshold=0.5, save_path=None, reconstruction=True,
                         onset=True, pseudo_onset=False, rule='rule2', VAT=False):
    """model/evaluate_functions.py:20-127: whole-song evaluation; returns a dict of lists with the reference's keys
    (losses, metric/note/*, metric/note-with-offsets/*, metric/frame/*, metric/MusicNet/micro_avg_P, and the *_2
    variants of the reconstruction pass).  ``save_path``: per song `<basename>.pred.mid` (the transcription, reconvat_amd/midi.py)
    and the `<basename>.label.png` / `.pred.png` piano rolls (model/utils.py:61-80), as the reference writes them (:119-126)."""
    from sklearn.metrics import average_precision_score
    metrics = defaultdict(list)
    for label in data:
        pred, losses, _ = model.run_on_batch(label, None, False) if VAT else model.run_on_batch(label)
        for key, loss in losses.items():
            metrics[key].append(loss.item())
        for key in ('frame', 'onset', 'frame2', 'onset2'):
            if pred.get(key) is not None:
                pred[key] = pred[key].detach().squeeze(0).relu()
        lab_on, lab_fr = label['onset'].squeeze(0), label['frame'].squeeze(0)
        if onset:
            p_ref, i_ref = extract_notes_wo_velocity(lab_on, lab_fr, rule=rule)
            p_est, i_est = extract_notes_wo_velocity(lab_on if pseudo_onset else pred['onset'], pred['frame'], onset_threshold,
                                                     frame_threshold, rule=rule)
        else:
            p_ref, i_ref = extract_notes_wo_velocity(lab_fr, lab_fr, rule=rule)
            p_est, i_est = extract_notes_wo_velocity(pred['frame'], pred['frame'], onset_threshold, frame_threshold, rule=rule)
        t_ref, f_ref = _frames_to_eval_units(*notes_to_frames(p_ref, i_ref, lab_fr.shape))
        t_est, f_est = _frames_to_eval_units(*notes_to_frames(p_est, i_est, pred['frame'].shape))
        p_ref, i_ref = _to_eval_units(p_ref, i_ref)
        p_est, i_est = _to_eval_units(p_est, i_est)

        def note_block(suffix, pe, ie):
            p, r, f, o = evaluate_notes(i_ref, p_ref, ie, pe, offset_ratio=None)
            for k, v in zip(('precision', 'recall', 'f1', 'overlap'), (p, r, f, o)):
                metrics[f'metric/note/{k}{suffix}'].append(v)
            p, r, f, o = evaluate_notes(i_ref, p_ref, ie, pe)
            for k, v in zip(('precision', 'recall', 'f1', 'overlap'), (p, r, f, o)):
                metrics[f'metric/note-with-offsets/{k}{suffix}'].append(v)

        note_block('', p_est, i_est)
        frame_metrics = evaluate_frames(t_ref, f_ref, t_est, f_est)
        metrics['metric/frame/f1'].append(hmean([frame_metrics['Precision'] + eps, frame_metrics['Recall'] + eps]) - eps)
        metrics['metric/MusicNet/micro_avg_P'].append(
            average_precision_score(lab_fr.cpu().flatten().numpy(), pred['frame'].cpu().flatten().numpy()))
        if reconstruction and pred.get('frame2') is not None:
            p2, i2 = extract_notes_wo_velocity(pred['onset2'], pred['frame2'], onset_threshold, frame_threshold)
            t2, f2 = _frames_to_eval_units(*notes_to_frames(p2, i2, pred['frame2'].shape))
            p2, i2 = _to_eval_units(p2, i2)
            note_block('_2', p2, i2)
            fm2 = evaluate_frames(t_ref, f_ref, t2, f2)
            frame_metrics['Precision_2'], frame_metrics['Recall_2'], frame_metrics['accuracy_2'] = \
                fm2['Precision'], fm2['Recall'], fm2['Accuracy']
            metrics['metric/frame/f1_2'].append(hmean([fm2['Precision'] + eps, fm2['Recall'] + eps]) - eps)
            metrics['metric/MusicNet/micro_avg_P2'].append(
                average_precision_score(lab_fr.cpu().flatten().numpy(), pred['frame2'].cpu().flatten().numpy()))
        for key, value in frame_metrics.items():
            metrics['metric/frame/' + key.lower().replace(' ', '_')].append(value)
        if save_path is not None:
            os.makedirs(save_path, exist_ok=True)
            path = label['path'][0] if isinstance(label['path'], (list, tuple)) else label['path']
            stem = os.path.join(save_path, os.path.basename(str(path)))
            save_pianoroll(stem + '.label.png', lab_on, lab_fr)
            save_pianoroll(stem + '.pred.png', pred['onset'] if pred.get('onset') is not None else pred['frame'], pred['frame'])
            save_midi(stem + '.pred.mid', p_est, i_est, [127] * len(p_est))
    return metrics


def save_pianoroll(path, onsets, frames, onset_threshold=0.5, frame_threshold=0.5, zoom=4):
    """model/utils.py:61-80: RGB piano-roll diagram (onsets / frames / both, pitch upwards, `zoom` x stretched).  Needs PIL; without
    it the diagram is skipped (the MIDI file and the metrics do not depend on it)."""
    try:
        from PIL import Image
    except ImportError:
        return False
    on = (1 - (onsets.t() > onset_threshold).to(torch.uint8)).cpu()
    fr = (1 - (frames.t() > frame_threshold).to(torch.uint8)).cpu()
    both = 1 - (1 - on) * (1 - fr)
    image = torch.stack([on, fr, both], dim=2).flip(0).mul(255).numpy()
    image = Image.fromarray(image, 'RGB')
    image.resize((image.size[0], image.size[1] * zoom)).save(path)
    return True
