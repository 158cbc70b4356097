"""Note decoding of the posteriorgrams (evaluation / inference path).

Same results as the reference's ``model/decoding.py`` (pinned by tests/golden/decoding.npz, produced by running the
reference functions), computed without the reference's per-note Python ``while`` loop: the end of every note is a
lookup in a per-pitch "next inactive frame" table built with one reverse cumulative minimum.
"""
import numpy as np
import torch


def _binarise(onsets, frames, onset_threshold, frame_threshold):
    on = (torch.as_tensor(onsets) > onset_threshold).cpu().numpy()
    fr = (torch.as_tensor(frames) > frame_threshold).cpu().numpy()
    return on, fr


def _note_ends(active):
    """next_off[t, p] = smallest t' >= t with active[t', p] == False (T if none)."""
    T = active.shape[0]
    idx = np.where(~active, np.arange(T)[:, None], T)
    return np.minimum.accumulate(idx[::-1], axis=0)[::-1]


def extract_notes_wo_velocity(onsets, frames, onset_threshold=0.5, frame_threshold=0.5, rule='rule1'):
    """model/decoding.py:4-56.  A note starts where the thresholded onset roll rises (rule1: and the frame roll is on)
    and lasts while the onset OR the frame roll stays on.  Returns (pitches [N], intervals [N, 2]) in frame units,
    ordered by (onset frame, pitch) like the reference."""
    if rule not in ('rule1', 'rule2'):
        raise NameError('Please enter the correct rule name')
    on, fr = _binarise(onsets, frames, onset_threshold, frame_threshold)
    onset_diff = np.concatenate([on[:1], on[1:] & ~on[:-1]], axis=0)
    if rule == 'rule1':
        onset_diff = onset_diff & fr
    t, p = np.nonzero(onset_diff)
    ends = _note_ends(on | fr)[t, p]
    keep = ends > t
    pitches = p[keep]
    intervals = np.stack([t[keep], ends[keep]], axis=1) if keep.any() else np.array([])
    return (pitches if keep.any() else np.array([])), intervals


def extract_notes(onsets, frames, velocity, onset_threshold=0.5, frame_threshold=0.5):
    """model/decoding.py:59-107: as above without the frame condition, plus the mean velocity over the note's
    onset-active frames."""
    on, fr = _binarise(onsets, frames, onset_threshold, frame_threshold)
    vel = torch.as_tensor(velocity).cpu().numpy()
    onset_diff = np.concatenate([on[:1], on[1:] & ~on[:-1]], axis=0)
    t, p = np.nonzero(onset_diff)
    ends = _note_ends(on | fr)[t, p]
    pitches, intervals, velocities = [], [], []
    for a, b, q in zip(t, ends, p):
        if b > a:
            m = on[a:b, q]
            pitches.append(q)
            intervals.append([a, b])
            velocities.append(float(np.mean(vel[a:b, q][m])) if m.any() else 0)
    return np.array(pitches), np.array(intervals), np.array(velocities)


def notes_to_frames(pitches, intervals, shape):
    """model/decoding.py:109-130: piano roll of the notes -> (frame indices, list of active-bin arrays)."""
    roll = np.zeros(tuple(shape))
    for pitch, (onset, offset) in zip(pitches, intervals):
        roll[onset:offset, pitch] = 1
    time = np.arange(roll.shape[0])
    freqs = [roll[t, :].nonzero()[0] for t in time]
    return time, freqs
