#!/usr/bin/env python
"""Transcribe audio files to MIDI with a trained model (reference transcribe_files.py:12-69).

    python transcribe_files.py with device=cuda:0 weight=runs/.../model-final.pt input=Application/Input output=Application/Output

Inputs: 16 kHz mono 16-bit ``.wav`` files or ``.pt`` track caches (dict with an int16 ``audio`` tensor).
"""
import os
import sys
import wave

import numpy as np
import torch

import reconvat_amd as ra
from reconvat_amd.constants import HOP_LENGTH, SAMPLE_RATE, MIN_MIDI
from reconvat_amd.decoding import extract_notes_wo_velocity
from reconvat_amd.evaluate import midi_to_hz
from reconvat_amd.midi import save_midi
from reconvat_amd.sacred_lite import parse_cli


def load_audio(path):
    if path.endswith('.pt'):
        return torch.load(path)['audio'].float().div(32768.0)
    with wave.open(path, 'rb') as w:
        assert w.getframerate() == SAMPLE_RATE and w.getsampwidth() == 2, f'{path}: need 16 kHz 16-bit PCM'
        x = np.frombuffer(w.readframes(w.getnframes()), dtype=np.int16).reshape(-1, w.getnchannels()).mean(axis=1)
    return torch.from_numpy(x.astype(np.float32) / 32768.0)


def transcribe2midi(files, model, device, out_dir, onset_threshold=0.5, frame_threshold=0.5, rule='rule2', tag='ReconVAT'):
    os.makedirs(out_dir, exist_ok=True)
    for path in files:
        audio = load_audio(path).to(device)
        with torch.no_grad():
            pred = model.transcribe({'audio': audio.unsqueeze(0)})
        onset, frame = pred['onset'].squeeze(0).relu(), pred['frame'].squeeze(0).relu()
        p_est, i_est = extract_notes_wo_velocity(onset, frame, onset_threshold, frame_threshold, rule=rule)
        scaling = HOP_LENGTH / SAMPLE_RATE
        i_est = (np.asarray(i_est) * scaling).reshape(-1, 2)
        p_est = np.array([midi_to_hz(MIN_MIDI + m) for m in p_est])
        midi_path = os.path.join(out_dir, tag + '-' + os.path.splitext(os.path.basename(path))[0] + '.mid')
        save_midi(midi_path, p_est, i_est, [127] * len(p_est))
        print(f'midi_path = {midi_path}  ({len(p_est)} notes)')


def main(argv):
    cfg = dict(device='cuda:0', weight=None, input='Application/Input', output='Application/Output')
    cfg.update(parse_cli(argv))
    model = ra.UNet((2, 2), (2, 2), log=True, reconstruction=True, mode='imagewise', spec='Mel', device=cfg['device'])
    if cfg['weight']:
        model.load_state_dict(torch.load(cfg['weight'], map_location='cpu'))
    model.to(cfg['device']).eval()
    files = sorted(os.path.join(cfg['input'], f) for f in os.listdir(cfg['input']) if f.endswith(('.wav', '.pt')))
    transcribe2midi(files, model, cfg['device'], cfg['output'])


if __name__ == '__main__':
    main(sys.argv[1:])
