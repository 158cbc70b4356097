"""Standard MIDI file export of transcribed notes (inference surface).

Same contract as the reference's ``save_midi`` (model/midi.py:53-83), which builds the file with the `mido` package
(not installed here): one track, 480 ticks per beat at the default 120 bpm (so 960 ticks per second), a note_on at every
onset and a note_off at every offset, events in time order, velocity = min(127, int(v * 127)), pitch = round(MIDI number
of the frequency).  The bytes are written directly (format-1 SMF, no running status, end-of-track meta event).
PARITY UNPINNED against mido's exact byte stream (mido is absent); the test parses the file back and checks the events.
"""
import struct

import numpy as np

TICKS_PER_BEAT = 480
TICKS_PER_SECOND = TICKS_PER_BEAT * 2.0          # 120 bpm default tempo


def hz_to_midi(freq):
    return 12.0 * (np.log2(np.asarray(freq, dtype=np.float64)) - np.log2(440.0)) + 69.0


def _varlen(n):
    out = [n & 0x7F]
    n >>= 7
    while n:
        out.append((n & 0x7F) | 0x80)
        n >>= 7
    return bytes(reversed(out))


def save_midi(path, pitches, intervals, velocities):
    """pitches: frequencies in Hz; intervals: [(onset_s, offset_s)]; velocities: 0..1 floats (or 127 for 'loud')."""
    events = []
    for p, (t0, t1), v in zip(pitches, intervals, velocities):
        events.append((float(t0), 0x90, p, v))
        events.append((float(t1), 0x80, p, v))
    events.sort(key=lambda e: e[0])              # stable: ties keep insertion order, like the reference's list.sort
    body = bytearray()
    last = 0
    for t, status, p, v in events:
        tick = int(t * TICKS_PER_SECOND)
        vel = min(127, int(v * 127))
        note = int(round(float(hz_to_midi(p))))
        body += _varlen(max(0, tick - last)) + bytes([status, note & 0x7F, vel & 0x7F])
        last = tick
    body += b'\x00\xff\x2f\x00'
    with open(path, 'wb') as f:
        f.write(b'MThd' + struct.pack('>IHHH', 6, 1, 1, TICKS_PER_BEAT))
        f.write(b'MTrk' + struct.pack('>I', len(body)) + bytes(body))
