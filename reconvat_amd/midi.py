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


# ------------------------------------------------------------------------------------------------------------------
# reading: Standard MIDI file -> note list (the corpus side: MAESTRO ships .midi, the datasets want onset/offset/note/velocity)
# ------------------------------------------------------------------------------------------------------------------
def _read_varlen(buf, i):
    n = 0
    while True:
        b = buf[i]
        i += 1
        n = (n << 7) | (b & 0x7F)
        if not b & 0x80:
            return n, i


def _track_events(buf):
    """(absolute tick, kind, a, b) for the channel-voice and tempo events of one MTrk chunk (running status handled)."""
    i, tick, status, out = 0, 0, 0, []
    while i < len(buf):
        delta, i = _read_varlen(buf, i)
        tick += delta
        first = buf[i]
        if first == 0xFF:                                   # meta
            kind = buf[i + 1]
            n, j = _read_varlen(buf, i + 2)
            if kind == 0x51 and n == 3:
                out.append((tick, 'tempo', int.from_bytes(buf[j:j + 3], 'big'), 0))
            i = j + n
            continue
        if first in (0xF0, 0xF7):                           # sysex
            n, j = _read_varlen(buf, i + 1)
            i = j + n
            continue
        if first & 0x80:
            status = first
            i += 1
        hi = status & 0xF0
        width = 1 if hi in (0xC0, 0xD0) else 2
        a, b = buf[i], (buf[i + 1] if width == 2 else 0)
        i += width
        if hi in (0x80, 0x90):
            out.append((tick, 'note', a, b if hi == 0x90 else 0))
        elif hi == 0xB0 and a == 64:
            out.append((tick, 'pedal', b, 0))
    return out


def read_smf(path):
    """All tracks of a format-0/1 file merged in time order (stable across tracks), ticks converted to seconds through the
    tempo map: [(seconds, kind, a, b)] with kind in {'note' (a = pitch, b = velocity, 0 = off), 'pedal' (a = CC64 value)}."""
    with open(path, 'rb') as fh:
        data = fh.read()
    if data[:4] != b'MThd':
        raise ValueError(f'{path}: not a Standard MIDI file')
    hlen, _fmt, ntracks, division = struct.unpack('>IHHH', data[4:14])
    if division & 0x8000:
        raise ValueError(f'{path}: SMPTE time division is not supported')
    pos, merged = 8 + hlen, []
    for t in range(ntracks):
        if data[pos:pos + 4] != b'MTrk':
            raise ValueError(f'{path}: bad track chunk')
        n = struct.unpack('>I', data[pos + 4:pos + 8])[0]
        merged += [(tick, t, k) + tuple(ev) for k, (tick, *ev) in enumerate(_track_events(data[pos + 8:pos + 8 + n]))]
        pos += 8 + n
    merged.sort(key=lambda e: (e[0], e[1], e[2]))
    out, seconds, last_tick, tempo = [], 0.0, 0, 500000
    for tick, _t, _k, kind, a, b in merged:
        seconds += (tick - last_tick) * tempo * 1e-6 / division
        last_tick = tick
        if kind == 'tempo':
            tempo = a
        else:
            out.append((seconds, kind, a, b))
    return out


def parse_midi(path):
    """np.array of (onset s, offset s, note, velocity) rows with the reference's pairing rule (model/midi.py:12-50): a note
    ends at the next event on its pitch (note-off, zero-velocity note-on or a re-strike); if the sustain pedal (CC64 >= 64)
    is down at that moment the note rings until the pedal is released; whatever is still open ends at the last event."""
    timeline, pedal_down = [], False                       # (seconds, pitch or None, velocity, pedal state)
    for seconds, kind, a, b in read_smf(path):
        if kind == 'pedal':
            if (a >= 64) != pedal_down:
                pedal_down = a >= 64
                timeline.append((seconds, None, 0, 'down' if pedal_down else 'up'))
        else:
            timeline.append((seconds, a, b, pedal_down))
    last = len(timeline) - 1
    notes = []
    for i, (t_on, pitch, velocity, _state) in enumerate(timeline):
        if pitch is None or velocity == 0:
            continue
        j = next((k for k in range(i + 1, last + 1) if timeline[k][1] == pitch), last)
        if j != last and timeline[j][3] is True:           # released under the pedal: ring until it comes up
            j = next((k for k in range(j + 1, last + 1) if timeline[k][3] == 'up'), last)
        notes.append((t_on, timeline[j][0], pitch, velocity))
    return np.array(notes)
