"""Step time of the Onsets&Frames BiLSTM baseline at the reference script's sizes: B_l = B_ul = 8 segments
of 327 680 samples, VAT on both groups, Adam + clip -- the same step definition as bench.py, for the second model family.
Prints one JSON line (audio-s/s) and, with --lstm, the isolated BiLSTM forward/backward launch times.

    python tools/bench_onf.py [--steps 20] [--warmup 5] [--no-graph] [--lstm]
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--no-graph', action='store_true')
    ap.add_argument('--no-vat', action='store_true')
    ap.add_argument('--lstm', action='store_true')
    ap.add_argument('--single-stream', action='store_true')
    args = ap.parse_args()
    import reconvat_amd as ra
    from reconvat_amd import ops
    from reconvat_amd.onset_frames import OnsetsAndFrames_VAT_full
    dev = torch.device('cuda:0')
    torch.manual_seed(0)
    if args.lstm:
        for (i, h) in ((768, 384), (176, 384)):
            x = torch.randn(8, 640, i, device=dev, requires_grad=True)
            lstm = torch.nn.LSTM(i, h, batch_first=True, bidirectional=True).to(dev)
            ps = [getattr(lstm, n + s) for s in ('', '_reverse') for n in ('weight_ih_l0', 'weight_hh_l0', 'bias_ih_l0', 'bias_hh_l0')]
            gy = torch.randn(8, 640, 2 * h, device=dev)
            for _ in range(2):
                y = ops.BiLstmFn.apply(x, *ps); y.backward(gy)
            torch.cuda.synchronize()
            e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
            e[0].record(); y = ops.BiLstmFn.apply(x, *ps); e[1].record(); y.backward(gy); e[2].record()
            torch.cuda.synchronize()
            print(f'BiLSTM {i}->{2 * h}, B=8, T=640: fwd {e[0].elapsed_time(e[1]):.3f} ms  bwd {e[1].elapsed_time(e[2]):.3f} ms '
                  f'({e[0].elapsed_time(e[1]) / 640 * 1e3:.2f} / {e[1].elapsed_time(e[2]) / 640 * 1e3:.2f} us per step incl. GEMMs)', file=sys.stderr)
    m = OnsetsAndFrames_VAT_full(229, 88, XI=1e-6, eps=1e-1).to(dev)
    g = torch.Generator().manual_seed(1)

    def batch():
        return {'audio': (torch.rand(8, 327680, generator=g) * 0.2 - 0.1).to(dev),
                'frame': (torch.rand(8, 640, 88, generator=g) > 0.95).float().to(dev),
                'onset': (torch.rand(8, 640, 88, generator=g) > 0.99).float().to(dev)}
    bl, bul = batch(), batch()
    opt = ra.FlatAdam(m.parameters(), lr=5e-4, step_size=10000, gamma=0.98)
    vat = not args.no_vat
    step = ra.TrainStep(m, opt, bl, bul if vat else None, alpha=1.0, VAT=vat, clip=3.0, graph=not args.no_graph,
                        dual_stream=not args.single_stream)
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.steps
    ops.lstm_check(dev)              # raises if any recurrence launch timed out
    audio_s = (16 if vat else 8) * 327680 / 16000 / dt
    print(json.dumps({'metric': 'training audio-sec/sec (1 GPU), Onsets&Frames BiLSTM baseline' + (' VAT' if vat else ''),
                      'value': round(audio_s, 1), 'unit': 'audio-s/s', 'ms_per_step': round(dt * 1e3, 3), 'steps': args.steps,
                      'warmup': args.warmup, 'hipgraph': not args.no_graph, 'two_stream_schedule': not args.single_stream, 'dtype': 'f32', 'data': 'synthetic',
                      'final_loss': round(float(step.loss), 5), 'lstm_timeout_flag': 0}))


if __name__ == '__main__':
    main()
