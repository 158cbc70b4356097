#!/usr/bin/env python
"""Headline benchmark: training audio-seconds per second of the ReconVAT U-Net+Onset VAT+reconstruction
step (BASELINE.json metric) on N MI355X GPUs of one node, one process per GPU.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python bench.py --gpus N --steps K --warmup W          (self-launching: spawns N fresh rank processes)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A "step" is one pass of the reference loop body (model/helper_functions.py:577-607) over one synthetic
batch: zero_grad, log-Mel front-end of 8 labelled + 8 unlabelled 327 680-sample segments, VAT power
iteration on both, transcriber -> reconstructor -> transcriber, the 11 losses, backward, [gradient
all-reduce], Adam + StepLR, post-step gradient clip.  Inputs are resident in HBM before the timed region.
Rank 0 prints ONE JSON line; at N=1 it also carries the conv-phase roofline (measured with HIP events on
the launch stream) and the CPU baseline (the oracle timed on the host cores, bounded sample).
"""
import argparse
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

SEG = 327680
SEG_SECONDS = SEG / 16000.0
MFMA_F32_PEAK_TFLOPS = 157.3          # /opt/skills/guides/MI355X_MICROARCH.md: FP32 matrix peak (dense)


def synthetic_batch(b, gen, device):
    """Seeded synthetic segments: audio ~ U(-0.1, 0.1), ~5 % frame / ~1 % onset labels."""
    audio = (torch.rand(b, SEG, generator=gen) * 0.2 - 0.1)
    u = torch.rand(b, SEG // 512, 88, generator=gen)
    return {'audio': audio.to(device), 'frame': (u > 0.95).float().to(device), 'onset': (u > 0.99).float().to(device)}


# ---------------------------------------------------------------------------------------------
# conv-phase roofline: every conv launch of one step is recorded (shapes only), then each distinct
# launch is re-issued back-to-back between two HIP events on the launch stream.
# ---------------------------------------------------------------------------------------------
def conv_flops(name, a):
    if name == 'rv_conv_fwd':
        mode, b, h, w, cin, ho, wo, cout = a[0], a[3], a[4], a[5], a[6], a[9], a[10], a[11]
        taps = {0: 9, 1: 1, 2: 4, 3: 1}[mode]
        pix = b * h * w * 4 if mode == 3 else b * ho * wo
        return 2.0 * pix * cin * cout * taps
    mode, ca, hv, wv, cb, b = a[0] & 0xff, a[5], a[8], a[9], a[10], a[11]          # (bit 8 of mode: bf16 operands)
    return 2.0 * b * hv * wv * ca * cb * {0: 9, 1: 1, 2: 4}[mode]


def conv_bytes(name, a):
    """Algorithmic HBM bytes of one conv launch: input + output (+ weights), fp32, every tensor touched once."""
    if name == 'rv_conv_fwd':
        mode, b, h, w, cin, ho, wo, cout = a[0], a[3], a[4], a[5], a[6], a[9], a[10], a[11]
        taps = {0: 9, 1: 1, 2: 4, 3: 4}[mode]
        return 4.0 * (b * h * w * cin + b * ho * wo * cout + taps * cin * cout)
    mode, hu, wu, ca, hv, wv, cb, b = a[0] & 0xff, a[3], a[4], a[5], a[8], a[9], a[10], a[11]
    return 4.0 * (b * hu * wu * ca + b * hv * wv * cb + {0: 9, 1: 1, 2: 4}[mode] * ca * cb)


def conv_is_bf16(name, a):
    return bool((a[15] >> 20) & 1) if name == 'rv_conv_fwd' else bool(a[0] & 0x100)


MFMA_BF16_PEAK_TFLOPS = 2500.0        # /opt/skills/guides/MI355X_MICROARCH.md: dense bf16 matrix peak
CONV_ENTRY_POINTS = ('rv_conv_fwd', 'rv_conv_wgrad', 'rv_conv_wgrad_deferred', 'rv_wgrad_reduce_table')
ROTATE_BYTES = 320 << 20              # operand sets of one timed launch are rotated until they total more than the 256 MiB Infinity Cache


def plain_wgrad(name, a):
    """Book-keeping form of a SEGMENTED weight-gradient launch (rv_conv_wgrad_seg / _deferred_seg: the (x, dY) pairs of several backward
    passes of one layer in one launch): the argument list of the plain entry point over nseg * Bseg images -- same kernel, same work, same
    time; the segment pointer arrays (host memory of the caller) are not kept."""
    if name == 'rv_conv_wgrad_seg':
        return 'rv_conv_wgrad', (a[0], 0, a[4], a[5], a[6], a[7], 0, a[8], a[9], a[10], a[11], a[1] * a[12]) + tuple(a[13:])
    if name == 'rv_conv_wgrad_deferred_seg':
        return 'rv_conv_wgrad_deferred', (a[0], 0, a[4], a[5], a[6], a[7], 0, a[8], a[9], a[10], a[11], a[1] * a[12]) + tuple(a[13:])
    return name, a


def record_launches(step_fn, names):
    """Run step_fn once with the library's launch hook installed; returns [(entry point, args)] of the launches in `names`."""
    from reconvat_amd import _lib
    records = []

    def hook(name, args, fn):
        pname, pargs = plain_wgrad(name, args)
        if pname in names:
            records.append((pname, pargs))
        return fn(*args)
    prev = _lib.HOOK[0]
    _lib.HOOK[0] = hook
    try:
        step_fn()
        torch.cuda.synchronize()
    finally:
        _lib.HOOK[0] = prev
    return records


def measure_conv_phase(step_fn, device):
    """Isolated figure: every distinct conv launch of one step re-issued back to back between two HIP events on the launch
    stream -- on ROTATING operand sets (>= 3 sets, together larger than the 256 MiB Infinity Cache), so that no launch finds
    its operands warm from the previous repetition (colder than the real step, where a conv's input was just written by its
    producer: a conservative figure)."""
    from reconvat_amd import _lib
    lib = _lib.load()
    # the recorded step runs the weight gradients in their immediate form so that every conv launch goes through the recorded
    # entry points; they are then TIMED the way the timed step executes them: the partial-sum kernel of each layer
    # (rv_conv_wgrad_deferred) plus the table launches that run all per-layer reductions of a backward pass at once
    # (rv_wgrad_reduce_table, one per chain: timed below with every reduction of the step in two tables)
    prev_defer = os.environ.get('RV_DEFER_WGRAD')
    os.environ['RV_DEFER_WGRAD'] = '0'
    try:
        step_fn()                  # (two steps first: the weight-gradient merger learns the pass counts of this step object's two modes ...
        step_fn()
        records = record_launches(step_fn, ('rv_conv_fwd', 'rv_conv_wgrad'))      # ... so the recorded step launches what the timed step launches)
    finally:
        if prev_defer is None:
            del os.environ['RV_DEFER_WGRAD']
        else:
            os.environ['RV_DEFER_WGRAD'] = prev_defer
    # group identical launches by their shape signature (pointers and stream excluded)
    groups = {}
    for name, a in records:
        if name == 'rv_conv_fwd':
            sig = (name, a[0]) + tuple(a[2:7]) + tuple(a[8:12]) + (a[15], ('bnbwd' if a[17] else 'bn') if a[16] else '')
        else:
            sig = (name, a[0]) + tuple(a[2:6]) + tuple(a[7:12])        # (a[0] carries the bf16 bit: bf16 launches are their own group)
        g = groups.setdefault(sig, {'count': 0, 'name': name, 'args': a})
        g['count'] += 1
    arena = torch.empty((3 << 30) // 4, device=device)             # 3 GiB of scratch operands
    arena.uniform_(-1, 1)
    base = arena.data_ptr()
    ws = torch.empty(256 * 1024 * 1024 // 4, device=device)
    st = torch.cuda.current_stream()
    eb = lib.rv_wgrad_table_entry_bytes()
    entry_scratch = torch.empty(eb, dtype=torch.uint8)

    def deferrable(a):           # mirrors ops.conv_wgrad: accumulating calls, except the 1 -> many 3x3 layer
        return bool(a[17]) and not ((a[0] & 0xff) == 0 and a[5] == 1 and a[10] > 16)

    def deferred_fn(*a):
        return 0 if lib.rv_conv_wgrad_deferred(*a) > 0 else -1

    def up(n):
        return (n + 4095) & ~4095

    total_ms, total_flops, per_kernel, min_sets = 0.0, 0.0, [], 1 << 30
    # per-launch bound of SURVEY 8(d): min(matrix peak of the launch's operand type, arithmetic intensity x HBM bandwidth)
    bound = {'all': [0.0, 0.0, 0.0], 'bf16': [0.0, 0.0, 0.0]}       # [measured ms, ideal ms at the bound, flops]
    split = {'hbm_bound': [0.0, 0.0, 0.0, 0], 'mfma_bound': [0.0, 0.0, 0.0, 0]}      # [ms, flops, algorithmic bytes, launches] per step
    bound['split'] = split
    for sig, g in groups.items():
        a0 = list(g['args'])
        if g['name'] == 'rv_conv_fwd':
            b_, h_, w_, ild, ho, wo, old = a0[3], a0[4], a0[5], a0[2], a0[9], a0[10], a0[8]
            sz = {'in': up(4 * b_ * h_ * w_ * ild), 'out': up(4 * b_ * ho * wo * old), 'w': up(4 << 20),
                  'z': up(4 * b_ * ho * wo * a0[18]) if a0[17] else 0, 'coef': up(1 << 16) if a0[17] else 0}
        else:
            uld, hu, wu, vld, hv, wv, b_ = a0[2], a0[3], a0[4], a0[7], a0[8], a0[9], a0[11]
            sz = {'in': up(4 * b_ * hu * wu * uld), 'out': up(4 * b_ * hv * wv * vld), 'w': up(4 << 20), 'z': 0, 'coef': 0}
        foot = sum(sz.values())
        nsets = max(3, min(64, -(-ROTATE_BYTES // foot) + 1, (3 << 30) // foot))
        min_sets = min(min_sets, nsets)
        calls = []
        for s_ in range(nsets):
            a = list(a0)
            o = base + s_ * foot
            if g['name'] == 'rv_conv_fwd':
                a[1], a[7], a[12], a[13] = o, o + sz['in'], o + sz['in'] + sz['out'], None
                if a[16]:
                    a[16] = ws.data_ptr()                      # fused BatchNorm statistics: any fp64 scratch
                if a[17]:
                    a[17] = o + sz['in'] + sz['out'] + sz['w']
                    a[19] = a[17] + sz['z']                    # z / coefficients: scratch
                a[-1] = st.cuda_stream
                fn = lib.rv_conv_fwd
            else:
                a[1], a[6], a[12], a[16], a[18] = o, o + sz['in'], o + sz['in'] + sz['out'], None, ws.data_ptr()
                a[19] = ws.numel() * 4
                a[-1] = st.cuda_stream
                fn = lib.rv_conv_wgrad
                if deferrable(a):
                    a = a[:17] + a[18:20] + [entry_scratch.data_ptr(), st.cuda_stream]      # (no `acc`; entry slot before the stream)
                    fn = deferred_fn
            calls.append((fn, a))
        for fn, a in calls[:2]:
            if fn(*a) != 0:
                raise RuntimeError(f"{g['name']} {sig}: {_lib.last_error()}")
        reps = max(6, nsets)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st)
        for i in range(reps):
            fn, a = calls[i % nsets]
            fn(*a)
        e1.record(st)
        e1.synchronize()
        ms = e0.elapsed_time(e1) / reps
        fl = conv_flops(g['name'], g['args'])
        total_ms += ms * g['count']
        total_flops += fl * g['count']
        is_bf = conv_is_bf16(g['name'], g['args'])
        peak = (MFMA_BF16_PEAK_TFLOPS if is_bf else MFMA_F32_PEAK_TFLOPS) * 1e12
        cbytes = conv_bytes(g['name'], g['args'])
        lim = min(peak, fl / cbytes * HBM_PEAK_TBS * 1e12)
        half = 'hbm_bound' if lim < peak else 'mfma_bound'          # which roof is the lower one for this launch
        split[half][0] += ms * g['count']
        split[half][1] += fl * g['count']
        split[half][2] += cbytes * g['count']
        split[half][3] += g['count']
        for key in (('all', 'bf16') if is_bf else ('all',)):
            bound[key][0] += ms * g['count']
            bound[key][1] += fl / lim * 1e3 * g['count']
            bound[key][2] += fl * g['count']
        per_kernel.append((ms * g['count'], g['count'], ms, fl / ms / 1e9, sig))
    # the batched reductions: every deferrable weight gradient of the step, split over two tables like the two chains of the step
    wg = [list(r[1]) for r in records if r[0] == 'rv_conv_wgrad' and deferrable(r[1])]
    ntab = 0
    if wg:
        halves = [wg[:len(wg) // 2], wg[len(wg) // 2:]]
        tables = []
        off = 0
        for half in halves:
            if not half:
                continue
            host = torch.empty(len(half) * eb, dtype=torch.uint8)
            for i, a in enumerate(half):
                a[1], a[6] = base, base + (512 << 20)
                a[12], a[16] = base + (1 << 30), None
                nbytes = lib.rv_conv_wgrad_workspace_bytes({0: 9, 1: 1, 2: 4}[a[0] & 0xff], a[11], a[8], a[5], a[10])
                if (off + nbytes) > ws.numel() * 4:
                    off = 0                                  # (scratch partial sums may alias: only the timing matters)
                da = a[:17] + [ws.data_ptr() + off, nbytes, host.data_ptr() + i * eb, st.cuda_stream]
                off += (nbytes + 255) & ~255
                if lib.rv_conv_wgrad_deferred(*da) <= 0:
                    raise RuntimeError('rv_conv_wgrad_deferred: ' + _lib.last_error())
            total_blocks = lib.rv_wgrad_table_finalize(host.data_ptr(), len(half))
            tables.append((host.to(device), len(half), total_blocks))
        torch.cuda.synchronize()
        for _ in range(2):
            for t, n, tb in tables:
                lib.rv_wgrad_reduce_table(t.data_ptr(), n, tb, st.cuda_stream)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st)
        for _ in range(5):
            for t, n, tb in tables:
                lib.rv_wgrad_reduce_table(t.data_ptr(), n, tb, st.cuda_stream)
        e1.record(st)
        e1.synchronize()
        ms = e0.elapsed_time(e1) / 5
        total_ms += ms
        ntab = len(tables)
        per_kernel.append((ms, ntab, ms / ntab, 0.0, ('rv_wgrad_reduce_table', f'{len(wg)} reductions in {ntab} launches')))
    per_kernel.sort(reverse=True)
    return total_ms, total_flops, per_kernel, len(records) + ntab, min_sets, bound


def measure_in_situ(step_fn, device):
    """In-situ figure: ONE eager single-stream step with every launch of the kernel families bracketed by two HIP events on
    the launch stream (the operands are the step's own: as warm or cold as the step leaves them).  Returns
    ({family: (ms, launches)}, event-bracket floor in us).  A bracket contains the dispatch gap behind the previous launch
    (the floor: a bracket around a one-workgroup kernel), so the sums are upper bounds of rocprofv3's kernel durations."""
    from reconvat_amd import _lib
    fam_of = dict(FAMILY)
    fam_of.update({n: 'convolutions' for n in CONV_ENTRY_POINTS})
    brackets = []

    def hook(name, args, fn):
        pname, pargs = plain_wgrad(name, args)
        f = fam_of.get(pname)
        if f is None:
            return fn(*args)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        rc = fn(*args)
        e1.record()
        brackets.append((f, pname, pargs, e0, e1))
        return rc
    prev = _lib.HOOK[0]
    _lib.HOOK[0] = hook
    try:
        step_fn()
        torch.cuda.synchronize()
    finally:
        _lib.HOOK[0] = prev
    out, flops = {}, 0.0
    for f, name, args, e0, e1 in brackets:
        ms, n = out.get(f, (0.0, 0))
        out[f] = (ms + e0.elapsed_time(e1), n + 1)
        if name in ('rv_conv_fwd', 'rv_conv_wgrad'):
            flops += conv_flops(name, args)
        elif name == 'rv_conv_wgrad_deferred':
            flops += conv_flops('rv_conv_wgrad', args)
    # floor of a bracket: the same two events around a one-workgroup kernel, issued while the queue is still busy (as in the
    # step, where the host runs ahead of the device): dispatch gap + event markers + a ~1 us kernel
    lib = _lib.load()
    cnt = torch.zeros(1, dtype=torch.int64, device=device)
    busy = torch.empty(1 << 28, device=device)
    floors = []
    for _ in range(8):
        busy.uniform_()                      # ~2-3 ms of queued work: the brackets below are enqueued behind it
    for _ in range(50):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        lib.rv_counter_add(cnt.data_ptr(), 1, None, torch.cuda.current_stream().cuda_stream)
        e1.record()
        floors.append((e0, e1))
    torch.cuda.synchronize()
    fl = sorted(a.elapsed_time(b) for a, b in floors)
    return out, flops, fl[len(fl) // 2] * 1e3


# ---------------------------------------------------------------------------------------------
# the other kernel families of the step (linear GEMMs, local attention, BatchNorm, front-end): every launch of one eager
# step is recorded with its arguments and each distinct launch is re-issued between two HIP events on the launch stream
# (the operands of the recorded step are still mapped -- the caching allocator keeps them -- and the kernels are
# idempotent or accumulate into scratch statistics; results are not used).
# ---------------------------------------------------------------------------------------------
HBM_PEAK_TBS = 8.0                    # /opt/skills/guides/MI355X_MICROARCH.md (6.3 TB/s measured for a copy)
FAMILY = {
    'rv_gemm': 'linear GEMMs', 'rv_gemm_table_run': 'linear GEMMs', 'rv_local_attn_fwd': 'local attention', 'rv_local_attn_bwd': 'local attention',
    'rv_bn_lrelu_fwd': 'BatchNorm + leaky-ReLU', 'rv_bn_lrelu_fwd_skip': 'BatchNorm + leaky-ReLU', 'rv_bn_lrelu_bwd': 'BatchNorm + leaky-ReLU',
    'rv_melspec_lognorm_fwd': 'log-Mel front-end',
}


def family_work(name, a):
    """(flops, algorithmic HBM bytes) of one launch from its argument list (include/reconvat_hip.h)."""
    if name == 'rv_gemm':
        m, n, k, batch = a[13], a[14], a[15], a[19]
        return 2.0 * m * n * k * batch, 4.0 * batch * (m * k + k * n + m * n)
    if name == 'rv_gemm_table_run':               # grouped launch of the deferred parameter-gradient GEMMs
        import reconvat_amd.ops as ops_
        return ops_.GEMM_TABLE_WORK.get(a[0], (0.0, 0.0))
    if name == 'rv_bn_lrelu_fwd':                 # reads z (+ residual), writes y
        p_, c = a[2], a[3]
        return 0.0, 4.0 * p_ * c * (3 if a[13] else 2)
    if name == 'rv_bn_lrelu_fwd_skip':            # reads z and x (cin floats per pixel: the block's skip conv is evaluated in place), writes y
        p_, c, cin = a[2], a[3], a[15]
        return 0.0, 4.0 * p_ * (2 * c + cin)
    if name == 'rv_bn_lrelu_bwd':                 # reads dy, z, writes dz
        p_, c = a[4], a[5]
        return 0.0, 4.0 * p_ * c * 3
    if name == 'rv_local_attn_fwd':               # q, k, v in, out + attention map out
        b, l, g, dh = a[7], a[8], a[9], a[10]
        return 2.0 * 2 * b * l * g * dh * 31, 4.0 * b * l * (4 * g * dh + g * 31)
    if name == 'rv_local_attn_bwd':
        b, l, g, dh = a[12], a[13], a[14], a[15]
        return 2.0 * 5 * b * l * g * dh * 31, 4.0 * b * l * (8 * g * dh + 2 * g * 31)
    if name == 'rv_melspec_lognorm_fwd':          # audio in, log-mel written, re-read and re-written by the normalisation
        b, nsamp, n_mels, t = a[2], a[3], a[10], a[15]
        return 0.0, 4.0 * b * (nsamp + 3 * t * n_mels)
    return 0.0, 0.0


def measure_families(step_fn, device):
    from reconvat_amd import _lib
    import reconvat_amd.ops as ops_
    lib = _lib.load()
    ops_.KEEP_TABLES[0] = True                     # the grouped GEMM tables (and their operands) of the recorded step stay alive
    try:
        records = record_launches(step_fn, FAMILY)
    finally:
        ops_.KEEP_TABLES[0] = False
    cur = torch.cuda.current_stream().cuda_stream
    groups = {}
    for name, a in records:
        sig = (name,) + tuple(v for v in a[:-1] if isinstance(v, (int, float)) and not (isinstance(v, int) and v > (1 << 32)))
        g = groups.setdefault(sig, {'count': 0, 'name': name, 'args': a})
        g['count'] += 1
    fam = {}
    # the grouped-GEMM device tables were allocated at the END of the recorded step, possibly in memory that an earlier launch of the
    # step used as its (since freed) output: re-launch them FIRST, before the re-launches below scribble over such memory
    ordered = sorted(groups.values(), key=lambda g_: g_['name'] != 'rv_gemm_table_run')
    for g in ordered:
        a = list(g['args'])
        a[-1] = cur
        fn = getattr(lib, g['name'])
        for _ in range(2):
            fn(*a)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            fn(*a)
        e1.record()
        e1.synchronize()
        ms = e0.elapsed_time(e1) / 5
        fl, by = family_work(g['name'], g['args'])
        if os.environ.get('RV_FAMILY_VERBOSE'):
            sig = [v for v in g['args'][:-1] if isinstance(v, (int, float)) and not (isinstance(v, int) and v > (1 << 32))]
            print(f"[fam] {ms * g['count']:8.3f} ms/step  x{g['count']:3d} {ms * 1e3:8.1f} us  {by / ms / 1e9 if by else fl / ms / 1e9:7.2f} "
                  f"{'TB/s' if by else 'TF/s'}  {g['name']} {sig}", file=sys.stderr)
        f = fam.setdefault(FAMILY[g['name']], {'ms_per_step': 0.0, 'launches': 0, 'gflop': 0.0, 'gbyte': 0.0})
        f['ms_per_step'] += ms * g['count']
        f['launches'] += g['count']
        f['gflop'] += fl * g['count'] / 1e9
        f['gbyte'] += by * g['count'] / 1e9
    out = []
    for name, f in fam.items():
        mfma = name == 'linear GEMMs'
        achieved = (f['gflop'] / f['ms_per_step']) if mfma else (f['gbyte'] / f['ms_per_step'])      # TFLOP/s or TB/s
        peak = MFMA_F32_PEAK_TFLOPS if mfma else HBM_PEAK_TBS
        out.append({'family': name, 'bound': 'mfma' if mfma else 'hbm', 'ms_per_step': round(f['ms_per_step'], 3),
                    'launches_per_step': f['launches'], 'achieved': round(achieved, 3), 'unit': 'TFLOP/s' if mfma else 'TB/s',
                    'peak': peak, 'frac': round(achieved / peak, 4),
                    'work_per_step': round(f['gflop'], 1) if mfma else round(f['gbyte'], 2),
                    'work_unit': 'GFLOP' if mfma else 'GB (algorithmic)'})
    out.sort(key=lambda r: -r['ms_per_step'])
    return out


def _cpu_steps(threads, budget_s, min_timed=3, max_timed=3, batch=1):
    """`min_timed`..`max_timed` oracle steps (after one warm-up) with torch.set_num_threads(threads)."""
    from oracle import fixture as fx, model as om
    torch.set_num_threads(threads)
    torch.manual_seed(0)
    params = fx.fixture_params('onset', True)
    g = torch.Generator().manual_seed(1)
    bl, bul = synthetic_batch(batch, g, 'cpu'), synthetic_batch(batch, g, 'cpu')
    state, times = {}, []
    t_start = time.time()
    for i in range(1 + max_timed):
        t0 = time.time()
        om.train_step(params, state, i, bl, bul, om.run_on_batch_onset, VAT=True, reconstruction=True, xi=1e-6, eps=2.0)
        times.append(time.time() - t0)
        if len(times) - 1 >= min_timed and time.time() - t_start > budget_s:
            break
    timed = times[1:]
    return sorted(timed)[len(timed) // 2], len(timed)


def cpu_baseline():
    """The oracle (CPU restatement pinned to the reference by tests/golden) timed on this host's cores at the workload's OWN batch
    (B_l = B_ul = 8 full-length segments per step; same step definition: front-end, 2 x VAT, forward, backward, Adam), SURVEY 8(d):
    1 warm-up + 3 timed steps at 8 threads and at 32 threads (the best count on every box of rounds 1-5), plus ONE run at all physical
    cores -- listed even where it is slower; that run stops after its first timed step once it has used its 75 s budget.  `value` is
    the fastest run, every run is listed."""
    try:
        import psutil
        physical = psutil.cpu_count(logical=False) or os.cpu_count()
    except Exception:  # noqa: BLE001
        physical = os.cpu_count()
    prev = torch.get_num_threads()
    runs = []
    for n in sorted({min(8, physical), min(32, physical)}):
        per_step, timed = _cpu_steps(n, budget_s=1e9, min_timed=3, max_timed=3, batch=8)
        runs.append({'threads': n, 'batch': '8+8', 's_per_step': round(per_step, 3), 'timed_steps': timed,
                     'audio_s_per_s': round(16 * SEG_SECONDS / per_step, 3)})
    if physical not in {r['threads'] for r in runs}:
        per_step, timed = _cpu_steps(physical, budget_s=75.0, min_timed=1, max_timed=3, batch=8)
        runs.append({'threads': physical, 'batch': '8+8', 's_per_step': round(per_step, 3), 'timed_steps': timed,
                     'audio_s_per_s': round(16 * SEG_SECONDS / per_step, 3), 'note': 'all physical cores'})
    torch.set_num_threads(prev)
    best = max(runs, key=lambda r: r['audio_s_per_s'])
    return {'value': best['audio_s_per_s'], 'unit': 'audio-s/s', 'cores': best['threads'], 'kind': 'port',
            'physical_cores': physical, 'logical_cpus': os.cpu_count(), 'runs': runs,
            'sample': f'B_l + B_ul = {best["batch"]} full 327680-sample segments per step, UNet_Onset VAT+recon fp32 (oracle = CPU port pinned '
                      f'to the reference), median of {best["timed_steps"]} timed steps after 1 warm-up at torch.set_num_threads('
                      f'{best["threads"]}) ({best["s_per_step"]:.2f} s/step); every thread count tried (8, 32, all physical cores) is in `runs`'}


def parity_leg(device):
    """After the timed loop, in the SAME process and kernel configuration: one frozen-weight step of the bench schedule
    (two-stream hipGraph TrainStep, VAT + reconstruction) on the full-length fixture at the bench's OWN batch (tests/golden/anchor_b8.npz,
    case onset_T640_B8: B_l = B_ul = 8 segments of 327 680 samples, closed-form weights / inputs / injected VAT noise; falls back to the
    B = 2 case of lds_spread.npz) against the REFERENCE's own loss values on those inputs.  `oracle.fixture` only regenerates the closed-form fixture tensors (checker input, not a
    compute path)."""
    import numpy as np
    import reconvat_amd as ra
    from reconvat_amd import plans
    from oracle import fixture as fx
    b8 = os.path.join(ROOT, 'tests', 'golden', 'anchor_b8.npz')
    if os.path.exists(b8):                      # the bench's own batch: B_l = B_ul = 8 (the reference at 8 threads / 1 thread, fp32)
        g, case, nb, src = np.load(b8), 'onset_T640_B8', 8, 'tests/golden/anchor_b8.npz'
    else:
        g, case, nb, src = np.load(os.path.join(ROOT, 'tests', 'golden', 'lds_spread.npz')), 'onset_T640', 2, 'tests/golden/lds_spread.npz'

    def mk(tag):
        onset, frame = fx.fixture_labels(nb, 640, tag)
        return {'audio': fx.fixture_audio(nb, 640 * 512, tag).to(device), 'onset': onset.to(device), 'frame': frame.to(device)}
    bl, bul = mk('L'), mk('UL')
    noise = [fx.fixture_noise((nb, 1, 640, 229), 'd0_ul').to(device), fx.fixture_noise((nb, 1, 640, 229), 'd0_l').to(device)]
    m = ra.UNet_Onset((2, 2), (2, 2), log=True, reconstruction=True, mode='imagewise', spec='Mel', XI=1e-6, eps=2)
    m.load_state_dict(fx.fixture_params('onset', True))
    m.to(device).train()
    opt = ra.FlatAdam(m.parameters(), lr=0.0)            # frozen weights: every step sees the fixture weights
    state = {'i': 0}

    def draw(t):
        state['i'] += 1
        return noise[(state['i'] - 1) % 2].clone()       # unlabelled first, labelled second (model/UNet_onset.py:425,445)
    m.vat_loss.noise = draw
    step = ra.TrainStep(m, opt, bl, bul, alpha=1.0, VAT=True, clip=3.0, graph=True, dual_stream=True)
    step()
    step()
    torch.cuda.synchronize()
    step.check()
    keys = [str(k) for k in g[case + '_keys']]
    spread = dict(zip(keys, (float(v) for v in g[case + '_spread'])))
    errs = {}
    for k, ref in zip(keys, g[case + '_f32_8t']):
        errs[k.split('/')[-1]] = abs(float(step.losses[k]) - float(ref)) / max(abs(float(ref)), 1e-6)
    vat = {k: v for k, v in errs.items() if 'LDS' in k or 'r_norm' in k}
    non = {k: v for k, v in errs.items() if k not in vat}
    return {'case': f'{case} ({src}: the reference at 8 threads fp32, B_l = B_ul = {nb} full segments), two-stream hipGraph TrainStep, frozen weights',
            'rel_err_non_vat_max': float(f'{max(non.values()):.3e}'), 'rel_err_vat_max': float(f'{max(vat.values()):.3e}'),
            'rel_err': {k: float(f'{v:.3e}') for k, v in errs.items()},
            'reference_own_spread_vat_max': float(f'{max(v for k, v in spread.items() if "LDS" in k or "r_norm" in k):.3e}'),
            'tolerance': '1e-3 relative (north_star) on every term; (the -m gpu tests widen the VAT terms to max(1e-3, 2 x the reference\'s own '
                         '8-thread / 1-thread / fp64 movement) on the small fixtures, where the reference itself is noisier than 1e-3)',
            'kernel_plan_table': plans.digest()}


def _time_steps(step, n):
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        step()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / n


def deterministic_leg(args, device):
    """The same step with RV_DETERMINISTIC=1 (ops.DETERMINISTIC): parameter gradients folded in a fixed order instead of by fp32 atomics
    (per-layer weight-gradient reductions, park + in-order fold split-K of the parameter-gradient GEMMs) -- what bit-exact replica / solo-run
    comparisons cost (tests/test_stress_gpu.py, tests/test_dp_gpu.py run in this mode).  A fresh model and capture; ms per step."""
    from reconvat_amd import ops as ops_
    prev = ops_.DETERMINISTIC[0]
    ops_.DETERMINISTIC[0] = True
    try:
        _m, _o, _b, _bu, dstep = make_rank_step(args.model, args.batch, args.batch, 0, device)
        dstep.capture()
        return round(_time_steps(dstep, 10), 3)
    except Exception as e:  # noqa: BLE001 -- a reported figure, never a reason to lose the line
        return f'failed: {type(e).__name__}: {e}'
    finally:
        ops_.DETERMINISTIC[0] = prev


def dp_seam_leg(step, args, device):
    """What the data-parallel seam of a step costs on ONE GPU: the step is graph replay -> [gradient all-reduce] -> Adam -> repack, and
    the collective is an eager RCCL call between a graph and a kernel.  Measured back to back in this process: K steps without a
    process group, then K steps with a single-rank RCCL group and the all-reduce forced (RV_DP_FORCE_ALLREDUCE=1: the real
    ncclAllReduce launch on the 14.6 MB bucket, in place, in stream order).  seam_ms = the difference."""
    import torch.distributed as dist_
    from reconvat_amd import dp
    if dp.active():
        return None
    prev = {k: os.environ.get(k) for k in ('RV_DP_FORCE_ALLREDUCE', 'MASTER_ADDR', 'MASTER_PORT', 'RANK', 'WORLD_SIZE', 'LOCAL_RANK')}
    try:
        base = _time_steps(step, 10)
        import socket
        with socket.socket() as s_:
            s_.bind(('127.0.0.1', 0))
            port = s_.getsockname()[1]
        os.environ.update(RV_DP_FORCE_ALLREDUCE='1', MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK='0', WORLD_SIZE='1', LOCAL_RANK=str(device.index or 0))
        dp.init(device)
        calls0 = int(getattr(step.opt, 'allreduce_calls', 0))
        forced = _time_steps(step, 10)
        calls = int(getattr(step.opt, 'allreduce_calls', 0)) - calls0
        backend = dist_.get_backend()
        dp.shutdown()
        step.opt.grad_scale = 1.0
        again = _time_steps(step, 10)
        return {'ms_per_step_no_group': round(base, 3), 'ms_per_step_forced_allreduce': round(forced, 3), 'ms_per_step_no_group_again': round(again, 3),
                'seam_ms': round(forced - 0.5 * (base + again), 3), 'allreduce_calls': calls, 'backend': backend, 'bucket_mb': round(step.opt.n * 4 / 1e6, 1)}
    except Exception as e:  # noqa: BLE001
        return {'failed': f'{type(e).__name__}: {e}'}
    finally:
        for k, v in prev.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def visible_gpus():
    """GPUs this process may use, WITHOUT initialising the HIP runtime: the *_VISIBLE_DEVICES lists if set, else the KFD topology
    (nodes with compute units); None when neither can be read (the ranks then fail on their own if a device is missing)."""
    for var in ('ROCR_VISIBLE_DEVICES', 'HIP_VISIBLE_DEVICES', 'CUDA_VISIBLE_DEVICES'):
        v = os.environ.get(var)
        if v is not None:
            return len([x for x in v.split(',') if x.strip() != ''])
    root = '/sys/class/kfd/kfd/topology/nodes'
    try:
        count = 0
        for node in os.listdir(root):
            with open(os.path.join(root, node, 'properties')) as fh:
                props = dict(line.split()[:2] for line in fh if len(line.split()) >= 2)
            if int(props.get('simd_count', '0')) > 0:
                count += 1
        return count
    except OSError:
        return None


def self_launch(args):
    """`python bench.py --gpus N` without a launcher: spawn N fresh rank processes (one per GPU) BEFORE this process touches
    the GPU, watch ALL of them, relay rank 0's JSON line; if any rank exits non-zero the others are terminated and the launcher
    exits non-zero (a rank that dies early would otherwise leave its peers blocked in the rendezvous / a collective)."""
    import socket
    import subprocess
    import tempfile
    n = args.gpus
    have = visible_gpus()                         # (the launcher never touches the HIP runtime: the ranks are fresh child processes)
    if os.environ.get('RV_DP_SAME_GPU') == '1':   # every rank on cuda:0 over gloo (rank logic with the real kernels on a one-GPU box)
        if os.environ.get('RV_DP_BACKEND', 'nccl').lower() != 'gloo':
            raise SystemExit('RV_DP_SAME_GPU=1 needs RV_DP_BACKEND=gloo (RCCL refuses two ranks on one device)')
        if have is not None and have < 1:
            raise SystemExit('bench.py: no GPU')
    elif have is not None and have < n:
        raise SystemExit(f'bench.py --gpus {n}: this node exposes {have} GPU(s)')
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    procs = []
    out0 = tempfile.TemporaryFile(mode='w+')
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
        env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        env.setdefault('OMP_NUM_THREADS', str(max(1, (os.cpu_count() or 8) // n)))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=out0 if r == 0 else subprocess.DEVNULL, text=True))
    codes = [None] * n
    while any(c is None for c in codes):
        for r, p_ in enumerate(procs):
            if codes[r] is None:
                codes[r] = p_.poll()
        if any(c not in (None, 0) for c in codes):
            for r, p_ in enumerate(procs):
                if codes[r] is None:
                    p_.terminate()
            for r, p_ in enumerate(procs):
                if codes[r] is None:
                    try:
                        codes[r] = p_.wait(timeout=20)
                    except subprocess.TimeoutExpired:
                        p_.kill()
                        codes[r] = p_.wait()
            break
        time.sleep(0.05)
    out0.seek(0)
    sys.stdout.write(out0.read())
    sys.stdout.flush()
    bad = [(r, c) for r, c in enumerate(codes) if c != 0]
    if bad:
        raise SystemExit(f'bench.py: rank(s) failed: {bad}')


def make_rank_step(model_name, batch_l, batch_ul_n, rank, device, graph=True, dual_stream=True, bf16_backward=False):
    """Model, optimiser, resident synthetic shard and TrainStep of data-parallel rank `rank`: identical seeded initial weights on
    every rank, a distinct data shard and a distinct VAT noise stream per rank.  tests/test_dp_gpu.py rebuilds a
    rank's step in a single process with this function."""
    import reconvat_amd as ra
    torch.manual_seed(1234)                       # identical initial weights on every rank
    cls = ra.UNet_Onset if model_name == 'onset' else ra.UNet
    model = cls((2, 2), (2, 2), log=True, reconstruction=True, mode='imagewise', spec='Mel', device=str(device), XI=1e-6, eps=2).to(device)
    opt = ra.FlatAdam(model.parameters(), lr=1e-3, step_size=1000, gamma=0.98)
    gen = torch.Generator().manual_seed(1000 + rank)      # distinct data shard per rank
    batch, batch_ul = synthetic_batch(batch_l, gen, device), synthetic_batch(batch_ul_n, gen, device)
    torch.manual_seed(77 + rank)                  # VAT noise stream of this rank
    step = ra.TrainStep(model, opt, batch, batch_ul, alpha=1.0, VAT=True, clip=3.0, graph=graph, dual_stream=dual_stream,
                        bf16_backward=bf16_backward)
    return model, opt, batch, batch_ul, step


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--batch', type=int, default=8, help='labelled AND unlabelled segments per GPU')
    ap.add_argument('--model', choices=('onset', 'unet'), default='onset',
                    help="'unet': the no-onset model of train_UNet_VAT.py (BASELINE config 2; with --batch-l 1 the script's own batch sizes)")
    ap.add_argument('--batch-l', type=int, default=None, help='labelled segments per GPU (default: --batch)')
    ap.add_argument('--no-graph', action='store_true', help='eager launches instead of hipGraph replay')
    ap.add_argument('--single-stream', action='store_true', help='disable the two-stream step schedule (A/B)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-roofline', action='store_true')
    ap.add_argument('--bf16-backward', action='store_true',
                    help='BASELINE config 3 as an opt-in experiment (NOT the headline): bf16 operands in the backward 3x3 convs of the final '
                         'graphs; every forward pass and the power iteration stay fp32 (loss terms / posteriorgrams unchanged)')
    ap.add_argument('--no-parity', action='store_true', help='skip the post-run parity leg (frozen-weight step vs the reference fixture)')
    ap.add_argument('--verbose', action='store_true', help='dump the per-launch conv table to stderr')
    ap.add_argument('--dp-dump', default=None, metavar='DIR',
                    help='test instrumentation (tests/test_dp_gpu.py): every rank writes DIR/rank<r>.pt -- its data / VAT-noise checksums, its '
                         'gradient bucket before and after the FIRST all-reduce and its parameters after the first optimiser step')
    args = ap.parse_args()

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if 'WORLD_SIZE' not in os.environ and args.gpus > 1:
        return self_launch(args)
    if args.gpus != world:
        raise SystemExit(f'--gpus {args.gpus} does not match WORLD_SIZE={world}')
    from reconvat_amd import dp
    # one slice of the host cores per rank, set before anything touches the GPU (next to the launcher's OMP_NUM_THREADS split)
    rank_cpus = dp.pin_rank_cpus() if world > 1 else None
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs an MI355X (no CPU fallback for the product path)')
    device = dp.local_device() if world > 1 else torch.device('cuda', local)
    torch.cuda.set_device(device)
    if world > 1 or os.environ.get('RV_DP_FORCE_ALLREDUCE') == '1':
        # (RV_DP_FORCE_ALLREDUCE=1 under a launcher with one rank: a single-rank RCCL group, so that a one-GPU box executes the
        # gradient all-reduce call for real; RV_DP_BACKEND=gloo RV_DP_SAME_GPU=1: N ranks on cuda:0 -- reconvat_amd/dp.py)
        dp.init(device)

    import reconvat_amd as ra
    from reconvat_amd import plans, ops as ops_mod, _lib as _lib_mod
    cls = ra.UNet_Onset if args.model == 'onset' else ra.UNet
    batch_l = args.batch if args.batch_l is None else args.batch_l
    model, opt, batch, batch_ul, step = make_rank_step(args.model, batch_l, args.batch, rank, device, graph=not args.no_graph,
                                                       dual_stream=not args.single_stream, bf16_backward=args.bf16_backward)
    used_graph = not args.no_graph
    if used_graph:
        try:
            step.capture()
        except Exception as e:                     # noqa: BLE001 -- report and fall back to eager launches
            if rank == 0:
                print(f'[bench] hipGraph capture failed ({type(e).__name__}: {e}); falling back to eager', file=sys.stderr)
            torch.cuda.synchronize()
            step.use_graph, step.graph, used_graph = False, None, False

    def barrier():
        dp.barrier()
        torch.cuda.synchronize()

    dump = None
    if args.dp_dump:
        # the first optimiser step of this rank, seen from inside FlatAdam.step(): bucket before / after the collective, then the
        # parameters that step produced (written below, after the warm-up steps have been issued and synchronised)
        dump = {'rank': rank, 'world': world, 'audio_checksum_l': float(batch['audio'].double().sum()),
                'audio_checksum_ul': float(batch_ul['audio'].double().sum()), 'cuda_seed': int(torch.cuda.initial_seed()),
                'omp_num_threads': os.environ.get('OMP_NUM_THREADS'), 'device': str(device), 'pid': os.getpid(),
                'cpu_affinity': sorted(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else None, 'rank_cpus': rank_cpus}

        def dp_hook(when, o):
            if when + '_bucket' not in dump:
                dump[when + '_bucket'] = o.flat_grad.detach().cpu().clone()
        opt.dp_hook = dp_hook

        def grab_params():
            if 'params_after_step1' not in dump:
                dump['params_after_step1'] = opt.flat_param.detach().cpu().clone()
                dump['losses_step1'] = {k: float(v) for k, v in step.losses.items()}

    for i in range(args.warmup):
        step()
        if dump is not None and i == 0:
            grab_params()
    barrier()
    # K + 1 HIP events on the launch stream: event i sits behind step i, so consecutive differences are the per-step device times
    # (`ms_per_step` = their median; the barrier-to-barrier wall clock of the same K steps stays the basis of `value`)
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    ev0, ev1 = evs[0], evs[-1]
    ev0.record()
    t0 = time.perf_counter()
    # test instrumentation (tests/test_dp_gpu.py): this rank dies in the middle of the timed loop -- its peers are then blocked in the
    # next collective and it is the LAUNCHER's job to notice, terminate them and exit non-zero
    fail_at = int(os.environ.get('RV_TEST_FAIL_AT', '0')) if os.environ.get('RV_TEST_FAIL_RANK') == str(rank) else None
    for i in range(args.steps):
        if fail_at is not None and i == fail_at:
            torch.cuda.synchronize()
            os._exit(17)
        step()
        evs[i + 1].record()
        if dump is not None and i == 0:
            grab_params()
    barrier()
    elapsed = time.perf_counter() - t0
    device_ms = ev0.elapsed_time(ev1)             # the same K steps on the device's own clock (HIP events on the launch stream)
    per_step_ms = sorted(evs[i].elapsed_time(evs[i + 1]) for i in range(args.steps))
    median_ms = per_step_ms[len(per_step_ms) // 2] if len(per_step_ms) % 2 else 0.5 * (per_step_ms[len(per_step_ms) // 2 - 1] + per_step_ms[len(per_step_ms) // 2])
    if dp.active():
        t = torch.tensor([elapsed, median_ms], device=device, dtype=torch.float64)
        dp.all_reduce(t, dist.ReduceOp.MAX)
        elapsed, median_ms = t[0].item(), t[1].item()
    loss = float(step.loss.item())
    nan_flag = int(model.vat_loss.nan_flag.item())
    loss_terms = {k: round(float(v), 6) for k, v in step.losses.items()}      # the 11 loss values of the last timed step
    # replica check: identical seeded weights + one summed gradient bucket + identical Adam => bit-identical parameters on
    # every rank.  Checksum = wrapping int64 sum of the parameter bit patterns; MAX - MIN over ranks must be 0.
    dp_ranks, replicas_equal = 1, True
    checksum = opt.flat_param.view(torch.int32).sum(dtype=torch.int64).reshape(1)
    if dp.active():
        dp_ranks = dist.get_world_size()
        hi, lo = checksum.clone(), checksum.clone()
        dp.all_reduce(hi, dist.ReduceOp.MAX)
        dp.all_reduce(lo, dist.ReduceOp.MIN)
        replicas_equal = bool((hi - lo).item() == 0)
    if dump is not None:
        dump.update(allreduce_calls=int(getattr(opt, 'allreduce_calls', 0)), optimizer_steps=int(opt.step_count.item()),
                    params_final=opt.flat_param.detach().cpu().clone(), offsets=list(opt.offsets),
                    names=[n for n, p_ in model.named_parameters() if p_.requires_grad])
        os.makedirs(args.dp_dump, exist_ok=True)
        torch.save(dump, os.path.join(args.dp_dump, f'rank{rank}.pt'))
    ms = elapsed / args.steps * 1e3
    assert _lib_mod.load().rv_source_digest().decode() == _lib_mod.source_digest(), 'stale libreconvat_hip.so'
    audio_s = world * (batch_l + args.batch) * SEG_SECONDS * args.steps / elapsed

    line = {
        'metric': 'training audio-sec/sec (node)', 'value': round(audio_s, 2), 'unit': 'audio-s/s', 'n_gpus': world,
        # ms_per_step: MEDIAN of the K per-step device times (HIP events on the launch stream, MAX over ranks);
        # mean_ms_per_step: barrier-to-barrier wall clock of the same K steps / K (MAX over ranks) -- what `value` is computed from
        'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': round(median_ms, 3), 'mean_ms_per_step': round(ms, 3),
        'min_ms_per_step': round(per_step_ms[0], 3), 'max_ms_per_step': round(per_step_ms[-1], 3), 'higher_is_better': True,
        'scaling': 'weak', 'vs_baseline': None,
        'dtype': 'f32' if not args.bf16_backward else 'f32 forward + power iteration, bf16-operand / f32-accumulate backward 3x3 convs (opt-in experiment)',
        'data': 'synthetic',
        'config': {'workload': f'ReconVAT {cls.__name__} VAT=True reconstruction=True, per-GPU B_l={batch_l} + '
                               f'B_ul={args.batch} segments of 327680 samples (640 frames x 229 mel), Adam+StepLR, fp32',
                   'parallelism': f'dp{world}', 'hipgraph': used_graph, 'two_stream_schedule': not args.single_stream, 'labelled_only_audio_s_per_s': round(audio_s * batch_l / (batch_l + args.batch), 2),
                   'kernel_plan_table': plans.digest(), 'kernel_plan_mode': str(ops_mod.AUTOTUNE),
                   # the library's own record of the sources it was built from; _lib.load() has already refused a library whose digest
                   # differs from the sources next to it, the assertion repeats that for the record
                   'source_digest': _lib_mod.load().rv_source_digest().decode(), 'source_digest_of_tree': _lib_mod.source_digest(),
                   'final_loss': round(loss, 5), 'vat_nan_flag': nan_flag, 'losses_last_step': loss_terms},
        # the same K steps timed by two HIP events on the launch stream (the record carries its own evidence that the device worked)
        'device_ms_per_step': round(device_ms / args.steps, 3),
        'dp_ranks': dp_ranks, 'dp_backend': (dist.get_backend() if dp.active() else None), 'replicas_equal': replicas_equal,
        'param_checksum': int(checksum.item()),
        # gradient all-reduces issued by FlatAdam.step (RCCL, or gloo through pinned host memory): one per optimiser step
        'dp_allreduce_calls': int(getattr(opt, 'allreduce_calls', 0)), 'optimizer_steps': int(opt.step_count.item()),
        # HBM footprint of this rank up to the end of the timed loop (caching allocator; the captured graph's private pool included):
        # weights + optimiser state 58 MB, the rest activations of the 8 + 1 passes, workspaces and the parked (x, dY) pairs of the merged weight gradients
        'device_memory_gb': {'max_allocated': round(torch.cuda.max_memory_allocated(device) / 2 ** 30, 2),
                             'max_reserved': round(torch.cuda.max_memory_reserved(device) / 2 ** 30, 2)},
    }
    if rank == 0 and world == 1 and args.model == 'onset' and batch_l == args.batch == 8:       # (the roofline / parity legs are written for the headline workload)
        if not args.no_roofline:
            eager = ra.TrainStep(model, opt, batch, batch_ul, alpha=1.0, VAT=True, clip=3.0, graph=False, bf16_backward=args.bf16_backward)
            conv_ms, conv_flops_total, per_kernel, nlaunch, min_sets, bound = measure_conv_phase(eager, device)
            achieved = conv_flops_total / conv_ms / 1e9
            # in situ: one eager single-stream step, every family launch bracketed by HIP events on the launch stream
            eager1 = ra.TrainStep(model, opt, batch, batch_ul, alpha=1.0, VAT=True, clip=3.0, graph=False, dual_stream=False,
                                  bf16_backward=args.bf16_backward)
            eager1()
            situ, situ_flops, floor_us = measure_in_situ(eager1, device)
            conv_situ_ms, conv_situ_n = situ.get('convolutions', (0.0, 0))
            achieved_situ = situ_flops / conv_situ_ms / 1e9 if conv_situ_ms else 0.0
            if args.verbose:
                for t, c, m_, tf, sg in per_kernel:
                    print(f'[conv] {t:8.3f} ms/step  x{c:3d}  {m_:8.4f} ms  {tf:7.1f} TF/s  {sg}', file=sys.stderr)
            # HBM bytes of the same launches from the committed PMC passes (rocprofv3 cannot run inside this process):
            # tools/pmc_traffic.py over separate FETCH_SIZE / WRITE_SIZE runs of `bench.py --no-graph`, gfx950-corrected
            traffic, traffic_src, traffic_digest, two_stream = None, None, None, None
            traffic_fam = None
            for rnd in ('r06', 'r05', 'r04', 'r03', 'r02', 'r01'):
                tpath = os.path.join(ROOT, 'profiles', f'{rnd}_pmc_traffic.json')
                if os.path.exists(tpath):
                    with open(tpath) as fh:
                        tj = json.load(fh)
                    traffic, traffic_digest = tj.get('traffic_bytes'), tj.get('kernel_plan_table')
                    traffic_fam = tj.get('families')        # round 5: measured bytes of EVERY family of the step, not only the convs
                    traffic_src = f"profiles/{rnd}_pmc_traffic.json (committed PMC pass, {tj.get('git', 'commit not recorded')})"
                    break
            # the conv fraction INSIDE the shipped two-stream schedule: rocprofv3 kernel-trace of the timed command, summed by
            # tools/rocpd_stats.py (committed table; a profiler cannot run inside this process)
            for rnd in ('r06', 'r05', 'r04'):
                spath = os.path.join(ROOT, 'profiles', f'{rnd}_two_stream_conv.json')
                if os.path.exists(spath):
                    with open(spath) as fh:
                        two_stream = json.load(fh)
                    two_stream['source'] = f'profiles/{rnd}_two_stream_conv.json (tools/rocpd_stats.py --json over profiles/{rnd}_step_kernel_stats.txt\'s trace)'
                    break
            # the committed trace describes the tile table (and library) it ran: with another table the kernel times are not this build's
            # (and the SOURCES it was built from: rv_source_digest(), round 6)
            src_digest = _lib_mod.load().rv_source_digest().decode()
            two_stream_ok = bool(two_stream) and two_stream.get('kernel_plan_table') == plans.digest() and two_stream.get('source_digest') == src_digest
            # the same table of `bench.py --single-stream` (no co-running kernels stretching each other): the rocprofv3 figure the isolated
            # HIP-event figure above has to agree with
            single_stream, single_src = None, None
            for rnd in ('r06', 'r05'):
                spath1 = os.path.join(ROOT, 'profiles', f'{rnd}_single_stream_conv.json')
                if os.path.exists(spath1):
                    with open(spath1) as fh:
                        single_stream = json.load(fh)
                    single_src = f'profiles/{rnd}_single_stream_conv.json (tools/profile_step.sh <tag> --single-stream; table: profiles/{rnd}_step_kernel_stats_single_stream.txt)'
                    break
            single_stream_ok = bool(single_stream) and single_stream.get('kernel_plan_table') == plans.digest() and single_stream.get('source_digest') == src_digest
            if two_stream:
                two_stream['kernel_plan_table_matches'] = two_stream_ok      # (plan table AND source digest)
            families = measure_families(eager, device)
            if traffic_fam:
                # measured HBM bytes (committed PMC passes) next to the algorithmic bytes each HBM-bound family is graded on
                key = {'BatchNorm + leaky-ReLU': 'BatchNorm + leaky-ReLU (HBM)', 'local attention': 'local attention',
                       'linear GEMMs': 'linear GEMMs (MFMA)', 'log-Mel front-end': 'front-end (HBM)'}
                for f in families:
                    t = traffic_fam.get(key.get(f['family'], ''))
                    if t:
                        f['measured_gb'] = round(t['traffic_bytes'] / 1e9, 2)
                        if f['bound'] == 'hbm' and f['work_per_step']:
                            f['measured_over_algorithmic'] = round(t['traffic_bytes'] / 1e9 / f['work_per_step'], 3)
            # `frac` / `achieved` describe the SHIPPED two-stream schedule: executed conv flops / the conv kernel time of
            # the committed rocprofv3 trace of this very command when that trace ran this build's plan table; otherwise the live
            # event-bracketed in-situ figure of this run (an upper bound of the kernel time).  The isolated re-launch figure measured live
            # by this run is `frac_isolated` / `achieved_isolated`.
            if two_stream_ok:
                achieved_shipped = conv_flops_total / (two_stream['conv_ms_per_step'] * 1e-3) / 1e12
                frac_src = 'frac_two_stream (committed rocprofv3 kernel trace of the shipped schedule, same plan table)'
            else:
                achieved_shipped = achieved_situ
                frac_src = 'frac_in_situ (live HIP-event brackets inside one eager single-stream step; no committed trace of this plan table)'
            line['roofline'] = {
                'bound': 'mfma', 'achieved': round(achieved_shipped, 2), 'peak': MFMA_F32_PEAK_TFLOPS, 'unit': 'TFLOP/s',
                'frac': round(achieved_shipped / MFMA_F32_PEAK_TFLOPS, 4), 'frac_source': frac_src,
                'frac_isolated': round(achieved / MFMA_F32_PEAK_TFLOPS, 4), 'achieved_isolated': round(achieved, 2), 'traffic': traffic,
                'traffic_scope': 'HBM bytes (FETCH_SIZE x2 + WRITE_SIZE) of all conv launches of one step; NOT measured by this run: '
                                 + str(traffic_src),
                # the PMC pass ran the tile table with this digest; a mismatch means the committed traffic figure describes other tiles
                'traffic_plan_table': traffic_digest, 'traffic_plan_table_matches': (traffic_digest == plans.digest()) if traffic_digest else None,
                'traffic_by_family': traffic_fam,
                # executed conv flops / summed conv kernel time of the SHIPPED two-stream schedule (kernels stretched by the concurrent chain)
                # (None when the committed trace was taken with another tile table: ADVICE r04)
                'frac_two_stream': (round(conv_flops_total / (two_stream['conv_ms_per_step'] * 1e-3) / 1e12 / MFMA_F32_PEAK_TFLOPS, 4)
                                    if two_stream_ok else None),
                'two_stream': two_stream,
                'frac_single_stream_rocprof': (round(conv_flops_total / (single_stream['conv_ms_per_step'] * 1e-3) / 1e12 / MFMA_F32_PEAK_TFLOPS, 4)
                                               if single_stream_ok else None),
                'single_stream_rocprof': ({'conv_ms_per_step': single_stream['conv_ms_per_step'], 'conv_launches_per_step': single_stream['conv_launches_per_step'],
                                           'kernel_ms_per_step': single_stream['kernel_ms_per_step'], 'kernel_plan_table': single_stream.get('kernel_plan_table'),
                                           'git': single_stream.get('git'),
                                           'source': single_src}
                                          if single_stream else None),
                'kernel': 'conv3x3_lds_k / conv_mfma_k / wgrad_mfma_k family (all conv launches of one step)',
                'conv_ms_source': 'isolated re-launch of every distinct conv launch of one step, HIP events on the launch stream, after the '
                                  f'timed loop, operands ROTATED through >= {min_sets} scratch sets totalling > 256 MiB per launch shape (no launch finds '
                                  'its operands warm from the previous repetition; per-kernel figure; the timed step overlaps two chains); weight '
                                  'gradients as the step runs them: per-layer partial-sum kernel + the per-chain reduction table launches',
                # the same family inside ONE eager single-stream step: every conv launch bracketed by two HIP events on the launch
                # stream, operands as warm / cold as the step leaves them.  A bracket includes the dispatch gap behind the previous
                # launch (`event_bracket_floor_us`: the same bracket around a one-workgroup kernel), so this is an upper bound of the
                # rocprofv3 kernel-duration sum of the same step (profiles/r05_step_kernel_stats_single_stream.txt; `frac_single_stream_rocprof`)
                'frac_in_situ': round(achieved_situ / MFMA_F32_PEAK_TFLOPS, 4), 'achieved_in_situ': round(achieved_situ, 2),
                'conv_ms_in_situ': round(conv_situ_ms, 3), 'conv_launches_in_situ': conv_situ_n,
                'event_bracket_floor_us': round(floor_us, 2),
                'conv_ms_in_situ_minus_floor': round(conv_situ_ms - conv_situ_n * floor_us * 1e-3, 3),
                'families_in_situ': {k: {'ms_per_step': round(v[0], 3), 'launches': v[1]} for k, v in sorted(situ.items())},
                'launches_per_step': nlaunch, 'conv_ms_per_step': round(conv_ms, 3),
                'executed_gflop_per_step': round(conv_flops_total / 1e9, 1),
                'reference_gflop_per_step': 1531.0,
                # SURVEY 8(d) counts the reference's conv work (1 531 GFLOP incl. the weight gradients of the power
                # iteration that this build provably does not need); `frac` above credits only what was executed
                'frac_counting_reference_work': round(1531.0e9 / (conv_ms * 1e-3) / 1e12 / MFMA_F32_PEAK_TFLOPS, 4),
                # achieved / min(MFMA peak of the launch's operand type, arithmetic intensity x HBM), launch by launch
                # (algorithmic bytes: input + output + weights once); `bf16_launches`: the opt-in bf16-operand launches alone
                'frac_of_min_mfma_ai_hbm': {k: {'frac': round(v[1] / v[0], 4), 'measured_ms': round(v[0], 3), 'ms_at_bound': round(v[1], 3),
                                                'achieved_tflops': round(v[2] / v[0] / 1e9, 1),
                                                'mfma_peak_tflops': MFMA_BF16_PEAK_TFLOPS if k == 'bf16_launches' else 'per launch: 157.3 (f32) / 2500 (bf16)',
                                                'hbm_tb_s': HBM_PEAK_TBS}
                                            for k, v in (('all_conv_launches', bound['all']), ('bf16_launches', bound['bf16'])) if v[0] > 0},
                # the same launches split by WHICH roof is the lower one: the HBM-bound ones (1x1 skip, 2x2 down / up,
                # C <= 2 layers, the 16-channel 3x3 layers) against HBM peak on their algorithmic bytes, the rest against the f32 MFMA peak
                'conv_launches_by_roof': {
                    'hbm_bound': {'ms_per_step': round(bound['split']['hbm_bound'][0], 3), 'launches': bound['split']['hbm_bound'][3],
                                  'algorithmic_gb': round(bound['split']['hbm_bound'][2] / 1e9, 2),
                                  'tb_s': round(bound['split']['hbm_bound'][2] / max(bound['split']['hbm_bound'][0], 1e-9) / 1e9, 3),
                                  'frac_of_hbm_peak': round(bound['split']['hbm_bound'][2] / max(bound['split']['hbm_bound'][0], 1e-9) / 1e9 / HBM_PEAK_TBS, 4)},
                    'mfma_bound': {'ms_per_step': round(bound['split']['mfma_bound'][0], 3), 'launches': bound['split']['mfma_bound'][3],
                                   'gflop': round(bound['split']['mfma_bound'][1] / 1e9, 1),
                                   'tflops': round(bound['split']['mfma_bound'][1] / max(bound['split']['mfma_bound'][0], 1e-9) / 1e9, 2),
                                   'frac_of_mfma_peak': round(bound['split']['mfma_bound'][1] / max(bound['split']['mfma_bound'][0], 1e-9) / 1e9 / MFMA_F32_PEAK_TFLOPS, 4)}},
                'top': [{'ms_per_step': round(t, 3), 'count': c, 'ms': round(m, 4), 'tflops': round(tf, 1), 'sig': list(map(str, s))}
                        for t, c, m, tf, s in per_kernel[:6]],
                # every other kernel family of the step against the roofline that bounds it (same isolated re-launch method)
                'families': [{'family': 'convolutions (3x3 / 1x1 / 2x2, fwd + dgrad + wgrad)', 'bound': 'mfma',
                              'ms_per_step': round(conv_ms, 3), 'launches_per_step': nlaunch, 'achieved': round(achieved, 2),
                              'unit': 'TFLOP/s', 'peak': MFMA_F32_PEAK_TFLOPS, 'frac': round(achieved / MFMA_F32_PEAK_TFLOPS, 4),
                              'work_per_step': round(conv_flops_total / 1e9, 1), 'work_unit': 'GFLOP'},
                             # the same launches split by which roof is the lower one
                             {'family': 'conv launches whose AI x HBM roof is the lower one (1x1, 2x2, C <= 2, 16-channel 3x3)', 'bound': 'hbm',
                              'ms_per_step': round(bound['split']['hbm_bound'][0], 3), 'launches_per_step': bound['split']['hbm_bound'][3],
                              'achieved': round(bound['split']['hbm_bound'][2] / max(bound['split']['hbm_bound'][0], 1e-9) / 1e9, 3), 'unit': 'TB/s', 'peak': HBM_PEAK_TBS,
                              'frac': round(bound['split']['hbm_bound'][2] / max(bound['split']['hbm_bound'][0], 1e-9) / 1e9 / HBM_PEAK_TBS, 4),
                              'work_per_step': round(bound['split']['hbm_bound'][2] / 1e9, 2), 'work_unit': 'GB (algorithmic)', 'subset_of': 'convolutions'},
                             {'family': 'conv launches whose MFMA roof is the lower one', 'bound': 'mfma',
                              'ms_per_step': round(bound['split']['mfma_bound'][0], 3), 'launches_per_step': bound['split']['mfma_bound'][3],
                              'achieved': round(bound['split']['mfma_bound'][1] / max(bound['split']['mfma_bound'][0], 1e-9) / 1e9, 2), 'unit': 'TFLOP/s',
                              'peak': MFMA_F32_PEAK_TFLOPS,
                              'frac': round(bound['split']['mfma_bound'][1] / max(bound['split']['mfma_bound'][0], 1e-9) / 1e9 / MFMA_F32_PEAK_TFLOPS, 4),
                              'work_per_step': round(bound['split']['mfma_bound'][1] / 1e9, 1), 'work_unit': 'GFLOP', 'subset_of': 'convolutions'}] + families,
            }
            line['deterministic_ms_per_step'] = deterministic_leg(args, device)
            line['dp_seam'] = dp_seam_leg(step, args, device)
        if not args.no_parity:
            line['parity'] = parity_leg(device)
        if not args.no_cpu_baseline:
            line['cpu_baseline'] = cpu_baseline()
    dp.shutdown()
    if rank == 0:
        # RCCL writes its version banner to the C-level stdout (block-buffered when piped, flushed at exit): push it out NOW so that
        # the JSON line is the LAST line of stdout
        try:
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except Exception:  # noqa: BLE001
            pass
        sys.stdout.write(json.dumps(line) + '\n')
        sys.stdout.flush()


if __name__ == '__main__':
    main()
