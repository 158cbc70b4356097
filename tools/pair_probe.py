"""Do two kernels from two HIP streams actually share the chip?  (round 6, VERDICT r05 item 1: co-residency.)

For a pair (X, Y): N launches of X captured into a hipGraph, N launches of Y into another, and ONE graph that forks -- X's chain on one
branch, Y's chain on the other -- i.e. exactly how the two chains of the training step meet on the device.  Reported per pair:

    T(X), T(Y), T(X || Y) per launch pair and   hidden = (T(X) + T(Y) - T(X || Y)) / min(T(X), T(Y))

hidden = 1: the shorter kernel disappears behind the longer one (perfect co-running); 0: the pair costs what the two cost back to back.

    python tools/pair_probe.py            # the built-in list (persistent Winograd convs at 171 / 230 / 146 / 190 VGPRs against BatchNorm,
                                          # 1x1 conv, attention, GEMM, and against a second conv)
Kernel specs: conv:<cin>:<cout>:<H>:<W>:<algo hex> | bn:<C>:<H>:<W> | c1:<cin>:<cout>:<H>:<W> | attn | gemm:<M>:<N>:<K>
"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from reconvat_amd import ops, _lib

dev = torch.device('cuda:0')
B = 8
N = int(os.environ.get('PAIR_N', '24'))
lib = _lib.load()


def make(spec, tag):
    """-> (name, launch()) ; every kernel gets its own operands (tag keeps the two branches apart)."""
    p = spec.split(':')
    if p[0] == 'conv':
        cin, cout, h, w, algo = int(p[1]), int(p[2]), int(p[3]), int(p[4]), int(p[5], 16)
        x = torch.rand(B, h, w, cin, device=dev) - 0.5
        wt = (torch.rand(cout, cin, 3, 3, device=dev) - 0.5) * 0.1
        bias = torch.zeros(cout, device=dev)
        y = torch.empty(B, h, w, cout, device=dev)
        pack = ops._pack('c3', wt, 'fwd')
        keep = (x, wt, bias, y, pack)

        def run():
            rc = lib.rv_conv_fwd(0, x.data_ptr(), cin, B, h, w, cin, y.data_ptr(), cout, h, w, cout, pack.data_ptr(), bias.data_ptr(), 0, algo,
                                 None, None, 0, None, 0.0, torch.cuda.current_stream().cuda_stream)
            assert rc == 0, (spec, lib.rv_last_error())
        run.keep = keep
        return f'conv3x3 {cin}->{cout} @{h}x{w} algo {algo:#x}', run
    if p[0] == 'bn':
        c, h, w = int(p[1]), int(p[2]), int(p[3])
        z = torch.rand(B, h, w, c, device=dev)
        g, bta = torch.ones(c, device=dev), torch.zeros(c, device=dev)
        rm, rv, nbt = torch.zeros(c, device=dev), torch.ones(c, device=dev), torch.zeros((), device=dev, dtype=torch.long)

        def run():
            with torch.no_grad():
                ops.BnActFn.apply(z, g, bta, rm, rv, nbt, None, True, ops.SLOPE, None, None)
        run.keep = (z, g, bta, rm, rv, nbt)
        return f'BatchNorm (statistics + apply) C={c} @{h}x{w}', run
    if p[0] == 'c1':
        cin, cout, h, w = int(p[1]), int(p[2]), int(p[3]), int(p[4])
        x = torch.rand(B, h, w, cin, device=dev) - 0.5
        wt = (torch.rand(cout, cin, 1, 1, device=dev) - 0.5) * 0.1
        bias = torch.zeros(cout, device=dev)
        y = torch.empty(B, h, w, cout, device=dev)

        def run():
            ops.conv_forward_into('c1', x, wt, bias, y)
        run.keep = (x, wt, bias, y)
        return f'conv1x1 {cin}->{cout} @{h}x{w}', run
    if p[0] == 'attn':
        f, groups, L = 768, 6, 640
        q = torch.rand(B * L, 3 * f, device=dev) - 0.5
        rel = (torch.rand(31, f, device=dev) - 0.5) * 0.1
        out = torch.empty(B, L, f, device=dev)
        att = torch.empty(B, L, groups, 31, device=dev)

        def run():
            rc = lib.rv_local_attn_fwd(q.data_ptr(), q.data_ptr() + 4 * f, q.data_ptr() + 8 * f, 3 * f, rel.data_ptr(), out.data_ptr(), att.data_ptr(),
                                       B, L, groups, f // groups, torch.cuda.current_stream().cuda_stream)
            assert rc == 0
        run.keep = (q, rel, out, att)
        return 'attn_fwd_k 768/6 heads, L=640', run
    if p[0] == 'gemm':
        m, n, k = int(p[1]), int(p[2]), int(p[3])
        a = torch.rand(m, k, device=dev) - 0.5
        b = torch.rand(k, n, device=dev) - 0.5
        c = torch.empty(m, n, device=dev)

        def run():
            ops.gemm(a, b, c, splitk=1)
        run.keep = (a, b, c)
        return f'gemm {m}x{n}x{k}', run
    if p[0] == 'mfma':
        # rv_debug_mfma_peak: 256-thread workgroups of pure f32 MFMA issue -- no LDS, ~40 registers, no memory traffic: the cleanest possible
        # MFMA-bound partner (if THIS does not hide a bandwidth-bound kernel, nothing will)
        import ctypes
        blocks, iters, nacc = int(p[1]), int(p[2]), int(p[3])
        fn = ctypes.CDLL(_lib.LIB_PATH).rv_debug_mfma_peak
        fn.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
        out = torch.empty(4096 * 256, device=dev)

        def run():
            fn(out.data_ptr(), blocks, iters, nacc, torch.cuda.current_stream().cuda_stream)
        run.keep = (out,)
        return f'mfma_peak_k {blocks} wgs x {iters} it, nacc {nacc}', run
    raise SystemExit(f'unknown kernel spec {spec}')


def graph_of(fns):
    """One graph: branch i runs N launches of fns[i] (branch 0 on the capture stream, the others forked)."""
    side = [torch.cuda.Stream() for _ in fns[1:]]
    for f in fns:                      # warm up outside the capture (weight packing, plan lookups)
        f(); f()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        cur = torch.cuda.current_stream()
        for s, f in zip(side, fns[1:]):
            s.wait_stream(cur)
            with torch.cuda.stream(s):
                for _ in range(N):
                    f()
        for _ in range(N):
            fns[0]()
        for s in side:
            cur.wait_stream(s)
    return g


def time_graph(g, reps=8):
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(reps):
        e0.record()
        g.replay()
        e1.record()
        e1.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3 / N)
    return sorted(ts)[len(ts) // 2]


def time_two_graphs(gx, gy, reps=8):
    """The two single-branch graphs replayed CONCURRENTLY on two streams (instead of one forked graph)."""
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    ts = []
    for _ in range(reps + 1):
        torch.cuda.synchronize()
        e0, e1, e2 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        cur = torch.cuda.current_stream()
        e0.record(cur)
        s1.wait_event(e0); s2.wait_event(e0)
        with torch.cuda.stream(s1):
            gx.replay()
            e1.record(s1)
        with torch.cuda.stream(s2):
            gy.replay()
            e2.record(s2)
        torch.cuda.synchronize()
        ts.append(max(e0.elapsed_time(e1), e0.elapsed_time(e2)) * 1e3 / N)
    return sorted(ts[1:])[len(ts[1:]) // 2]


def time_eager_two_streams(fx, fy, reps=5):
    """N eager launches of each on two streams, issued alternately by the host (no graph at all)."""
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    ts = []
    for _ in range(reps + 1):
        torch.cuda.synchronize()
        e0, e1, e2 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(torch.cuda.current_stream())
        s1.wait_event(e0); s2.wait_event(e0)
        for _i in range(N):
            with torch.cuda.stream(s1):
                fx()
            with torch.cuda.stream(s2):
                fy()
        e1.record(s1); e2.record(s2)
        torch.cuda.synchronize()
        ts.append(max(e0.elapsed_time(e1), e0.elapsed_time(e2)) * 1e3 / N)
    return sorted(ts[1:])[len(ts[1:]) // 2]


PAIRS3 = [('mfma:256:180:4', 'bn:64:160:57'), ('mfma:256:180:2', 'mfma:256:180:2'), ('conv:64:64:160:57:911', 'bn:64:160:57'), ('bn:64:160:57', 'bn:16:640:229'),
          ('conv:64:64:160:57:911', 'conv:64:64:160:57:911'), ('conv:64:64:160:57:911', 'attn'), ('attn', 'bn:16:640:229')]
# round 6, VERDICT r05 item 1: the half-CU family 0xE (four waves, <= 78 KiB of LDS, <= 256 registers, two workgroups per CU) against the
# shipped full-CU instances of the same layer -- alone, as a pair of two half-CU kernels from two chains, and beside a BatchNorm
PAIRS4 = [('conv:128:128:80:28:a11', 'conv:128:128:80:28:a11'), ('conv:128:128:80:28:e11', 'conv:128:128:80:28:e11'), ('conv:128:128:80:28:e11', 'bn:128:80:28'),
          ('conv:64:64:160:57:911', 'conv:64:64:160:57:911'), ('conv:64:64:160:57:e11', 'conv:64:64:160:57:e11'), ('conv:64:64:160:57:e11', 'bn:64:160:57'),
          ('conv:64:128:80:28:c11', 'conv:64:128:80:28:c11'), ('conv:64:128:80:28:e11', 'conv:64:128:80:28:e11'),
          ('conv:32:32:320:114:911', 'conv:32:32:320:114:911'), ('conv:32:32:320:114:e11', 'conv:32:32:320:114:e11')]
PAIRS2 = [
    # can ANY MFMA-bound kernel hide a bandwidth-bound one?  pure MFMA issue (no LDS, few registers) at 2 and 1 waves per SIMD
    ('mfma:512:90:4', 'bn:64:160:57'), ('mfma:256:180:4', 'bn:64:160:57'), ('mfma:512:90:4', 'bn:16:640:229'), ('mfma:512:90:4', 'attn'),
    ('mfma:512:90:4', 'mfma:512:90:4'), ('mfma:256:180:4', 'mfma:256:180:4'),
    # the LDS-free direct conv kernel (thousands of small workgroups) and the persistent kernel with a two-row band (96 KB of LDS)
    ('conv:64:64:160:57:1', 'bn:64:160:57'), ('conv:64:64:160:57:2911', 'bn:64:160:57'), ('conv:64:64:160:57:2911', 'conv:64:64:160:57:2911'),
]
PAIRS = [
    # the #1 kernel of the step (conv3x3_wino2_k<1,1,8,HALF>: 171 VGPRs -> 160 free per SIMD beside its two waves) and its 230-register sibling
    ('conv:64:64:160:57:911', 'bn:64:160:57'), ('conv:64:64:160:57:811', 'bn:64:160:57'),
    ('conv:16:16:640:229:911', 'bn:16:640:229'), ('conv:16:16:640:229:811', 'bn:16:640:229'),
    ('conv:128:128:80:28:a11', 'bn:128:80:28'), ('conv:128:128:80:28:611', 'bn:128:80:28'),
    ('conv:64:64:160:57:911', 'c1:32:64:160:57'), ('conv:64:64:160:57:911', 'attn'), ('conv:64:64:160:57:911', 'gemm:5120:768:176'),
    ('conv:64:64:160:57:911', 'conv:64:64:160:57:911'), ('conv:16:16:640:229:911', 'conv:64:64:160:57:911'),
    ('bn:64:160:57', 'bn:16:640:229'), ('attn', 'bn:16:640:229'),
]

if __name__ == '__main__':
    args = [a for a in sys.argv[1:]]
    pairs = PAIRS2 if args == ['set2'] else PAIRS3 if args == ['set3'] else PAIRS4 if args == ['set4'] else ([tuple(a.split('+')) for a in args] or PAIRS)
    print(f'# N = {N} launches per branch and graph, B = {B}; microseconds per launch (pair: per launch of each)')
    for xs, ys in pairs:
        nx, fx = make(xs, 'x')
        ny, fy = make(ys, 'y')
        try:
            fx(); fy()
        except AssertionError as e:
            print(f'{nx} | {ny}: skipped ({e})')
            continue
        gx, gy = graph_of([fx]), graph_of([fy])
        tx, ty = time_graph(gx), time_graph(gy)
        txy = time_graph(graph_of([fx, fy]))
        tser = time_graph(graph_of([lambda: (fx(), fy())]))
        t2g = time_two_graphs(gx, gy)
        teg = time_eager_two_streams(fx, fy)
        hidden = (tx + ty - txy) / min(tx, ty)
        print(f'{nx:44s} {tx:7.1f} | {ny:44s} {ty:7.1f} | forked graph {txy:7.1f} (hidden {hidden:5.2f}) | one branch, alternating {tser:7.1f} | '
              f'two graphs on two streams {t2g:7.1f} (hidden {(tx + ty - t2g) / min(tx, ty):5.2f}) | eager, two streams {teg:7.1f} (hidden {(tx + ty - teg) / min(tx, ty):5.2f})')
