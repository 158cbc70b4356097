"""rv_local_attn_fwd alone (B = 8, L = 640; the two head shapes of the step) under the library RECONVAT_HIP_LIB points at -- one line per run; used by
tools/attn_ablate.sh with the -DRV_ATTN_ABL=mask builds (mask: 1 no staging, 2 no score MFMAs, 4 no softmax, 8 no banded apply, 16 return at entry)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from reconvat_amd import _lib

mask = int(sys.argv[1]) if len(sys.argv) > 1 else 0
names = {0: 'full kernel', 16: 'return at entry (launch floor)', 1: 'no staging', 3: 'no staging, no scores', 7: 'no staging / scores / softmax',
         15: 'everything off (skeleton + barriers)', 2: 'no score MFMAs', 4: 'no softmax', 8: 'no banded apply'}
dev = torch.device('cuda:0')
lib = _lib.load()
st = torch.cuda.current_stream()
out = []
for g, dh in ((6, 128), (4, 229)):
    f = g * dh
    qkv = torch.rand(8 * 640, 3 * f, device=dev) - 0.5
    rel = torch.rand(31, f, device=dev) - 0.5
    o = torch.empty(8, 640, f, device=dev)
    att = torch.empty(8, 640, g, 31, device=dev)
    args = (qkv.data_ptr(), qkv.data_ptr() + 4 * f, qkv.data_ptr() + 8 * f, 3 * f, rel.data_ptr(), o.data_ptr(), att.data_ptr(), 8, 640, g, dh, st.cuda_stream)
    for _ in range(3):
        assert lib.rv_local_attn_fwd(*args) == 0
    best = None
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st)
        for _ in range(10):
            lib.rv_local_attn_fwd(*args)
        e1.record(st)
        e1.synchronize()
        t = e0.elapsed_time(e1) / 10 * 1e3
        best = t if best is None else min(best, t)
    out.append(f'G={g} dh={dh}: {best:6.1f} us')
print(f'mask {mask:>2} {names.get(mask, ""):<40} ' + '   '.join(out))
