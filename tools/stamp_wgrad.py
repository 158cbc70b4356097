"""s_memtime stamps of wgrad_mfma_k (RV_ABLATION build): where one workgroup's time goes.
    RECONVAT_HIP_LIB=reconvat_amd/libreconvat_hip_abl.so python tools/stamp_wgrad.py c3 32 32 320 114"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
dev = torch.device('cuda:0')
dbg = torch.zeros(64, dtype=torch.int64, device=dev)
os.environ['RV_DBG_PTR'] = str(dbg.data_ptr())
from reconvat_amd import ops
kind, cin, cout, h, w = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
B = 8
x = torch.rand(B, h, w, cin, device=dev) - 0.5
wshape = {'c3': (cout, cin, 3, 3), 'c1': (cout, cin, 1, 1), 'down': (cout, cin, 2, 2)}[kind]
wt = (torch.rand(*wshape, device=dev) - 0.5) * 0.1
ho, wo = ops._out_hw(kind, h, w, None)
dy = torch.rand(B, ho, wo, cout, device=dev) - 0.5
for _ in range(3):
    ops.conv_wgrad(kind, x, dy, wt, True)
torch.cuda.synchronize()
dbg.zero_()
ops.conv_wgrad(kind, x, dy, wt, True)
torch.cuda.synchronize()
d = dbg.cpu().tolist()
t0 = d[0]
print(f'{kind} {cin}->{cout} {h}x{w}  (s_memtime ticks; 100 MHz constant clock if the values look small, else shader cycles)')
print(f'  entry->zero-fill done      {d[1] - t0}')
print(f'  ->first row resident       {d[2] - d[1]}')
print(f'  ->row loop done            {d[3] - d[2]}   (waiting {d[6]}, issuing DMA {d[8]}, k-steps {d[7]})')
print(f'  ->fold done                {d[4] - d[3]}')
print(f'  ->stores done              {d[5] - d[4]}')
print(f'  total                      {d[5] - t0}')
