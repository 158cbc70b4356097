"""Data-parallel plumbing of the hot path: one process per GPU, ONE all-reduce (sum) of the flat gradient bucket per
optimiser step (`train.allreduce_gradients`), the 1/world mean folded into the Adam kernel.  The reference has no distributed
code at all (train_UNet_Onset_VAT.py:34: one `device='cuda:0'`); this is the build's own surface.

Backends (`RV_DP_BACKEND`, default `nccl`):
  * `nccl`  -- RCCL over xGMI, the collective runs on the device buffers in stream order (what an 8-GPU node runs);
  * `gloo`  -- the same call sites with the bucket staged through ONE pinned host buffer.  It exists so that the rank logic can be
    exercised with the REAL kernels on a box with a single GPU: RCCL refuses two ranks on one device, gloo does not
    (`RV_DP_SAME_GPU=1` puts every rank on `cuda:0`).  Never the configuration to measure.
"""
import os
from datetime import timedelta

import torch
import torch.distributed as dist

_WAIT_GROUP = [None]
_HOST = {}


def backend():
    b = os.environ.get('RV_DP_BACKEND', 'nccl').lower()
    if b not in ('nccl', 'gloo'):
        raise SystemExit(f'RV_DP_BACKEND={b!r}: expected nccl or gloo')
    return b


def same_gpu():
    return os.environ.get('RV_DP_SAME_GPU') == '1'


def local_device():
    """The device of this rank: cuda:LOCAL_RANK, or cuda:0 for every rank with RV_DP_SAME_GPU=1 (gloo only)."""
    if same_gpu():
        if backend() != 'gloo':
            raise SystemExit('RV_DP_SAME_GPU=1 needs RV_DP_BACKEND=gloo (RCCL refuses two ranks on one device)')
        return torch.device('cuda', 0)
    return torch.device('cuda', int(os.environ.get('LOCAL_RANK', '0')))


def pin_rank_cpus(local_rank=None, local_world=None):
    """Give this rank its own slice of the host cores (`os.sched_setaffinity`): rank r of n gets CPUs [r*k, (r+1)*k) of the process's
    current affinity set, k = len // n -- next to the OMP_NUM_THREADS split of the launcher, so eight launch threads (and their
    torch intra-op pools) do not migrate over each other.  Call it BEFORE anything touches the GPU (it is plain host state, but the
    HIP runtime's helper threads inherit the mask of the thread that creates them).  Returns the sorted CPU list (or None when the
    platform has no affinity API / the set is smaller than the rank count)."""
    if not hasattr(os, 'sched_setaffinity'):
        return None
    r = int(os.environ.get('LOCAL_RANK', '0')) if local_rank is None else local_rank
    n = int(os.environ.get('LOCAL_WORLD_SIZE', os.environ.get('WORLD_SIZE', '1'))) if local_world is None else local_world
    cpus = sorted(os.sched_getaffinity(0))
    k = len(cpus) // max(n, 1)
    if n <= 1 or k < 1:
        return cpus
    mine = cpus[r * k:(r + 1) * k]
    try:
        os.sched_setaffinity(0, mine)
    except OSError:                 # (a sandbox that does not allow it: the ranks simply share the launcher's CPU set)
        return cpus
    return mine


def init(device, long_wait_group=False):
    """Join the process group the launcher described (RANK / WORLD_SIZE / MASTER_*; rendezvous on 127.0.0.1 by default).  The
    training group keeps the DEFAULT collective timeout, so a rank that dies leaves its peers blocked for minutes, not hours;
    `long_wait_group=True` additionally creates a gloo group with a 4 h timeout for `wait_for_rank0` (rank 0's whole-song
    validation passes)."""
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('MASTER_PORT', '29541')
    os.environ.setdefault('RANK', '0')
    os.environ.setdefault('WORLD_SIZE', '1')
    if backend() == 'nccl':
        dist.init_process_group('nccl', device_id=device)
    else:
        dist.init_process_group('gloo')
    if long_wait_group and dist.get_world_size() > 1:
        _WAIT_GROUP[0] = dist.new_group(backend='gloo', timeout=timedelta(hours=4))


def active():
    return dist.is_available() and dist.is_initialized()


def all_reduce(t, op=None):
    """dist.all_reduce on a device tensor.  RCCL: in place, in stream order.  gloo: device -> pinned host (synchronises the
    current stream) -> all-reduce on the host -> back (asynchronous copy on the current stream; the pinned buffer is reused only
    after the next device -> host copy, which waits for that stream again)."""
    op = dist.ReduceOp.SUM if op is None else op
    if dist.get_backend() != 'gloo' or not t.is_cuda:
        dist.all_reduce(t, op=op)
        return
    key = (t.numel(), t.dtype)
    host = _HOST.get(key)
    if host is None:
        host = _HOST[key] = torch.empty(t.numel(), dtype=t.dtype).pin_memory()
    flat = t.reshape(-1)
    host.copy_(flat)
    dist.all_reduce(host, op=op)
    flat.copy_(host, non_blocking=True)


def barrier():
    if active():
        dist.barrier()


def wait_for_rank0():
    """Peers wait here while rank 0 runs a validation pass (long-timeout gloo group when `init` made one)."""
    if not active() or dist.get_world_size() < 2:
        return
    if _WAIT_GROUP[0] is not None:
        dist.barrier(group=_WAIT_GROUP[0])
    else:
        dist.barrier()


def shutdown():
    if active():
        dist.destroy_process_group()
    _WAIT_GROUP[0] = None
