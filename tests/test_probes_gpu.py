"""The two hardware facts the Winograd staging and the unaligned 16-byte accesses rely on, checked on the device under test
(tools/probes/*.hip are stand-alone HIP programs: compiled with hipcc here, run on cuda:0)."""
import os
import shutil
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run_probe(name, tmp_path):
    hipcc = shutil.which('hipcc') or '/opt/rocm/bin/hipcc'
    exe = str(tmp_path / name)
    src = os.path.join(ROOT, 'tools', 'probes', name + '.hip')
    r = subprocess.run([hipcc, '--offload-arch=gfx950', '-O2', src, '-o', exe], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    return r.returncode, r.stdout


def test_buffer_lds_dma_zero_fills_out_of_range_lanes(tmp_path):
    """conv3x3_wino_k / wgrad_wino_k: halo columns, padded channel quads and rows outside the image are out-of-range buffer offsets."""
    rc, out = _run_probe('buffer_lds_oob', tmp_path)
    assert rc == 0 and 'zero-filled (0 mismatches)' in out, out


def test_lds_dma_16_bytes_takes_4_byte_aligned_sources(tmp_path):
    """attention staging of 229-float rows: 16-byte DMA lanes on sources that are only 4-byte aligned."""
    rc, out = _run_probe('glds16_unaligned', tmp_path)
    assert rc == 0 and out.count(': ok') == 4, out


def test_vector_alu_instructions_are_not_free_next_to_f32_mfma(tmp_path):
    """Round 5: the fact behind the shape of the conv kernels' remaining inefficiency.  v_mfma_f32_16x16x4_f32 runs at the vector-ALU
    rate on this chip (157.3 TFLOP/s for both), and VALU instructions issued between such MFMAs are NOT hidden behind them: four
    v_add_f32 per MFMA stretch the loop by tens of percent (measured +60..80 %), i.e. a kernel's time is (MFMA time + VALU time), and the
    Winograd transforms / epilogues / address arithmetic are paid in full."""
    rc, out = _run_probe('mfma_valu_overlap', tmp_path)
    assert rc == 0, out
    line = [l for l in out.splitlines() if l.startswith('f32 MFMA: 4 VALU per MFMA cost')][-1]
    pct = float(line.split('cost')[1].split('%')[0])
    assert pct > 25.0, out


def test_f32_mfma_is_a_fused_multiply_add_chain_in_k_order(tmp_path):
    """Round 6: what the fused skip conv of the BatchNorm apply kernel relies on (rv_bn_lrelu_fwd_skip must reproduce the MFMA conv kernel bit for bit):
    v_mfma_f32_16x16x4_f32 rounds like d = fma(a3, b3, fma(a2, b2, fma(a1, b1, fma(a0, b0, c)))) -- not like the reversed chain, a pairwise tree or unfused products."""
    rc, out = _run_probe('mfma_f32_order', tmp_path)
    assert rc == 0 and 'fma chain k = 0,1,2,3 from C                           : 0 of' in out, out
    assert 'IS a chain of fused multiply-adds in k order' in out, out
