"""CPU oracle for the ReconVAT per-segment training hot path.

TEST INFRASTRUCTURE ONLY.  Nothing in ``reconvat_amd`` (the product) imports
this package.  The only legal importers are ``tests/``, ``__graft_entry__.smoke()``
and the ``cpu_baseline`` leg of ``bench.py`` -- and there only as the checker /
the timed CPU baseline, never as the thing shipped.

The oracle is a functional, pure-PyTorch (CPU, fp32) restatement of the
reference algorithm; every function cites the reference ``file:line`` it
follows.  It is pinned against outputs of the reference itself (imported in the
build container by ``tests/golden/make_golden.py``) through the fixtures under
``tests/golden/``.

Parity status
-------------
* everything from log/normalise onwards: PINNED --
  golden vectors were produced by running ``/root/reference`` itself.
* the three nnAudio-0.2.0 helpers the reference's vendored ``Spectrogram.py``
  star-imports but does not vendor (``create_fourier_kernels``,
  ``broadcast_dim``, ``mel``; pip package ``nnAudio==0.2.0``,
  ``requirements.txt:5``): PARITY UNPINNED -- restated from the published
  algorithm (Hann-windowed DFT kernels, librosa-0.7 Slaney mel filterbank) and
  cross-checked against ``torch.stft``; the reference holds no test or fixture
  for them.
* the Onsets&Frames BiLSTM baseline (``oracle/onset_frames.py``): PINNED -- ``tests/golden/onset_frames.npz`` holds outputs of
  the reference's own ``OnsetsAndFrames_VAT_full`` (Dropout probabilities set to 0 on the instances).
* the data-item rule (``oracle/dataset.py``, integer / byte work): PINNED bit-exactly -- ``tests/golden/dataset.npz``
  holds the outputs of the reference's own ``PianoRollAudioDataset.__getitem__`` on in-memory tracks.
"""
