"""CPU oracle for the SeLaVi hot path -- TEST INFRASTRUCTURE ONLY.

Everything under ``oracle/`` is a CPU restatement of the reference algorithm
(facebookresearch/selavi: ``model.py``, ``utils.get_loss``, ``src/sk_utils.py``
and the torchvision 0.4.2 networks those files instantiate).  It exists to
*check* the HIP path.  Only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import it; the product package
``selavi_amd`` never does and fails loudly when its HIP library is missing.

Parity status
-------------
* ``sk_ref``  : pinned against the reference's own ``optimize_L_sk_gpu`` executed
  in the build container (``tests/golden/make_golden.py`` -> ``tests/golden/sk_*.npz``).
* ``model_ref`` heads / AVModel.forward / get_loss : pinned against the reference's
  own ``model.py`` / ``utils.py`` executed in the build container.
* ``model_ref`` trunks (R(2+1)D-18, ResNet-9): the arithmetic lives in torchvision
  0.4.2, which is NOT vendored in the reference and not installed here -> this part is
  "parity unpinned" by the reference; it is pinned instead by the documented parameter
  counts (31 505 325 / 11 689 512) and state-dict key layout, see tests/test_oracle.py.
* ``input_ref.clip_augmentation_ref`` : pinned against the reference's own
  ``datasets/video_transforms.clip_augmentation`` executed in the build container
  (``tests/golden/make_input_golden.py`` -> ``clip_aug.npz``), bit-identical at production sizes.
* ``input_ref.logfbank_ref`` : restates python_speech_features 0.6 (absent) -> "parity unpinned" by
  reference outputs; the arithmetic underneath is numpy.fft.rfft.
"""
