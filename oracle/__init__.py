"""CPU oracle for the U-Net / CAE hot path -- TEST INFRASTRUCTURE ONLY.

This package is a plain ``torch`` (CPU, fp32/fp64) functional restatement of
the reference's hot path (``common/model/Unet3D.py``, ``common/model/Cae3D.py``,
``common/metrics.py:8-28``, the loss recipes of the two learners and the
three-line optimiser core of ``learner/Learner.py:120-122``).

Rules ("Oracle"):

* only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
  ``cpu_baseline`` leg may import it -- never the product package;
* it is pinned against the real reference: ``tests/golden/make_golden.py``
  imports ``/root/reference`` in the build container, runs the reference
  modules on deterministic weights/inputs and stores the results as small
  fixtures under ``tests/golden/``; ``tests/test_oracle_golden.py`` replays the
  same inputs through this restatement.  Parity status: **pinned**.
"""
