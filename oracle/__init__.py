"""CPU oracle for the JAMUN walk-jump sampling path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``jamun_amd/`` (the product) may
import this package; only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` use it, and there only as the checker.

The oracle is an op-for-op PyTorch-CPU restatement of the reference's
algorithm (materialised per-edge tensor-product weights, einsum tensor
product, ``index_add_`` mean aggregation, brute-force radius graph, Python
BAOAB loop, one extra denoiser forward per saved frame).  Every function cites
the reference ``file:line`` it follows (paths relative to ``/root/reference``).

PARITY STATUS
-------------
* Integrator (``walk.py``): PINNED.  ``tests/golden/baoab_*.npz`` and
  ``aboba_*.npz`` were produced by importing the reference's own
  ``src/jamun/sampling/mcmc/functional/_splitting.py`` and
  ``src/jamun/sampling/walkjump/_single_measurement.py`` in the build
  container (``tests/golden/make_golden.py``).
* Integer encodings (``residue_metadata``): PINNED the same way.
* Denoiser forward (``e3.py``, ``denoiser.py``, ``graph.py``): PARITY UNPINNED.
  The arithmetic lives in third-party wheels that are absent from
  ``/root/reference`` and from this image: e3nn 0.5.4, torch_geometric 2.6.1 /
  torch_cluster 1.6.3, torch_scatter 2.1.2 (``env/requirements.txt``).  Their
  published algorithms are restated here and anchored on the reference's call
  sites; the reference ships no tests or golden vectors for this boundary
  (SURVEY.md §4).  Pins available: analytic known-answer properties
  (zero gain, identity noise scaling, SE(3) equivariance, variance
  preservation of e3nn's normalisation), all in ``tests/test_oracle.py``.
"""
