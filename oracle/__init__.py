"""CPU oracle for the ProMP hot path -- TEST INFRASTRUCTURE ONLY.

This package is a float64 NumPy restatement of the reference algorithm
(jonasrothfuss/ProMP) for the path

    MetaSampleProcessor.process_samples -> MAMLAlgo._adapt -> ProMP.optimize_policy

It is the *checker*, never the product: only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may
import it.  Nothing under ``promp_amd/`` imports it, and the product path
fails loudly when the HIP library is missing.

Pinning status (see DESIGN.md "Oracle"):

* ``oracle.sample_processing`` (reference rows a1-a7) is PINNED: it is checked
  against the reference's own NumPy/SciPy code run in the build container
  (``oracle/gen_golden.py`` imports ``/root/reference`` and writes
  ``tests/golden/sample_proc_*.npz``) and against the reference's property
  tests (tests/test_samplers.py:216-219, :326-361).
* ``oracle.policy`` / ``oracle.promp`` (rows a8-a13) restate a TensorFlow-1
  graph.  TensorFlow is absent from the image and the reference holds no
  numeric golden vectors for these rows (only properties: ratio == 1 at
  unchanged params, tests/test_integration.py:128-175).  The restatement is
  cross-checked by finite differences and by ``torch.autograd`` double
  backward (tests/test_oracle_policy.py); relative to the reference itself
  the autodiff / Adam parts are "unpinned by reference vectors".  The part of
  these rows the reference also implements in NumPy IS pinned by its outputs
  (``tests/golden/dist_reference.npz``): DiagonalGaussian.kl / log_likelihood
  / entropy (row a9) and the KL-coefficient rule of ProMP.optimize_policy.
"""
