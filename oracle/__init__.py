"""CPU oracle for the SAE / transcoder train step.

TEST INFRASTRUCTURE ONLY. Nothing under ``sparsify_amd/`` imports this package.
The only allowed users are ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py``; there it is the checker (or the timed CPU
baseline), never the product path.

The oracle restates, with hand-derived backward formulas (no autograd), the
algorithm of the reference path
``SparseCoder.encode -> top-k -> decode -> FVU/AuxK -> backward -> unit-norm /
projected-grad / optimizer`` (reference: sparsify/sparse_coder.py:189-295,
sparsify/fused_encoder.py:18-87, sparsify/utils.py:81-85, sparsify/sign_sgd.py:15-23,
sparsify/muon.py:11-40,115-181).

Pinning: every function here is checked by ``tests/test_oracle_golden.py`` against
golden vectors produced by the real reference modules imported in the build
container (``tools/make_golden.py`` -> ``tests/golden/*.npz``).
"""
