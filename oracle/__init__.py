"""CPU oracle for the UniVST SD-v1.5 hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``univst_amd/`` (the product) may import
this package; only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` do, and only as the checker.

Every function is a plain-PyTorch (fp32 by default) restatement of the
reference algorithm and cites the reference file:line it follows (paths are
relative to the reference checkout).  The restatement is pinned against outputs
of the reference's own modules imported in the build container
(``tests/golden/make_golden.py`` -> ``tests/golden/*.pt``).

Parity status of third-party pieces (diffusers 0.35.1 ``Attention``,
``FeedForward``, ``Timesteps``, ``DDIMScheduler``; OpenCV 4.9 ``remap``):
**parity unpinned by the reference** (it ships no tests and does not vendor
them); they are restated from their published semantics in
``oracle/_stubs/diffusers`` (generator-side) and here (checker-side).
"""
