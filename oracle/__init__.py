"""CPU oracle for the PointsToWood inference hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is part of the product:
only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import it, and there only as the checker / the timed CPU baseline.
``pointstowood_amd`` never imports this package.

What it restates (reference = harryjfowen/PointsToWood @ 2025-09-12):

* ``oracle.ops``   - the eight third-party operators the reference forward calls
                     (torch-cluster / torch-scatter / torch-geometric; call sites
                     ``pointstowood/src/model.py:104,105,118,120,136,149`` and
                     ``pointstowood/src/pointnet.py:108,122``).
* ``oracle.net``   - ``Net.forward`` (``pointstowood/src/model.py:226-245``) and the
                     modules it calls, as a functional fp32 CPU forward over a state dict.
* ``oracle.host``  - the feed/consume pieces of ``pointstowood/src/predicter.py:78-105,193-215``.
* ``oracle.preprocess`` / ``oracle.backproject`` - the voxeliser (``preprocessing.py:18-127``) and the
                     back-projection vote (``predicter.py:107-142``) for the "next" rows.

Parity pinning: the reference has no tests/golden vectors of its own and the
third-party wheels are absent and un-pinned (only the wheel index
``torch-2.5.1+cu121`` is named, ``README.md:42-47``), so the *operator-level*
semantics are **parity unpinned** by the reference and are defined by
``oracle.ops`` (upstream's documented CUDA behaviour).  The *model-level*
restatement IS pinned: ``tests/golden/make_golden.py`` imports the reference's own
``src/model.py`` + ``src/pointnet.py`` in the authoring container over
``oracle/stubs`` (thin re-exports of ``oracle.ops`` under the PyG module paths),
and ``tests/test_oracle_golden.py`` checks ``oracle.net`` against those vectors.
"""
