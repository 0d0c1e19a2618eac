"""CPU oracle for the variational-layer forward hot path — TEST INFRASTRUCTURE ONLY.

Nothing under ``bayesian_torch_amd`` imports this package.  Allowed importers: ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py``.
"""
