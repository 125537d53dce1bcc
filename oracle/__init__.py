"""TEST INFRASTRUCTURE ONLY — CPU oracle for the StreamingFlow GRU-ODE hot path.

Nothing under ``oracle/`` is product code.  Only ``tests/``, ``__graft_entry__.smoke()`` and
``bench.py``'s ``cpu_baseline`` leg may import it, and only as the checker / reported baseline.
The product (``streamingflow_amd``) never imports this package and has no CPU fallback.

Parity status: the reference has no tests or golden vectors for this path, so the
oracle is pinned against outputs of the reference itself, imported in the build container by
``oracle/gen_golden.py`` (fixtures committed under ``tests/golden/``).
"""
