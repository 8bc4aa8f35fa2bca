"""CPU oracle for the PC-GNN pick / choose / aggregate hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is part of the product:
only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import it, and there only as the checker / the timed CPU
baseline - never as the thing shipped.  The product path (``pc-gnn_amd``)
never imports this package and fails loudly when its HIP library is missing.

Parity status: PINNED.  ``tests/golden/*.npz`` were produced by importing the
reference's own modules (``/root/reference/src/{layers,model,utils,graphsage}.py``,
``cuda=False``) in the build container with ``tests/golden/make_golden.py``;
``tests/test_oracle_golden.py`` checks every function here against them.
The reference ships no tests / golden vectors of its own.
"""
