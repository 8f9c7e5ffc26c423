"""TEST INFRASTRUCTURE ONLY -- CPU restatement ("oracle") of the SVision hot path.

Nothing under ``oracle/`` is product code.  Only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may
import it, and only as the checker.  The product (``svision_amd``) never
imports this package and fails loudly when its HIP library is missing.
"""
