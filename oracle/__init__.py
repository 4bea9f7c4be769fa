"""CPU oracle for the RGB+T detection-and-fusion path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is product code: only
``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import it, and there only as the checker / timed CPU
baseline.  The product path (``proben_amd``) never imports this package and
fails loudly when its HIP library is missing.

Every function cites the reference file:line (relative to the upstream
repository) whose arithmetic it restates.  How each piece is pinned (golden
fixtures generated from the imported reference code, reference test tables, or
"parity unpinned") is listed in DESIGN.md section "Oracle".
"""
