"""CPU oracle for the Detect hot path.  TEST INFRASTRUCTURE: importable only from tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg (never from yolact_amd)."""
