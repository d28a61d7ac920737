"""CPU oracle for the assembly path — TEST INFRASTRUCTURE ONLY (parity unpinned, see SPEC.md).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package.
"""
from .binding import CpuMt, Oracle, build_oracle, oracle_fit  # noqa: F401
