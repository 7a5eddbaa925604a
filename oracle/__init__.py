"""ORACLE -- test infrastructure only.  CPU restatement of the reference's GP-expert path
(see oracle/gp.py for the citation rules and the "parity unpinned" statement).  Importable from
tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg; never from the product package."""
from . import gp, spn  # noqa: F401
