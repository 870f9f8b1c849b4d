"""CPU ORACLE package -- test infrastructure only (see oracle/clip_oracle.py and oracle/oracle.c headers).
Importable only from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg."""
