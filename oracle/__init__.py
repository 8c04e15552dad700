"""TEST INFRASTRUCTURE ONLY — CPU oracle for the GenASM/Scrooge path.

Importable only from tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  The product package (scrooge_amd) never imports this.
"""
