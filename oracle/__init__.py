"""ORACLE package -- CPU restatement of the reference's hot path.  Test infrastructure only:
imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg; never by mlsp_amd/."""
