"""TEST INFRASTRUCTURE.  CPU restatement of the reference's MAC cell (see mac_oracle.py header).
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package."""
