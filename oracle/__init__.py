"""Test-infrastructure package: CPU oracle of the MIMO U-Net hot path.
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import it."""
