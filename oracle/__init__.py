"""Test infrastructure only (CPU oracle).  See oracle/cal_oracle.py header."""
