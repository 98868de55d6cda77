"""CPU oracle package -- test infrastructure only (see proxgrad_oracle.py header)."""
