"""CPU oracle (test infrastructure only; see sisr_oracle.py header)."""
