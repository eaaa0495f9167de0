"""CPU oracle -- test infrastructure only (see oracle/ssc_oracle.py)."""
