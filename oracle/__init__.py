"""CPU oracle -- TEST INFRASTRUCTURE ONLY (see oracle/vs_oracle.cpp header)."""
