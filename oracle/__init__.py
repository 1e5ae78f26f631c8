"""CPU oracle for the d2frontend hot path -- TEST INFRASTRUCTURE ONLY (see d2fe_oracle.c header)."""
