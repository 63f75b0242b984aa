"""CPU oracle for the chordal Newton-KKT path -- TEST INFRASTRUCTURE ONLY (see chordal_oracle.c)."""
