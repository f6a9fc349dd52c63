"""CPU oracle (test infrastructure only; never imported by the shipped package)."""
