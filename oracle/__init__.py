"""CPU oracle (test infrastructure only) -- see oracle.py header."""
