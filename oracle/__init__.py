"""CPU oracle (test infrastructure only): see als_ref.py, als_ref.c and c_oracle.py."""
