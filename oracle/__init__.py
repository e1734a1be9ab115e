"""CPU oracle (test infrastructure only): see fbank_oracle.py and sanm_oracle.py."""
