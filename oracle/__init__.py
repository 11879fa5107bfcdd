"""CPU oracle (test infrastructure).  See oracle/path_b.py — never imported by helicon_amd/."""
