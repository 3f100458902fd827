"""Test-infrastructure oracle package (see gingr_oracle.py header). Not part of the product."""
