"""Test infrastructure (CPU oracle). Not part of the product; see oracle/oracle.py header."""
