"""The pieces of bench.py (see benchlib/common.py)."""
