"""Shared constants (reference easy_rec/python/utils/constant.py)."""
SAMPLE_WEIGHT = 'SAMPLE_WEIGHT'
