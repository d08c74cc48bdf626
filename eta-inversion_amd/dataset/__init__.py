"""Evaluation datasets (reference dataset/): only PIE-Bench, the one the headline metric is quoted on."""
from .pie_bench_data import PieBenchData  # noqa: F401
