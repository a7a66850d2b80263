"""Data-format helpers either side of the hot path."""
from .heatmaps import generate_input_heatmaps  # noqa: F401
