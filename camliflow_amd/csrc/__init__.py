"""Operator boundary -- same four names the reference exports from ``models/csrc/__init__.py:1``."""
from .wrapper import correlation2d, furthest_point_sampling, squared_distance, k_nearest_neighbor

__all__ = ["correlation2d", "furthest_point_sampling", "squared_distance", "k_nearest_neighbor"]
