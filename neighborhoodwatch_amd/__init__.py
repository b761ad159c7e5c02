"""neighborhoodwatch_amd -- MI355X-native drop-in for the GPU hot path of phact/neighborhoodwatch.

Only the brute-force kNN path is here (cu_knn.compute_knn -> knn() -> merge.merge_indices_and_distances);
see DESIGN.md for scope and INTEGRATION.md for how the reference binds to it.
"""
__version__ = "0.1.0"

from .engine import KnnEngine, ShardedKnnEngine, knn, device_count, mem_info, resolve_metric  # noqa: F401
