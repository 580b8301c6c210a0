from .collation import batched_coordinates, sparse_collate, batch_sparse_collate, SparseCollation  # noqa: F401
from .quantization import (sparse_quantize, quantize, quantize_label, unique_coordinate_map,  # noqa: F401
                           segment_reduce)
from .scene_prefetch import ScenePrefetcher  # noqa: F401
