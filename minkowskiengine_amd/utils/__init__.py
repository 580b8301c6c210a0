from .collation import batched_coordinates, sparse_collate  # noqa: F401
