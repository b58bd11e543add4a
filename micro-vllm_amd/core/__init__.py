"""Host-side mirror of the reference's `core` package for the paged-KV decode path."""
