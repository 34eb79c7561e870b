"""variantstore_amd -- MI355X-native region-query engine for VariantStore indexes.

Only the position-indexed variation-graph lookup path is here (query types 6
and 4 of `variantstore query`), as HIP kernels behind a C ABI
(include/variantstore_hip.h).  See DESIGN.md.
"""
from .api import Comm, DeviceArray, QueryResult, Variant, VariantStore, VariantStoreError  # noqa: F401
