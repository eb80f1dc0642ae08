"""heamd -- Python host-side mirror of the reference's PolyContext / Context<Bfv> interface over the C ABI.

Thin ctypes layer over ``lib/libhe_amd.so`` (include/he_amd.h).  Device memory is borrowed from PyTorch tensors
(``int64`` storage reinterpreted as UInt64 words); PyTorch is plumbing only -- every operation is one of the
hand-written HIP kernels behind the C ABI.  There is NO CPU fallback: if the library is missing or no GPU is present
the calls raise.

Names follow the reference (Sources/HomomorphicEncryption): PolyContext.forwardNtt -> PolyContext.forward_ntt_, etc.
"""
from .binding import (  # noqa: F401
    BfvContext,
    BfvContext32,
    DeviceGroup,
    HeError,
    PolyContext,
    current_device,
    device_count,
    galois_element_rotating_columns,
    galois_element_swapping_rows,
    generate_primes,
    library_path,
    load_library,
    narrow_u64,
    set_device,
    scratch_cached_bytes,
    shard_bounds,
    set_scratch_cache,
    stream_copy,
    to_device,
    to_device32,
    to_host,
    to_host32,
    trim_scratch,
    version,
    widen_u32,
)
