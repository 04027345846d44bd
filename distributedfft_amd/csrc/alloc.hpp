// alloc.hpp -- device memory of the library (alloc.hip): hipMalloc, the HIP virtual-memory API in physical chunks, and the
// placement-aware default backing.  See include/dfft_c.h (dfft_malloc) for the contract.
#pragma once
#include <stddef.h>

namespace dfft {
// chunk_mib == 0: hipMalloc; otherwise one virtual range at a never-used address, backed by physical chunks of chunk_mib MiB
// (spread > 1: every spread-th of spread times as many chunks is kept).  Returns 0 or an error code (message: dfft_last_error)
int dev_alloc(size_t bytes, size_t chunk_mib, void **out, int spread = 1);
// the default backing of library-owned memory and of dfft_malloc(DFFT_CHUNK_DEFAULT): placement by measurement for buffers >= 1 GiB
int dev_alloc_default(size_t bytes, void **out);
// frees what dev_alloc / dev_alloc_default returned (any other pointer: hipFree); drains the owning device first
int dev_free(void *ptr);
size_t default_chunk_mib();      // DFFT_DEFAULT_CHUNK_MIB (0 = hipMalloc)
// what the last placement-aware allocation of the process did, as a JSON object (dfft_last_placement_info)
int placement_info_json(char *buf, size_t capacity);
}  // namespace dfft
