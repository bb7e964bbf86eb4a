// sort_topk.hip -- ranking a whole shard: the device-wide sort behind a search for more results than passes of the
// streaming kernel are worth (api.hip: search_sorted_k).
//
// The reference keeps a heap of k + 1 entries whatever k is (examples/similarity_search.rs:143, :159-168) and sorts what is
// left (:172-173); for k in the thousands and beyond, the cheapest exact equivalent on the device is to let the streaming
// kernel write every row's composite (order key << 32 | row: 8 bytes per row, beside the dim * es it reads) and sort them.
// The sort is the library's (rocPRIM onesweep radix sort, as hipBLASLt is for a plain GEMM): 4 digit passes over 8 bytes
// per row -- 10M rows in ~0.5 ms next to the 4.4 ms scan that produced them.  Composites are distinct and their order is
// the result order (score key, then row position): no comparator, and the position half needs no sorting (sort_composites).
#include "aux_kernels.h"

#include <rocprim/device/device_radix_sort.hpp>

namespace mvf {

hipError_t sort_composites(void* tmp, size_t* tmp_bytes, uint64_t* a, uint64_t* b, size_t n, uint64_t** sorted, hipStream_t s) {
    rocprim::double_buffer<uint64_t> db(a, b);
    // The entries are rank entries (mvf_common.h: position << 32 | key): only the KEY half, bits 0..31, is sorted -- both
    // callers produce them in ascending position (the dump writes row r to slot r; the merge's slot index) and the radix
    // sort is stable, so equal keys keep that order: the order of the composites in four digit passes instead of eight.
    hipError_t e = rocprim::radix_sort_keys(tmp, *tmp_bytes, db, n, 0u, 32u, s, false);
    if (tmp && sorted) *sorted = db.current();
    return e;
}

}  // namespace mvf
