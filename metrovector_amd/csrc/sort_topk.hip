// sort_topk.hip -- ranking a whole shard: the device-wide sort behind a search for more results than passes of the
// streaming kernel are worth (api.hip: search_sorted_k).
//
// The reference keeps a heap of k + 1 entries whatever k is (examples/similarity_search.rs:143, :159-168) and sorts what is
// left (:172-173); for k in the thousands and beyond, the cheapest exact equivalent on the device is to let the streaming
// kernel write every row's composite (order key << 32 | row: 8 bytes per row, beside the dim * es it reads) and sort them.
// The sort is the library's (rocPRIM onesweep radix sort, as hipBLASLt is for a plain GEMM): 8 digit passes over 8 bytes
// per row -- 10M rows in ~0.5 ms next to the 4.4 ms scan that produced them.  Composites are distinct and their order is
// the result order (score key, then row position), so the sort needs no comparator and no stability.
#include "aux_kernels.h"

#include <rocprim/device/device_radix_sort.hpp>

namespace mvf {

hipError_t sort_composites(void* tmp, size_t* tmp_bytes, uint64_t* a, uint64_t* b, size_t n, uint64_t** sorted, hipStream_t s) {
    rocprim::double_buffer<uint64_t> db(a, b);
    hipError_t e = rocprim::radix_sort_keys(tmp, *tmp_bytes, db, n, 0u, 64u, s, false);
    if (tmp && sorted) *sorted = db.current();
    return e;
}

}  // namespace mvf
