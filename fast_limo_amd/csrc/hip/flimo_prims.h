// fast_limo_amd/csrc/hip/flimo_prims.h -- the device-wide primitives the map build / insert / voxel code uses, straight on rocPRIM
// (stable LSD radix sort of (key, value) pairs, prefix sums, running maximum).  Same temporary-storage protocol as rocPRIM:
// a call with tmp == nullptr returns the bytes needed.
#pragma once
#include <hip/hip_runtime.h>
#include <string.h>
#include <rocprim/functional.hpp>
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>
#include <stdint.h>

namespace flimo {

inline hipError_t sort_pairs_u32(void* tmp, size_t& bytes, const uint32_t* keys_in, uint32_t* keys_out, const uint32_t* vals_in,
                                 uint32_t* vals_out, size_t n, unsigned begin_bit, unsigned end_bit, hipStream_t st) {
  return rocprim::radix_sort_pairs(tmp, bytes, keys_in, keys_out, vals_in, vals_out, n, begin_bit, end_bit, st);
}
inline hipError_t sort_pairs_u64(void* tmp, size_t& bytes, const unsigned long long* keys_in, unsigned long long* keys_out,
                                 const uint32_t* vals_in, uint32_t* vals_out, size_t n, hipStream_t st) {
  return rocprim::radix_sort_pairs(tmp, bytes, keys_in, keys_out, vals_in, vals_out, n, 0, 64, st);
}
template <class In, class Out>
inline hipError_t exclusive_sum(void* tmp, size_t& bytes, In* in, Out* out, size_t n, hipStream_t st) {
  return rocprim::exclusive_scan(tmp, bytes, in, out, Out(0), n, rocprim::plus<Out>(), st);
}

}  // namespace flimo
