// Stable device radix sort of 64-bit keys (hipCUB / rocPRIM), kept in its own translation unit: the deterministic
// mode of the MSDeformAttn-3D backward (include/transoar_msda3d.h, TRANSOAR_MSDA3D_DETERMINISTIC) orders the sampling
// points by (cell, canonical point index) with it instead of by the arrival order of atomic cursors.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

namespace transoar {

// Sorts n keys on bits [0, end_bit).  keys / alt: two buffers of n keys (the input in `keys`); returns the buffer that
// holds the result in *sorted.  temp == nullptr: only reports the scratch size in *temp_bytes.
int sort_keys64(unsigned long long* keys, unsigned long long* alt, long n, int end_bit, void* temp, size_t* temp_bytes,
                unsigned long long** sorted, hipStream_t st) {
  if (n < 0 || n >= (1L << 31)) return static_cast<int>(hipErrorInvalidValue);      // hipCUB takes the count as an int
  hipcub::DoubleBuffer<unsigned long long> buf(keys, alt);
  const hipError_t e = hipcub::DeviceRadixSort::SortKeys(temp, *temp_bytes, buf, static_cast<int>(n), 0, end_bit, st);
  if (sorted != nullptr) *sorted = buf.Current();
  return static_cast<int>(e);
}

}  // namespace transoar
