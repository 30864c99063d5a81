// tests/emu/emu_stubs.cpp — TEST INFRASTRUCTURE: entry points of bvg_transpose.hip (rocPRIM radix sort: not emulated); the transposition feed
// reports failure in the emulated library.
#include "bvg_kernels.h"
namespace bvg {
void launch_union_count(const uint64_t*, const int64_t*, const uint64_t*, const int64_t*, int64_t, int32_t*, hipStream_t) { abort(); }
void launch_union_write(const uint64_t*, const int64_t*, const uint64_t*, const int64_t*, int64_t, const uint64_t*, int64_t*, hipStream_t) { abort(); }
size_t transpose_temp_bytes(uint64_t, int64_t) { return 0; }
hipError_t transpose_pairs(const uint64_t*, int64_t, uint64_t, const int64_t*, int64_t*, uint64_t*, void*, size_t, uint64_t*, int64_t*, unsigned*, hipStream_t) { return hipErrorInvalidValue; }
}
