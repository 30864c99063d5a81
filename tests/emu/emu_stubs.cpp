// tests/emu/emu_stubs.cpp — TEST INFRASTRUCTURE: entry points of bvg_transpose.hip (rocPRIM radix sort: not emulated); the transposition feed
// reports failure in the emulated library.
#include "bvg_kernels.h"
namespace bvg {
void launch_union_count(const uint64_t*, const int64_t*, const uint64_t*, const int64_t*, int64_t, int32_t*, hipStream_t) { abort(); }
void launch_union_write(const uint64_t*, const int64_t*, const uint64_t*, const int64_t*, int64_t, const uint64_t*, int64_t*, hipStream_t) { abort(); }
size_t transpose_temp_bytes(uint64_t, int64_t) { return 0; }
// the other experimental kernels (workgroup row kernel, streaming kernel, flow kernel) are not emulated
void launch_rows_wg_decode(const DecodeArgs&, uint32_t, int, hipStream_t) { abort(); }
size_t rows_wg_static_lds(int) { return 0; }
void launch_stream_decode(const DecodeArgs&, uint32_t, bool, bool, hipStream_t) { abort(); }
size_t flow_scratch_bytes_per_wave(int) { return 0; }
size_t flow_lds_bytes(uint32_t) { return 0; }
void launch_flow_scan(const DecodeArgs&, uint32_t, uint32_t, void*, uint32_t, hipStream_t) { abort(); }
hipError_t transpose_pairs(const uint64_t*, int64_t, uint64_t, const int64_t*, int64_t*, uint64_t*, void*, size_t, uint64_t*, int64_t*, unsigned*, hipStream_t) { return hipErrorInvalidValue; }
}
