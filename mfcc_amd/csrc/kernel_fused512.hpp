// Specialised 512/170/32 float kernel (placeholder until the fused kernel lands: reports
// "unsupported", so MFCC_HIP_IMPL_AUTO uses the generic kernel).
#pragma once
#include <hip/hip_runtime.h>
#include <vector>
#include "kernels_generic.hpp"

namespace mfcc_fused {
struct FusedTables { const void *blob; };
inline bool supported(int, int, int, int) { return false; }
inline bool build_tables(int, double, double, int, std::vector<char> &) { return false; }
inline void bind_tables(const char *b, FusedTables &t) { t.blob = b; }
inline void launch(const mfcc_k::StreamDesc &, const FusedTables &, float *, int, hipStream_t) {}
inline const char *kernel_name() { return "mfcc_fused512_kernel"; }
}  // namespace mfcc_fused
