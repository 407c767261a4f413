// Helpers shared by the fused float kernels (kernel_fused512.hpp, kernel_fused1024.hpp): the uniform tile
// cursor, the geometry of a tile's sample window, exact integer pre-emphasis, the LDS-only barrier.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "kernels_generic.hpp"

namespace mfcc_fc {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

// Workgroup barrier that orders LDS traffic only.  __syncthreads() is a full workgroup fence: it also
// drains vmcnt, i.e. waits for the prefetch loads of the next tile and for role 0's output stores,
// which nothing on the other side of the barrier depends on.
__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// Uniform (SGPR) cursor over the workgroup's tiles: tile = ch * tiles_per_ch + t_in.  Advancing by
// the grid size is a handful of scalar adds with one carry -- no multiply or division in the loop
// (a wave's scalar instructions issue ~10 clocks apart; the multiply form of this cost ~450 clocks
// per tile in every wave).
struct Cursor {
    int ch, t_in;
    const int16_t *ptr;      // the tile's first sample: s.pcm + ch * ch_stride + t_in * kTileHop
};

struct LaunchGeom {
    int tiles_per_ch, n_ch, grid_div, grid_mod;      // grid = grid_div * tiles_per_ch + grid_mod
    long long step_ptr, wrap_ptr;                    // samples: ptr step per grid stride / extra step on carry
    int t_lo, t_hi;                                  // tiles t_lo <= t_in <= t_hi have their window inside the channel
};

__device__ __forceinline__ void advance(Cursor &c, const LaunchGeom &g) {
    c.t_in += g.grid_mod;
    c.ch += g.grid_div;
    c.ptr += g.step_ptr;
    if (c.t_in >= g.tiles_per_ch) {
        c.t_in -= g.tiles_per_ch;
        ++c.ch;
        c.ptr += g.wrap_ptr;
    }
}

// uniform (scalar) geometry of a tile's window; every wave computes it, fetchers or not
struct Window {
    const int16_t *ptr;      // the tile's first sample
    int t_in;
    int shift;
    bool inside;             // whole window (and the dword in front of it) lies inside the channel
};

__device__ __forceinline__ Window window_of(const Cursor &c, const LaunchGeom &g) {
    Window w;
    w.ptr = c.ptr;
    w.t_in = c.t_in;
    const int mis = (int)((reinterpret_cast<uintptr_t>(c.ptr) & 15) >> 1);   // samples past alignment
    // t_lo / t_hi (host): first - 7 - 2 >= -halo (the dword in front of piece 0) and first + kSUsed <= n_samples
    w.inside = c.t_in >= g.t_lo && c.t_in <= g.t_hi;
    w.shift = w.inside ? mis : 0;
    return w;
}

// e[k] = 32 x[k] - 31 x[k-1] for the 8 samples packed in v, x[-1] = high half of prev.  One
// v_dot2_i32_i16 per sample (the three-operand form: for the builtin hipcc picks v_dot2c, which costs
// an extra v_mov 0 per sample); the 1/32 is in the window table.
__device__ __forceinline__ void preemph8(int prev, const i32x4 &v, float *__restrict__ dst) {
    const int c3132 = 0x0020ffe1;                  // (int16 -31, int16 32)
    float e[8];
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        const int before = m ? v[m - 1] : prev;
        const int pe = (int)__builtin_amdgcn_alignbit((unsigned)v[m], (unsigned)before, 16u);   // (x[2m-1], x[2m])
        int e0, e1;
        asm("v_dot2_i32_i16 %0, %1, %2, 0" : "=v"(e0) : "v"(pe), "s"(c3132));
        asm("v_dot2_i32_i16 %0, %1, %2, 0" : "=v"(e1) : "v"(v[m]), "s"(c3132));
        e[2 * m] = (float)e0;
        e[2 * m + 1] = (float)e1;
    }
    reinterpret_cast<f32x4 *>(dst)[0] = (f32x4){e[0], e[1], e[2], e[3]};
    reinterpret_cast<f32x4 *>(dst)[1] = (f32x4){e[4], e[5], e[6], e[7]};
}

}  // namespace mfcc_fc
