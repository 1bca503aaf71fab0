// gfx950 helpers shared by the conv kernels (conv.hip) and the patch-staged 3x3 weight gradient (wgrad3.hip): exact division by
// a runtime constant, raw buffer descriptors, the LDS-DMA load (`buffer_load_dwordx4 ... lds`) and counted vmcnt waits.
#pragma once
#include "common.h"

// exact unsigned division by a runtime constant (Granlund-Montgomery round-up form), all 32-bit numerators
struct FastDiv { unsigned m, s1, s2; };
static inline FastDiv make_fastdiv(unsigned d) {
    FastDiv f;
    unsigned l = 0;
    if (d < 1) d = 1;
    while ((1ull << l) < d) ++l;
    f.m = (unsigned)(((1ull << 32) * ((1ull << l) - d)) / d + 1);
    f.s1 = l < 1 ? l : 1;
    f.s2 = l > 0 ? l - 1 : 0;
    return f;
}

#define G_OOB 0x80000000u          // buffer offset beyond any descriptor (tensors are < 2 GiB, checked on the host)

typedef unsigned int v2u32 __attribute__((ext_vector_type(2)));
typedef unsigned int v4u32 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;

template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// (ScalarTouch / AY_KERNARG_TOUCH, the scalar-cache prefetch of a kernel's parameter block, live in common.h: the elementwise
// kernels use them too)

// Buffer descriptor (raw, stride 0, `bytes` records) held in 4 SGPRs, and the LDS-DMA load itself.  The DMA is issued
// from inline asm on purpose: hipcc (ROCm 7.2) orders every LDS read behind ALL LDS-DMA it knows to be in flight with a
// vmcnt(0), which would drain the two-steps-ahead pipeline at each step; hidden from its scoreboard, completion is
// tracked by the counted wait_vm<N>() + s_barrier of the step loop alone.  (The compiler's own vmcnt for its loads and
// stores stays correct: completion is in order, so extra younger operations only make its counts conservative.)
typedef int v4i32 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ v4i32 make_srd(const void* ptr, unsigned bytes) {
    const unsigned long long a = (unsigned long long)ptr;
    v4i32 r;
    r.x = __builtin_amdgcn_readfirstlane((int)(unsigned)a);
    r.y = __builtin_amdgcn_readfirstlane((int)(unsigned)(a >> 32));
    r.z = __builtin_amdgcn_readfirstlane((int)bytes);
    r.w = 0x00020000;
    return r;
}
// 64 lanes x 16 B: global (srd base + voff, zero when voff is beyond the descriptor) -> LDS [lds_addr + lane*16, +16)
__device__ __forceinline__ void glds16(v4i32 srd, unsigned lds_addr, unsigned voff) {
    unsigned keep;
    // s_nop 2: (a) one wait state between the M0 write and the LDS-DMA; (b) with the two s_mov it makes five wait states
    // between any VALU that wrote a descriptor SGPR just before this statement (v_readlane of a spilled SGPR) and the
    // VMEM instruction reading it -- hipcc does not look inside the string
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 2\n\tbuffer_load_dwordx4 %1, %3, 0 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(voff), "s"(lds_addr), "s"(srd)
                 : "memory");
}

// Predicated form: the load is issued with EXEC = all ones or zero (`on`, uniform) and EXEC is restored to all ones -- a VMEM
// instruction with EXEC = 0 is skipped by the hardware, so a loader can be straight-line code without a branch per piece (a
// taken branch is an instruction-buffer refill: the branchy piece cursor of k_wgrad3's first versions cost 300-400 cycles per
// piece, profiles/r05_w3_probe_v3_experiments.txt).  Only for kernels whose wavefronts are always fully active outside this
// statement.
__device__ __forceinline__ void w3_glds_if(v4i32 srd, unsigned lds_addr, unsigned voff, unsigned on) {
    unsigned keep;
    const unsigned on_s = (unsigned)__builtin_amdgcn_readfirstlane((int)on);     // (hipcc sometimes keeps a uniform bool in a VGPR)
    asm volatile("s_mov_b32 %0, m0\n\ts_cmp_lg_u32 %4, 0\n\ts_cselect_b64 exec, -1, 0\n\ts_mov_b32 m0, %2\n\ts_nop 2\n\t"
                 "buffer_load_dwordx4 %1, %3, 0 offen lds\n\ts_mov_b64 exec, -1\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(voff), "s"(lds_addr), "s"(srd), "s"(on_s)
                 : "memory", "scc");
}

// The same statement with a different text.  Two glds16 calls in the arms of an if / else are otherwise MERGED by hipcc into one
// statement behind the join, its descriptor / LDS address operands selected per arm -- through VGPRs, which the "s" constraints
// then reject ("invalid operand for instruction").  Use this one in the second arm.
__device__ __forceinline__ void glds16_b(v4i32 srd, unsigned lds_addr, unsigned voff) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 2\n\tbuffer_load_dwordx4 %1, %3, 0 offen lds\n\ts_mov_b32 m0, %0 ; arm b"
                 : "=&s"(keep)
                 : "v"(voff), "s"(lds_addr), "s"(srd)
                 : "memory");
}

// Lane-masked form: only the lanes of `mask` load and write their 16 bytes of the piece (LDS position = lane * 16 whatever the
// mask), EXEC is all ones before and after.  k_pw's two-segment loader: a piece holds whole pixel rows, i.e. chunks of both input
// segments -- two loads with complementary masks and the segments' own descriptors fill it.
__device__ __forceinline__ void glds16_exec(v4i32 srd, unsigned lds_addr, unsigned voff, unsigned long long mask) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b64 exec, %4\n\ts_mov_b32 m0, %2\n\ts_nop 2\n\tbuffer_load_dwordx4 %1, %3, 0 offen lds\n\t"
                 "s_mov_b64 exec, -1\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(voff), "s"(lds_addr), "s"(srd), "s"(mask)
                 : "memory");
}

__device__ __forceinline__ unsigned fdiv(unsigned n, const FastDiv f) {
    const unsigned t = __umulhi(f.m, n);
    return (t + ((n - t) >> f.s1)) >> f.s2;
}

